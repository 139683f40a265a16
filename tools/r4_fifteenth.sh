# Round 4: a 1e9-pair call over 1 024 names as ONE launch (scratch cap / sub-launch bound lifted) against the default two
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4q}; mkdir -p $OUT
for i in 1 2; do
python tools/sweep.py --samples 1e9 --pairs 1024 --reps 8 --dists lognormal,lognormal,constant 2>&1 | cut -c1-130 | sed "s/^/two  /" | tee -a $OUT/c3_onelaunch.txt
python tools/sweep.py --samples 1e9 --pairs 1024 --reps 8 --dists lognormal,lognormal,constant --opt 6=17179869184 --opt 7=1073741824 2>&1 | cut -c1-130 | sed "s/^/one  /" | tee -a $OUT/c3_onelaunch.txt
done
python tools/sweep.py --samples 1e9 --pairs 8192 --reps 6 --dists lognormal,lognormal 2>&1 | cut -c1-130 | sed "s/^/two  /" | tee -a $OUT/c3_onelaunch.txt
python tools/sweep.py --samples 1e9 --pairs 8192 --reps 6 --dists lognormal,lognormal --opt 6=17179869184 --opt 7=1073741824 2>&1 | cut -c1-130 | sed "s/^/one  /" | tee -a $OUT/c3_onelaunch.txt
true
