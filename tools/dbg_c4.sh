R=$GRAFT_REPO_ROOT; cd $R
python tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 3 --dists lognormal 2>&1 | cut -c1-300
python tools/sweep.py --samples 1e9 --pairs 65536 --reps 2 --dists lognormal 2>&1 | cut -c1-300
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pk
rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 3 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|lh::" | cut -c1-175
