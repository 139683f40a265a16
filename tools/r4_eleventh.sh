# Round 4: K1 with per-workgroup windows against the previous K1 (fixed window, no floating windows), alternating on one box
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4m}; mkdir -p $OUT
D=lognormal,lognormal,constant,kvalues2,kvalues4,uniform,normal,lognormal25
for i in 1 2 3; do
  python tools/sweep.py --samples 1e9 --reps 10 --dists $D 2>&1 | cut -c1-118 | sed "s/^/new $i /" | tee -a $OUT/k1_ab.txt
  python tools/sweep.py --samples 1e9 --reps 10 --dists $D --lib loghisto_amd/build/liblhgpu_tuning_k1old.so 2>&1 | cut -c1-118 | sed "s/^/old $i /" | tee -a $OUT/k1_ab.txt
done
true
