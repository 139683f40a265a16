# Round 4: do the mixed paths have a cliff on streams outside their windows?  (K1 had one: profiles/r04_k1_wide_streams.txt)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4n}; mkdir -p $OUT
D=lognormal,lognormal,loguniform,far_1e30,negative_far,signed_wide,thin_far_tail
for M in 4 32 1024 65536; do
  python tools/sweep.py --samples 2.5e8 --pairs $M --reps 4 --dists $D 2>&1 | cut -c1-150 | tee -a $OUT/mixed_wide.txt
done
true
