# Kernel split of config 4's slice (1.25e8 pairs over 65 536 names) and of 1e9 pairs, first-generation two-level path.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2c4}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for N in 1.25e8 1e9; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples $N --pairs 65536 --reps 3 --dists lognormal > $OUT/sweep_$N.json 2>/dev/null
echo "== n=$N: $(cut -c1-130 $OUT/sweep_$N.json)"; python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|lh::" | cut -c1-150 | tee $OUT/trace_$N.txt
done
