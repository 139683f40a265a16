# P1 / P1b ablation at 65 536 names, 1e9 samples: LH_DEBUG_FLAGS=1 drops the record stores of both scatter kernels
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for F in 0 1; do
  rm -rf /tmp/pa; LH_DEBUG_FLAGS=$F rocprofv3 --kernel-trace -d /tmp/pa -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 3 --dists lognormal > /dev/null 2>&1
  echo "== LH_DEBUG_FLAGS=$F"; python $R/profiles/summarize_rocpd.py stats /tmp/pa/t_results.db | grep -E "k_part_hist|k_scatter|k_plan" | cut -c1-150
done
