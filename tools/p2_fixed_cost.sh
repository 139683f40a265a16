cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for N in 1e6 1e7 1.25e8; do
  rm -rf /tmp/pp; rocprofv3 --kernel-trace -d /tmp/pp -o t -- python $R/tools/sweep.py --samples $N --pairs 65536 --reps 3 --dists lognormal > /dev/null 2>&1
  echo "== n=$N"; python $R/profiles/summarize_rocpd.py stats /tmp/pp/t_results.db | grep -E "k_part_hist|k_scatter|k_plan|k_ingest_pairs" | cut -c1-130
done
