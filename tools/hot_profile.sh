cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for H in 1 0; do
  rm -rf /tmp/ph; LH_PART_HOT=$H rocprofv3 --kernel-trace -d /tmp/ph -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal > /dev/null 2>&1
  echo "== LH_PART_HOT=$H"; python $R/profiles/summarize_rocpd.py stats /tmp/ph/t_results.db | grep -E "k_part_hist|k_scatter|k_plan" | cut -c1-160
done
