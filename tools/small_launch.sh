# fixed cost of lane-sized mixed launches (what lh_submit_pairs issues per pinned half-buffer)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for N in 262144 1048576 4194304; do for M in 1024 1280; do
  rm -rf /tmp/ps; rocprofv3 --kernel-trace -d /tmp/ps -o t -- python $R/tools/sweep.py --samples $N --pairs $M --reps 5 --dists lognormal > /tmp/ps.out 2>/dev/null
  echo "== n=$N names=$M $(tail -1 /tmp/ps.out | cut -c1-120)"; python $R/profiles/summarize_rocpd.py stats /tmp/ps/t_results.db | grep -E "lh::" | cut -c1-132
done; done
