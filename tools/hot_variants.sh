# hot-name windows in P1 (k_scatter_samples<true>) on/off, Zipf and uniform name distributions
R=$GRAFT_REPO_ROOT; cd $R
for H in 1 0; do for I in zipf zipf0.5 uniform; do
  echo "== LH_PART_HOT=$H ids=$I"
  LH_PART_HOT=$H python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal,constant --ids $I 2>/dev/null | cut -c1-170
done; done
LH_PART_HOT=1 python tools/sweep.py --samples 1e9 --pairs 4096 --reps 3 --dists lognormal 2>/dev/null | cut -c1-170
LH_PART_HOT=0 python tools/sweep.py --samples 1e9 --pairs 4096 --reps 3 --dists lognormal 2>/dev/null | cut -c1-170
