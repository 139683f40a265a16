# P1/P1b staging line experiment: LINE = 16 (64 B) vs 32 (128 B = one L2 line) records; run on the GPU box.
R=$GRAFT_REPO_ROOT; cd $R
for V in "16 3" "32 2" "32 3"; do set -- $V
  LH_EXTRA_CXXFLAGS="-DLH_LINE=$1 -DLH_P1_WGS_PER_CU=$2" python -m loghisto_amd.build --force > /dev/null 2>&1 || { echo "build failed $V"; continue; }
  echo "== LINE=$1 wgs/cu=$2"
  python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal 2>/dev/null | cut -c1-140
  python tools/sweep.py --samples 1e9 --pairs 65536 --reps 3 --dists lognormal 2>/dev/null | cut -c1-140
done
python -m loghisto_amd.build --force > /dev/null 2>&1
