# Tuning helper (GPU box): rebuild liblhgpu.so with K1 variants and time the single-metric kernel.
R=$GRAFT_REPO_ROOT; cd $R
for v in "" "-DLH_K1_UNROLL=4" "-DLH_K1_UNROLL=2"; do
  LH_EXTRA_CXXFLAGS="$v" python -m loghisto_amd.build --force > /dev/null 2>&1
  echo "== variant [$v]"
  python tools/sweep.py --samples 1e9 --reps 8 --dists lognormal,constant,loguniform 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('   %-12s %.4f ms  %.1f GB/s' % (j['dist'], j['avg_ms'], j['GBps']))"
done
python -m loghisto_amd.build --force > /dev/null 2>&1
