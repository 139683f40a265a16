R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2m}; mkdir -p $OUT; cd $R
echo "== ablations (tuning build; results wrong by design): 1 no record stores, 2 no compress, 4 no P2, 16 phase 1 only, 32 no LDS atomics, 64 no name-table gather"
for SH in 0 1; do for D in 4 20 132 260 5; do
  timeout 300 python tools/sweep.py --lib loghisto_amd/build/liblhgpu_tuning.so --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 11=$SH --opt 100=$D --dists lognormal 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('shape $SH dbg $D avg_ms', round(j['avg_ms'],3))" | tee -a $OUT/ablate.txt
done; done
