# K1 (single-metric ingest): is it HBM-bound or issue-bound?  base vs a tuning build whose loads hit L2.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2k1}; shift; mkdir -p $OUT; cd $R
for rep in 1 2; do for V in "$@"; do
  LIB=loghisto_amd/build/liblhgpu_tuning_$V.so; [ "$V" = base ] && LIB=loghisto_amd/build/liblhgpu_tuning.so
  timeout 300 python tools/sweep.py --lib $LIB --samples 1e9 --reps 5 --dists loguniform,lognormal 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$V', j['dist'], 'avg_ms', round(j['avg_ms'],4), 'min_ms', round(j['min_ms'],4))" | tee -a $OUT/k1.txt
done; done
