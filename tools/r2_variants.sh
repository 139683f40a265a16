# A/B of tuning-build variants of the region scatter on the C3 stream (same box, same call); the survey-path parity
# tests run against every variant first.
# usage: bash tools/r2_variants.sh TAG name1 name2 ...   (names of build/liblhgpu_tuning_<name>.so; "base" = liblhgpu_tuning.so)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2x}; shift; mkdir -p $OUT; cd $R
for V in "$@"; do
  LIB=loghisto_amd/build/liblhgpu_tuning_$V.so; [ "$V" = base ] && LIB=loghisto_amd/build/liblhgpu_tuning.so
  echo "== parity $V: $(timeout 600 python tools/run_tests_with_lib.py $LIB 2>&1 | tail -1)" | tee -a $OUT/variants.txt
done
for rep in 1 2; do
for V in "$@"; do
  LIB=loghisto_amd/build/liblhgpu_tuning_$V.so; [ "$V" = base ] && LIB=loghisto_amd/build/liblhgpu_tuning.so
  for D in ${DISTS:-lognormal constant loguniform}; do
  timeout 300 python tools/sweep.py --lib $LIB --samples 1e9 --pairs 1024 --reps 5 --opt 11=2 --dists $D 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$V', '$D', 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'ovf', j.get('region_overflows'))" | tee -a $OUT/variants.txt
  done
done; done
