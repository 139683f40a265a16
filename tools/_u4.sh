R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/u4; cd $R
run() { timeout 400 python tools/sweep.py --samples 1000000000 --pairs $1 --reps 4 --warmup 10 --dists lognormal,normal,lognormal25,lognormal50,loguniform,signed_wide,thin_far_tail,kvalues8 --lib loghisto_amd/build/liblhgpu_tuning_$2.so 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); v = j['v3']; n = max(1, v['samples_partitioned_v3'])
    print('$1', '$2', j['dist'], 'avg_ms %.3f min %.3f' % (j['avg_ms'], j['min_ms']), 'logw', v['window_log2'], 'ovf', j['region_overflows'], 'l1cum %.3f' % (v['records_level1'] / n))
"; }
for m in 65536 20000; do
for lib in base prod sm1 sm2 base prod; do run $m $lib; done; done 2>&1 | tee $R/gpurun_out/u4/ab.txt
