R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/u2
(cd $R; timeout 1500 python -m pytest tests/test_gpu_part3.py tests/test_gpu_mixed_streams.py tests/test_gpu_options.py tests/test_gpu_pairs16.py -q 2>&1 | tail -8 | cut -c1-400)
cd $R
for m in 65536 20000; do timeout 500 python tools/sweep.py --samples 1000000000 --pairs $m --reps 5 --dists lognormal,normal,loguniform,signed_wide,thin_far_tail,lognormal 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); v = j['v3']
    print(j['names'], j['dist'], 'avg_ms %.3f min %.3f' % (j['avg_ms'], j['min_ms']), 'logw', v['window_log2'], 'p2miss(cum)', v['reduce_window_misses'])
"; done 2>&1 | tee $R/gpurun_out/u2/sweep.txt
