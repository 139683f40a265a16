# Round 4, third GPU call: K1 with two LDS copies (+ variants), the bench's new N-rank code under stub ranks, bench.py.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4c}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_bench_ranks.py tests/test_gpu_merge.py tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_cpp_host.py -m gpu -q) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
show() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('$1', j['dist'], 'names', j['names'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'frac', round(j['frac_hbm_peak'],3))"; }
D=lognormal,lognormal,constant,kvalues2,kvalues3,kvalues4,kvalues8,kvalues16,kvalues3_skewed,bimodal,uniform,loguniform,normal
timeout 600 python tools/sweep.py --samples 1e9 --reps 6 --dists $D 2>/dev/null | tee $OUT/k1_c2.jsonl | show k1-c2
for v in c1 c2roll c2b1024u4 c4 c2agg64 c2agg16; do
timeout 600 python tools/sweep.py --samples 1e9 --reps 6 --dists lognormal,lognormal,constant,kvalues2,kvalues3,kvalues4,kvalues3_skewed,uniform --lib loghisto_amd/build/liblhgpu_tuning_$v.so 2>/dev/null | tee $OUT/k1_$v.jsonl | show k1-$v
done
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
