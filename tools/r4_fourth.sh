# Round 4, fourth GPU call: third generation with 128-byte pieces at level 1 (vs 64-byte: -DLH_V3_PIECE=1), atomic reduce flush,
# survey every 32 calls; single-pass wave extract + wave clear; kernel split of the slice and the 1e9-pair call.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4d}; mkdir -p $OUT; cd $R
(timeout 1500 python -m pytest tests/test_gpu_pairs16.py tests/test_gpu_part3.py tests/test_gpu_options.py tests/test_gpu_fullsize.py tests/test_gpu_merge.py tests/test_gpu_sharded.py tests/test_gpu_small.py tests/test_gpu_parity.py -q) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
run() { timeout 300 python tools/sweep.py --samples $1 --pairs $2 --reps ${5:-5} --dists ${4:-lognormal} $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); v=j['v3']; n=max(1,v['samples_partitioned_v3'])
    print('n=$1 names=$2 $3', j['dist'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'ovf', j['region_overflows'], 'logw', v['window_log2'], 'l1 %.3f l2 %.3f l2ovf %.5f p2miss %.5f' % (v['records_level1']/n, v['records_level2']/n, v['level2_overflows']/n, v['reduce_window_misses']/n))" | tee -a $OUT/v3.txt; }
run 1.25e8 65536 "" lognormal 24; run 1e9 65536
run 1.25e8 65536 "--lib loghisto_amd/build/liblhgpu_tuning_piece1.so" lognormal 24; run 1e9 65536 "--lib loghisto_amd/build/liblhgpu_tuning_piece1.so"
run 1e9 65536 "" constant,loguniform,bimodal
loghisto_amd/build/latency 300 4194304 65536 1 2>&1 | tail -3 | tee $OUT/latency.txt
timeout 600 python bench.py --workload c4 --steps 10 --warmup 3 > $OUT/c4_bench.json 2> $OUT/c4_bench.err; tail -c 1500 $OUT/c4_bench.json
cd /tmp; export TMPDIR=/tmp
for sz in 1.25e8 1e9; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples $sz --pairs 65536 --reps 4 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_split|k_survey|k_ingest_pairs|k_extract|k_clear|k_v3" | cut -c1-160 | tee $OUT/trace_$sz.txt
done
