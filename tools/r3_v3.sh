# Third-generation path (8 193 .. 65 536 names) after a change: its parity tests, then config 4's slice and 1e9 pairs
# over 65 536 / 16 384 names with the default dispatch, and the kernel split of both sizes.
# usage: gpurun -- 'bash tools/r3_v3.sh <tag> [notest] [quick]'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3v3}; mkdir -p $OUT; cd $R
if [ "$2" != "notest" ]; then
(timeout 1500 python -m pytest tests/test_gpu_part3.py -x -q) > $OUT/pytest_v3.log 2>&1
tail -15 $OUT/pytest_v3.log
if ! grep -q " passed" $OUT/pytest_v3.log || grep -q "failed\|Aborted\|Fatal\|error" $OUT/pytest_v3.log; then echo "TESTS FAILED: no timings"; exit 1; fi
fi
run() { timeout 300 python tools/sweep.py --samples $1 --pairs $2 --reps 5 --dists ${4:-lognormal} $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); v=j['v3']; n=max(1,v['samples_partitioned_v3'])
    print('n=$1 names=$2 $3', j['dist'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'ovf', j['region_overflows'], 'logw', v['window_log2'], 'l1 %.3f l2 %.3f l2ovf %.5f p2miss %.5f' % (v['records_level1']/n, v['records_level2']/n, v['level2_overflows']/n, v['reduce_window_misses']/n))" | tee -a $OUT/v3.txt; }
run 1.25e8 65536; run 1e9 65536
if [ "$3" != "quick" ]; then
run 1.25e8 16384; run 1e9 16384
run 1e9 65536 "" constant,loguniform,lognormal25,uniform
run 1e9 65536 "--ids uniform"
fi
cd /tmp; export TMPDIR=/tmp
for sz in 1.25e8 1e9; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples $sz --pairs 65536 --reps 3 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_split|k_survey|k_ingest_pairs" | cut -c1-160 | tee $OUT/trace_$sz.txt
done
