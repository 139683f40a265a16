# Round 4, sixth GPU call: reduce pass with half-width windows (two workgroups per CU): tests, slice / 1e9 timings on and off, kernel split.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4f}; mkdir -p $OUT; cd $R
(timeout 1500 python -m pytest tests/test_gpu_part3.py tests/test_gpu_options.py tests/test_gpu_fullsize.py tests/test_gpu_pairs16.py -q) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log | cut -c1-300
run() { timeout 300 python tools/sweep.py --samples $1 --pairs $2 --reps ${5:-5} --dists ${4:-lognormal} $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); v=j['v3']; n=max(1,v['samples_partitioned_v3'])
    print('n=$1 names=$2 $3', j['dist'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'ovf', j['region_overflows'], 'logw', v['window_log2'], 'l1 %.3f l2 %.3f l2ovf %.5f p2miss %.5f' % (v['records_level1']/n, v['records_level2']/n, v['level2_overflows']/n, v['reduce_window_misses']/n))" | tee -a $OUT/v3.txt; }
run 1.25e8 65536 "" lognormal,lognormal 24; run 1e9 65536 "" lognormal,lognormal
run 1.25e8 65536 "--opt 17=0" lognormal,lognormal 24; run 1e9 65536 "--opt 17=0" lognormal,lognormal
run 1e9 65536 "" constant,kvalues8,bimodal,lognormal25
cd /tmp; export TMPDIR=/tmp
for sz in 1.25e8 1e9; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples $sz --pairs 65536 --reps 4 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_split|k_survey_count" | cut -c1-160 | tee $OUT/trace_$sz.txt
done
