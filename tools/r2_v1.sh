# First-generation (two-level) path after a change: its parity tests, then 65 536 / 16 384 names at 1e9 pairs and
# config 4's slice with the default dispatch, and the kernel split of the slice.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2v1}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_twolevel.py tests/test_gpu_hot.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q) > $OUT/pytest_v1.log 2>&1
tail -2 $OUT/pytest_v1.log
if ! grep -q " passed" $OUT/pytest_v1.log || grep -q "failed\|Aborted\|Fatal" $OUT/pytest_v1.log; then echo "TESTS FAILED: no timings"; exit 1; fi
run() { timeout 300 python tools/sweep.py --samples $1 --pairs $2 --reps 5 --dists lognormal $3 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('n=$1 names=$2 $3', 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3))" | tee -a $OUT/v1.txt; }
for rep in 1 2; do run 1.25e8 65536; run 1e9 65536; run 4194304 1024; run 1.25e8 16384; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 3 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan" | cut -c1-150 | tee $OUT/trace_slice.txt
