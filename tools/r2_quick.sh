# Quick loop for the mixed-stream path: parity of the second generation, then timings with the default dispatch and
# the per-kernel split.  Usage: gpurun -- 'bash tools/r2_quick.sh TAG'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2q}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_part2.py tests/test_gpu_fuzz.py tests/test_gpu_options.py -x -q) > $OUT/pytest_part2.log 2>&1
tail -3 $OUT/pytest_part2.log
if ! grep -q " passed" $OUT/pytest_part2.log || grep -q "failed" $OUT/pytest_part2.log; then echo "TESTS FAILED: no timings"; exit 1; fi
for D in lognormal constant loguniform normal; do
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 5 --dists $D 2>/dev/null | cut -c1-125 | tee -a $OUT/sweep_default.jsonl
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_survey" | cut -c1-175 | tee $OUT/kernel_trace_1024.txt
