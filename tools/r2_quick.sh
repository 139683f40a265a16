# Quick loop for the mixed-stream path: parity of the second generation, then timings with the default dispatch.
# Usage: gpurun -- 'bash tools/r2_quick.sh TAG'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2q}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_part2.py tests/test_gpu_fuzz.py tests/test_gpu_options.py -x -q) > $OUT/pytest_part2.log 2>&1
tail -3 $OUT/pytest_part2.log
if ! grep -q " passed" $OUT/pytest_part2.log || grep -q "failed" $OUT/pytest_part2.log; then echo "TESTS FAILED: no timings"; exit 1; fi
for IDS in zipf sorted; do
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 5 --ids $IDS --dists lognormal,lognormal25 2>/dev/null | cut -c1-120,380-600 | tee -a $OUT/sweep_default.jsonl
done
