# Round 2, mixed-stream path (BASELINE config 3): correctness of the second generation first, then timings of both
# generations on the same streams, then the per-kernel breakdown.  Results under gpurun_out/r2b/.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_part2.py tests/test_gpu_fuzz.py -x -q) > $OUT/pytest_part2.log 2>&1
tail -15 $OUT/pytest_part2.log
if ! grep -q " passed" $OUT/pytest_part2.log || grep -q "failed" $OUT/pytest_part2.log; then echo "TESTS FAILED: no timings"; exit 1; fi
for V2 in 1 0; do
  timeout 600 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 9=$V2 \
      --dists lognormal,constant,uniform,exponential,normal,loguniform,lognormal25 2>/dev/null | cut -c1-260 | tee -a $OUT/sweep_1024_v2_$V2.jsonl
done
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --dists lognormal,constant --ids uniform 2>/dev/null | cut -c1-260 | tee -a $OUT/sweep_1024_uniform_ids.jsonl
for M in 64 256 4096 8192; do
  timeout 300 python tools/sweep.py --samples 1e9 --pairs $M --reps 3 --opt 9=1 --dists lognormal 2>/dev/null | cut -c1-260 | tee -a $OUT/sweep_names.jsonl
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --dists lognormal,constant,loguniform > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_survey" | cut -c1-175 | tee $OUT/kernel_trace_1024.txt
