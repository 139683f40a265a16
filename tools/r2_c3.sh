# Round 2, mixed-stream path (BASELINE config 3): correctness of the second generation first, then timings of both
# generations on the same streams, the per-kernel breakdown, and (tuning build) the ablations of the scatter pass.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2c}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_part2.py tests/test_gpu_fuzz.py -x -q) > $OUT/pytest_part2.log 2>&1
tail -5 $OUT/pytest_part2.log
if ! grep -q " passed" $OUT/pytest_part2.log || grep -q "failed" $OUT/pytest_part2.log; then echo "TESTS FAILED: no timings"; exit 1; fi
for SH in 0 1; do
timeout 600 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 11=$SH \
      --dists lognormal,constant,uniform,exponential,normal,loguniform,lognormal25 2>/dev/null | cut -c1-200 | tee -a $OUT/sweep_1024_v2_shape$SH.jsonl
done
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 9=0 --dists lognormal 2>/dev/null | cut -c1-200 | tee -a $OUT/sweep_1024_v1.jsonl
for SH in 0 1; do
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 11=$SH --dists lognormal --ids uniform 2>/dev/null | cut -c1-200 | tee -a $OUT/sweep_1024_uniform_ids.jsonl
for M in 64 256 4096 8192; do
  timeout 300 python tools/sweep.py --samples 1e9 --pairs $M --reps 3 --opt 9=1 --opt 11=$SH --dists lognormal 2>/dev/null | cut -c1-200 | tee -a $OUT/sweep_names.jsonl
done; done
echo "== ablations (tuning build; results wrong by design): 1 no record stores, 2 no compress, 4 no P2, 16 phase 1 only, 32 no LDS atomics, 64 no name-table gather"
for D in 0 4 16 18 48 112 114; do
  timeout 300 python tools/sweep.py --lib loghisto_amd/build/liblhgpu_tuning.so --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 100=$D --dists lognormal 2>/dev/null | cut -c1-200 | tee -a $OUT/ablate_v2.jsonl
done
cd /tmp; export TMPDIR=/tmp
for SH in 0 1; do
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 11=$SH --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_survey" | cut -c1-175 | tee $OUT/kernel_trace_1024_shape$SH.txt
done
