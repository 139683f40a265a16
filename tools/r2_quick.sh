# Quick loop for the mixed-stream path: parity of the second generation, then scatter-pass ablations of the
# shapes / tuning builds named below.  Usage: gpurun -- 'bash tools/r2_quick.sh TAG'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2q}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests/test_gpu_part2.py tests/test_gpu_fuzz.py tests/test_gpu_options.py -x -q) > $OUT/pytest_part2.log 2>&1
tail -3 $OUT/pytest_part2.log
if ! grep -q " passed" $OUT/pytest_part2.log || grep -q "failed" $OUT/pytest_part2.log; then echo "TESTS FAILED: no timings"; exit 1; fi
run() { # lib shape dbg dist
  timeout 300 python tools/sweep.py --lib $1 --samples 1e9 --pairs 1024 --reps 3 --opt 9=1 --opt 11=$2 --opt 100=$3 --dists $4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('$1 shape $2 dbg $3 $4 avg_ms', round(j['avg_ms'],3))" | tee -a $OUT/ablate.txt
}
for LIB in loghisto_amd/build/liblhgpu_tuning.so loghisto_amd/build/liblhgpu_tuning_b8.so; do
for SH in 2 3; do for D in 0 4 20; do run $LIB $SH $D lognormal; done; done
run $LIB 2 0 constant; run $LIB 2 0 loguniform
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 3 --opt 11=2 --dists lognormal > /dev/null 2>&1
python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_plan|k_survey" | cut -c1-175 | tee $OUT/kernel_trace_1024.txt
