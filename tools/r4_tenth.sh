# Round 4, K1 windows (main window placed per workgroup, floating windows): parity on streams outside the default window, timings
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4j}; mkdir -p $OUT
python -m pytest tests/test_gpu_k1_window.py -x -q -s 2>&1 | tail -8 > $OUT/pytest_k1.log; cat $OUT/pytest_k1.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_hot.py -x -q 2>&1 | tail -3 | tee $OUT/pytest_parity.log
D=lognormal,lognormal,constant,kvalues2,kvalues4,uniform,normal,lognormal25,loguniform,far_1e30,negative_far,signed_wide,thin_far_tail
python tools/sweep.py --samples 1e9 --reps 6 --dists $D 2>&1 | cut -c1-140 | tee $OUT/k1_windows.txt
for L in "$@"; do [ -f "$L" ] && python tools/sweep.py --samples 1e9 --reps 6 --dists $D --lib $L 2>&1 | cut -c1-140 | tee $OUT/k1_$(basename $L .so).txt; done
