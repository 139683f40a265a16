# Round-end check in one call: the whole GPU suite, smoke(), then tools/profile_round.sh (bench lines, traces, PMC).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; mkdir -p $OUT; cd $R
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=5) > $OUT/pytest_gpu.log 2>&1; tail -9 $OUT/pytest_gpu.log
(timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; tail -30 $OUT/profile_round.log
