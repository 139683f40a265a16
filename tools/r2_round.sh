# Round-2 evidence, part 1: the GPU tests and the extract latency at 1 / 1 024 / 65 536 names (C ABI, tools/latency.cc).
# Part 2 is tools/profile_round.sh (bench lines, kernel traces, PMC passes).  Results under gpurun_out/$1.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2j}; mkdir -p $OUT; cd $R
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8) > $OUT/pytest_gpu.log 2>&1; tail -14 $OUT/pytest_gpu.log
: > $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 3000 1048576 1 >> $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 1000 4194304 1024 >> $OUT/latency.jsonl
timeout 300 loghisto_amd/build/latency 300 4194304 65536 >> $OUT/latency.jsonl
timeout 300 loghisto_amd/build/latency 300 4194304 65536 1 >> $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 1000 4194304 1024 1 >> $OUT/latency.jsonl
cat $OUT/latency.jsonl
