# Extract latency at 1 024 names against the zero-copy limit (kernel stores into pinned memory + completion word) and
# the copy path (D2H + stream sync).  usage: bash tools/r2_lat.sh TAG
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2lat}; mkdir -p $OUT; cd $R
for rep in 1 2; do
for ZC in 1 262144 1048576; do for V in 0 1; do
timeout 120 loghisto_amd/build/latency 1000 4194304 1024 $V $ZC | tee -a $OUT/latency_1024.jsonl | cut -c60-260
done; done
timeout 120 loghisto_amd/build/latency 600 4194304 4096 1 1 | tee -a $OUT/latency_1024.jsonl | cut -c60-260
timeout 120 loghisto_amd/build/latency 600 4194304 4096 1 1048576 | tee -a $OUT/latency_1024.jsonl | cut -c60-260
done
