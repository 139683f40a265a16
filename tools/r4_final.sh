# Round-4 closing evidence in one call: the whole GPU suite, smoke(), the profile round, K1's SQ counters per distribution on
# the final kernel, and the worst case of the bucket index (every sample on a bucket boundary).
R=$GRAFT_REPO_ROOT; cd $R
bash tools/r2_final.sh
PARTS=k1 bash tools/r4_counters.sh final > /dev/null 2>&1; cut -c1-200 gpurun_out/final/k1_lds_counters.jsonl | grep '"set": "B"'
python tools/sweep.py --samples 1e9 --reps 6 --dists lognormal,lognormal,on_thresholds 2>&1 | cut -c1-140 | tee gpurun_out/final/on_thresholds.txt
