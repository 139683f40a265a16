# K1 evidence: the un-profiled HIP-event average, then the same command under
# rocprofv3 --kernel-trace with every k_ingest_single launch listed in start order.  usage: bash tools/r3_k1trace.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-k1trace}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --workload c2 --no-secondary --steps 30 --warmup 5 --no-cpu-baseline --no-parity --latency-flips 0"
$CMD 2>/dev/null | tail -1 > $OUT/unprofiled.json
rm -rf /tmp/k1t; rocprofv3 --kernel-trace -d /tmp/k1t -o t -- $CMD 2>/dev/null | tail -1 > $OUT/under_trace.json
$CMD 2>/dev/null | tail -1 > $OUT/unprofiled_after.json
{ echo "# rocprofv3 --kernel-trace -- $CMD"
  python $R/profiles/summarize_rocpd.py stats /tmp/k1t/t_results.db --min-ns 500000 | grep -E "^#|^kernel|k_ingest_single" | cut -c1-170
  python $R/profiles/summarize_rocpd.py list /tmp/k1t/t_results.db k_ingest_single --min-ns 500000 --skip 5
  for f in unprofiled under_trace unprofiled_after; do python -c "
import json; j=json.loads(open('$OUT/$f.json').read()); r=j['roofline']; print('HIP events, $f run: avg_launch_ms %.4f frac %.4f ms_per_step %.4f' % (r['avg_launch_ms'], r['frac'], j['ms_per_step']))"; done; } | tee $OUT/kernel_trace.txt
