# Round 4: third generation below 2^20 pairs; host-fed 65 536 names after the threshold change; then the round's final evidence.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4i}; mkdir -p $OUT; cd $R
for M in 16384 65536; do for n in 262144 524288; do
for o in "" "--opt 13=131072"; do
timeout 120 python tools/sweep.py --samples $n --pairs $M --reps 40 --dists lognormal $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('names $M n $n [$o]', 'avg_ms', round(j['avg_ms'],4), 'min_ms', round(j['min_ms'],4), 'Gpairs_per_s', round(j['Gsamples_per_s'],2), 'v3', j['v3']['samples_partitioned_v3']>0)" | tee -a $OUT/v3_small.txt
done; done; done
loghisto_amd/build/hostfed_native 16 8e8 65536 2>&1 | tee $OUT/hostfed_native_65536.jsonl
loghisto_amd/build/latency 300 4194304 65536 1 2>&1 | tail -1 | tee $OUT/latency.txt
bash tools/r2_final.sh
