# Round 4: where the third generation starts to win over the first for 8 193 .. 65 536 names now that a survey serves 32 calls.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4h}; mkdir -p $OUT; cd $R
for M in 16384 65536; do for n in 1048576 2097152 4194304 8388608; do
for o in "" "--opt 13=131072"; do
timeout 120 python tools/sweep.py --samples $n --pairs $M --reps 40 --dists lognormal $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('names $M n $n [$o]', 'avg_ms', round(j['avg_ms'],4), 'min_ms', round(j['min_ms'],4), 'Gpairs_per_s', round(j['Gsamples_per_s'],2), 'v3', j['v3']['samples_partitioned_v3']>0)" | tee -a $OUT/v3_small.txt
done; done; done
