# Round 4: small mixed launches over many names -- partitioned (first generation) against one global atomic per sample.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4g}; mkdir -p $OUT; cd $R
for M in 1024 16384 65536; do for n in 262144 524288 1048576 2097152 4194304 8388608 16777216; do
for o in "" "--opt 17=1073741824"; do
timeout 120 python tools/sweep.py --samples $n --pairs $M --reps 12 --dists lognormal $o 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('names $M n $n [$o]', 'avg_ms', round(j['avg_ms'],4), 'min_ms', round(j['min_ms'],4), 'Gpairs_per_s', round(j['Gsamples_per_s'],2))" | tee -a $OUT/small_launches.txt
done; done; done
