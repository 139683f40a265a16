# Mixed ingest over 1 024 Zipf names against the launch size: survey path (default from 2^24 pairs) vs the
# first-generation path (--opt 9=0) around the switch-over.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2sz}; mkdir -p $OUT; cd $R
for N in 4194304 16777216 33554432 67108864 134217728 268435456; do for O in "" "--opt 9=0" "--opt 10=1048576"; do
timeout 300 python tools/sweep.py --samples $N --pairs 1024 --reps 5 --dists lognormal $O 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('n=$N', '$O', 'avg_ms', round(j['avg_ms'],4), 'ns_per_1000', round(j['avg_ms']*1e6/j['n']*1000,3), 'v2', j['v2_samples']>0)" | tee -a $OUT/sizes.txt
done; done
