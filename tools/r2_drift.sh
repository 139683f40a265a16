# A stream whose name skew changes within the launch: the ranking of the names is reversed half way.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2dr}; mkdir -p $OUT; cd $R
for IDS in zipf drift uniform; do
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 5 --ids $IDS --dists lognormal 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('ids=$IDS survey path', 'avg_ms', round(j['avg_ms'],3), 'region_overflows', j['region_overflows'])" | tee -a $OUT/drift.txt
timeout 300 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 5 --ids $IDS --dists lognormal --opt 9=0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('ids=$IDS first generation (per-workgroup hot names from the first tile)', 'avg_ms', round(j['avg_ms'],3))" | tee -a $OUT/drift.txt
done
