# Round 4, second GPU call: whole GPU suite (no -x), K1 wave aggregation v2 (two groups, one ds_add) and K1 shapes.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4b}; mkdir -p $OUT; cd $R
(timeout 1200 python -m pytest tests -m gpu -q) > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
show() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('$1', j['dist'], 'names', j['names'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'frac', round(j['frac_hbm_peak'],3))"; }
D=lognormal,lognormal,constant,kvalues2,kvalues3,kvalues4,kvalues8,kvalues16,bimodal,uniform
timeout 600 python tools/sweep.py --samples 1e9 --reps 6 --dists $D 2>/dev/null | tee $OUT/k1_agg16.jsonl | show k1-agg16
for v in agg12 noagg b1024u8 b1024u4 b512u4; do
timeout 600 python tools/sweep.py --samples 1e9 --reps 6 --dists lognormal,lognormal,constant,kvalues2,kvalues3,kvalues4,uniform --lib loghisto_amd/build/liblhgpu_tuning_$v.so 2>/dev/null | tee $OUT/k1_$v.jsonl | show k1-$v
done
timeout 600 python tools/sweep.py --samples 1e9 --pairs 16 --reps 3 --dists lognormal,constant,kvalues2,kvalues4,bimodal 2>/dev/null | tee $OUT/small16.jsonl | show small16
