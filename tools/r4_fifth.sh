# Round 4, fifth GPU call: copy-out piece sizes (k_scatter3: 128-byte pieces; k_scatter4: 256-byte), hot-window width rule.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4e}; mkdir -p $OUT; cd $R
show() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('$1', j['dist'], 'names', j['names'], 'n', j['n'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'frac', round(j['frac_hbm_peak'],3), 'ovf', j['region_overflows'])"; }
for v in sc3p2 sc3p2c5 v3p4 hotspan; do
(timeout 600 python tools/run_tests_with_lib.py loghisto_amd/build/liblhgpu_tuning_$v.so tests/test_gpu_part2.py tests/test_gpu_part3.py -k "exact or threshold or clustered") > $OUT/pytest_$v.log 2>&1; echo "$v: $(tail -1 $OUT/pytest_$v.log)"
done
D3=lognormal,lognormal,constant,kvalues8,kvalues16,loguniform,uniform
timeout 900 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 4 --dists $D3 2>/dev/null | tee $OUT/c3_base.jsonl | show c3-base
for v in sc3p2 sc3p2c5 hotspan; do
timeout 900 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 4 --dists $D3 --lib loghisto_amd/build/liblhgpu_tuning_$v.so 2>/dev/null | tee $OUT/c3_$v.jsonl | show c3-$v
done
D4=lognormal,lognormal,kvalues8,loguniform
timeout 900 python tools/sweep.py --samples 1e9 --pairs 65536 --reps 4 --dists $D4 2>/dev/null | tee $OUT/n65536_base.jsonl | show 65536-base
for v in v3p4 hotspan; do
timeout 900 python tools/sweep.py --samples 1e9 --pairs 65536 --reps 4 --dists $D4 --lib loghisto_amd/build/liblhgpu_tuning_$v.so 2>/dev/null | tee $OUT/n65536_$v.jsonl | show 65536-$v
done
timeout 600 python tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 24 --dists lognormal,lognormal --lib loghisto_amd/build/liblhgpu_tuning_v3p4.so 2>/dev/null | tee $OUT/slice_v3p4.jsonl | show slice-v3p4
timeout 600 python tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 24 --dists lognormal,lognormal 2>/dev/null | tee $OUT/slice_base.jsonl | show slice-base
