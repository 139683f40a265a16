# Round 4, first GPU call: the GPU suite on the new library (ADVICE fixes, few-valued cases), the pure-read ceiling, the
# VALU issue costs, few-valued sweeps of K1 / C3 / 65 536 names (wave aggregation on / off), the float32 index variant.
#   usage: bash tools/r4_first.sh [tag]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r4a}; mkdir -p $OUT; cd $R
(timeout 900 python -m pytest tests -m gpu -x -q) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
loghisto_amd/build/read_ceiling --reps 20 > $OUT/read_ceiling.jsonl 2>&1; cat $OUT/read_ceiling.jsonl | cut -c1-220
loghisto_amd/build/valu_rates > $OUT/valu_rates.jsonl 2>&1; cat $OUT/valu_rates.jsonl
FEW=lognormal,constant,kvalues2,kvalues4,kvalues8,kvalues16,bimodal
show() { python -c "
import sys,json
for l in sys.stdin:
    if not l.startswith('{'): continue
    j=json.loads(l); print('$1', j['dist'], 'names', j['names'], 'avg_ms', round(j['avg_ms'],3), 'min_ms', round(j['min_ms'],3), 'frac', round(j['frac_hbm_peak'],3))"; }
# K1 at 1e9 samples: aggregation (default) vs round 3's all-equal shortcut only
timeout 600 python tools/sweep.py --samples 1e9 --reps 5 --dists $FEW,uniform,normal,loguniform 2>/dev/null | tee $OUT/k1_agg.jsonl | show k1-agg
timeout 600 python tools/sweep.py --samples 1e9 --reps 5 --dists $FEW --lib loghisto_amd/build/liblhgpu_tuning_noagg.so 2>/dev/null | tee $OUT/k1_noagg.jsonl | show k1-noagg
# few names (single-pass kernel)
timeout 600 python tools/sweep.py --samples 1e9 --pairs 16 --reps 3 --dists $FEW 2>/dev/null | tee $OUT/small16.jsonl | show small16
# C3 and 65 536 names, default library and the float32 index variant
timeout 900 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 4 --dists $FEW 2>/dev/null | tee $OUT/c3.jsonl | show c3
(timeout 600 python tools/run_tests_with_lib.py loghisto_amd/build/liblhgpu_tuning_f32.so tests/test_gpu_part2.py tests/test_gpu_part3.py -k "threshold or exact") > $OUT/pytest_f32.log 2>&1; tail -2 $OUT/pytest_f32.log
timeout 600 python tools/sweep.py --samples 1e9 --pairs 1024 --reps 4 --dists lognormal,kvalues2,constant --lib loghisto_amd/build/liblhgpu_tuning_f32.so 2>/dev/null | tee $OUT/c3_f32.jsonl | show c3-f32
timeout 900 python tools/sweep.py --samples 1e9 --pairs 65536 --reps 4 --dists lognormal,constant,kvalues2,kvalues8,bimodal 2>/dev/null | tee $OUT/n65536.jsonl | show 65536
timeout 600 python tools/sweep.py --samples 1e9 --pairs 65536 --reps 4 --dists lognormal --lib loghisto_amd/build/liblhgpu_tuning_f32.so 2>/dev/null | tee $OUT/n65536_f32.jsonl | show 65536-f32
timeout 600 python tools/sweep.py --samples 1.25e8 --pairs 65536 --reps 16 --dists lognormal,kvalues2 2>/dev/null | tee $OUT/slice.jsonl | show slice
