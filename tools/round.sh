#!/bin/bash
# tools/round.sh -- the GPU calls of a round as sub-commands of ONE script (each is one `gpurun` call; outputs go to
# gpurun_out/<tag>/, and the summaries worth keeping are copied into profiles/ by hand).
#
#   gpurun --timeout 1500 -- 'bash tools/round.sh <sub-command> [tag] [args...]'
#
#   tests <tag> [pytest args]   the GPU suite (or a part of it) + smoke()
#   p3spill <tag>               the third generation's suites on a build whose packed reduce-pass cells hand on at 2^6
#   cells32 <tag>               the GPU suite on engines of 32-bit cells, and on ones that widen in the middle of every test
#   profile <tag>               the round's evidence (tools/profile_round.sh: bench lines, kernel traces, PMC passes;
#                               written to gpurun_out/round/)
#   ab <tag> <suffix,...>       mixed-ingest A/B on ONE box: config 4's slice and the 1e9-pair call (65 536 names) and
#                               config 3 (1 024 names) timed with the product library and with every
#                               build/liblhgpu_tuning_<suffix>.so (tools/build_tuning.py -D... --name <suffix>), then the
#                               product's kernel split at the slice and at config 3
#   abl <tag> <suffix,...>      timing of ablation builds (sweep.py --nocheck: their counts are wrong by construction)
#   rows <tag>                  tools/row_stride.hip: the row store's access shapes at several row strides
#   lanes <tag>                 host-fed path: its tests + tools/hostfed_native.cc at 1 024 / 20 000 / 65 536 names
#                               (LH_OPT_LANE_GEN3 off / on / level-1 workgroup caps, 8 and 16 lane blocks)
#   lanetrace <tag>             rocprofv3 --kernel-trace of hostfed_native at 65 536 names, lanes' third generation on / off
#   direct <tag>                the small launches' reduce pass without windows (k_part_direct3): its tests, hostfed_native with
#                               it on / off, and where it stops paying (sweep at 2^20 .. 2^23 pairs, both passes)
#   merge <tag>                 lh_snapshot_merge: its tests (stub ranks, bench ranks as threads), then the direct reduce pass again
#   hotwin <tag> <suffix>       the hot-window rule: tests of the second / third generation, then 1 024 names x 1e9 pairs over
#                               lognormal / few-valued streams and 65 536 names, product against build/liblhgpu_tuning_<suffix>.so
#   extract <tag>               K2 at 65 536 names: its tests, config 4's extract_roofline, the kernel under the tracer
#   order <tag> <libs> [pairs names reps dists]   one sweep with several builds in a given order (A/B/A: build or box?)
#   counters <tag>              tools/sq_counters.sh: SQ instruction / LDS counters per distribution
#   level1 <tag> <base> <abl,...>   level 1 of the mixed ingest (k_scatter3 / k_scatter4): its parity tests, an A/B/A/B of the
#                               product against build/liblhgpu_tuning_<base>.so, then the level-1 kernel's time under
#                               rocprofv3 --kernel-trace for the product and every ablation build
#                               build/liblhgpu_tuning_<abl>.so (tools/build_tuning.py -DLH_ABL=bits --name <abl>), then
#                               the kernel's SQ wait / issue counters (two --pmc passes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SUB=$1; TAG=${2:-r5}; shift 2
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R

sweep() { # sweep <pairs> <names> <reps> <extra sweep.py args...>: one line per distribution
    local n=$1 m=$2 reps=$3; shift 3
    timeout 400 python tools/sweep.py --samples $n --pairs $m --reps $reps "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); v = j['v3']; n = max(1, v['samples_partitioned_v3'])
    print('n=$n names=$m', '$*', j['dist'], 'avg_ms %.3f min_ms %.3f' % (j['avg_ms'], j['min_ms']), 'frac %.3f' % j['frac_hbm_peak'],
          'ovf', j['region_overflows'], 'logw', v['window_log2'],
          'l1 %.3f l2 %.3f l2ovf %.5f p2miss %.5f' % (v['records_level1'] / n, v['records_level2'] / n, v['level2_overflows'] / n, v['reduce_window_misses'] / n))"
}
ktrace() { # ktrace <pairs> <names> <file>: per-kernel averages of a sweep under rocprofv3 --kernel-trace
    (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pk
     timeout 400 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples $1 --pairs $2 --reps 4 --dists lognormal > /dev/null 2>&1
     python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "kernel|k_scatter|k_part|k_split|k_survey|k_plan|k_v3|k_ingest" | cut -c1-170) | tee $3
}
suite() { # suite <pytest args...>
    (timeout 2400 python -m pytest "$@" -x -q) > $OUT/pytest.log 2>&1
    tail -6 $OUT/pytest.log | cut -c1-300; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest.log | head -20 | cut -c1-300
}

case $SUB in
tests)
    if [ $# -gt 0 ]; then suite "$@"; else suite tests -m gpu; fi
    (timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") 2>&1 | tail -2 | tee $OUT/smoke.log
    ;;
profile)
    bash tools/profile_round.sh
    ;;
p3spill)
    # the reduce pass's 16-bit window cells (k_part_hist3): the third generation's suites against a build whose fields hand
    # their counts on at 2^6 instead of 2^14 (python tools/build_tuning.py -DLH_P3_SPILL_LOG=6 --name p3s6) -- every busy
    # cell of every slot takes the hand-off path hundreds of times
    (timeout 1500 python tools/run_tests_with_lib.py loghisto_amd/build/liblhgpu_tuning_p3s6.so tests/test_gpu_part3.py \
        tests/test_gpu_fullsize.py tests/test_gpu_cells32.py tests/test_gpu_lane_blocks.py) > $OUT/p3spill.log 2>&1
    echo "LH_P3_SPILL_LOG=6: $(grep -E " passed| failed| error" $OUT/p3spill.log | tail -1)" | tee $OUT/summary.txt
    grep -hE "^(FAILED|ERROR)" $OUT/p3spill.log | cut -c1-250 | tee -a $OUT/summary.txt
    ;;
cells32)
    # the GPU suite on engines of 32-bit cells at EVERY name count, and on ones that move to uint64 cells in the middle of every
    # test (the knobs are the Python test wrapper's, loghisto_amd/engine.py; the library reads no environment).  Not in the
    # second run: four 65 536-name engines on one GPU, each with a wide store beside its narrow ones, do not fit 288 GB.
    (LH_TEST_CELL_BITS=32 timeout 1200 python -m pytest tests -m gpu -q) > $OUT/bits32.log 2>&1
    echo "LH_TEST_CELL_BITS=32: $(grep -E " passed| failed| error" $OUT/bits32.log | tail -1)" | tee $OUT/summary.txt
    (LH_TEST_CELL_BITS=32 LH_TEST_WIDEN_AT=200000 timeout 1200 python -m pytest tests -m gpu -q \
        --deselect "tests/test_gpu_bench_ranks.py::test_c4_step_with_ranks_as_threads[4-65536]") > $OUT/widen.log 2>&1
    echo "LH_TEST_CELL_BITS=32 LH_TEST_WIDEN_AT=200000: $(grep -E " passed| failed| error" $OUT/widen.log | tail -1)" | tee -a $OUT/summary.txt
    grep -hE "^(FAILED|ERROR)" $OUT/bits32.log $OUT/widen.log | cut -c1-250 | tee -a $OUT/summary.txt
    ;;
ab)
    for sfx in "" $(echo ${1:-} | tr ',' ' '); do
        lib=""; [ -n "$sfx" ] && lib="--lib loghisto_amd/build/liblhgpu_tuning_$sfx.so"
        echo "== ${sfx:-product}" | tee -a $OUT/ab.txt
        sweep 1.25e8 65536 24 --dists lognormal $lib | tee -a $OUT/ab.txt
        sweep 1e9 65536 5 --dists lognormal $lib | tee -a $OUT/ab.txt
        sweep 1e9 1024 6 --dists lognormal,kvalues8 $lib | tee -a $OUT/ab.txt
    done
    ktrace 1.25e8 65536 $OUT/trace_slice.txt
    ktrace 1e9 1024 $OUT/trace_c3.txt
    ;;
abl)
    for sfx in $(echo ${1:-} | tr ',' ' '); do
        echo "== $sfx" | tee -a $OUT/abl.txt
        sweep 1.25e8 65536 16 --dists lognormal --nocheck --lib loghisto_amd/build/liblhgpu_tuning_$sfx.so | tee -a $OUT/abl.txt
        sweep 1e9 65536 4 --dists lognormal --nocheck --lib loghisto_amd/build/liblhgpu_tuning_$sfx.so | tee -a $OUT/abl.txt
    done
    ;;
rows)
    loghisto_amd/build/row_stride --reps 10 2>&1 | cut -c1-250 | tee $OUT/row_stride.jsonl
    ;;
lanes)
    suite tests/test_gpu_lane_blocks.py tests/test_gpu_faults.py tests/test_gpu_pairs16.py
    # hostfed_native [threads] [pairs] [names] [batch] [lane_gen3: 0 off, 1 on, n >= 2: on with n level-1 workgroups] [survey_every] [lane_blocks]
    for args in "16 8e8 1024 1048576 1 0 16" "16 8e8 1024 1048576 1 0 8" "16 8e8 20000 1048576 1 0 16" "16 8e8 65536 1048576 1 0 16" \
                "16 8e8 65536 1048576 1 0 8" "16 8e8 65536 1048576 0 0 16" "16 8e8 65536 1048576 16 0 16" "16 8e8 65536 1048576 256 0 16" \
                "8 8e8 65536 1048576 1 0 16"; do
        loghisto_amd/build/hostfed_native $args 2>&1 | grep -v "^counters" | sed -e "s/^{/{\"args\": \"$args\", /" | cut -c1-360 | tee -a $OUT/hostfed_native.jsonl
    done
    ;;
lanetrace)
    for g3 in 1 0; do
        (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pk
         timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- $R/loghisto_amd/build/hostfed_native 16 4e8 65536 1048576 $g3 > $OUT/run_$g3.txt 2>&1
         python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | cut -c1-170) | grep -v "^[WE]2" | tee $OUT/trace_gen3_$g3.txt | head -24
        grep -v "^[WE]2" $OUT/run_$g3.txt | cut -c1-400
    done
    ;;
direct)
    suite tests/test_gpu_part3.py tests/test_gpu_lane_blocks.py tests/test_gpu_faults.py tests/test_gpu_pairs16.py tests/test_gpu_options.py
    for args in "16 8e8 65536 1048576 1 0 16 0" "16 8e8 65536 1048576 1 0 16 1" "16 8e8 65536 2097152 1 0 16 0" "16 8e8 20000 1048576 1 0 16 0" \
                "16 8e8 20000 1048576 1 0 16 1" "16 8e8 1024 1048576 1 0 16 0"; do
        loghisto_amd/build/hostfed_native $args 2>&1 | grep -v "^counters" | sed -e "s/^{/{\"args\": \"$args\", /" | cut -c1-360 | tee -a $OUT/hostfed_native.jsonl
    done
    for n in 1048576 2097152 4194304 8388608; do
        for dm in 1 1073741824; do
            echo "direct_max=$dm" | tee -a $OUT/sweep.txt
            sweep $n 65536 30 --dists lognormal --opt 21=$dm --opt 13=131072 | tee -a $OUT/sweep.txt
        done
    done
    ;;
merge)
    suite tests/test_gpu_merge.py tests/test_gpu_bench_ranks.py tests/test_gpu_part3.py::test_small_launches_reduce_without_windows tests/test_gpu_lane_blocks.py
    for args in "16 8e8 65536 1048576 1 0 16 0" "16 8e8 65536 1048576 1 0 16 1" "16 8e8 20000 1048576 1 0 16 0"; do
        loghisto_amd/build/hostfed_native $args 2>&1 | grep -v "^counters" | sed -e "s/^{/{\"args\": \"$args\", /" | cut -c1-360 | tee -a $OUT/hostfed_native.jsonl
    done
    for n in 2097152 4194304; do
        for dm in 1 1073741824; do
            echo "direct_max=$dm" | tee -a $OUT/sweep.txt
            sweep $n 65536 30 --dists lognormal --opt 21=$dm --opt 13=131072 | tee -a $OUT/sweep.txt
        done
    done
    ;;
hotwin)
    suite tests/test_gpu_part2.py tests/test_gpu_part3.py tests/test_gpu_lane_blocks.py tests/test_gpu_options.py
    for sfx in "" ${1:-}; do
        lib=""; [ -n "$sfx" ] && lib="--lib loghisto_amd/build/liblhgpu_tuning_$sfx.so"
        echo "== ${sfx:-product}" | tee -a $OUT/ab.txt
        sweep 1e9 1024 5 --dists lognormal,kvalues4,kvalues8,kvalues16,lognormal25,bimodal $lib | tee -a $OUT/ab.txt
        sweep 1e9 65536 4 --dists lognormal,kvalues8 $lib | tee -a $OUT/ab.txt
    done
    ;;
extract)
    # K2 at 65 536 names: the tests that hold k_extract_wave against the oracle and against k_extract, then config 4 on one
    # rank (extract_roofline: kernel and copy apart; extract latency over 300 flips) and the kernel under the tracer
    suite tests/test_gpu_extract_thresholds.py tests/test_gpu_options.py tests/test_gpu_part3.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_merge.py
    python bench.py --workload c4 --steps 5 --warmup 2 --no-parity --latency-flips 300 2> $OUT/c4.err | grep "^{" | tail -1 > $OUT/c4_bench.json
    python - <<PY | tee $OUT/extract.txt
import json
j = json.load(open("$OUT/c4_bench.json"))
x = j["extract_roofline"]
print("k_extract_wave kernel_ms %.4f copy_ms %.4f kernel_frac %.3f  extract_owned_ms %.3f  latency %s" % (
    x["kernel_ms"], x["copy_ms"], x["kernel_frac"], j.get("extract_owned_ms", -1), j.get("extract_latency_us")))
PY
    (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pk
     timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/bench.py --workload c4 --steps 3 --warmup 1 --no-parity --latency-flips 0 > /dev/null 2>&1
     python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "^kernel|k_extract|k_clear|k_pack|k_unpack|k_merge" | cut -c1-170) | tee $OUT/trace.txt
    ;;
order)
    # order <tag> <lib,lib,...> <pairs> <names> <reps> <dists>: the same sweep with several libraries IN THE GIVEN ORDER
    # ("product" = loghisto_amd/liblhgpu.so, anything else = build/liblhgpu_tuning_<name>.so; a name may repeat): tells a
    # difference between two builds from what the box does over the minutes of a call
    for sfx in $(echo ${1:-product} | tr ',' ' '); do
        lib=""; [ "$sfx" != product ] && lib="--lib loghisto_amd/build/liblhgpu_tuning_$sfx.so"
        echo "== $sfx" | tee -a $OUT/order.txt
        sweep ${2:-1e9} ${3:-1024} ${4:-5} --dists ${5:-lognormal} $lib | tee -a $OUT/order.txt
    done
    ;;
counters)
    bash tools/sq_counters.sh $TAG
    ;;
level1)
    BASE=${1:-}; ABLS=$(echo ${2:-} | tr ',' ' ')
    if [ -n "${L1_TESTS:-}" ]; then suite $L1_TESTS; else suite tests/test_gpu_part2.py tests/test_gpu_part3.py tests/test_gpu_lane_blocks.py tests/test_gpu_pairs16.py; fi
    if [ -n "$BASE" ]; then
        for sfx in $BASE product $BASE product; do
            lib=""; [ "$sfx" != product ] && lib="--lib loghisto_amd/build/liblhgpu_tuning_$sfx.so"
            echo "== $sfx" | tee -a $OUT/ab.txt
            sweep 1e9 1024 6 --dists lognormal $lib | tee -a $OUT/ab.txt
            sweep 1e9 65536 5 --dists lognormal $lib | tee -a $OUT/ab.txt
            sweep 1.25e8 65536 24 --dists lognormal $lib | tee -a $OUT/ab.txt
        done
    fi
    l1trace() { # l1trace <label> <lib args...>: the level-1 kernels' rows of a kernel trace at both name counts
        for m in 1024 65536; do
            (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pk
             timeout 300 rocprofv3 --kernel-trace -d /tmp/pk -o t -- python $R/tools/sweep.py --samples 1e9 --pairs $m --reps 4 --dists lognormal --nocheck "${@:2}" > /dev/null 2>&1
             python $R/profiles/summarize_rocpd.py stats /tmp/pk/t_results.db | grep -E "k_scatter|k_part_hist|k_split|k_hot|k_survey" | cut -c1-150 | sed -e "s/^/$1 names=$m  /")
        done
    }
    l1trace product | tee -a $OUT/l1trace.txt
    for sfx in $ABLS; do l1trace $sfx --lib $R/loghisto_amd/build/liblhgpu_tuning_$sfx.so | tee -a $OUT/l1trace.txt; done
    # SQ counters of the product's level-1 kernels: what the waves wait for
    SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
          "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
          "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
    for m in 1024 65536; do
        for i in 0 1 2; do
            (cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kc
             timeout 300 rocprofv3 --pmc ${SETS[$i]} -d /tmp/kc -o t -- python $R/tools/sweep.py --samples 1e9 --pairs $m --reps 2 --dists lognormal > /tmp/kc.out 2>&1
             for k in k_scatter3 k_scatter4; do
                 python $R/profiles/summarize_rocpd.py pmc /tmp/kc/t_results.db $k | python -c "import sys, json; j = json.load(sys.stdin); print(json.dumps(dict(names=$m, kernel=j['kernel'], counters={k: round(v['avg']) for k, v in j['counters'].items()}, avg_us=(list(j['counters'].values()) or [dict(avg_duration_us_profiled=0)])[0]['avg_duration_us_profiled'])))"
             done) | grep -v '"counters": {}' | tee -a $OUT/l1_counters.jsonl
        done
    done
    ;;
*)
    echo "unknown sub-command: $SUB"; exit 2;;
esac
