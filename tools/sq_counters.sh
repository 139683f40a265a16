# SQ counters of the ingest kernels on the few-valued and the usual value distributions (tools/round.sh counters).
#   K1 (k_ingest_single), n = 1e9: LDS bank conflicts / LDS instructions / VALU instructions per distribution  -> k1_lds_counters.jsonl
#   1 024 names (k_scatter3, k_part_hist2) and 65 536 names (k_scatter4, k_split_waves, k_part_hist3), 1e9 pairs:
#   instruction mix and LDS conflicts, lognormal + kvalues2                                        -> mixed_counters.jsonl
# One rocprofv3 --pmc pass per (distribution, counter set); no tracing domains in the same run.
#   k_extract_wave at 65 536 names: instructions per name, VALU activity (PARTS=extract)             -> extract_counters.jsonl
# usage: [PARTS="k1 mixed extract"] bash tools/sq_counters.sh OUTDIR   (round.sh counters <tag>)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUTD=$R/gpurun_out/${1:-counters}; mkdir -p $OUTD
PARTS=${PARTS:-k1 mixed}
K1=$OUTD/k1_lds_counters.jsonl; MX=$OUTD/mixed_counters.jsonl

row() {  # row DB KERNEL SAMPLES_PER_LAUNCH LABEL... : one JSON line with every counter of that kernel in the db
  python - "$@" <<PY
import json, subprocess, sys
db, kern, spl = sys.argv[1], sys.argv[2], float(sys.argv[3])
label = dict(kv.split("=", 1) for kv in sys.argv[4:])
j = json.loads(subprocess.check_output(["python", "$R/profiles/summarize_rocpd.py", "pmc", db, kern]))
c = j["counters"]
if c:
    any_c = next(iter(c.values()))
    out = dict(label, kernel=kern, launches=any_c["launches"], avg_duration_us_under_pmc=round(any_c["avg_duration_us_profiled"], 1))
    for k, v in sorted(c.items()):
        out[k] = v["avg"]
    if spl > 0:
        # counters are wave-level instruction counts summed over the device: x 64 lanes / samples of a launch
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
            if k in c:
                out[k.lower() + "_per_wave_sample"] = round(c[k]["avg"] * 64.0 / spl, 2)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        out["bank_conflict_over_idx_active"] = round(c["SQ_LDS_BANK_CONFLICT"]["avg"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]["avg"]), 4)
    print(json.dumps(out))
PY
}

SET_A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SET_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"

case " $PARTS " in *" k1 "*) : > $K1;; esac
for D in lognormal constant kvalues2 kvalues4 kvalues16 bimodal loguniform signed_wide far_1e30; do
  case " $PARTS " in *" k1 "*) ;; *) break;; esac
  for S in A B; do
    eval "SET=\$SET_$S"
    rm -rf /tmp/kc; timeout 300 rocprofv3 --pmc $SET -d /tmp/kc -o t -- python $R/tools/sweep.py --samples 1e9 --reps 3 --dists $D > /tmp/kc.out 2>&1
    row /tmp/kc/t_results.db k_ingest_single 1e9 dist=$D names=1 set=$S >> $K1
  done
done

case " $PARTS " in *" mixed "*) : > $MX;; esac
for D in lognormal kvalues2; do
  case " $PARTS " in *" mixed "*) ;; *) break;; esac
  for S in A B; do
    eval "SET=\$SET_$S"
    rm -rf /tmp/kc; timeout 300 rocprofv3 --pmc $SET -d /tmp/kc -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 1024 --reps 2 --dists $D > /tmp/kc.out 2>&1
    # a 1e9-pair call over <= 8 192 names is two sub-launches (2^29 + the rest): 5e8 pairs per launch on average
    for k in k_scatter3 k_part_hist2; do row /tmp/kc/t_results.db $k 5e8 dist=$D names=1024 set=$S >> $MX; done
    rm -rf /tmp/kc; timeout 300 rocprofv3 --pmc $SET -d /tmp/kc -o t -- python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 2 --dists $D > /tmp/kc.out 2>&1
    # level 1 sees every pair; the later levels see the records level 1 forwards (their per-sample figures are per PAIR of the call)
    for k in k_scatter4 k_split_waves k_part_hist3; do row /tmp/kc/t_results.db $k 1e9 dist=$D names=65536 set=$S >> $MX; done
  done
done
# k_extract_wave at 65 536 names (tools/extract_time.py: 33 extracts of one snapshot per run): instructions per name = per wave
EX=$OUTD/extract_counters.jsonl
case " $PARTS " in *" extract "*) : > $EX;; esac
for NP in 9 1; do
  case " $PARTS " in *" extract "*) ;; *) break;; esac
  for S in A B; do
    eval "SET=\$SET_$S"
    rm -rf /tmp/kc; timeout 300 rocprofv3 --pmc $SET -d /tmp/kc -o t -- python $R/tools/extract_time.py --np $NP --reps 10 > /tmp/kc.out 2>&1
    row /tmp/kc/t_results.db k_extract_wave 0 names=65536 percentiles=$NP set=$S | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
        if k in j: j[k.lower() + '_per_name'] = round(j[k] / 65536.0, 1)
    if 'SQ_ACTIVE_INST_VALU' in j and 'SQ_BUSY_CYCLES' in j: j['valu_active_over_busy'] = round(j['SQ_ACTIVE_INST_VALU'] / max(1.0, j['SQ_BUSY_CYCLES']), 3)
    print(json.dumps(j))" >> $EX
  done
done
cat $K1 $MX $EX 2>/dev/null
