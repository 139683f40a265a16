# LDS contention counters of K1 (k_ingest_single) across the contention sweep's distributions.
# One rocprofv3 --pmc pass per distribution (no tracing domains mixed in), n = 1e8 samples.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/k1_lds.jsonl; : > $OUT
for D in lognormal constant uniform exponential normal loguniform lognormal25; do
  rm -rf /tmp/kl; timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/kl -o t -- python $R/tools/sweep.py --samples 1e8 --reps 3 --dists $D > /tmp/kl.out 2>/dev/null
  python - "$D" >> $OUT <<PY
import json, subprocess, sys
d = sys.argv[1]
j = json.loads(subprocess.check_output(["python", "$R/profiles/summarize_rocpd.py", "pmc", "/tmp/kl/t_results.db", "k_ingest_single"]))
c = j["counters"]
row = {"dist": d, "n": 100000000, "kernel": "k_ingest_single", "launches": c["SQ_LDS_IDX_ACTIVE"]["launches"]}
for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES"):
    row[k] = c[k]["avg"]
row["bank_conflict_over_idx_active"] = row["SQ_LDS_BANK_CONFLICT"] / max(1.0, row["SQ_LDS_IDX_ACTIVE"])
row["avg_duration_us_under_pmc"] = c["SQ_LDS_IDX_ACTIVE"]["avg_duration_us_profiled"]
print(json.dumps(row))
PY
done
cat $OUT
