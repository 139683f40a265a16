# HBM bytes of one 65 536-name, 1e9-pair call (third-generation kernels summed): separate FETCH_SIZE / WRITE_SIZE passes.
# usage: bash tools/r3_pmc_bytes.sh <tag>
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r3bytes}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 3 --dists lognormal"
rm -rf /tmp/pf /tmp/pw
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o t -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o t -- $CMD > /dev/null 2>&1
python - <<PY
import json, subprocess
R="$R"
def pmc(db, k):
    return json.loads(subprocess.check_output(["python", R+"/profiles/summarize_rocpd.py", "pmc", db, k]))["counters"]
ks = ["k_survey_count_h", "k_survey_pick", "k_survey_plan_h", "k_survey_remap", "k_v3_prepare", "k_scatter4", "k_split_waves",
      "k_split_records", "k_part_hist3", "k_plan_count", "k_plan_scan", "k_plan_scatter", "k_v3_report", "k_ingest_pairs"]
calls = None
per = {}; rd = wr = 0.0
sc = pmc("/tmp/pf/t_results.db", "k_scatter4")["FETCH_SIZE"]["launches"]
calls = sc  # one k_scatter4 launch per call
for k in ks:
    cf = pmc("/tmp/pf/t_results.db", k).get("FETCH_SIZE"); cw = pmc("/tmp/pw/t_results.db", k).get("WRITE_SIZE")
    if not cf: continue
    r = cf["avg"]*cf["launches"]*2048/calls; w = (cw["avg"]*cw["launches"]*1024/calls) if cw else 0.0
    per[k] = {"launches_per_call": cf["launches"]/calls, "read_bytes_per_call": r, "write_bytes_per_call": w,
              "avg_duration_us_under_pmc": cf["avg_duration_us_profiled"]}
    rd += r; wr += w
json.dump({"workload": "65 536 Zipf(1.0) names, 1e9 (uint32 id, float64 value) pairs, lognormal values: one lh_submit_pairs_device call",
 "pairs_per_call": 1000000000, "names": 65536, "calls_in_the_run": calls,
 "commands": ["rocprofv3 --pmc FETCH_SIZE -- $CMD", "rocprofv3 --pmc WRITE_SIZE -- $CMD"],
 "corrections": "FETCH_SIZE is in KiB and on gfx950 counts the 128-B requests of a 16-B/lane coalesced stream as 64 B: read bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md 'HBM'). WRITE_SIZE*1024, uncalibrated.",
 "kernels": per, "hbm_read_bytes_per_call": rd, "hbm_write_bytes_per_call": wr, "hbm_bytes_per_call": rd+wr,
 "algorithmic_bytes_per_call": 12e9, "traffic_over_algorithmic": (rd+wr)/12e9}, open("$OUT/pmc_bytes.json", "w"), indent=1)
print(json.dumps({"read": rd, "write": wr, "ratio": (rd+wr)/12e9, "calls": calls}))
PY
