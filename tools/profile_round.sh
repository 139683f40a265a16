# Per-round evidence under gpurun_out/round/ (copy into profiles/ as r06_* afterwards: tools/collect_profiles.py).  One gpurun call.
# EVERY file written here carries the tree stamp (bench.tree_stamp(): one digest of the sources a result can depend on) --
# JSON: "tree_stamp"; JSON lines: a first line {"tree_stamp": ...}; text: a first line "# tree_stamp: ..." -- and
# tests/test_profiles_fresh.py holds the committed copies to the tree.  The sweep over the value distributions runs LAST
# (on the kernels everything else was measured on) and behind a warm-up launch series (its first line used to be a
# cold-clock draw).
# Every PMC summary records the hashes of the kernel sources it was measured on (bench.source_hashes): bench.py refuses a
# summary whose hashes differ from the tree's (traffic_source: "stale ...").
#   bench.json              python bench.py (default flags: headline C2 + secondary C3 / C4 / host-fed / C5)
#   kernel_trace.txt        rocprofv3 --kernel-trace of the C2 bench command (no cpu baseline), summarised
#   k1_pmc.json             FETCH_SIZE / WRITE_SIZE passes for k_ingest_single, corrected per MI355X_MICROARCH.md
#   c3_kernel_trace.txt     rocprofv3 --kernel-trace of bench.py --workload c3
#   c3_pmc.json             FETCH_SIZE / WRITE_SIZE passes of the same command, every kernel of the call summed
#   c4_bench.json           bench.py --workload c4 (one rank)
#   c4_kernel_trace.txt / c4_pmc.json   the same for bench.py --workload c4 (65 536 names, one rank's 1.25e8-pair slice)
#   first_calls.txt         tools/first_call.py: calls 0 .. 3 of fresh engines on wide, sorted and run-clustered streams
# PMC passes are separate runs with no tracing domain mixed in.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round; mkdir -p $OUT; cd $R
STAMP=$(python -c "import bench; print(bench.tree_stamp())")
{ echo "{\"tree_stamp\": \"$STAMP\"}"; loghisto_amd/build/read_ceiling --reps 20 2>&1; } > $OUT/read_ceiling.jsonl
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --workload c2 --no-secondary --steps 30 --warmup 5 --no-cpu-baseline --no-parity --latency-flips 0"
rm -rf /tmp/pr_trace; rocprofv3 --kernel-trace -d /tmp/pr_trace -o t -- $CMD 2>/dev/null | grep "^{" | tail -1 > $OUT/bench_under_trace.json
$CMD 2>/dev/null | grep "^{" | tail -1 > $OUT/bench_unprofiled_after.json
{ echo "# tree_stamp: $STAMP"; echo "# rocprofv3 --kernel-trace -- $CMD"
  python $R/profiles/summarize_rocpd.py stats /tmp/pr_trace/t_results.db --min-ns 500000 | grep -E "^#|^kernel|k_ingest_single" | cut -c1-170
  python $R/profiles/summarize_rocpd.py list /tmp/pr_trace/t_results.db k_ingest_single --min-ns 500000 --skip 30
  for f in bench_under_trace bench_unprofiled_after; do python -c "
import json; j=json.loads(open('$OUT/$f.json').read()); r=j['roofline']; print('HIP events, $f run: avg_launch_ms %.4f frac %.4f ms_per_step %.4f' % (r['avg_launch_ms'], r['frac'], j['ms_per_step']))"; done; } > $OUT/kernel_trace.txt
CMD2="python $R/bench.py --workload c2 --no-secondary --steps 3 --warmup 1 --no-cpu-baseline --no-parity --latency-flips 0"
rm -rf /tmp/pr_fetch /tmp/pr_write
rocprofv3 --pmc FETCH_SIZE -d /tmp/pr_fetch -o t -- $CMD2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pr_write -o t -- $CMD2 > /dev/null 2>&1
CMD3="python $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --latency-flips 0"
rm -rf /tmp/pr_c3t /tmp/pr_c3f /tmp/pr_c3w
rocprofv3 --kernel-trace -d /tmp/pr_c3t -o t -- $CMD3 > $OUT/c3_under_trace.json 2>/dev/null
{ echo "# tree_stamp: $STAMP"; echo "# rocprofv3 --kernel-trace -- $CMD3"; python $R/profiles/summarize_rocpd.py stats /tmp/pr_c3t/t_results.db | grep -E "^#|^kernel|lh::" | cut -c1-170; } > $OUT/c3_kernel_trace.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pr_c3f -o t -- $CMD3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pr_c3w -o t -- $CMD3 > /dev/null 2>&1
CMD4="python $R/bench.py --workload c4 --steps 3 --warmup 1 --no-parity --latency-flips 0"
rm -rf /tmp/pr_c4t /tmp/pr_c4f /tmp/pr_c4w
rocprofv3 --kernel-trace -d /tmp/pr_c4t -o t -- $CMD4 > $OUT/c4_under_trace.json 2>/dev/null
{ echo "# tree_stamp: $STAMP"; echo "# rocprofv3 --kernel-trace -- $CMD4"; python $R/profiles/summarize_rocpd.py stats /tmp/pr_c4t/t_results.db | grep -E "^#|^kernel|lh::" | cut -c1-170
  # the small merge / plan kernels launch by launch, in start order (10 - 60 x spreads): the run makes 4 SERIAL
  # steps (every phase drained: the kernel alone on the GPU) and then 4 PIPELINED ones (the kernel beside the next step's level 1,
  # which holds every CU's LDS: its workgroups wait for a CU)
  for k in k_merge_widths k_merge_prep k_plan_scan k_extract_wave; do python $R/profiles/summarize_rocpd.py list /tmp/pr_c4t/t_results.db $k | cut -c1-200; done; } > $OUT/c4_kernel_trace.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pr_c4f -o t -- $CMD4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pr_c4w -o t -- $CMD4 > /dev/null 2>&1
python - <<PY
import json, subprocess
R="$R"
import sys
sys.path.insert(0, R)
import bench
def pmc(db, k, min_ns=0):
    return json.loads(subprocess.check_output(["python", R+"/profiles/summarize_rocpd.py", "pmc", db, k, "--min-ns", str(min_ns)]))["counters"]
CORR = ("FETCH_SIZE is in KiB and on gfx950 counts the 128-B requests of a 16-B/lane coalesced stream as 64 B: read "
        "bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md 'HBM'). WRITE_SIZE*1024, uncalibrated.")
f=pmc("/tmp/pr_fetch/t_results.db","k_ingest_single",500000)["FETCH_SIZE"]; w=pmc("/tmp/pr_write/t_results.db","k_ingest_single",500000)["WRITE_SIZE"]
out={"kernel":"lh::k_ingest_single","workload":"1e9 float64 samples, lognormal(ln 1e5, 1), one metric",
 "commands":["rocprofv3 --pmc FETCH_SIZE -- $CMD2","rocprofv3 --pmc WRITE_SIZE -- $CMD2"],
 "FETCH_SIZE_KiB_per_launch":f["avg"],"WRITE_SIZE_KiB_per_launch":w["avg"],"launches":f["launches"],
 "corrections":CORR, "sources": bench.source_hashes("k1"), "tree_stamp": bench.tree_stamp(),
 "hbm_read_bytes_per_launch":f["avg"]*2048,"hbm_write_bytes_per_launch":w["avg"]*1024,"algorithmic_bytes_per_launch":8e9,
 "read_over_algorithmic":f["avg"]*2048/8e9,
 "avg_duration_us_under_pmc":{"FETCH_SIZE pass":f["avg_duration_us_profiled"],"WRITE_SIZE pass":w["avg_duration_us_profiled"]}}
json.dump(out, open("$OUT/k1_pmc.json","w"), indent=1)
# C3: every kernel of one lh_submit_pairs_device call (4 calls in the run: 1 warmup + 3 timed)
calls = 4
kernels = ["k_survey_count", "k_survey_plan", "k_scatter3", "k_scatter2", "k_scatter_clustered", "k_hot_reduce", "k_plan_count", "k_plan_scan",
           "k_plan_scatter", "k_part_hist2", "k_ingest_pairs"]
per = {}; rd = wr = 0.0
for k in kernels:
    cf = pmc("/tmp/pr_c3f/t_results.db", k).get("FETCH_SIZE"); cw = pmc("/tmp/pr_c3w/t_results.db", k).get("WRITE_SIZE")
    if not cf: continue
    r = cf["avg"]*cf["launches"]*2048/calls; ww = (cw["avg"]*cw["launches"]*1024/calls) if cw else 0.0
    per[k] = {"launches_per_call": cf["launches"]/calls, "read_bytes_per_call": r, "write_bytes_per_call": ww,
              "avg_duration_us_under_pmc": cf["avg_duration_us_profiled"]}
    rd += r; wr += ww
json.dump({"workload":"C3: 1e9 (uint32 id, float64 value) pairs over 1 024 Zipf(1.0) names, lognormal values",
 "pairs_per_call": 1000000000, "names": 1024, "sources": bench.source_hashes("c3"), "tree_stamp": bench.tree_stamp(),
 "commands":["rocprofv3 --pmc FETCH_SIZE -- $CMD3","rocprofv3 --pmc WRITE_SIZE -- $CMD3"], "corrections": CORR,
 "kernels": per, "hbm_read_bytes_per_call": rd, "hbm_write_bytes_per_call": wr, "hbm_bytes_per_call": rd+wr,
 "algorithmic_bytes_per_call": 12e9, "traffic_over_algorithmic": (rd+wr)/12e9}, open("$OUT/c3_pmc.json","w"), indent=1)
# C4: every kernel of one lh_submit_pairs_device call of the slice.  The run makes 1 warm-up + 3 pipelined + a serial
# pass of timed calls: one k_scatter4 launch per call (the extract-latency leg, whose small intervals take the same
# kernels since round 4, is switched off: --latency-flips 0).
calls = pmc("/tmp/pr_c4f/t_results.db", "k_scatter4")["FETCH_SIZE"]["launches"]
k4 = ["k_survey_count_h", "k_survey_pick", "k_survey_plan_h", "k_survey_remap", "k_scatter4", "k_scatter_clustered", "k_hot_reduce", "k_split_waves",
      "k_split_records", "k_part_hist3", "k_v3_report"]
per = {}; rd = wr = 0.0
for k in k4:
    cf = pmc("/tmp/pr_c4f/t_results.db", k).get("FETCH_SIZE"); cw = pmc("/tmp/pr_c4w/t_results.db", k).get("WRITE_SIZE")
    if not cf: continue
    r = cf["avg"]*cf["launches"]*2048/calls; ww = (cw["avg"]*cw["launches"]*1024/calls) if cw else 0.0
    per[k] = {"launches_per_call": cf["launches"]/calls, "read_bytes_per_call": r, "write_bytes_per_call": ww,
              "avg_duration_us_under_pmc": cf["avg_duration_us_profiled"]}
    rd += r; wr += ww
json.dump({"workload":"C4 one rank: 1.25e8 (uint32 id, float64 value) pairs over 65 536 Zipf(1.0) names, lognormal values",
 "pairs_per_call": 125000000, "names": 65536, "calls_in_the_run": calls, "sources": bench.source_hashes("c4"), "tree_stamp": bench.tree_stamp(),
 "commands":["rocprofv3 --pmc FETCH_SIZE -- $CMD4","rocprofv3 --pmc WRITE_SIZE -- $CMD4"], "corrections": CORR,
 "note": "the plan kernels (k_plan_count / k_plan_scan / k_plan_scatter: chunk descriptors only) and memsets are not in the sum",
 "kernels": per, "hbm_read_bytes_per_call": rd, "hbm_write_bytes_per_call": wr, "hbm_bytes_per_call": rd+wr,
 "algorithmic_bytes_per_call": 1.5e9, "traffic_over_algorithmic": (rd+wr)/1.5e9}, open("$OUT/c4_pmc.json","w"), indent=1)
PY
# 65 536 names, 1e9 pairs per call: bytes of every kernel of one call (tools/sweep.py), and its kernel split
CMD5="python $R/tools/sweep.py --samples 1e9 --pairs 65536 --reps 3 --dists lognormal"
rm -rf /tmp/pr_5t /tmp/pr_5f /tmp/pr_5w
rocprofv3 --kernel-trace -d /tmp/pr_5t -o t -- $CMD5 > /dev/null 2>&1
{ echo "# tree_stamp: $STAMP"; echo "# rocprofv3 --kernel-trace -- $CMD5"; python $R/profiles/summarize_rocpd.py stats /tmp/pr_5t/t_results.db | grep -E "^#|^kernel|lh::" | cut -c1-170; } > $OUT/c4_names_1e9_kernel_trace.txt
rocprofv3 --pmc FETCH_SIZE -d /tmp/pr_5f -o t -- $CMD5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pr_5w -o t -- $CMD5 > /dev/null 2>&1
python - <<PY
import json, subprocess, sys
R="$R"
sys.path.insert(0, R)
import bench
def pmc(db, k):
    return json.loads(subprocess.check_output(["python", R+"/profiles/summarize_rocpd.py", "pmc", db, k]))["counters"]
ks = ["k_survey_count_h", "k_survey_pick", "k_survey_plan_h", "k_survey_remap", "k_v3_prepare", "k_scatter4", "k_scatter_clustered", "k_hot_reduce", "k_split_waves",
      "k_split_records", "k_part_hist3", "k_plan_count", "k_plan_scan", "k_plan_scatter", "k_v3_report", "k_ingest_pairs"]
per = {}; rd = wr = 0.0
calls = pmc("/tmp/pr_5f/t_results.db", "k_scatter4")["FETCH_SIZE"]["launches"]
for k in ks:
    cf = pmc("/tmp/pr_5f/t_results.db", k).get("FETCH_SIZE"); cw = pmc("/tmp/pr_5w/t_results.db", k).get("WRITE_SIZE")
    if not cf: continue
    r = cf["avg"]*cf["launches"]*2048/calls; w = (cw["avg"]*cw["launches"]*1024/calls) if cw else 0.0
    per[k] = {"launches_per_call": cf["launches"]/calls, "read_bytes_per_call": r, "write_bytes_per_call": w,
              "avg_duration_us_under_pmc": cf["avg_duration_us_profiled"]}
    rd += r; wr += w
json.dump({"workload": "65 536 Zipf(1.0) names, 1e9 (uint32 id, float64 value) pairs, lognormal values: one lh_submit_pairs_device call",
 "pairs_per_call": 1000000000, "names": 65536, "calls_in_the_run": calls, "sources": bench.source_hashes("c4"), "tree_stamp": bench.tree_stamp(),
 "commands": ["rocprofv3 --pmc FETCH_SIZE -- $CMD5", "rocprofv3 --pmc WRITE_SIZE -- $CMD5"],
 "corrections": "FETCH_SIZE is in KiB and on gfx950 counts the 128-B requests of a 16-B/lane coalesced stream as 64 B: read bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md 'HBM'). WRITE_SIZE*1024, uncalibrated.",
 "kernels": per, "hbm_read_bytes_per_call": rd, "hbm_write_bytes_per_call": wr, "hbm_bytes_per_call": rd+wr,
 "algorithmic_bytes_per_call": 12e9, "traffic_over_algorithmic": (rd+wr)/12e9}, open("$OUT/c4_names_1e9_pmc.json", "w"), indent=1)
print(json.dumps({"c4 1e9 read": rd, "write": wr, "ratio": (rd+wr)/12e9, "calls": calls}))
PY
# The bench lines, AFTER the counter passes: bench.py takes its `traffic` fields from profiles/r06_*_pmc.json and refuses
# summaries measured on other kernel sources -- the ones just written are this tree's.
cd $R
for f in k1_pmc c3_pmc c4_pmc c4_names_1e9_pmc; do cp $OUT/$f.json $R/profiles/r06_$f.json; done
python bench.py 2> $OUT/bench.err | grep "^{" | tail -1 > $OUT/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench20.err | grep "^{" | tail -1 > $OUT/bench_steps20_warmup5.json
python bench.py --workload c4 2> $OUT/c4_bench.err | grep "^{" | tail -1 > $OUT/c4_bench.json
# first calls of fresh engines and clustered streams (sweeps drop their first calls): per-call times of four calls each
cd $R
{ echo "# tree_stamp: $STAMP"; echo "# tools/first_call.py <dists> <ids> 1e9 (NAMES=..., ONLY0=1): ms per call, calls 0 .. 3 of a fresh engine"
  for c in "65536 zipf lognormal,loguniform" "65536 sorted lognormal,loguniform" "65536 runs256 lognormal" "8192 sorted lognormal,loguniform" "1024 sorted lognormal"; do
    set -- $c; echo "== names $1, ids $2"; NAMES=$1 ONLY0=1 timeout 300 python tools/first_call.py $3 $2 1e9 2>&1 | grep " call " | sed -e "s/ logw_opt 0//" -e "s/'records_level2'.*'region_overflows'/'region_overflows'/" | cut -c1-200
  done; } > $OUT/first_calls.txt
# LAST: the final kernels over the value distributions, each on its own survey, every sweep process behind its own
# warm-up series (--warmup 40: K1 launches in the SAME process -- a warm-up in a process of its own left the next one cold)
cd $R
{ echo "{\"tree_stamp\": \"$STAMP\"}"
  python tools/sweep.py --warmup 40 --samples 1e9 --reps 5 2>/dev/null
  python tools/sweep.py --warmup 40 --samples 1e9 --pairs 1024 --reps 4 --dists lognormal,constant,uniform,exponential,normal,loguniform,lognormal25,kvalues2,kvalues4,kvalues8,kvalues16,bimodal,signed_wide 2>/dev/null
  python tools/sweep.py --warmup 40 --samples 1e9 --pairs 65536 --reps 4 --dists lognormal,constant,normal,kvalues2,kvalues8,bimodal,lognormal25,lognormal50,loguniform,signed_wide,thin_far_tail 2>/dev/null
  for m in 8192 4096; do python tools/sweep.py --warmup 40 --samples 1e9 --pairs $m --reps 4 --dists lognormal,normal,lognormal25,lognormal50,loguniform,signed_wide,thin_far_tail 2>/dev/null; done
  python tools/sweep.py --warmup 40 --samples 1e9 --pairs 65536 --reps 4 --ids sorted --dists lognormal 2>/dev/null
  python tools/sweep.py --warmup 40 --samples 1e9 --pairs 1024 --reps 4 --ids sorted --dists lognormal 2>/dev/null
} | cut -c1-700 > $OUT/sweep_final.jsonl
cat $OUT/c4_kernel_trace.txt | cut -c1-150
ls -la $OUT; tail -c 1500 $OUT/bench.json; grep -E "k_ingest_single" $OUT/kernel_trace.txt | cut -c1-140; cat $OUT/c3_kernel_trace.txt | cut -c1-150; python -c "
import json; j=json.load(open('$OUT/c3_pmc.json')); print({k:j[k] for k in ('hbm_read_bytes_per_call','hbm_write_bytes_per_call','traffic_over_algorithmic')})"
