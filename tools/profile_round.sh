# Produces the per-round evidence under gpurun_out/round/ (copy into profiles/ afterwards):
#   bench.json            python bench.py (default flags)
#   kernel_trace.txt      rocprofv3 --kernel-trace of the same command (no cpu baseline), summarised
#   k1_pmc.json           FETCH_SIZE / WRITE_SIZE passes for k_ingest_single, corrected per MI355X_MICROARCH.md
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round; mkdir -p $OUT; cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp; export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency-flips 50"
rocprofv3 --kernel-trace -d /tmp/pr_trace -o t -- $CMD > $OUT/bench_under_trace.json 2>/dev/null
{ echo "# rocprofv3 --kernel-trace -- $CMD"; python $R/profiles/summarize_rocpd.py stats /tmp/pr_trace/t_results.db | cut -c1-170
  echo; echo "## full-size launches only (duration >= 0.5 ms)"; python $R/profiles/summarize_rocpd.py stats /tmp/pr_trace/t_results.db --min-ns 500000 | cut -c1-170; } > $OUT/kernel_trace.txt
CMD2="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --latency-flips 0"
rocprofv3 --pmc FETCH_SIZE -d /tmp/pr_fetch -o t -- $CMD2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pr_write -o t -- $CMD2 > /dev/null 2>&1
python - <<PY
import json, subprocess
R="$R"
def pmc(db): return json.loads(subprocess.check_output(["python", R+"/profiles/summarize_rocpd.py", "pmc", db, "k_ingest_single", "--min-ns", "500000"]))
f=pmc("/tmp/pr_fetch/t_results.db")["counters"]["FETCH_SIZE"]; w=pmc("/tmp/pr_write/t_results.db")["counters"]["WRITE_SIZE"]
out={"kernel":"lh::k_ingest_single","workload":"1e9 float64 samples, lognormal(ln 1e5, 1), one metric",
 "commands":["rocprofv3 --pmc FETCH_SIZE -- $CMD2","rocprofv3 --pmc WRITE_SIZE -- $CMD2"],
 "FETCH_SIZE_KiB_per_launch":f["avg"],"WRITE_SIZE_KiB_per_launch":w["avg"],"launches":f["launches"],
 "corrections":"FETCH_SIZE is in KiB and on gfx950 counts the 128-B requests of a 16-B/lane coalesced stream as 64 B: read bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md 'HBM'). WRITE_SIZE*1024, uncalibrated (flush atomics), 3 orders of magnitude below the read side.",
 "hbm_read_bytes_per_launch":f["avg"]*2048,"hbm_write_bytes_per_launch":w["avg"]*1024,"algorithmic_bytes_per_launch":8e9,
 "read_over_algorithmic":f["avg"]*2048/8e9,
 "avg_duration_us_under_pmc":{"FETCH_SIZE pass":f["avg_duration_us_profiled"],"WRITE_SIZE pass":w["avg_duration_us_profiled"]}}
json.dump(out, open("$OUT/k1_pmc.json","w"), indent=1)
PY
ls -la $OUT; tail -c 1200 $OUT/bench.json; grep -E "k_ingest_single" $OUT/kernel_trace.txt | cut -c1-140
