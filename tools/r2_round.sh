# Round-2 evidence in one call: GPU tests, the default bench line (headline + secondary), latency at 1 / 1 024 / 65 536
# names.  Results under gpurun_out/$1.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r2j}; mkdir -p $OUT; cd $R
(time timeout 1200 python -m pytest tests -m gpu -x -q --durations=6) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
(time timeout 900 python bench.py) > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.err; python - <<PY
import json
try:
    j=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("parity"), j["extract_latency_us"])
    for k,v in j.get("secondary",{}).items():
        print(k, {a:v.get(a) for a in ("value","ms_per_step","failed","skipped")}, (v.get("roofline") or {}).get("frac"), v.get("parity"), v.get("extract_latency_us"), v.get("merge"))
except Exception as e: print("bench parse failed", e)
PY
: > $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 3000 1048576 1 >> $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 1000 4194304 1024 >> $OUT/latency.jsonl
timeout 300 loghisto_amd/build/latency 300 4194304 65536 >> $OUT/latency.jsonl
timeout 300 loghisto_amd/build/latency 300 4194304 65536 1 >> $OUT/latency.jsonl
timeout 120 loghisto_amd/build/latency 1000 4194304 1024 1 >> $OUT/latency.jsonl
cat $OUT/latency.jsonl
