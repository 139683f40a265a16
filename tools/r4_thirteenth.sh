# Round 4: calls above the per-launch caps (4.5e9 samples, 2.5e9 pairs); extract / merge of 65 536 wide rows (21-decade values)
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r4o}; mkdir -p $OUT
(time python -m pytest tests/test_gpu_fullsize.py -x -q -k larger_than_one_launch) 2>&1 | tail -8 | tee $OUT/pytest_huge.log
python -m pytest tests/test_gpu_k1_window.py -x -q -s 2>&1 | tail -4 | tee $OUT/pytest_k1.log
python bench.py --workload c4 --dist loguniform --steps 5 --warmup 2 --no-cpu-baseline --latency-flips 0 2> $OUT/c4_wide.err | grep "^{" | tail -1 > $OUT/c4_wide.json
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r4o/c4_wide.json").read())
print({k: j.get(k) for k in ("value", "ms_per_step")})
for k in ("ingest_ms", "merge", "extract_owned_ms", "extract_roofline", "parity"):
    print(k, json.dumps(j.get(k) or j.get("config", {}).get(k))[:600])
PY
true
