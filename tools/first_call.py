#!/usr/bin/env python3
"""The FIRST mixed call of a fresh engine at 65 536 names, per distribution and window-width setting (sweeps drop their
first calls and never see it): python tools/first_call.py loguniform,lognormal  ->  profiles/r06_first_call.txt.  GPU only."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, loghisto_amd
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
M = int(os.environ.get('NAMES', '65536'))
IDS = sys.argv[2] if len(sys.argv) > 2 else "zipf"   # sorted | uniform | drift | runs: see below
NS = [int(float(x)) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [int(1e8), int(1e9)]
w = torch.arange(1, M + 1, dtype=torch.float64, device="cuda") ** -1.0
for dist in sys.argv[1].split(","):
  for n in NS:
    ids = torch.multinomial(w / w.sum(), n, replacement=True).to(torch.int32)
    if IDS == "sorted": ids = torch.sort(ids).values.contiguous()
    if IDS == "uniform": ids = torch.randint(0, M, (n,), device="cuda", dtype=torch.int32)
    if IDS == "drift": ids[n // 2:] = (M - 1) - ids[n // 2:]
    if IDS.startswith("runs"):  # runsR: R pairs of one name in a row
        R = int(IDS[4:] or 4096)
        ids = ids[::R].repeat_interleave(R)[:n].contiguous()
    data = bench.make_samples(n, dist, 7)
    for logw in ((0, 10, 13) if M > 8192 and not os.environ.get("ONLY0") else (0,)):
        eng = loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
        if logw: eng.set_option(14, logw)  # LH_OPT_PART_V3_LOG_W
        for kv in os.environ.get("OPTS", "").split(","):
            if kv: eng.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        prev = eng.counters()
        for r in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); eng.submit_pairs_device(ids, data, n, stream=stream); e1.record(stream)
            torch.cuda.synchronize()
            c = eng.counters()
            print(dist, n, "logw_opt", logw, "call", r, "ms %.2f" % e0.elapsed_time(e1), "logw", c["window_log2"],
                  {k: c[k] - prev[k] for k in ("records_level1", "records_level2", "level2_overflows", "reduce_window_misses", "region_overflows", "samples_partitioned_v3", "regions_disabled")}, flush=True)
            prev = c
            s = eng.flip(); s.extract([0.5], M); s.release()
        eng.close()
    del ids, data
