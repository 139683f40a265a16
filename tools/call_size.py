#!/usr/bin/env python3
"""Device-resident mixed calls of 2^k pairs: ms per call and pairs/s per name count (where the path changes: lh_dispatch).
python tools/call_size.py [names,...] [kmin] [kmax]; OPTS=id=value,... sets lh_set_option.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, loghisto_amd
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
names = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1024,8192,65536").split(",")]
kmin, kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 14, int(sys.argv[3]) if len(sys.argv) > 3 else 28
for M in names:
    w = torch.arange(1, M + 1, dtype=torch.float64, device="cuda") ** -1.0
    nmax = 1 << kmax
    ids = torch.multinomial(w / w.sum(), nmax, replacement=True).to(torch.int32)
    data = bench.make_samples(nmax, os.environ.get("DIST", "lognormal"), 7)
    if os.environ.get("IDS") == "sorted": ids = torch.sort(ids).values.contiguous()
    if os.environ.get("IDS") == "zero": ids.zero_()
    eng = loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    for kv in os.environ.get("OPTS", "").split(","):
        if kv: eng.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
    for k in range(kmin, kmax + 1):
        n = 1 << k
        reps = 24
        for r in range(4): eng.submit_pairs_device(ids, data, n, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0 = eng.counters()
        e0.record(stream)
        for r in range(reps): eng.submit_pairs_device(ids, data, n, stream=stream)
        e1.record(stream)
        torch.cuda.synchronize()
        c1 = eng.counters()
        ms = e0.elapsed_time(e1) / reps
        path = {k2: c1[k2] - c0[k2] for k2 in ("samples_partitioned_v3", "samples_partitioned_v2", "samples_partitioned", "samples_fallback")}
        print(f"OPTS {os.environ.get('OPTS', '')} {os.environ.get('DIST', '')} {os.environ.get('IDS', '')} names {M} n 2^{k} ms_per_call {ms:.4f} Gpairs_per_s {n / ms / 1e6:.2f} {path}", flush=True)
        eng.flip().release()
    eng.close()
