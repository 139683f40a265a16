"""Tuning helper: 3 launches of the mixed ingest over 1e9 samples / 1024 Zipf names (no checks:
LH_DEBUG_FLAGS ablations produce wrong results by design)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, loghisto_amd
n, M = int(float(os.environ.get("ABL_N", "1e9"))), 1024
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
eng = loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
data = bench.make_samples(n, os.environ.get("ABL_DIST", "lognormal"), 7)
w = 1.0 / torch.arange(1, M + 1, dtype=torch.float64, device="cuda")
ids = torch.multinomial(w / w.sum(), n, replacement=True).to(torch.int32)
for _ in range(3):
    eng.submit_pairs_device(ids, data, n, stream=s)
    torch.cuda.synchronize()
    eng.flip().release()
torch.cuda.synchronize()
