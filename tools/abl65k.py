import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, loghisto_amd
n, M = int(1e9), 65536
torch.cuda.set_device(0)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
eng = loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
data = bench.make_samples(n, "lognormal", 7)
w = 1.0 / torch.arange(1, M + 1, dtype=torch.float64, device="cuda")
ids = torch.multinomial(w / w.sum(), n, replacement=True).to(torch.int32)
for _ in range(3):
    eng.submit_pairs_device(ids, data, n, stream=s)
    torch.cuda.synchronize()
    eng.flip().release()
torch.cuda.synchronize()
