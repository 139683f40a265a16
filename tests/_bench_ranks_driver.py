"""bench.py's N-rank paths with N ranks as THREADS on the one reachable GPU (test infrastructure for
test_gpu_bench_ranks.py).

The driver runs `bench.py --gpus N` on an 8-GPU node at round end; here one GPU is reachable.  The N-rank code of the
headline (run_c2: every rank's slice of the ONE metric, all-reduce of the row at the flip, merged row against the oracle
over the concatenated slices) and of config 4 (run_c4: owned-block bookkeeping, the all-gathered bounds, per-name
counts, the probe rows compared cell by cell after the merge, the merge report) is exercised with
tests/cpp/rccl_stub.cc as the RCCL of lh_snapshot_merge and a thread-rendezvous stand-in for the few
torch.distributed calls they make.
usage: python _bench_ranks_driver.py N names slice        (config 4)
       python _bench_ranks_driver.py N c2 samples         (the headline)
       python _bench_ranks_driver.py N job samples        (bench.run_job: what main() runs -- the headline, then config 4 with
                                                           its one-rank reference under secondary.c4)"""
import ctypes as C
import json
import os
import sys
import threading
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class ThreadDist:
    """all_reduce / all_gather / barrier / broadcast between the threads of one process (tensors on one device)."""

    class ReduceOp:
        SUM, MAX = "sum", "max"

    def __init__(self, n):
        self.n, self.bar, self.slots, self.tls = n, threading.Barrier(n), [None] * n, threading.local()

    def bind(self, rank):
        self.tls.rank = rank

    def barrier(self):
        self.bar.wait()

    def _exchange(self, t):
        import torch
        torch.cuda.current_stream().synchronize()
        self.slots[self.tls.rank] = t.clone()
        self.bar.wait()
        got = [x.clone() for x in self.slots]
        self.bar.wait()
        return got

    def all_reduce(self, t, op="sum"):
        import torch
        got = torch.stack(self._exchange(t))
        t.copy_(got.max(0).values if op == "max" else got.sum(0))

    def all_gather(self, out, t):
        for o, g in zip(out, self._exchange(t)):
            o.copy_(g)

    def broadcast(self, t, src):
        t.copy_(self._exchange(t)[src])


def main():
    c2, job = sys.argv[2] == "c2", sys.argv[2] == "job"
    nranks, names, slice_pairs = int(sys.argv[1]), (1 if c2 else 9000 if job else int(sys.argv[2])), float(sys.argv[3])
    import torch
    import bench
    import loghisto_amd as la
    from loghisto_amd import _native as N
    stub_path = os.path.join(ROOT, "loghisto_amd", "build", "librccl_stub.so")
    N.check(N.lib().lh_set_rccl_library(stub_path.encode()), "lh_set_rccl_library")
    stub = C.CDLL(stub_path)
    comms = (C.c_void_p * nranks)()
    assert stub.stub_comm_create(nranks, comms) == 0
    args = types.SimpleNamespace(names=names, c4_slice=slice_pairs, no_parity=False, samples=slice_pairs, dist="lognormal",
                                 steps=2, warmup=1, latency_flips=3, no_cpu_baseline=True, no_secondary=False)
    ref_comms = []
    if job:                           # one-rank communicators (stub) for config 4's one-rank reference on every rank
        for _ in range(nranks):
            c1 = (C.c_void_p * 1)()
            assert stub.stub_comm_create(1, c1) == 0
            ref_comms.append(c1)
    dist = ThreadDist(nranks)
    results, errors = [None] * nranks, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            dist.bind(r)
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            if job:
                results[r] = bench.run_job(args, la, stream, r, nranks, dist, comms[r], "c-abi: lh_snapshot_merge -> RCCL (stub)",
                                           "", "c2", ref_comm=ref_comms[r][0])
            elif c2:
                results[r] = bench.run_c2(args, la, stream, r, nranks, dist if nranks > 1 else None,
                                          comms[r] if nranks > 1 else 0, "c-abi: lh_snapshot_merge -> RCCL (stub)")
            else:
                results[r] = bench.run_c4(args, la, stream, r, nranks, dist, steps=2, warmup=1, comm_override=comms[r])
        except BaseException as exc:  # noqa: BLE001
            errors.append((r, repr(exc)))
            dist.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errors, errors
    res = results[0]
    if job:
        res, clean = res
        assert all(c for _, c in results), [c for _, c in results]
        c4 = res["secondary"]["c4"]
        print(json.dumps({"ok": True, "ranks": nranks, "parity": res["parity"], "config": res["config"],
                          "c4_parity": c4["parity"], "c4_ranks": c4["config"]["ranks"], "c4_merge": c4["config"]["merge"],
                          "c4_one_rank_reference": c4["one_rank_reference"], "c4_efficiency": c4.get("efficiency_vs_one_rank"),
                          "values": [r[0]["value"] for r in results]}))
        return
    if c2:
        print(json.dumps({"ok": True, "ranks": nranks, "parity": res["parity"], "merge": res.get("merge"),
                          "config": res["config"], "values": [r["value"] for r in results],
                          "samples_per_step": slice_pairs * nranks}))
        return
    print(json.dumps({"ok": True, "ranks": nranks, "parity": res["parity"], "merge": res["merge"],
                      "owned_rows": [r["config"]["owned_rows"] for r in results]}))


if __name__ == "__main__":
    main()
