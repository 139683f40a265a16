"""bench.run_c4 with N ranks as THREADS on the one reachable GPU (test infrastructure for test_gpu_bench_ranks.py).

The driver runs `bench.py --gpus N` on an 8-GPU node at round end; here one GPU is reachable.  The N-rank code of
run_c4 (owned-block bookkeeping, the all-gathered bounds, per-name counts, the probe rows compared cell by cell after
the merge, the merge report) is exercised with tests/cpp/rccl_stub.cc as the RCCL of lh_snapshot_merge and a
thread-rendezvous stand-in for the few torch.distributed calls run_c4 makes.  usage: python _bench_ranks_driver.py N names slice"""
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
    nranks, names, slice_pairs = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    import torch
    import bench
    import loghisto_amd as la
    from loghisto_amd import _native as N
    stub_path = os.path.join(ROOT, "loghisto_amd", "build", "librccl_stub.so")
    N.check(N.lib().lh_set_rccl_library(stub_path.encode()), "lh_set_rccl_library")
    stub = C.CDLL(stub_path)
    comms = (C.c_void_p * nranks)()
    assert stub.stub_comm_create(nranks, comms) == 0
    args = types.SimpleNamespace(names=names, c4_slice=slice_pairs, no_parity=False)
    dist = ThreadDist(nranks)
    results, errors = [None] * nranks, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            dist.bind(r)
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            results[r] = bench.run_c4(args, la, stream, r, nranks, dist, steps=2, warmup=1, comm_override=comms[r])
        except BaseException as exc:  # noqa: BLE001
            errors.append((r, repr(exc)))
            dist.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errors, errors
    res = results[0]
    print(json.dumps({"ok": True, "ranks": nranks, "parity": res["parity"], "merge": res["merge"],
                      "owned_rows": [r["config"]["owned_rows"] for r in results]}))


if __name__ == "__main__":
    main()
