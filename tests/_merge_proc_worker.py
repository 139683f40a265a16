"""One rank of lh_snapshot_merge as a PROCESS of its own (run by tests/test_gpu_merge_procs.py, two or more at a time on
the one GPU): buckets its slice of the seeded stream for all names, flips, joins the other ranks through the
process-shared mode of tests/cpp/rccl_stub.cc and calls the C-ABI merge; writes what it ends up owning.

usage: python tests/_merge_proc_worker.py RANK NRANKS NROWS PLAN SHM_NAME OUT_DIR"""
import ctypes as C
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stream(M, nranks):
    """The whole stream (the parent checks against the oracle run on it); rank r owns [n r / N, n (r + 1) / N)."""
    rng = np.random.default_rng(7 * nranks + M)
    n = 300_000
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(math.log(1e5) + 0.002 * ids, 1.0) * np.where(rng.random(n) < 0.05, -1.0, 1.0)
    return ids, v


def main():
    rank, nranks, M, plan, shm, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    import loghisto_amd
    from loghisto_amd import _native as N
    stub_path = os.path.join(ROOT, "loghisto_amd", "build", "librccl_stub.so")
    N.check(N.lib().lh_set_rccl_library(stub_path.encode()), "lh_set_rccl_library")
    stub = C.CDLL(stub_path)
    stub.stub_comm_create_shm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    stub.stub_comm_destroy_shm.argtypes = [C.c_char_p, C.c_void_p]
    comm = C.c_void_p(0)
    rc = stub.stub_comm_create_shm(shm.encode(), nranks, rank, 64 << 20, C.byref(comm))
    assert rc == 0, f"stub_comm_create_shm: {rc}"
    ids, v = stream(M, nranks)
    n = ids.size
    lo, hi = n * rank // nranks, n * (rank + 1) // nranks
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit_pairs(ids[lo:hi], v[lo:hi])
        snap = e.flip()
        first, last = snap.merge_rccl(comm.value, nranks, rank, M, plan=plan)
        info = snap.merge_info()
        off, keys, counts = snap.buckets_all(last - first, first=first)
        st = snap.extract([0.5, 0.99], last - first, first=first)
        snap.release()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), first=first, last=last, off=off, keys=keys, counts=counts,
             count=st["count"], pkeys=st["pkeys"], info=json.dumps(info))
    stub.stub_comm_destroy_shm(shm.encode(), comm)


if __name__ == "__main__":
    main()
