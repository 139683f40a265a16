"""GPU side of the multi-GPU merge plumbing.  Only one GPU is reachable here, so this
drives loghisto_amd.merge.merge_snapshot through a single-rank RCCL group: it checks
the zero-copy aliasing of the snapshot's HBM rows as torch tensors, the stream
hand-off between torch's stream and the snapshot's extract stream, and that
extract after the merge sees the merged cells.  The world-size-2 arithmetic is
covered on CPU by tests/test_merge_gloo.py."""
import math
import os
import socket

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_merge_snapshot_single_rank_rccl(native_lib, torch_cuda):
    torch = torch_cuda
    import torch.distributed as dist
    import loghisto_amd
    from loghisto_amd import merge

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        M = 6
        rng = np.random.default_rng(8)
        n = 500_000
        ids = rng.integers(0, M, n).astype(np.uint32)
        v = rng.lognormal(math.log(1e5), 1.0, n)
        want = oracle.histogram_pairs(ids, v, M)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
                eng.submit_pairs_device(torch.from_numpy(ids.view(np.int32)).cuda(), torch.from_numpy(v).cuda(),
                                        stream=stream)
                snap = eng.flip()
                rows, ranges = merge.snapshot_tensors(snap, M)
                for plan in ("allreduce", "reduce_scatter"):
                    first, last = merge.merge_snapshot(snap, M, plan=plan)
                    assert (first, last) == (0, M)
                # the aliased tensors ARE the snapshot's memory
                torch.cuda.synchronize()
                assert np.array_equal(rows.cpu().numpy().view(np.uint64), want)
                r = ranges.cpu().numpy()
                for m in range(M):
                    nz = np.nonzero(want[m])[0]
                    assert (r[m][0], r[m][1]) == (nz[0], nz[-1])
                # a cell injected through the alias (as a peer's contribution would be) is seen by extract
                rows[2, 40000] += 5
                snap.mark_dirty(2, 1, 40000, 40000)
                merge.merge_snapshot(snap, M, plan="allreduce")
                got = snap.extract([1.0], M)
                assert int(got["count"][2]) == int(want[2].sum()) + 5
                assert int(got["pkeys"][2][0]) == 40000 - 32768
                snap.release()
    finally:
        dist.destroy_process_group()


def test_native_rccl_merge_through_the_c_abi(native_lib, torch_cuda):
    """K4 without torch.distributed: lh_snapshot_merge drives RCCL directly (what a cgo caller would use).
    One GPU is reachable, so the communicator has a single rank; that still runs the whole path --
    range merge through the bit-flipped MIN all-reduce, window pack / collective / unpack for both plans --
    and must leave the cells exactly as they were."""
    import ctypes as C
    torch = torch_cuda
    import loghisto_amd
    from loghisto_amd import _native

    rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(rccl_path):
        rccl_path = "/opt/rocm/lib/librccl.so"
    rc = _native.lib().lh_set_rccl_library(rccl_path.encode())
    assert rc in (0, 7)          # 7: already resolved by an earlier test in this process
    rccl = C.CDLL(rccl_path)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p(0)
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        M = 5
        rng = np.random.default_rng(12)
        n = 400_000
        ids = rng.integers(0, M, n).astype(np.uint32)
        ids[ids == 3] = 1                          # row 3 stays empty
        v = rng.normal(0, 1e4, n)                  # signed: window spans both sides of key 0
        want = oracle.histogram_pairs(ids, v, M)
        with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
            eng.submit_pairs(ids, v)
            snap = eng.flip()
            assert snap.merge_rccl(comm.value, 1, 0, M, plan="allreduce") == (0, M)
            assert snap.merge_rccl(comm.value, 1, 0, M, plan="reduce_scatter") == (0, M)
            assert snap.merge_rccl(comm.value, 1, 0, 1, plan="allreduce") == (0, 1)   # single-row in-place form
            got = snap.extract([0.0, 0.5, 1.0], M)
            for m in range(M):
                assert np.array_equal(snap.dense_row(m), want[m]), m
                ref = oracle.process_dense(want[m], [0.0, 0.5, 1.0])
                assert int(got["count"][m]) == ref["count"]
                if ref["count"]:
                    assert np.array_equal(got["pkeys"][m], ref["pkeys"])
            snap.release()
            # an empty snapshot: every rank sees an empty window and returns without a collective
            snap = eng.flip()
            assert snap.merge_rccl(comm.value, 1, 0, M, plan="reduce_scatter") == (0, M)
            assert snap.extract([0.5], M)["count"].sum() == 0
            snap.release()
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("nranks,nrows,plan,outliers,narrow", [
    (2, 5, "allreduce", 0, 1),
    (2, 5, "reduce_scatter", 0, 1),     # ragged, one row empty everywhere; name 1's single cell needs 32 bits
    (4, 37, "reduce_scatter", 1, 1),    # outliers in the first and the last row: those two rows are blocks of their own
    (8, 64, "reduce_scatter", 1, 1),    # config 4's rank count; name 1's single cell needs 16 bits
    (8, 5, "reduce_scatter", 0, 1),     # more ranks than rows: some ranks own nothing
    (8, 512, "reduce_scatter", 0, 1),   # Zipf names ranked by id: equal-word blocks keep the padding under 1.3 x
    (4, 64, "allreduce", 1, 1),
    (3, 130, "reduce_scatter", 1, 1),   # odd window widths against 2- and 4-cell words
    (4, 64, "reduce_scatter", 0, 0),    # LH_OPT_MERGE_NARROW_CELLS = 0 on every rank: every cell a whole word
    (4, 64, "reduce_scatter", 0, 2),    # ... on ONE rank: the all-reduced bound makes every rank send whole words
    (4, 64, "reduce_scatter", 1, 3),    # uint64 wire words (one rank's sample count unknown): the sums may pass 2^32
    (3, 37, "allreduce", 0, 4),         # ... on engines of 32-bit cells: every rank's snapshot moves to uint64 cells first
])
def test_c_abi_merge_with_n_ranks_on_one_gpu(native_lib, torch_cuda, nranks, nrows, plan, outliers, narrow):
    """lh_snapshot_merge beyond one rank.  N engines on the one reachable GPU play N ranks;
    tests/cpp/rccl_stub.cc stands in for RCCL (thread rendezvous + host reduction with RCCL's reduce-scatter
    placement).  Runs in a subprocess because the RCCL library of a process can be chosen only once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if narrow == 4:  # (the test wrapper's knob, loghisto_amd/engine.py: the driver's engines count in uint32 cells)
        narrow, env["LH_TEST_CELL_BITS"] = 3, "32"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_stub_merge_driver.py"), str(nranks), str(nrows),
                        plan, str(outliers), str(narrow)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=env)
    print(r.stdout)
    assert r.returncode == 0, r.stdout
    import json
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"]
    if outliers:
        # two rows are ~47 000 and ~17 000 cells wide; the others ~1 000: per-row windows keep the exchange small
        assert res["widest_row"] > 20_000
        assert res["packed_cells"] < 0.25 * nrows * res["widest_row"]
