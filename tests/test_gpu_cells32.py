"""Epoch buffers of 32-bit cells (ABI 7, loghisto_amd/csrc/lh_cells.h): above 8 192 names -- or with lh_config.cell_bits = 32
-- a buffer counts in uint32 cells while its interval holds fewer than 2^32 samples and moves to a uint64 store of its own
before a submit could pass that (lh_engine.cc: widen_buffer).  The reference's cell is a *uint64 (metrics.go:278): whatever
the width in HBM, every count must be exact -- below the bound, across the move, and for a single cell beyond 2^32.

The whole GPU suite also runs on 32-bit engines (LH_TEST_CELL_BITS=32) and on engines that widen in the middle of every test
(LH_TEST_WIDEN_AT=200000): tools/round.sh cells32, profiles/r06_cells32_suite.txt."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from loghisto_amd import _native as N
from tests.test_gpu_part3 import PCTS, _dev, _ids, _values, check

pytestmark = pytest.mark.gpu


def _cell_bytes(e):
    return int(N.lib().lh_cell_bytes(e._h))


def test_default_width_follows_the_name_count(native_lib, torch_cuda):
    import loghisto_amd
    for M, bits, want in [(1024, 0, 8), (8192, 0, 8), (8193, 0, 4), (8193, 64, 8), (300, 32, 4)]:
        with loghisto_amd.Engine(max_metrics=M, num_lanes=1, lane_samples=1 << 12, max_counters=0, cell_bits=bits) as e:
            assert _cell_bytes(e) == want, (M, bits)
            c = e.counters()
            assert c["store_bytes"] == 2 * M * N.lib().lh_row_stride() * want and c["widenings"] == 0


@pytest.mark.parametrize("M,n,kind", [(300, 700_001, "lognormal"), (1024, 2_200_000, "signed"), (5000, 2_500_000, "edge"),
                                      (30, 400_000, "kvalues8"), (20000, 3_500_000, "lognormal"), (65536, 3_200_000, "sigma25")])
def test_every_path_is_exact_on_32_bit_cells_and_across_the_move(native_lib, torch_cuda, M, n, kind):
    """Three intervals on one engine: all on the narrow store; one that moves to uint64 cells between its calls (the bound set
    to a fraction of the interval); one on the narrow store again -- the widening left it clean, and the wide store is reused
    by the fourth."""
    import loghisto_amd
    rng = np.random.default_rng(M + n)
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, kind, ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    h = n // 2 & ~1
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=1 << 16, cell_bits=32) as e:
        assert _cell_bytes(e) == 4
        for interval, widen_at in enumerate([0xffffffff, n + 10, 0xffffffff, n // 3, n + 10]):
            e.set_option(N.OPT_WIDEN_AT_SAMPLES, widen_at)
            before = e.counters()["widenings"]
            e.submit_pairs_device(d_ids[:h], d_v[:h])
            e.submit_pairs(ids[h:h + 5000], v[h:h + 5000])             # host-fed lanes add to the same buffer
            e.submit(int(ids[0]), v[:3000])
            e.submit_pairs_device(d_ids[h:], d_v[h:])
            with e.flip() as snap:
                moved = e.counters()["widenings"] - before
                assert moved == (1 if widen_at < 2 * n else 0), (interval, moved)
                _, _, cb = snap.device_cells()
                assert cb == (8 if moved else 4)
                all_ids = np.concatenate([ids, ids[h:h + 5000], np.full(3000, ids[0], np.uint32)])
                all_v = np.concatenate([v, v[h:h + 5000], v[:3000]])
                check(snap, all_ids, all_v, M, snap.extract(PCTS, M))
                got_one = snap.buckets(int(ids[0]))                     # lh_buckets reads the row at the store's width
                want_one = oracle.histogram_dense(all_v[all_ids == ids[0]])
                assert np.array_equal(snap.dense_row(int(ids[0])), want_one) and int(got_one[1].sum()) == int(want_one.sum())
        c = e.counters()
        # two buffers alternate.  Buffer 0 (intervals 0, 2, 4) moved once: it holds both stores.  Buffer 1 (intervals 1, 3) moved in
        # two consecutive intervals: it stays on its wide store and has given the narrow one back (next test).
        assert c["store_bytes"] == M * N.lib().lh_row_stride() * (4 + 8 + 8)


def test_a_buffer_that_keeps_passing_the_bound_stays_wide_and_comes_back(native_lib, torch_cuda):
    """At rates where every interval passes 2^32 samples the narrow store only ever holds an interval's first milliseconds: after
    two such intervals in a row a buffer stays on its wide store and frees the narrow one (the engine then holds what a 64-bit
    engine holds, not 1.5 x that); sixteen intervals in a row below 2^31 samples bring the narrow store back.  Exact throughout."""
    import loghisto_amd
    M, n = 9000, 300_000
    rng = np.random.default_rng(11)
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, "lognormal", ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    unit = M * N.lib().lh_row_stride()
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16, cell_bits=0) as e:
        def interval(check_it=False):
            e.submit_pairs_device(d_ids, d_v)
            e.submit_pairs_device(d_ids, d_v)
            with e.flip() as snap:
                cb = snap.device_cells()[2]
                if check_it:
                    check(snap, np.concatenate([ids, ids]), np.concatenate([v, v]), M, snap.extract(PCTS, M))
            return cb
        assert e.counters()["store_bytes"] == unit * 8                                   # 4 + 4
        e.set_option(N.OPT_WIDEN_AT_SAMPLES, n + 1)                                      # every interval passes the bound
        assert [interval(i == 3) for i in range(4)] == [8, 8, 8, 8]                      # (both buffers, twice each)
        assert e.counters()["widenings"] == 4 and e.counters()["store_bytes"] == unit * 16   # 8 + 8: the narrow stores are gone
        assert [interval(i == 1) for i in range(4)] == [8, 8, 8, 8] and e.counters()["widenings"] == 4   # nothing left to move
        e.set_option(N.OPT_WIDEN_AT_SAMPLES, 0xffffffff)
        got = [interval(i in (29, 33)) for i in range(36)]                               # sixteen quiet intervals per buffer
        assert got[:32] == [8] * 32 and got[32:] == [4] * 4, got
        assert e.counters()["store_bytes"] == unit * 8 and e.counters()["widenings"] == 4


def test_one_cell_beyond_two_to_the_32(native_lib, torch_cuda):
    """A constant stream on one name: 17 x 2^28 samples into ONE cell of a 32-bit engine (K1's rate gets there in 6 ms).  The
    buffer moves to uint64 cells before the sixteenth call could wrap the cell; the count is exact."""
    import loghisto_amd
    torch = torch_cuda
    n = 1 << 28
    d_v = torch.full((n,), 123.0, dtype=torch.float64, device="cuda")
    with loghisto_amd.Engine(max_metrics=9000, num_buffers=2, num_lanes=1, lane_samples=1 << 12, max_counters=0, cell_bits=0) as e:
        assert _cell_bytes(e) == 4
        for _ in range(17):
            e.submit_device(4242, d_v)
        e.submit_device(7, d_v[:1000])
        with e.flip() as snap:
            assert e.counters()["widenings"] == 1
            got = snap.extract([0.5], 1, 4242)
            assert int(got["count"][0]) == 17 * n and 17 * n > 1 << 32
            keys, counts = snap.buckets(4242)
            assert len(keys) == 1 and int(counts[0]) == 17 * n and int(keys[0]) == int(oracle.compress_many(np.array([123.0]))[0])
            assert int(snap.extract([0.5], 1, 7)["count"][0]) == 1000
        # the next interval starts narrow again and is exact
        e.submit_device(4242, d_v[:5000])
        with e.flip() as snap:
            _, _, cb = snap.device_cells()
            assert cb == 4 and int(snap.extract([0.5], 1, 4242)["count"][0]) == 5000


def test_row_view_and_cell_view(native_lib, torch_cuda):
    """lh_snapshot_rows keeps handing out uint64 rows (the snapshot moves to its wide store first); lh_snapshot_cells hands out
    the cells as they are."""
    import loghisto_amd
    from loghisto_amd import merge
    torch = torch_cuda
    M, n = 9000, 600_000
    rng = np.random.default_rng(5)
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, "lognormal", ids, n)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16, cell_bits=0) as e:
        e.submit_pairs_device(_dev(torch, ids), _dev(torch, v))
        with e.flip() as snap:
            ptr, nrows, cb = snap.device_cells()
            assert (nrows, cb) == (M, 4)
            hot = int(np.bincount(ids).argmax())
            want = oracle.histogram_dense(v[ids == hot])
            stride = snap.row_stride()
            flat = torch.as_tensor(merge._DeviceArray(ptr + hot * stride * 4, (N.NKEYS,), "<i4"), device="cuda")
            torch.cuda.synchronize()
            e.sync()
            assert np.array_equal(flat.cpu().numpy().view(np.uint32).astype(np.uint64), want)
            rows, ranges = merge.snapshot_tensors(snap, M)             # lh_snapshot_rows: the wide store from here on
            assert snap.device_cells()[2] == 8 and e.counters()["widenings"] == 1
            torch.cuda.current_stream().wait_stream(torch.cuda.ExternalStream(snap.stream()))
            assert np.array_equal(rows[hot].cpu().numpy().view(np.uint64), want)
            rows[hot, 40000] += (1 << 33)                              # a caller's own merge may exceed 2^32 in the view
            torch.cuda.synchronize()
            snap.mark_dirty(hot, 1, 40000, 40000)
            assert int(snap.extract([0.5], 1, hot)["count"][0]) == int(want.sum()) + (1 << 33)


def test_merge_on_uint64_words_widens_the_snapshot_first(native_lib, torch_cuda):
    """lh_snapshot_merge picks uint64 wire words when nranks x the ranks' sample counts may pass 2^32 (here: the count is
    unknown after lh_snapshot_mark_dirty); the merged sums may then not fit a narrow store's cells either."""
    import loghisto_amd
    torch = torch_cuda
    rccl_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(rccl_path):
        rccl_path = "/opt/rocm/lib/librccl.so"
    assert N.lib().lh_set_rccl_library(rccl_path.encode()) in (0, 7)
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
        M, n = 8200, 500_000
        rng = np.random.default_rng(6)
        ids = _ids(rng, M, n, 1.0)
        v = _values(rng, "signed", ids, n)
        with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16, cell_bits=0) as e:
            for widen in (False, True):
                e.submit_pairs_device(_dev(torch, ids), _dev(torch, v))
                with e.flip() as snap:
                    if widen:
                        snap.mark_dirty(0, 1, 100, 100)     # "cells of the caller's own": the sample count is unknown from here
                    assert snap.merge_rccl(comm.value, 1, 0, M, plan="reduce_scatter") == (0, M)
                    info = snap.merge_info()
                    assert info["cell_bytes"] == (8 if widen else 4) and snap.device_cells()[2] == (8 if widen else 4)
                    check(snap, ids, v, M, snap.extract(PCTS, M))
            assert e.counters()["widenings"] == 1
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_concurrent_submitters_across_the_move(native_lib, torch_cuda):
    """Eight host threads submit through their lanes while a ninth submits device-resident pairs; the bound falls in the middle of
    the interval, so one of them moves the buffer while the others are enqueueing (cells_mu: a step holds the cells shared from
    reading the pointer until its launches are enqueued, the move holds them unique behind a device synchronisation)."""
    import threading
    import loghisto_amd
    M, per, T = 12000, 400_000, 8
    rng = np.random.default_rng(21)
    parts = []
    for t in range(T + 1):
        ids = _ids(rng, M, per, 1.0)
        parts.append((ids, _values(rng, "lognormal", ids, per)))
    d_ids, d_v = _dev(torch_cuda, parts[T][0]), _dev(torch_cuda, parts[T][1])
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=4, lane_samples=1 << 15, cell_bits=0) as e:
        for interval in range(3):
            e.set_option(N.OPT_WIDEN_AT_SAMPLES, [per * 4, 0xffffffff, per * 2][interval])
            errs = []

            def host(t):
                try:
                    ids, v = parts[t]
                    for k in range(0, per, 50_000):
                        e.submit_pairs(ids[k:k + 50_000], v[k:k + 50_000])
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex)

            def dev():
                try:
                    for k in range(0, per, 100_000):
                        e.submit_pairs_device(d_ids[k:k + 100_000], d_v[k:k + 100_000])
                except Exception as ex:  # noqa: BLE001
                    errs.append(ex)

            th = [threading.Thread(target=host, args=(t,)) for t in range(T)] + [threading.Thread(target=dev)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            assert not errs, errs
            with e.flip() as snap:
                assert snap.device_cells()[2] == (4 if interval == 1 else 8)
                check(snap, np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), M, snap.extract(PCTS, M))
        assert e.counters()["widenings"] == 2
