"""A scratch block that cannot be had must not lose or double-count a pair, and must not fail the call.

The reference's Histogram returns nothing and never fails (metrics.go:251, 273); loss is only ever allowed at EMISSION
(metrics.go:18-23).  Up to ABI 4 a failed hipMalloc of the mixed ingest's scratch block returned LH_ENOMEM in the
middle of a call -- after the call's samples had been counted towards the interval and after earlier sub-launches had
been enqueued -- so the caller could not tell what had been ingested.  Now the sub-launch whose
block cannot be had goes through the scratch-free kernel (one global atomic per sample: exact, slower) and
lh_counters says so.  LH_OPT_FAIL_SCRATCH_ALLOCS (include/loghisto_gpu_tuning.h) makes the next N allocations fail."""
import math
import threading

import numpy as np
import pytest

from tests.conftest import thresholds_until_round_6

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .99, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def _stream(M, n, seed):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(math.log(1e5) + 2e-3 * (ids % 4096), 1.0)
    return ids, v


def _cells(ids, v):
    bins = oracle.key_to_bin(oracle.compress_many(v)).astype(np.uint64)
    return np.unique((ids.astype(np.uint64) << np.uint64(16)) | bins, return_counts=True)


def _check(snap, ids, v, M):
    """every occupied cell of every row, as sorted (name << 16 | bin, count) lists"""
    want_cells, want_counts = _cells(ids, v)
    off, keys, counts = snap.buckets_all(M)
    rows = np.repeat(np.arange(M, dtype=np.uint64), np.diff(off.astype(np.int64)))
    cells = (rows << np.uint64(16)) | oracle.key_to_bin(keys).astype(np.uint64)
    assert np.array_equal(cells, want_cells) and np.array_equal(counts.astype(np.int64), want_counts)
    got = snap.extract(PCTS, M)
    assert np.array_equal(got["count"].astype(np.int64), np.bincount(ids, minlength=M))


@pytest.mark.parametrize("M,n,opts,counter", [
    (1024, 1_500_001, {N.OPT_PART_V2_MIN_PAIRS: 1 << 17}, "samples_partitioned_v2"),    # second generation
    (65536, 2_000_000, {N.OPT_PART_V3_MIN_PAIRS: 1 << 17}, "samples_partitioned_v3"),   # third
    (300, 900_000, {N.OPT_PART_V2: 0, N.OPT_PART_MIN_PAIRS: 1 << 17}, "samples_partitioned"),   # first (no call's default since round 6)
])
def test_device_resident_call_falls_back_and_recovers(native_lib, torch_cuda, M, n, opts, counter):
    import loghisto_amd
    ids, v = _stream(M, n, M + n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        for k, val in opts.items():
            e.set_option(k, val)
        e.set_option(N.OPT_FAIL_SCRATCH_ALLOCS, 1)
        e.submit_pairs_device(d_ids, d_v)                      # must not raise
        e.sync()
        c = e.counters()
        assert c["scratch_alloc_failures"] == 1 and c["samples_fallback"] >= n - 1 and c[counter] == 0
        assert c["samples_direct"] == n and c["scratch_bytes"] == 0
        with e.flip() as snap:
            _check(snap, ids, v, M)
        # the next call gets its block: the partitioned path again, into the other epoch buffer
        e.submit_pairs_device(d_ids, d_v)
        e.sync()
        c = e.counters()
        assert c["scratch_alloc_failures"] == 1 and c[counter] >= n - 1 and c["scratch_bytes"] > 0
        with e.flip() as snap:
            _check(snap, ids, v, M)


def test_only_the_sub_launch_without_a_block_falls_back(native_lib, torch_cuda):
    """A call cut into sub-launches: the first one finds no block, the following ones allocate theirs -- every pair once."""
    import loghisto_amd
    M, n = 512, (3 << 22) + 77_001
    ids, v = _stream(M, n, 11)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_SUBLAUNCH_PAIRS, 1 << 22)
        e.set_option(N.OPT_FAIL_SCRATCH_ALLOCS, 1)
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        c = e.counters()
        assert c["samples_fallback"] == 1 << 22 and c["scratch_alloc_failures"] == 1
        assert c["samples_partitioned"] + c["samples_direct"] == n and c["samples_partitioned"] >= 2 << 22
        with e.flip() as snap:
            _check(snap, ids, v, M)


def test_a_block_that_cannot_grow(native_lib, torch_cuda):
    """The shared block exists but is too small for the next call, and neither the new block nor -- after the old one
    was given up -- a second attempt succeeds: the call runs without scratch, the engine is left without a block, and
    the call after that allocates a fresh one."""
    import loghisto_amd
    M = 1024
    small, big = _stream(M, 400_000, 1), _stream(M, 3_000_000, 2)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        thresholds_until_round_6(e)
        e.submit_pairs_device(_dev(torch_cuda, small[0]), _dev(torch_cuda, small[1]))
        e.sync()
        first = e.counters()["scratch_bytes"]
        assert first > 0
        e.set_option(N.OPT_FAIL_SCRATCH_ALLOCS, 2)
        e.submit_pairs_device(_dev(torch_cuda, big[0]), _dev(torch_cuda, big[1]))
        e.sync()
        c = e.counters()
        assert c["scratch_alloc_failures"] == 2 and c["samples_fallback"] == 3_000_000 and c["scratch_bytes"] == 0
        ids, v = np.concatenate([small[0], big[0]]), np.concatenate([small[1], big[1]])
        with e.flip() as snap:
            _check(snap, ids, v, M)
        e.submit_pairs_device(_dev(torch_cuda, big[0]), _dev(torch_cuda, big[1]))
        e.sync()
        assert e.counters()["scratch_bytes"] > first and e.counters()["scratch_alloc_failures"] == 2
        with e.flip() as snap:
            _check(snap, big[0], big[1], M)


@pytest.mark.parametrize("width", [4, 2])
def test_host_fed_lanes_keep_every_pair_and_no_lane_stays_reserved(native_lib, torch_cuda, width):
    """lh_submit_pairs* / lh_reserve_pairs* from four threads while the lanes' scratch blocks cannot be allocated: the
    half-buffers that find no block go through the direct kernel; afterwards every lane can still be reserved, committed,
    flushed and flipped."""
    import loghisto_amd
    M, per = 2000, 700_000
    parts = [_stream(M, per, 100 + t) for t in range(4)]
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=4, lane_samples=1 << 18) as e:
        e.set_option(N.OPT_LANE_SCRATCH_BLOCKS, 16)             # (the default, 0, needs no block at all since round 6)
        e.set_option(N.OPT_FAIL_SCRATCH_ALLOCS, 5)
        errs = []

        def work(t):
            try:
                ids, v = parts[t]
                ids = ids.astype(np.uint16) if width == 2 else ids
                half = per // 2
                e.submit_pairs(ids[:half], v[:half])
                e.submit_pairs_in_place(ids[half:], v[half:])
            except Exception as exc:  # noqa: BLE001
                errs.append(exc)

        th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        e.sync()
        c = e.counters()
        assert c["scratch_alloc_failures"] == 5 and c["samples_fallback"] > 0
        assert c["samples_partitioned"] + c["samples_direct"] + c["samples_small"] == 4 * per
        ids, v = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        with e.flip() as snap:
            _check(snap, ids, v, M)
        # no lane was left reserved: every one of them hands out its buffer again
        toks = []
        for _ in range(4):
            di, dv, tok = e.reserve_pairs(16, 16 if width == 2 else 32)
            di[:16] = 7
            dv[:16] = 123.0
            toks.append(tok)
        assert len(set(toks)) == 4
        for tok in toks:
            e.commit_pairs(tok, 16)
        e.sync()
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
            assert int(got["count"][7]) == 64 and int(got["count"].sum()) == 64


def test_lane_tables_that_cannot_be_had(native_lib, torch_cuda):
    """65 536 names from the host: the first lane launch needs its block AND the lanes' two table sets (third generation);
    with any of the three allocations failing the half-buffer goes through the direct kernel, and the next one tries again."""
    import loghisto_amd
    M, per = 65536, 1 << 18
    ids, v = _stream(M, 4 * per, 5)
    ids16 = ids.astype(np.uint16)
    for fail in (1, 2, 3):
        with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=per) as e:
            e.set_option(N.OPT_LANE_SCRATCH_BLOCKS, 16)
            e.set_option(N.OPT_FAIL_SCRATCH_ALLOCS, fail)
            e.submit_pairs(ids16, v)
            e.sync()
            c = e.counters()
            assert c["scratch_alloc_failures"] == fail and c["samples_fallback"] >= per, c
            assert c["samples_partitioned_v3"] >= per and c["samples_partitioned_v3"] + c["samples_fallback"] == 4 * per, c
            with e.flip() as snap:
                _check(snap, ids, v, M)
