"""GPU parity tests: the HIP path, called through the C ABI, against the oracle.

Bar: bucket counts bit-exact, `_count` exact, every
percentile value bit-identical to the oracle's decompress(k), `_sum`/`_avg`
within relative 1e-12 of the oracle's ascending-key sum (the reference's own sum
order is Go map order, i.e. not reproducible -- SURVEY.md 7.4).
"""
import math
import threading

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

SUM_RTOL = 1e-12
PCTS = [0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0]  # metrics.go:145-155


@pytest.fixture(scope="module")
def la(native_lib, torch_cuda):
    import loghisto_amd
    return loghisto_amd


@pytest.fixture()
def engine(la):
    e = la.Engine(max_metrics=16, num_buffers=2, num_lanes=2, lane_samples=1 << 16)
    yield e
    e.close()


def dev(torch, a, dtype=None):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)  # same bits; torch's uint32 support is partial
    t = torch.from_numpy(a)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def dists(n, seed):
    """SURVEY.md 8(d) synthetic inputs."""
    rng = np.random.default_rng(seed)
    return {
        "lognormal_s1": rng.lognormal(math.log(1e5), 1.0, n),
        "constant": np.full(n, 123.0),
        "uniform": rng.uniform(0, 1e9, n),
        "exponential": rng.exponential(1e6, n),
        "normal_signed": rng.normal(0, 1e3, n),
        "loguniform": 10.0 ** rng.uniform(-3, 18, n),
        "lognormal_s2.5": rng.lognormal(math.log(1e5), 2.5, n),
        "tiny": rng.uniform(-0.6, 0.6, n),
        # few-valued streams (quantised timers, status codes, queue depths -- what TimerToken.Stop produces,
        # metrics.go:242-246): k distinct buckets, every wave holds 64 / k same-address LDS atomics per value
        "kvalues2": 1e3 * 1.5 ** rng.integers(0, 2, n),
        "kvalues4": 1e3 * 1.5 ** rng.integers(0, 4, n),
        "kvalues8": 1e3 * 1.5 ** rng.integers(0, 8, n),
        "kvalues16": 1e3 * 1.5 ** rng.integers(0, 16, n),
        "kvalues3_skewed": 1e3 * 1.5 ** rng.choice(3, n, p=[0.9, 0.09, 0.01]),
        # two lognormal lobes 10x apart, 90 / 10
        "bimodal": rng.lognormal(math.log(1e5), 1.0, n) * np.where(rng.random(n) < 0.1, 10.0, 1.0),
        # streams that leave K1's main LDS window (keys -4096 .. 4095, |v| < 6.1e17) for its floating windows or the
        # global row: everything far above it, far below it, a little beyond it on both sides, and a thin far tail
        "far_1e30": rng.lognormal(math.log(1e30), 1.0, n),
        "negative_far": -rng.lognormal(math.log(1e25), 0.5, n),
        "signed_wide": 10.0 ** rng.uniform(-3, 20, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0),
        "thin_far_tail": np.where(rng.random(n) < 1e-3, 10.0 ** rng.uniform(19, 60, n), rng.lognormal(math.log(1e5), 1.0, n)),
    }


def check_stats(row, got, m=0, pcts=PCTS):
    want = oracle.process_dense(row, pcts)
    assert int(got["count"][m]) == want["count"]
    assert int(got["nbuckets"][m]) == want["nbuckets"]
    assert bool(got["present"][m]) == (want["count"] > 0)
    if want["count"]:
        # _sum is a float64 accumulation whose order the reference itself does not fix (Go map
        # order, metrics.go:342); the bound is the usual one for reordered sums: relative to the
        # sum of |terms| (== relative to the sum for the non-negative streams of BASELINE.json;
        # for signed streams the terms cancel and only this form is meaningful).
        mag = float(np.sum(np.abs(oracle.decompress_table()) * np.asarray(row, dtype=np.float64)))
        assert abs(got["sum"][m] - want["sum"]) <= SUM_RTOL * mag
        assert abs(got["avg"][m] - want["avg"]) <= SUM_RTOL * mag / want["count"]
        # bit-identical percentile values and keys
        assert np.array_equal(got["pvalid"][m], want["pvalid"])
        assert np.array_equal(got["pkeys"][m], want["pkeys"])
        assert np.array_equal(got["pvals"][m].view(np.uint64), want["pvals"].view(np.uint64))
    else:
        assert not got["pvalid"][m].any()


# ---- codec -------------------------------------------------------------------

def test_device_tables_equal_oracle_tables(engine):
    tx, d = engine.codec_tables()
    otx = oracle.thresholds(len(tx))
    assert np.array_equal(tx.view(np.uint64), otx.view(np.uint64))
    assert np.array_equal(d.view(np.uint64), oracle.decompress_table().view(np.uint64))


def test_vlog_error_bound_supports_guard_band(engine):
    # guard = 2/16384 buckets; error budget = 69.32*(trunc 2^-23/ln2 + vlog err)
    err = engine.selftest_vlog()
    t_err = 69.3147 * (2.0 ** -23 / math.log(2) + err)
    assert err < 1e-6
    assert t_err < 0.5 * (2 / 16384), (err, t_err)


def _compress_both_routes(engine, torch, v):
    dv = dev(torch, v)
    k1 = torch.empty(v.size, dtype=torch.int16, device="cuda")
    k2 = torch.empty(v.size, dtype=torch.int16, device="cuda")
    engine.compress_device(dv, k1, v.size)
    engine.compress_device(dv, k2, v.size, golog=True)
    engine.sync()
    torch.cuda.synchronize()
    return k1.cpu().numpy(), k2.cpu().numpy()


def test_compress_known_answers_and_specials(engine, torch_cuda):
    v = np.array([33, 59, 330000, 123, 1, -1, 0.0, -0.0, 0.005, 0.00502, 0.5, 0.51, 1e9, 1e12, 9.2e18,
                  -421408208120481.0, 214141241241241.0, 1e142, 2.0196e142, 2.03e142, -2.03e142, 3e142, -3e142,
                  1e200, 1.7976931348623157e308, -1.7976931348623157e308, math.inf, -math.inf, math.nan,
                  4.9e-324, 2.2250738585072014e-308, 1e-300, -1e-17])
    want = oracle.compress_many(v)
    fast, golog = _compress_both_routes(engine, torch_cuda, v)
    assert np.array_equal(fast, want)
    assert np.array_equal(golog, want)


def test_compress_at_every_threshold_neighbourhood(engine, torch_cuda):
    """The adversarial case: +-3 ulp around every one of the 70978 bucket
    thresholds, both signs.  Exercises the guard-band slow path on every sample."""
    tx = oracle.thresholds()[1: oracle.KEXT_MAX + 1]
    cand = []
    for d in range(-3, 4):
        x = (tx.view(np.int64) + d).view(np.float64)
        cand.append(x - 1.0)          # v with 1+v close to the threshold
        cand.append(np.nextafter(x - 1.0, np.inf))
        cand.append(np.nextafter(x - 1.0, -np.inf))
    v = np.concatenate(cand)
    v = np.concatenate([v, -v])
    want = oracle.compress_many(v)
    fast, golog = _compress_both_routes(engine, torch_cuda, v)
    assert np.array_equal(fast, want)
    assert np.array_equal(golog, want)


def test_compress_equals_the_go_handoff_fixture(engine, torch_cuda):
    """tests/golden/thresholds_x.bin is the file integration/compress_thresholds_test.go feeds to real Go: the
    HIP path must give exactly the keys recorded there (both signs), through both device routes -- so a `go test`
    PASS on that file pins the GPU, not just the oracle, to Go's compress at every threshold."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_thresholds as mt
    v, keys, _ = mt.read()
    assert v.size == 3 * oracle.KEXT_MAX
    fast, golog = _compress_both_routes(engine, torch_cuda, v)
    assert np.array_equal(fast, keys) and np.array_equal(golog, keys)
    neg = (-keys.astype(np.int32)).astype(np.int16)                 # -1 * i in int16 arithmetic, metrics.go:319
    nz = v != 0
    fast, golog = _compress_both_routes(engine, torch_cuda, -v[nz])
    assert np.array_equal(fast, neg[nz]) and np.array_equal(golog, neg[nz])


@pytest.mark.parametrize("name", list(dists(4, 0)))
def test_compress_random(engine, torch_cuda, name):
    v = dists(300_000, 5)[name]
    want = oracle.compress_many(v)
    fast, golog = _compress_both_routes(engine, torch_cuda, v)
    assert np.array_equal(fast, want)
    assert np.array_equal(golog, want)


# ---- single-metric ingest + extract ---------------------------------------------

@pytest.mark.parametrize("name", list(dists(4, 0)))
def test_ingest_single_bit_exact(engine, torch_cuda, name):
    v = dists(2_000_003, 2)[name]
    engine.submit_device(0, dev(torch_cuda, v))
    snap = engine.flip()
    got = snap.extract(PCTS, 1)
    row = snap.dense_row(0)
    snap.release()
    want = oracle.histogram_dense(v)
    assert np.array_equal(row, want)
    check_stats(want, got)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 63, 64, 65, 4095, 4096, 4097, 8191, 12289])
@pytest.mark.parametrize("offset", [0, 1])
def test_ingest_ragged_sizes_and_alignment(engine, torch_cuda, n, offset):
    v = np.random.default_rng(n).lognormal(5, 3, n + offset)
    d = dev(torch_cuda, v)
    engine.submit_device(0, d[offset:], n)   # offset=1 => 8-byte-aligned only
    with engine.flip() as snap:
        row = snap.dense_row(0)
        got = snap.extract(PCTS, 1)
    want = oracle.histogram_dense(v[offset:])
    assert np.array_equal(row, want)
    check_stats(want, got)


def test_out_of_window_and_wrapped_keys(engine, torch_cuda):
    rng = np.random.default_rng(9)
    v = np.concatenate([10.0 ** rng.uniform(30, 308, 50000), -(10.0 ** rng.uniform(30, 308, 50000)),
                        [math.inf, -math.inf, math.nan, 1.7976931348623157e308], rng.normal(0, 10, 1000)])
    engine.submit_device(0, dev(torch_cuda, v))
    with engine.flip() as snap:
        row = snap.dense_row(0)
        got = snap.extract(PCTS, 1)
    want = oracle.histogram_dense(v)
    assert np.array_equal(row, want)
    check_stats(want, got)


def test_extract_percentile_edge_cases(engine, torch_cuda):
    v = np.array([33.0, 59.0, 330000.0])
    engine.submit_device(0, dev(torch_cuda, v))
    pcts = [0.0, 1.0, 1.0000001, 2.0, math.nan, -1.0, 1 / 3, 2 / 3]
    with engine.flip() as snap:
        got = snap.extract(pcts, 2)
    want = oracle.process_dense(oracle.histogram_dense(v), pcts)
    assert np.array_equal(got["pvalid"][0], want["pvalid"])
    assert list(got["pvalid"][0]) == [1, 1, 0, 0, 0, 1, 1, 1]
    assert np.array_equal(got["pvals"][0].view(np.uint64), want["pvals"].view(np.uint64))
    # metric 1 never received a sample: absent, like a name missing from histogramCache
    assert got["present"][1] == 0 and got["count"][1] == 0 and not got["pvalid"][1].any()
    assert math.isnan(got["avg"][1])


# ---- reference tests restated against the engine ------------------------------------

def test_reference_processed_broadcast(engine):
    # TestProcessedBroadcast, metrics_test.go:289-319 (host submit path)
    hid = engine.intern("histogram1")
    for s in (33, 59, 330000):
        engine.submit(hid, [float(s)])
    with engine.flip() as snap:
        got = snap.extract(PCTS, engine.num_metrics())
    assert int(got["sum"][hid]) == 331132
    assert int(got["count"][hid]) == 3
    assert int(got["agg_sum_add"][hid]) // int(got["count"][hid]) == 110377  # _agg_avg, metrics.go:601-602


def test_reference_percentile_table(engine):
    # TestPercentile, metrics_test.go:111-149, through compress/decompress (1 % tolerance as in the reference)
    metrics = {10: 9000, 25: 900, 33: 90, 47: 9, 500: 1}
    expected = {0: 10, .99: 25, .999: 33, .9991: 47, .9999: 47, 1: 500}
    v = np.concatenate([np.full(c, float(x)) for x, c in metrics.items()])
    engine.submit(0, v)
    with engine.flip() as snap:
        got = snap.extract(list(expected), 1)
    for i, (p, exp) in enumerate(expected.items()):
        assert got["pvalid"][0][i]
        assert abs(exp / got["pvals"][0][i] - 1) <= 0.01, (p, exp, got["pvals"][0][i])


def test_reference_compress_roundtrip(engine, torch_cuda):
    # TestCompress, metrics_test.go:151-172: decompress(compress(f)) within 1 %
    v = np.array([-421408208120481.0, -1.0, 0.0, 1.0, 214141241241241.0])
    keys, _ = _compress_both_routes(engine, torch_cuda, v)
    _, d = engine.codec_tables()
    for f, k in zip(v, keys):
        result = d[(int(k) & 0xFFFF) ^ 0x8000]
        diff = abs(f - result) if result == 0 else abs(f / result - 1)
        assert diff <= 0.01


# ---- mixed (id, value) streams -------------------------------------------------------

def test_ingest_pairs_zipf(engine, torch_cuda):
    rng = np.random.default_rng(3)
    n, M = 1_500_001, 16
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(math.log(1e5) + 0.002 * ids, 1.0)
    engine.submit_pairs_device(dev(torch_cuda, ids), dev(torch_cuda, v))
    with engine.flip() as snap:
        got = snap.extract(PCTS, M)
        rows = [snap.dense_row(m) for m in range(M)]
    want = oracle.histogram_pairs(ids, v, M)
    for m in range(M):
        assert np.array_equal(rows[m], want[m]), m
        check_stats(want[m], got, m)


def test_pairs_bad_id_is_reported_not_silently_dropped(engine, torch_cuda, la):
    ids = np.array([0, 1, 99, 2], dtype=np.uint32)
    with pytest.raises(la.LhError) as ei:
        engine.submit_pairs(ids, np.ones(4))
    assert ei.value.code == 6
    engine.submit_pairs_device(dev(torch_cuda, ids), dev(torch_cuda, np.ones(4)))
    with pytest.raises(la.LhError) as ei:
        engine.sync()
    assert ei.value.code == 6
    engine.flip().release()


# ---- host staging path, threads, epochs ------------------------------------------------

def test_host_submit_multithreaded_lossless(engine):
    rng = np.random.default_rng(21)
    T, per = 8, 120_001
    chunks = [rng.lognormal(8, 2, per) for _ in range(T)]
    idc = [rng.integers(0, 16, per).astype(np.uint32) for _ in range(T)]

    def work(t):
        # odd batch sizes cross the 65536-sample half-buffers and switch lane modes
        for lo in range(0, per, 7001):
            if (lo // 7001) % 2:
                engine.submit(3, chunks[t][lo:lo + 7001])
            else:
                engine.submit_pairs(idc[t][lo:lo + 7001], chunks[t][lo:lo + 7001])

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    with engine.flip() as snap:
        rows = np.stack([snap.dense_row(m) for m in range(16)])
    want = np.zeros((16, oracle.NKEYS), dtype=np.uint64)
    for t in range(T):
        for lo in range(0, per, 7001):
            if (lo // 7001) % 2:
                oracle.histogram_dense(chunks[t][lo:lo + 7001], want[3])
            else:
                oracle.histogram_pairs(idc[t][lo:lo + 7001], chunks[t][lo:lo + 7001], 16, want)
    assert int(rows.sum()) == T * per
    assert np.array_equal(rows, want)


@pytest.mark.parametrize("zero_copy", [1, 0])
def test_in_place_staging_reserve_commit(la, torch_cuda, zero_copy):
    """lh_reserve_pairs / lh_commit_pairs (SURVEY.md 8b Ownership: the producer writes into the C-allocated pinned
    buffer in place; call shape of Histogram, metrics.go:273): threads that reserve, fill a part, commit -- mixed with
    lh_submit*, partial and empty commits, reservations that outlast other producers' calls, and flips in between.
    Every cell equals the oracle's; with and without the kernels reading the pinned buffers in place."""
    from loghisto_amd import _native as N
    rng = np.random.default_rng(77)
    T, per, M = 6, 90_011, 24
    vals = [rng.lognormal(9, 2, per) for _ in range(T)]
    ids = [rng.integers(0, M, per).astype(np.uint32) for _ in range(T)]
    with la.Engine(max_metrics=M, num_buffers=2, num_lanes=3, lane_samples=1 << 14) as eng:   # fewer lanes than threads
        eng.set_option(N.OPT_LANE_ZERO_COPY, zero_copy)
        snaps_rows = []
        stop = threading.Event()

        def work(t):
            lo, k = 0, 0
            while lo < per:
                k += 1
                want = min(per - lo, 1 + (k * 2477) % 9001)
                if k % 5 == 0:
                    eng.submit_pairs(ids[t][lo:lo + want], vals[t][lo:lo + want])
                    lo += want
                    continue
                di, dv, tok = eng.reserve_pairs(want)
                n = di.size if k % 3 else di.size // 2          # partial commits; n = 0 gives the reservation back
                di[:n] = ids[t][lo:lo + n]
                dv[:n] = vals[t][lo:lo + n]
                eng.commit_pairs(tok, n)
                lo += n

        def flipper():
            while not stop.is_set():
                try:
                    with eng.flip() as snap:
                        snaps_rows.append(np.stack([snap.dense_row(m) for m in range(M)]))
                except la.LhError:
                    pass

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        fl = threading.Thread(target=flipper)
        fl.start()
        [x.start() for x in th]
        [x.join() for x in th]
        stop.set()
        fl.join()
        with eng.flip() as snap:
            snaps_rows.append(np.stack([snap.dense_row(m) for m in range(M)]))
        with pytest.raises(la.LhError):
            eng.commit_pairs(1, 0)                                # nothing reserved on that buffer: LH_ESTATE
    got = np.sum(snaps_rows, axis=0)
    want = np.zeros((M, oracle.NKEYS), dtype=np.uint64)
    for t in range(T):
        oracle.histogram_pairs(ids[t], vals[t], M, want)
    assert int(got.sum()) == T * per
    assert np.array_equal(got, want)


def test_epoch_semantics(engine, la, torch_cuda):
    # a sample belongs to exactly one interval (metrics.go:460-463)
    a = np.full(1000, 10.0)
    b = np.full(500, 1e6)
    engine.submit(0, a)
    s1 = engine.flip()
    engine.submit(0, b)
    with pytest.raises(la.LhError) as ei:   # both buffers in use: the epoch keeps accumulating
        engine.flip()
    assert ei.value.code == 5
    assert np.array_equal(s1.dense_row(0), oracle.histogram_dense(a))
    s1.release()
    s2 = engine.flip()
    assert np.array_equal(s2.dense_row(0), oracle.histogram_dense(b))
    s2.release()
    # recycled buffers come back clean
    for _ in range(3):
        with engine.flip() as s:
            assert s.extract(PCTS, 1)["count"][0] == 0
            assert s.buckets(0)[0].size == 0


def test_linearity_and_buckets_listing(engine, torch_cuda):
    rng = np.random.default_rng(5)
    a, b = rng.normal(0, 1e4, 300_000), rng.exponential(50, 200_000)
    engine.submit_device(1, dev(torch_cuda, a))
    engine.submit_device(1, dev(torch_cuda, b))
    with engine.flip() as snap:
        keys, counts = snap.buckets(1)
    want = oracle.histogram_dense(a) + oracle.histogram_dense(b)
    nz = np.nonzero(want)[0]
    assert np.array_equal(keys, oracle.bin_to_key(nz))
    assert np.array_equal(counts, want[nz])
    assert np.all(np.diff(keys.astype(np.int32)) > 0)


def test_names_intern(engine, la):
    a = engine.intern("some_ipc")
    assert engine.intern("some_ipc") == a
    b = engine.intern("other")
    assert b != a and engine.lookup("other") == b and engine.lookup("missing") is None
    assert engine.metric_name(a) == "some_ipc"
    for i in range(engine.max_metrics - engine.num_metrics()):
        engine.intern(f"fill{i}")
    with pytest.raises(la.LhError) as ei:
        engine.intern("one too many")
    assert ei.value.code == 6


# ---- partitioned mixed ingest (lh_kernels_part.hip) -----------------------------------

def _zipf_ids(rng, n, M):
    w = 1.0 / np.arange(1, M + 1)
    return rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)


@pytest.mark.parametrize("M,n,kind", [
    (1024, 3_000_001, "lognormal"),      # BASELINE config 3 shape: 256 partitions x 4 names
    (1024, 1_000_000, "constant"),       # one cell per name: the wave-uniform path
    (77, 700_001, "signed_wide"),        # odd name count, keys on both sides, out-of-window records
    (2, 200_000, "lognormal"),           # single partition
    (300, 131_072, "loguniform"),        # exactly the partitioned-path threshold
    (300, 131_071, "loguniform"),        # one below: direct-atomic path
])
def test_ingest_pairs_partitioned(la, torch_cuda, M, n, kind):
    rng = np.random.default_rng(M * 7 + n)
    ids = _zipf_ids(rng, n, M)
    if kind == "lognormal":
        v = rng.lognormal(math.log(1e5) + 0.002 * ids, 1.0)
    elif kind == "constant":
        v = np.full(n, 123.0)
    elif kind == "signed_wide":
        v = rng.normal(0, 1e3, n) * 10.0 ** rng.integers(0, 60, n)
        v[::1000] = math.inf
        v[1::1000] = 1e300
    else:
        v = 10.0 ** rng.uniform(-3, 18, n)
    e = la.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    try:
        e.submit_pairs_device(dev(torch_cuda, ids), dev(torch_cuda, v))
        e.sync()
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
            want = oracle.histogram_pairs(ids, v, M)
            assert int(got["count"].sum()) == n
            # EVERY row, cell by cell (no sampled rows): the device-compacted CSR listing of
            # all occupied cells against the oracle's matrix
            off, keys, counts = snap.buckets_all(M)
            dense = np.zeros((M, 65536), dtype=np.uint64)
            rows = np.repeat(np.arange(M), np.diff(off.astype(np.int64)))
            dense[rows, oracle.key_to_bin(keys)] = counts
            assert np.array_equal(dense, want)
            for m in range(M):
                wc = int(want[m].sum())
                assert int(got["count"][m]) == wc, m
                if wc:
                    check_stats(want[m], got, m)
    finally:
        e.close()


def test_pairs_partitioned_bad_ids_and_reuse(la, torch_cuda):
    rng = np.random.default_rng(99)
    M, n = 96, 400_000
    ids = _zipf_ids(rng, n, M)
    v = rng.lognormal(10, 1, n)
    bad = ids.copy()
    bad[[5, 77777, n - 1]] = [M, 0xFFFFFFFF, 123456]
    e = la.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    try:
        for rep in range(3):     # scratch and chunk pools are reused across launches
            e.submit_pairs_device(dev(torch_cuda, ids), dev(torch_cuda, v))
            e.sync()
            with e.flip() as snap:
                got = snap.extract([0.5], M)
                assert int(got["count"].sum()) == n
                assert np.array_equal(snap.dense_row(0), oracle.histogram_pairs(ids, v, M)[0])
        e.submit_pairs_device(dev(torch_cuda, bad), dev(torch_cuda, v))
        with pytest.raises(la.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        with e.flip() as snap:   # the three bad samples are skipped, the rest land exactly
            keep = np.ones(n, dtype=bool)
            keep[[5, 77777, n - 1]] = False
            want = oracle.histogram_pairs(ids[keep], v[keep], M)
            try:
                got = snap.extract([0.5], M)
            except la.LhError:
                got = snap.extract([0.5], M)   # the sticky id error is reported once
            assert int(got["count"].sum()) == n - 3
            assert np.array_equal(snap.dense_row(1), want[1])
    finally:
        e.close()


def test_create_reports_out_of_memory(la, torch_cuda):
    """An engine that cannot fit in HBM (2 TiB of rows per epoch buffer) fails with LH_ENOMEM, it does not abort;
    the process keeps working afterwards."""
    with pytest.raises(la.LhError) as ei:
        la.Engine(max_metrics=1 << 22, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    assert ei.value.code == 2
    with la.Engine(max_metrics=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit(0, np.array([1.0, 2.0, 3.0]))
        with e.flip() as snap:
            assert int(snap.extract([], 1)["count"][0]) == 3
