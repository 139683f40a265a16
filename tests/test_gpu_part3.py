"""Third generation of the partitioned mixed ingest (loghisto_amd/csrc/lh_kernels_part3.h): 8 193 .. 65 536 names
-- BASELINE config 4's name count -- through the hashed survey, the region scatter of 4-byte records
(k_scatter4), the second level that counts each partition's frequent names in place (k_split_records) and the
reduce pass (k_part_hist3).

The survey, the hot-name hash, the per-partition ranking and the window width only decide WHERE a sample is
counted; every cell of every row must equal the oracle's whatever they estimate.  A dense oracle matrix would be
32 GiB at 65 536 names, so both sides are compared as sorted (name << 16 | bin, count) lists: the oracle's from
compress_many + unique, the engine's from lh_buckets_all (device-compacted CSR of all occupied cells).
Reference semantics: metrics.go:273-295 (fan-in), 316-322 (compress)."""
import math

import numpy as np
import pytest

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .9, .99, .999, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def _ids(rng, M, n, skew, permute=True):
    w = np.arange(1, M + 1, dtype=np.float64) ** -skew
    perm = rng.permutation(M) if permute else np.arange(M)
    return perm[rng.choice(M, size=n, p=w / w.sum())].astype(np.uint32)


def _values(rng, kind, ids, n):
    if kind == "lognormal":
        return rng.lognormal(math.log(1e5) + 2e-3 * (ids % 4096), 1.0)
    if kind == "constant":
        return 1000.0 + (ids % 5)
    if kind.startswith("kvalues"):               # few-valued: k distinct buckets shared by all names
        return 1e3 * 1.5 ** rng.integers(0, int(kind[7:]), n)
    if kind == "bimodal":                        # two lognormal lobes 10x apart, 90 / 10
        return rng.lognormal(math.log(1e5), 1.0, n) * np.where(rng.random(n) < 0.1, 10.0, 1.0)
    if kind == "allsame":
        return np.full(n, 123.0)
    if kind == "signed":                         # two lobes of bins per name, the mean bin between them
        return rng.normal(0, 1e4, n)
    if kind == "loguniform":                     # 4 147 occupied buckets per name
        return 10.0 ** rng.uniform(-3, 18, n)
    if kind == "huge":                           # spans the whole key space: most samples miss every window
        return 10.0 ** rng.uniform(-6, 140, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    if kind == "signed_wide":                    # +-10^U(-3, 20): 9 211 bins across key 0 -- wider than an 8 192-bin window
        return 10.0 ** rng.uniform(-3, 20, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    if kind == "sigma25":
        return rng.lognormal(math.log(1e5), 2.5, n)
    if kind == "drift":                          # the distribution moves during the launch: windows go stale
        return rng.lognormal(math.log(1e3) + 9.0 * np.arange(n) / n, 0.5)
    assert kind == "edge"
    v = rng.lognormal(math.log(1e5), 1.0, n)
    hot = int(np.bincount(ids).argmax())
    sel = np.nonzero(ids == hot)[0]
    v[sel[:3000]] = 2.0196e142                   # key +32767 on a hot name
    v[sel[3000:6000]] = -2.0196e142
    v[sel[6000:6100]] = float("nan")
    v[sel[6100:6200]] = float("inf")
    v[sel[6200:6300]] = 0.0
    v[sel[6300:6400]] = 3e142                    # beyond the int16 domain: amd64 wrap
    last = int(ids.max())                        # the record 0xffffffff when the last name is 65 535
    v[ids == last] = 2.0196e142
    return v


def oracle_cells(ids, v):
    """Sorted (name << 16 | bin) of every occupied cell and its count, from the oracle's compress."""
    bins = oracle.key_to_bin(oracle.compress_many(v)).astype(np.uint64)
    return np.unique((ids.astype(np.uint64) << np.uint64(16)) | bins, return_counts=True)


def engine_cells(snap, M):
    off, keys, counts = snap.buckets_all(M)
    rows = np.repeat(np.arange(M, dtype=np.uint64), np.diff(off.astype(np.int64)))
    cells = (rows << np.uint64(16)) | oracle.key_to_bin(keys).astype(np.uint64)
    return cells, counts.astype(np.int64)


def check(snap, ids, v, M, got):
    want_cells, want_counts = oracle_cells(ids, v)
    cells, counts = engine_cells(snap, M)
    if not (np.array_equal(cells, want_cells) and np.array_equal(counts, want_counts)):
        a = dict(zip(cells.tolist(), counts.tolist()))
        b = dict(zip(want_cells.tolist(), want_counts.tolist()))
        bad = sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))
        names = sorted({k >> 16 for k in bad})
        raise AssertionError(f"{len(bad)} cells of {len(names)} names differ, e.g. " +
                             ", ".join(f"name {k >> 16} bin {k & 0xffff}: {a.get(k)} != {b.get(k)}" for k in bad[:6]))
    per_name = np.bincount(ids, minlength=M)
    assert np.array_equal(got["count"].astype(np.int64), per_name)
    for m in np.nonzero(per_name)[0][:: max(1, M // 48)]:
        ref = oracle.process_dense(oracle.histogram_dense(v[ids == m]), PCTS)
        assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m
        assert np.array_equal(got["pkeys"][m], ref["pkeys"]), m


CASES = [
    # names, pairs, values, Zipf exponent, permuted ids, window (0 = follow the survey)
    (65536, 3_000_001, "lognormal", 1.0, False, 0),   # config 4's shape (id = rank); odd length
    (65536, 2_500_000, "lognormal", 1.0, True, 0),    # the same with names in arbitrary order
    (65536, 2_000_000, "lognormal", 0.0, True, 10),   # no skew: nothing is hot, nothing is frequent in its partition
    (65536, 2_000_000, "edge", 1.0, False, 10),       # key +-32767, NaN, Inf, wrap; the last name; record 0xffffffff
    (65536, 1_500_000, "loguniform", 1.0, True, 10),  # 4 147 buckets per name against 1 024-bin windows: overflow paths
    (65536, 1_500_000, "loguniform", 1.0, True, 0),   # ... and with the width the survey reports
    (65536, 1_500_000, "huge", 1.5, True, 13),
    (65536, 2_000_000, "constant", 1.0, True, 11),
    (40000, 2_200_000, "allsame", 1.0, True, 12),     # ragged: 157 names per partition, fine partitions partly empty
    (16384, 2_600_000, "sigma25", 1.0, True, 0),      # 64 names per partition
    (8193, 2_000_000, "signed", 1.5, True, 0),        # the smallest name count of this path: 33 names per partition
    (20000, 1_800_000, "drift", 0.5, False, 0),
    (65536, 2_000_000, "kvalues2", 1.0, False, 0),    # few-valued streams (quantised timers): two cells per name
    (65536, 2_000_000, "kvalues8", 1.0, True, 0),
    (65536, 2_000_000, "bimodal", 1.0, False, 0),
    (65536, 2_400_000, "signed_wide", 1.0, True, 0),  # the survey asks for 16 384-bin windows: one reduce slot per CU
    (65536, 2_000_001, "edge", 1.0, False, 14),       # ... pinned: the key-space ends inside such a window
    (40000, 2_200_000, "huge", 1.2, True, 14),
    (20000, 2_000_000, "signed_wide", 0.5, False, 14),
]


@pytest.mark.parametrize("M,n,kind,skew,permute,log_w", CASES)
def test_third_generation_is_exact(native_lib, torch_cuda, M, n, kind, skew, permute, log_w):
    import loghisto_amd
    rng = np.random.default_rng(M * 7 + n)
    ids = _ids(rng, M, n, skew, permute)
    v = _values(rng, kind, ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.set_option(N.OPT_PART_V3_LOG_W, log_w)
        for rep in range(2):                     # scratch and survey tables are reused; rep 1 runs with rep 0's window report
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            c = e.counters()
            assert c["samples_partitioned_v3"] == n * (rep + 1), c
            assert 10 <= c["window_log2"] <= 14
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                check(snap, ids, v, M, got)


DIRECT_CASES = [CASES[i] for i in (0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 13)]


@pytest.mark.parametrize("M,n,kind,skew,permute,log_w", DIRECT_CASES)
def test_small_launches_reduce_without_windows(native_lib, torch_cuda, M, n, kind, skew, permute, log_w):
    """Launches of at most 2^22 pairs (a host-fed lane's half-buffer) end in k_part_direct3: the level-2 chunks' records
    go to their cells with one global atomic each, no plan and no LDS windows.  The same stream shapes as above, once at
    the default bound (the call in two slices) and once with the bound at its maximum."""
    import loghisto_amd
    rng = np.random.default_rng(M * 11 + n)
    ids = _ids(rng, M, n, skew, permute)
    v = _values(rng, kind, ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_LOG_W, log_w)
        for rep, bound in enumerate((0, 1 << 30)):
            e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, bound)
            if bound == 0:                       # default bound, two launches
                h = (n // 2) & ~1
                e.submit_pairs_device(d_ids[:h], d_v[:h])
                e.submit_pairs_device(d_ids[h:], d_v[h:])
            else:
                e.submit_pairs_device(d_ids, d_v)
            e.sync()
            c = e.counters()
            assert c["samples_partitioned_v3"] == n * (rep + 1), c
            assert c["reduce_window_misses"] == 0, c      # nothing has a window to miss
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                check(snap, ids, v, M, got)


def test_one_bucket_of_a_frequent_name_without_hot_windows(native_lib, torch_cuda):
    """With the first level's hot windows off the most frequent name reaches the second level with every sample in one
    of two adjacent buckets: ~60 000 records of a 62 000-record work slot land in ONE LDS cell there, its neighbour gets
    3 % of that.  (Written for a 16-bit-cell form of that pass -- measured slower and dropped, profiles/
    r05_level23_experiments.txt -- and kept: no other case puts 10^4 .. 10^5 counts into single cells of levels 2 and 3.)"""
    import loghisto_amd
    M, n = 65536, 4_000_000
    rng = np.random.default_rng(99)
    ids = _ids(rng, M, n, 1.0, permute=False)
    v = np.where(rng.random(n) < 0.97, 123.0, 124.0)      # bins 482 and 483 of every name: the two halves of one word
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.set_option(N.OPT_HOT_WINDOWS, 0)
        for log_w in (10, 11):
            e.set_option(N.OPT_PART_V3_LOG_W, log_w)
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                check(snap, ids, v, M, got)
        assert e.counters()["samples_partitioned_v3"] == 2 * n


def test_survey_is_reused_while_the_stream_looks_the_same(native_lib, torch_cuda):
    """LH_OPT_SURVEY_EVERY (default 8): calls run on the previous call's survey while its tables are still in the
    scratch block, the window width is unchanged and the self-metrics stay healthy; a stream that changes under a stale
    survey is still bucketed exactly (every cell against the oracle) and makes the next call survey again."""
    import loghisto_amd
    rng = np.random.default_rng(11)
    M, n = 65536, 1_500_000
    ids = _ids(rng, M, n, 1.0)
    v1 = _values(rng, "lognormal", ids, n)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.set_option(N.OPT_PART_V3_LOG_W, 10)
        d_ids, d_v1 = _dev(torch_cuda, ids), _dev(torch_cuda, v1)
        for rep in range(5):
            e.submit_pairs_device(d_ids, d_v1)
            e.sync()
            with e.flip() as snap:
                check(snap, ids, v1, M, snap.extract(PCTS, M))
        c = e.counters()
        assert c["surveys_reused"] == 4, c                       # one survey, four calls on it
        # the stream changes completely (other names are frequent, values 9 decades wide): the stale survey sends most
        # records down the overflow paths -- exact all the same -- and the call after that surveys again
        ids2 = (M - 1 - ids).astype(np.uint32)
        v2 = _values(rng, "loguniform", ids2, n)
        d_ids2, d_v2 = _dev(torch_cuda, ids2), _dev(torch_cuda, v2)
        for rep in range(3):
            e.submit_pairs_device(d_ids2, d_v2)
            e.sync()
            with e.flip() as snap:
                check(snap, ids2, v2, M, snap.extract(PCTS, M))
        c2 = e.counters()
        assert c2["surveys_reused"] - c["surveys_reused"] <= 2, (c, c2)   # at least one of the three surveyed
        e.set_option(N.OPT_SURVEY_EVERY, 1)
        before = e.counters()["surveys_reused"]
        e.submit_pairs_device(d_ids, d_v1)
        e.sync()
        assert e.counters()["surveys_reused"] == before
        e.flip().release()


def test_names_without_skew_go_back_to_the_first_generation(native_lib, torch_cuda):
    """Nothing is frequent among uniform names: the second level counts next to nothing in place and forwards > 3/4 of
    the pairs.  The engine reads that from the self-metrics of the completed calls and routes the following calls
    through the first generation (7.9 against 9.7 ms per 1e9 pairs at 65 536 names); exact either way."""
    import loghisto_amd
    rng = np.random.default_rng(23)
    M, n = 65536, 4_500_000
    ids = _ids(rng, M, n, 0.0)
    v = _values(rng, "lognormal", ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        seen = []
        for rep in range(4):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            seen.append(e.counters()["samples_partitioned_v3"])
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
        assert seen[0] == n and seen[-1] == seen[-2] < 4 * n, seen   # the later calls did not take the third generation
        assert e.counters()["samples_partitioned"] == 4 * n


def test_window_width_follows_the_stream(native_lib, torch_cuda):
    """The survey's report (the smallest window within half of which, around their name's mean, 99 % of the samples lie):
    lognormal sigma = 1 is 100 bins wide (1 024-bin windows), sigma = 2.5 is 251 (2 048), 21 decades span 4 147 (8 192), 23 on either side of key 0 span 9 211 (16 384)."""
    import loghisto_amd
    rng = np.random.default_rng(5)
    M, n = 65536, 1_000_000
    ids = _ids(rng, M, n, 1.0)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.set_option(N.OPT_SURVEY_EVERY, 1)      # every call surveys: the report follows the stream call by call
        for kind, want in (("lognormal", 10), ("loguniform", 13), ("sigma25", 11), ("signed_wide", 14), ("lognormal", 10)):
            e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, _values(rng, kind, ids, n)))
            e.sync()
            assert e.counters()["window_log2"] == want, (kind, e.counters())
            e.flip().release()


def test_bad_ids_sublaunches_and_two_launches_per_epoch(native_lib, torch_cuda):
    import loghisto_amd
    rng = np.random.default_rng(78)
    M, n = 30000, 9_000_001
    ids = _ids(rng, M, n, 1.0)
    v = rng.lognormal(10, 1.2, n)
    bad = ids.copy()
    where = [3, 4_200_000, n - 1]
    bad[where] = [M, 0xFFFFFFFF, M + 5]
    keep = np.ones(n, dtype=bool)
    keep[where] = False
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.set_option(N.OPT_SUBLAUNCH_PAIRS, 1 << 22)       # 3 sub-launches per call, one survey per call
        # (the device arrays must outlive the launches: the engine's stream is not one torch's allocator knows about)
        d_ids, d_bad, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, bad), _dev(torch_cuda, v)
        e.submit_pairs_device(d_ids, d_v)
        e.submit_pairs_device(d_bad, d_v)
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        assert e.counters()["sublaunches"] == 6 and e.counters()["samples_partitioned_v3"] == 2 * n
        with e.flip() as snap:
            try:
                got = snap.extract(PCTS, M)
            except loghisto_amd.LhError:
                got = snap.extract(PCTS, M)
            check(snap, np.concatenate([ids, ids[keep]]), np.concatenate([v, v[keep]]), M, got)


def test_threshold_fixture_through_the_hashed_scatter(native_lib, torch_cuda):
    """Every bucket threshold, the double below and the double above it (tests/golden/thresholds_x.bin, the file real
    Go is checked against), both signs, through k_scatter4's branch-free bucket index."""
    import os
    import sys
    import loghisto_amd
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_thresholds as mt
    x, _, _ = mt.read()
    v = np.concatenate([x, -x, x[::-1]])
    M = 9000
    ids = ((np.arange(v.size, dtype=np.uint64) * 7919) % M).astype(np.uint32)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        assert e.counters()["samples_partitioned_v3"] == v.size
        with e.flip() as snap:
            check(snap, ids, v, M, snap.extract(PCTS, M))


def test_clustered_stream_falls_back_to_the_first_generation(native_lib, torch_cuda):
    """A stream sorted by name puts whole tiles into one level-1 partition: the regions overflow, the kernel reports
    it, and the engine takes the exact-layout scatter for the following intervals.  Exact either way."""
    import loghisto_amd
    rng = np.random.default_rng(9)
    M, n = 65536, 4_194_304
    ids = np.sort(_ids(rng, M, n, 0.0))
    v = rng.lognormal(10, 1.0, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=3, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        seen_v3 = 0
        for rep in range(3):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            c = e.counters()
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
            if rep == 0:
                assert c["samples_partitioned_v3"] == n
                seen_v3 = c["samples_partitioned_v3"]
        c = e.counters()
        assert c["region_overflows"] > n // 50 and c["regions_disabled"] == 1, c
        assert c["samples_partitioned_v3"] < 3 * n and c["samples_partitioned"] == 3 * n, (c, seen_v3)


def test_concurrent_streams_into_the_same_names(native_lib, torch_cuda):
    """a third-generation launch on one stream while other streams add to the SAME cells of the same
    epoch buffer (a device-resident producer on stream A, small launches of the direct-atomic kernel and the host
    lanes on others).  Every cell update of every pass must be atomic: round 3's reduce pass flushed the slots that own
    their whole partition with a plain load + store and could lose the other stream's increment.  metrics.go:273-295:
    Histogram is called from any number of goroutines and never loses a sample."""
    import loghisto_amd
    torch = torch_cuda
    rng = np.random.default_rng(404)
    M, n = 65536, 6_000_000
    ids = _ids(rng, M, n, 1.0, permute=False)
    v = _values(rng, "constant", ids, n)              # few cells per name: both streams hit the same ones
    small = 100_000                                   # < 131 072 pairs: the direct-atomic kernel
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    d_ids, d_v = _dev(torch, ids), _dev(torch, v)
    rounds, per_round = 3, 12
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        torch.cuda.synchronize()
        for r in range(rounds):
            e.submit_pairs_device(d_ids, d_v, stream=sa)               # survey + three passes on stream A
            for k in range(per_round):                                  # ... while stream B and a host lane add too
                lo = (r * per_round + k) * small
                e.submit_pairs_device(d_ids[lo:lo + small], d_v[lo:lo + small], stream=sb)
                e.submit_pairs(ids[lo:lo + 4096], v[lo:lo + 4096])
        e.sync()
        torch.cuda.synchronize()
        assert e.counters()["samples_partitioned_v3"] == rounds * n
        parts_i = [ids] * rounds + [ids[: rounds * per_round * small]]
        parts_v = [v] * rounds + [v[: rounds * per_round * small]]
        for k in range(rounds * per_round):
            parts_i.append(ids[k * small:k * small + 4096])
            parts_v.append(v[k * small:k * small + 4096])
        all_i, all_v = np.concatenate(parts_i), np.concatenate(parts_v)
        with e.flip() as snap:
            check(snap, all_i, all_v, M, snap.extract(PCTS, M))


def test_a_width_change_is_not_mistaken_for_names_without_skew(native_lib, torch_cuda):
    """A stream whose value span changes (lognormal -> 21 decades) runs ONE call with windows that are too narrow: most
    records are forwarded whatever the names' skew.  The survey of that call reports the wider window; the engine must not
    read the forwarded share of a call that ran at another width as "no skew" and send the following calls to the first
    generation (measured: 12.4 instead of 8.8 ms per 1e9 pairs for 64 flips).  Exact throughout."""
    import loghisto_amd
    rng = np.random.default_rng(31)
    M, n = 65536, 4_500_000
    ids = _ids(rng, M, n, 1.0)
    v1 = _values(rng, "lognormal", ids, n)
    v2 = _values(rng, "loguniform", ids, n)
    d_ids, d_v1, d_v2 = _dev(torch_cuda, ids), _dev(torch_cuda, v1), _dev(torch_cuda, v2)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)    # launches this small keep the windowed reduce pass under test
        for d_v, v in ((d_v1, v1), (d_v1, v1), (d_v2, v2), (d_v2, v2), (d_v2, v2), (d_v2, v2)):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
        c = e.counters()
        assert c["samples_partitioned_v3"] == 6 * n, sorted(c.items())   # every call took the third generation
        assert c["window_log2"] == 13, sorted(c.items())


def test_a_value_shift_under_a_kept_survey_ends_its_reuse(native_lib, torch_cuda):
    """As tests/test_gpu_part2.py's test of the same name, 65 536 names: the share of a launch's pairs that level 1 did
    NOT turn into records against the first launch on the survey's tables (k_v3_report).  The shifted stream keeps its
    values inside every reduce window (no overflow or miss count moves: only the new word tells), and every cell is exact."""
    import loghisto_amd
    rng = np.random.default_rng(43)
    M, n = 65536, 3_000_000
    ids = _ids(rng, M, n, 1.0)
    v1 = rng.lognormal(math.log(1e5), 0.3, n)
    v2 = rng.lognormal(math.log(1e3), 0.3, n)                  # 460 bins lower: outside the hot windows, inside 1 024 bins of them
    d_ids, d_v1, d_v2 = _dev(torch_cuda, ids), _dev(torch_cuda, v1), _dev(torch_cuda, v2)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V3_DIRECT_MAX_PAIRS, 1)

        def call(d_v, v):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
            return e.counters()

        for _ in range(4):
            c = call(d_v1, v1)
        assert c["surveys_reused"] == 3 and c["survey_stale_pairs"] == 0, c
        c = call(d_v2, v2)
        assert c["surveys_reused"] == 4 and c["survey_stale_pairs"] > n // 10, c
        c = call(d_v2, v2)
        assert c["surveys_reused"] == 4, c                     # surveyed again
        stale = c["survey_stale_pairs"]
        for _ in range(3):
            c = call(d_v2, v2)
        assert c["surveys_reused"] == 7 and c["survey_stale_pairs"] == stale, c


def test_the_first_large_call_takes_the_width_of_its_own_stream(native_lib, torch_cuda):
    """Before any survey has reported, the engine has no window width to lay a call out for.  A wide stream (21 decades:
    4 147 bins per name) on the default 1 024-bin windows is exact but counts most of its records through the overflow
    tables and global atomics (a 1e9-pair call took 636 ms instead of 8.9: profiles/r06_first_call.txt).  The first call of
    at least 2^24 device-resident pairs therefore runs the survey's first kernels alone and waits for their report
    (lh_engine.cc: probe_width): this call already runs at 8 192-bin windows, and nearly nothing misses them."""
    import loghisto_amd
    rng = np.random.default_rng(77)
    M, n = 16384, (1 << 24) + 3
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, "loguniform", ids, n)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        c = e.counters()
        assert c["samples_partitioned_v3"] >= n - 8192 and c["window_log2"] == 13, sorted(c.items())
        assert c["reduce_window_misses"] + c["level2_overflows"] < n // 50, sorted(c.items())
        with e.flip() as snap:
            check(snap, ids, v, M, snap.extract(PCTS, M))
    # a pinned width is the caller's: no probe, the call runs on it (and stays exact)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3_LOG_W, 10)
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        c = e.counters()
        assert c["window_log2"] == 10 and c["reduce_window_misses"] > n // 4, sorted(c.items())
        with e.flip() as snap:
            check(snap, ids, v, M, snap.extract(PCTS, M))


@pytest.mark.parametrize("id16", [False, True])
def test_a_clustered_call_is_finished_by_the_cell_table(native_lib, torch_cuda, id16):
    """Inside ONE call: a level-1 workgroup whose last two tiles overflowed by more than an eighth leaves the rest of its turn
    to k_scatter4_clustered, whose LDS is a (name, bin) -> count table (a 1e9-pair call sorted by name took 1.9 s through
    the overflow path: profiles/r06_first_call.txt).  25 M pairs = 12 tiles per workgroup.  Every cell exact; bad ids in
    the part the table counts are reported and skipped like everywhere else."""
    import loghisto_amd
    rng = np.random.default_rng(123)
    M, n = (16384, 3 << 23) if not id16 else (12000, 3 << 23)
    ids = np.sort(_ids(rng, M, n, 1.0))
    v = rng.lognormal(10, 1.0, n)
    v[-5000:] = 10.0 ** rng.uniform(-3, 18, 5000)             # wide tail inside the table's part
    bad = np.arange(n - 100_000, n - 99_000)                  # ids out of range inside the table's part
    ids_in = ids.copy()
    ids_in[bad] = M + 7
    keep = np.ones(n, bool)
    keep[bad] = False
    d_ids = _dev(torch_cuda, ids_in.astype(np.uint16) if id16 else ids_in)
    d_v = _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit_pairs_device(d_ids, d_v)
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6                              # ids out of range were seen (and skipped)
        c = e.counters()
        assert c["samples_partitioned_v3"] >= n - 8192, sorted(c.items())
        with e.flip() as snap:
            try:
                got = snap.extract(PCTS, M)
            except loghisto_amd.LhError:
                got = snap.extract(PCTS, M)
            check(snap, ids[keep], v[keep], M, got)
        c = e.counters()
        # (a workgroup stops after the first two tiles in a row that overflow: at most a few tiles each, of 12)
        assert 8192 < c["region_overflows"] <= 256 * 4 * 8192, sorted(c.items())
