"""Second generation of the partitioned mixed ingest (loghisto_amd/csrc/lh_kernels_part2.h): one survey per
launch, hot windows sized from the measured spread, 2-byte records, line-granular copy-out.

The survey only decides WHERE a sample is counted (hot LDS window, 2-byte record into a 4 096-bin cold window,
or the exact out-of-window path); every cell must equal the oracle's whatever it estimates.  The engine takes
this path for launches of >= 2^25 pairs over 33 .. 8 192 names; lh_set_option(LH_OPT_PART_V2_MIN_PAIRS) lowers
the bar so that the cases below (a few million pairs, seconds of oracle time) run through it.  Every row of
every case is compared, cell by cell."""
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
        return rng.lognormal(math.log(1e5) + 2e-3 * ids, 1.0)
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
    if kind == "loguniform":                     # 4 147 occupied buckets: ~1 % of the samples miss a 4 096-bin window
        return 10.0 ** rng.uniform(-3, 18, n)
    if kind == "huge":                           # spans the whole key space: most samples miss every window
        return 10.0 ** rng.uniform(-6, 140, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
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
    return v


def _check(snap, ids, v, M, got):
    want = oracle.histogram_pairs_mt(ids, v, M)
    off, keys, counts = snap.buckets_all(M)
    dense = np.zeros((M, N.NKEYS), dtype=np.uint64)
    rows = np.repeat(np.arange(M), np.diff(off.astype(np.int64)))
    dense[rows, oracle.key_to_bin(keys)] = counts
    bad = np.nonzero((dense != want).any(axis=1))[0]
    assert bad.size == 0, f"rows {bad[:8]} differ ({bad.size} rows)"
    assert np.array_equal(got["count"].astype(np.int64), want.sum(axis=1).astype(np.int64))
    for m in np.nonzero(want.sum(axis=1))[0][:: max(1, M // 64)]:
        ref = oracle.process_dense(want[m], PCTS)
        assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m
        assert np.array_equal(got["pkeys"][m], ref["pkeys"]), m


CASES = [
    (1024, 3_000_001, "lognormal", 1.0),     # config 3's shape; odd length
    (1024, 2_500_000, "lognormal", 0.0),     # no skew: (almost) nothing is hot, everything takes the record path
    (1024, 2_000_000, "constant", 1.0),      # 64-bin hot windows, many hot names
    (37, 2_200_000, "allsame", 1.0),         # just above the single-pass kernel's 32 names; one bin per name
    (300, 2_200_000, "signed", 1.5),         # hot windows centred between two lobes: they catch nothing
    (1000, 2_400_000, "loguniform", 1.0),    # cold-window misses: out-of-window table + global atomics
    (200, 1_500_000, "huge", 1.0),
    (2048, 2_600_000, "sigma25", 1.0),       # 8 names per partition: 2 048-bin cold windows
    (5000, 3_000_000, "lognormal", 1.0),     # 20 names per partition: 512-bin cold windows
    (8192, 3_100_001, "edge", 1.0),          # the largest name count of this path
    (33, 1_000_000, "drift", 0.5),
    (1024, 2_000_000, "edge", 1.0),
    (1024, 2_000_000, "kvalues2", 1.0),      # few-valued: every hot name's samples land on two cells
    (1024, 2_000_000, "kvalues16", 1.0),
    (1024, 2_000_000, "bimodal", 1.0),
]


@pytest.mark.parametrize("shape", [0, 1, 2, 3, 6],
                         ids=["1024x256", "512x128", "1024x256-direct", "512x128-direct", "1024x512-direct-wide"])
@pytest.mark.parametrize("M,n,kind,skew", CASES)
def test_survey_path_is_exact(native_lib, torch_cuda, M, n, kind, skew, shape):
    import loghisto_amd
    rng = np.random.default_rng(M * 13 + n)
    ids = _ids(rng, M, n, skew)
    v = _values(rng, kind, ids, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V2_SHAPE, shape)
        for rep in range(2):                     # scratch, survey tables and ranges are reused across launches / epochs
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            c = e.counters()
            assert c["samples_partitioned_v2"] == n * (rep + 1), c
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                _check(snap, ids, v, M, got)


@pytest.mark.parametrize("shape", [0, 1, 2, 3, 6],
                         ids=["1024x256", "512x128", "1024x256-direct", "512x128-direct", "1024x512-direct-wide"])
def test_survey_path_bad_ids_sublaunches_and_two_launches_per_epoch(native_lib, torch_cuda, shape):
    import loghisto_amd
    rng = np.random.default_rng(77)
    M, n = 512, 9_000_001
    ids = _ids(rng, M, n, 1.0)
    v = rng.lognormal(10, 1.2, n)
    bad = ids.copy()
    where = [3, 4_200_000, n - 1]
    bad[where] = [M, 0xFFFFFFFF, M + 5]
    keep = np.ones(n, dtype=bool)
    keep[where] = False
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V2_SHAPE, shape)
        e.set_option(N.OPT_SUBLAUNCH_PAIRS, 1 << 22)       # 3 sub-launches per call, one survey per call
        # (the device arrays must outlive the launches: the engine's stream is not one torch's allocator knows about)
        d_ids, d_bad, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, bad), _dev(torch_cuda, v)
        e.submit_pairs_device(d_ids, d_v)
        e.submit_pairs_device(d_bad, d_v)
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        assert e.counters()["sublaunches"] == 6 and e.counters()["samples_partitioned_v2"] == 2 * n
        with e.flip() as snap:
            try:
                got = snap.extract(PCTS, M)
            except loghisto_amd.LhError:
                got = snap.extract(PCTS, M)
            _check(snap, np.concatenate([ids, ids[keep]]), np.concatenate([v, v[keep]]), M, got)


@pytest.mark.parametrize("shape", [0, 2], ids=["1024x256", "1024x256-direct"])
def test_threshold_fixture_through_the_survey_path(native_lib, torch_cuda, shape):
    """Every bucket threshold, the double below and the double above it (tests/golden/thresholds_x.bin, the file
    real Go is checked against), both signs, through the scatter pass's branch-free bucket index: the guard band
    must hand exactly the right samples to the exact compare."""
    import os
    import sys
    import loghisto_amd
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_thresholds as mt
    x, _, _ = mt.read()
    v = np.concatenate([x, -x, x[::-1]])
    M = 64
    ids = (np.arange(v.size, dtype=np.uint32) * 7) % M
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        e.set_option(N.OPT_PART_V2_SHAPE, shape)
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        assert e.counters()["samples_partitioned_v2"] == v.size
        with e.flip() as snap:
            _check(snap, ids, v, M, snap.extract(PCTS, M))


def test_clustered_stream_falls_back_to_exact_layout(native_lib, torch_cuda):
    """A stream sorted by name: every tile belongs to one or two names, so a cold name's records overflow the LDS
    region the survey sized for its share.  They are still counted exactly (out-of-window path); the engine sees
    the overflow count at the next flip and sends later calls through the exact-layout scatter."""
    import loghisto_amd
    rng = np.random.default_rng(11)
    M, n = 512, 3_000_000
    ids = np.sort(_ids(rng, M, n, 1.0))
    v = rng.lognormal(10, 2.5, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        e.submit_pairs_device(d_ids, d_v)
        e.sync()
        with e.flip() as snap:
            _check(snap, ids, v, M, snap.extract(PCTS, M))
        c = e.counters()
        assert c["region_overflows"] * 50 > n and c["regions_disabled"] == 1, c
        e.submit_pairs_device(d_ids, d_v)
        e.sync()
        with e.flip() as snap:
            _check(snap, ids, v, M, snap.extract(PCTS, M))
        c2 = e.counters()
        assert c2["region_overflows"] == c["region_overflows"] and c2["samples_partitioned_v2"] == 2 * n, c2


def test_idle_intervals_do_not_switch_the_scatter(native_lib, torch_cuda):
    """A random stream leaves a few hundred statistical-tail overflows per launch, and they can be seen one flip after
    the launch's samples were counted: intervals without region-path traffic must not turn the regions off."""
    import loghisto_amd
    rng = np.random.default_rng(12)
    M, n = 512, 3_000_000
    ids = _ids(rng, M, n, 1.0)
    v = rng.lognormal(10, 2.5, n)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        for _ in range(3):
            e.submit_pairs_device(d_ids, d_v)      # not synchronised: the flip comes before the kernel's count
            with e.flip() as snap:
                snap.extract(PCTS, M)
            for _ in range(3):                     # idle intervals collect the late counts
                with e.flip() as snap:
                    snap.extract(PCTS, M)
            assert e.counters()["regions_disabled"] == 0, e.counters()


def test_old_and_new_generation_agree(native_lib, torch_cuda):
    """The same stream through both generations of the partitioned path: identical cells."""
    import loghisto_amd
    rng = np.random.default_rng(5)
    M, n = 700, 2_300_000
    ids = _ids(rng, M, n, 1.0)
    v = rng.lognormal(9, 1.5, n) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    out = []
    for v2 in (0, 1):
        with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
            e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
            e.set_option(N.OPT_PART_V2, v2)
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            assert (e.counters()["samples_partitioned_v2"] == n) == bool(v2)
            with e.flip() as snap:
                out.append(snap.buckets_all(M))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_a_value_shift_under_a_kept_survey_ends_its_reuse(native_lib, torch_cuda):
    """A survey is shared by up to LH_OPT_SURVEY_EVERY calls, and no overflow count moves when only the VALUES of a
    stream shift under it: the hot windows sit where the samples no longer are and every sample becomes a record (1 024
    names, lognormal survey, four-valued stream: 4.34 instead of 3.0 ms per 1e9 pairs for up to 31 calls).  The launches
    therefore compare the share of their pairs the hot windows took with the share of the first launch on the tables
    (stale_judge, lh_kernels_part2.h): the call after a shifted one surveys again.  Exact at every step."""
    import loghisto_amd
    rng = np.random.default_rng(41)
    M, n = 1024, 3_000_000
    ids = _ids(rng, M, n, 1.0)
    v1 = rng.lognormal(math.log(1e5), 1.0, n)
    v2 = _values(rng, "kvalues4", ids, n)                      # bins 691 .. 812: below every window of the first survey
    d_ids, d_v1, d_v2 = _dev(torch_cuda, ids), _dev(torch_cuda, v1), _dev(torch_cuda, v2)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)

        def call(d_v, v):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                _check(snap, ids, v, M, snap.extract(PCTS, M))
            return e.counters()

        for _ in range(4):
            c = call(d_v1, v1)
        assert c["surveys_reused"] == 3 and c["survey_stale_pairs"] == 0, c   # a stationary stream: one survey, nothing stale
        c = call(d_v2, v2)                                    # runs on the lognormal survey: its windows take nothing
        assert c["surveys_reused"] == 4 and c["survey_stale_pairs"] > n // 4, c
        c = call(d_v2, v2)                                    # ... which the launch reported: this call surveys again
        assert c["surveys_reused"] == 4, c
        stale = c["survey_stale_pairs"]
        for _ in range(3):
            c = call(d_v2, v2)
        assert c["surveys_reused"] == 7 and c["survey_stale_pairs"] == stale, c   # and its survey is kept
        c = call(d_v1, v1)                                    # back again: the same
        assert c["survey_stale_pairs"] > stale, c
        before = c["surveys_reused"]
        assert call(d_v1, v1)["surveys_reused"] == before


def test_wide_value_spans_take_the_wide_shape_by_themselves(native_lib, torch_cuda):
    """A two-signed stream over 40 decades (+-10^U(-3, 20): 9 211 bins per name) does not fit the 8 192-bin reduce windows of
    four names per partition: a tenth of its pairs took the window-miss path, global atomics (1 024 names: 7.2 ms per 1e9
    pairs against 2.9 for lognormal values).  The survey reports the spans it saw (k_survey_plan -> the engine's pinned word),
    and the calls that follow run 512 partitions of two names x 16 384 bins (shape 6) -- until a survey reports narrow spans
    again.  Every cell exact at every step; `window_misses` tells the two shapes apart."""
    import loghisto_amd
    rng = np.random.default_rng(91)
    M, n = 1024, 3_000_000
    ids = _ids(rng, M, n, 1.0)
    wide = 10.0 ** rng.uniform(-3, 20, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    narrow = rng.lognormal(math.log(1e5), 1.0, n)
    d_ids, d_w, d_n = _dev(torch_cuda, ids), _dev(torch_cuda, wide), _dev(torch_cuda, narrow)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)

        def call(d_v, v):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                _check(snap, ids, v, M, snap.extract(PCTS, M))
            return e.counters()

        c0 = call(d_w, wide)                       # surveyed on the default shape: reports wide spans
        c1 = call(d_w, wide)                       # the shape changed: surveys again, on 512 partitions
        c2 = call(d_w, wide)
        assert c2["surveys_reused"] == c1["surveys_reused"] + 1, (c1, c2)   # ... and keeps that survey
        assert c0["samples_partitioned_v2"] == n and c2["samples_partitioned_v2"] == 3 * n, c2
        c3 = call(d_n, narrow)                     # narrow values under the wide survey: stale, or simply all in window
        c4 = call(d_n, narrow)
        c5 = call(d_n, narrow)
        assert c5["samples_partitioned_v2"] == 6 * n, c5


@pytest.mark.parametrize("M,shape,kind", [(8192, 2, "lognormal"), (1024, 2, "sigma25"), (1024, 6, "loguniform"), (3000, 0, "lognormal")],
                         ids=["8192-direct", "1024-direct", "1024-wide", "3000-exact-layout"])
def test_a_clustered_call_is_finished_by_the_cell_table(native_lib, torch_cuda, M, shape, kind):
    """Inside ONE call (as at 8 193+ names, tests/test_gpu_part3.py): a region-scatter workgroup whose last two tiles
    overflowed by more than an eighth leaves the rest of its turn to k_scatter_clustered (an LDS table of (name, bin) -> count).
    1e9 pairs sorted by name over 8 192 names took 68 ms on their first call and 42 ms per call on the exact layout after
    it (profiles/r06_first_call.txt).  25 M pairs = 12 tiles per workgroup; every cell exact, bad ids reported and skipped.
    The exact-layout shape (0) has no regions: nothing overflows, the table never runs."""
    import loghisto_amd
    rng = np.random.default_rng(321)
    n = 3 << 23
    ids = np.sort(_ids(rng, M, n, 1.0))
    v = _values(rng, kind, ids, n)
    bad = np.arange(n - 100_000, n - 99_000)
    ids_in = ids.copy()
    ids_in[bad] = M + 7
    keep = np.ones(n, bool)
    keep[bad] = False
    d_ids, d_v = _dev(torch_cuda, ids_in), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V2_SHAPE, shape)
        e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
        e.submit_pairs_device(d_ids, d_v)
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        c = e.counters()
        assert c["samples_partitioned_v2"] == n, sorted(c.items())
        with e.flip() as snap:
            try:
                got = snap.extract(PCTS, M)
            except loghisto_amd.LhError:
                got = snap.extract(PCTS, M)
            _check(snap, ids[keep], v[keep], M, got)
        c = e.counters()
        if shape & 2 and M > 1024:
            assert 8192 < c["region_overflows"] <= 256 * 4 * 8192, sorted(c.items())
        elif shape & 2:     # up to 1 024 names the tiles left to the table are reported too: the engine leaves the path
            assert c["region_overflows"] > n // 4 and c["regions_disabled"] == 1, sorted(c.items())
        else:
            assert c["region_overflows"] == 0, sorted(c.items())
