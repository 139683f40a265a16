"""Hot-name windows inside the partition scatter (k_scatter_samples<true>, lh_kernels_part.hip): samples of a
workgroup's most frequent names are counted in LDS windows instead of becoming records.  Which names are hot
is a per-workgroup heuristic (first tile, hashed heavy-hitter table above 2 048 names); the result must be
bit-exact whatever it picks.  lh_set_option(LH_OPT_HOT_MIN_TILES, 1) turns the path on for inputs of a few million samples
(the engine itself only uses it for launches of >= 67 M samples)."""
import math

import numpy as np
import pytest

from tests.conftest import thresholds_until_round_6

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .9, .99, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def _ids(rng, M, n, skew):
    w = np.arange(1, M + 1, dtype=np.float64) ** -skew
    perm = rng.permutation(M)                       # hot names anywhere in the id space, not only 0..15
    return perm[rng.choice(M, size=n, p=w / w.sum())].astype(np.uint32)


@pytest.mark.parametrize("M,n,kind,skew", [
    (1024, 3_000_001, "lognormal", 1.0),     # config 3's shape; odd length
    (37, 2_500_000, "constant", 1.0),        # few names (just above the single-pass kernel's 32) + same-bin contention
    (5000, 3_000_000, "signed", 1.0),        # hashed selection (> 2 048 names); two clusters of bins per name
    (65536, 4_000_000, "lognormal", 1.0),    # hot windows + second partition level
    (300, 2_200_000, "wide", 1.5),           # hot names whose samples mostly miss their 512-bin window
    (1024, 2_200_000, "lognormal", 0.0),     # no skew at all
    (20000, 2_300_000, "edge", 1.0),
])
def test_hot_windows_are_exact(native_lib, torch_cuda, M, n, kind, skew, monkeypatch):
    import loghisto_amd
    rng = np.random.default_rng(M * 7 + n)
    ids = _ids(rng, M, n, skew)
    if kind == "lognormal":
        v = rng.lognormal(math.log(1e5) + 1e-4 * ids, 1.0)
    elif kind == "constant":
        v = 1000.0 + (ids % 3)
    elif kind == "signed":
        v = rng.normal(0, 1e4, n)
    elif kind == "wide":
        v = 10.0 ** rng.uniform(-3, 18, n) * np.where(rng.random(n) < 0.3, -1.0, 1.0)
    else:
        v = rng.lognormal(math.log(1e5), 1.0, n)
        hot = int(np.bincount(ids, minlength=M).argmax())
        sel = np.nonzero(ids == hot)[0]
        v[sel[:3000]] = 2.0196e142               # key +32767 on a hot name: far outside its window
        v[sel[3000:6000]] = -2.0196e142
        v[sel[6000:6100]] = float("nan")
        v[sel[6100:6200]] = float("inf")
        v[sel[6200:6300]] = 0.0
        ids[:4000] = M - 1                       # the last name becomes frequent in the first tile
    counts = np.bincount(ids, minlength=M)
    order = np.argsort(-counts)
    sample = sorted({int(order[0]), int(order[1]), int(order[7]), int(order[15]), int(order[16]), int(order[40 % M]),
                     int(order[M // 2]), int(order[-1]), 0, M - 1})
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        thresholds_until_round_6(e)
        e.set_option(N.OPT_HOT_MIN_TILES, 1)
        e.set_option(N.OPT_HOT_WINDOWS, 1)
        e.set_option(N.OPT_PART_V3, 0)      # this file is about the FIRST generation's hot windows (above 8 192 names the third takes >= 2^18 pairs)
        for rep in range(2):                     # scratch, ranges and windows are reused across launches and epochs
            e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
            e.sync()
            assert e.counters()["samples_partitioned"] == n * (rep + 1)
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                rows = {m: snap.dense_row(m) for m in sample}
            assert np.array_equal(got["count"].astype(np.int64), counts)
            for m in sample:
                want = oracle.histogram_dense(v[ids == m])
                assert np.array_equal(rows[m], want), (rep, m)
                ref = oracle.process_dense(want, PCTS)
                if ref["count"]:
                    assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m


def test_hot_windows_with_bad_ids_and_two_launches_per_epoch(native_lib, torch_cuda, monkeypatch):
    import loghisto_amd
    rng = np.random.default_rng(99)
    M, n = 512, 2_400_000
    ids = _ids(rng, M, n, 1.0)
    v = rng.lognormal(10, 1.2, n)
    bad = ids.copy()
    bad[[3, 1_000_000, n - 1]] = [M, 0xFFFFFFFF, M + 5]
    keep = np.ones(n, dtype=bool)
    keep[[3, 1_000_000, n - 1]] = False
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_V3, 0)
        e.set_option(N.OPT_HOT_MIN_TILES, 1)
        # (the device arrays must outlive the launches: the engine's stream is not one torch's allocator knows about)
        d_ids, d_bad, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, bad), _dev(torch_cuda, v)
        e.submit_pairs_device(d_ids, d_v)      # two launches into one epoch:
        e.submit_pairs_device(d_bad, d_v)      # windows flush twice into the same rows
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        with e.flip() as snap:
            try:
                got = snap.extract([0.5], M)
            except loghisto_amd.LhError:
                got = snap.extract([0.5], M)
            want = np.bincount(ids, minlength=M) + np.bincount(ids[keep], minlength=M)
            assert np.array_equal(got["count"].astype(np.int64), want)
            hot = int(np.bincount(ids, minlength=M).argmax())
            row = oracle.histogram_dense(v[ids == hot]) + oracle.histogram_dense(v[keep & (ids == hot)])
            assert np.array_equal(snap.dense_row(hot), row)
