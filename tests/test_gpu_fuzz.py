"""Seeded differential fuzz of the mixed ingest against the oracle: random name counts, name skews, value
distributions, launch splits (device and host submits interleaved), with the hot-name windows and the second
partition level forced on for small inputs in half of the cases.  Every case checks per-name counts for all
names and bit-exact rows for a sample of names (the hottest, random ones, the first and the last)."""
import math
import os

import numpy as np
import pytest

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .99, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


def _values(rng, kind, n, ids):
    if kind == 0:
        return rng.lognormal(math.log(1e5) + 1e-4 * (ids % 977), 1.0)
    if kind == 1:
        return rng.normal(0, 10.0 ** rng.uniform(0, 9), n)
    if kind == 2:
        return np.full(n, float(rng.choice([0.0, 1.0, 123.0, -5e6, 2.0196e142])))
    if kind == 3:
        return 10.0 ** rng.uniform(-6, 30, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    if kind == 4:
        return rng.exponential(1e6, n)
    v = rng.lognormal(10, 2.5, n)
    k = max(1, n // 50)
    v[rng.integers(0, n, k)] = rng.choice([np.nan, np.inf, -np.inf, 0.0, -0.0, 5e-324, 1.7976931348623157e308], k)
    return v


@pytest.mark.parametrize("case", range(int(os.environ.get("LH_FUZZ_CASES", "24"))))  # LH_FUZZ_CASES=400: a longer one-off run
def test_mixed_ingest_fuzz(native_lib, torch_cuda, case, monkeypatch):
    import loghisto_amd
    rng = np.random.default_rng(1000 + case)
    M = int(rng.choice([1, 2, 3, 16, 17, 40, 255, 256, 257, 1000, 2048, 2049, 5000, 8193, 20000, 65536]))
    n = int(rng.integers(200_000, 3_000_000))
    skew = float(rng.choice([0.0, 0.5, 1.0, 1.5]))
    two_level_above = int(rng.choice([0, 4, 32]))
    w = np.arange(1, M + 1, dtype=np.float64) ** -skew
    perm = rng.permutation(M)
    ids = perm[rng.choice(M, size=n, p=w / w.sum())].astype(np.uint32)
    v = _values(rng, case % 6, n, ids)
    # split into 1..4 launches of uneven, sometimes odd / unaligned lengths; some go through the host path
    cuts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, int(rng.integers(0, 4)))]))
    counts = np.bincount(ids, minlength=M)
    order = np.argsort(-counts)
    sample = sorted({int(order[0]), int(order[min(3, M - 1)]), int(order[min(15, M - 1)]), int(order[min(16, M - 1)]),
                     int(order[-1]), 0, M - 1} | {int(x) for x in rng.integers(0, M, 3)})
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    torch_cuda.cuda.synchronize()
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=1 << 18) as e:
        if case % 2:
            e.set_option(N.OPT_HOT_MIN_TILES, 1)
            e.set_option(N.OPT_TWO_LEVEL_ABOVE, two_level_above)
        if case % 3:                         # survey + 2-byte records: region scatter (2, 3) or exact layout (0, 1)
            e.set_option(N.OPT_PART_V2_MIN_PAIRS, 1 << 17)
            e.set_option(N.OPT_PART_V2_SHAPE, [2, 3, 0, 1][(case // 3) % 4])
        for a, b in zip(cuts[:-1], cuts[1:]):
            if rng.random() < 0.25 and b - a < 600_000:
                e.submit_pairs(ids[a:b], v[a:b])
            else:
                # slicing the device tensors keeps odd offsets: the launcher must fall back when unaligned
                e.submit_pairs_device(d_ids[a:b], d_v[a:b], b - a)
        e.sync()
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
            rows = {m: snap.dense_row(m) for m in sample}
    assert np.array_equal(got["count"].astype(np.int64), counts), (case, M, n)
    for m in sample:
        want = oracle.histogram_dense(v[ids == m])
        assert np.array_equal(rows[m], want), (case, M, n, m)
        ref = oracle.process_dense(want, PCTS)
        if ref["count"]:
            assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), (case, m)
