"""lh_buckets_all: RawMetricSet.Histograms (metrics.go:54-60) for every name in one crossing,
compacted on the device, against the oracle's dense rows."""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,first,count", [(40, 0, 40), (40, 7, 20), (3, 0, 3), (1, 0, 1)])
def test_buckets_all_matches_oracle(native_lib, torch_cuda, M, first, count):
    import loghisto_amd
    rng = np.random.default_rng(M * 31 + first)
    n = 300_000
    ids = rng.integers(0, M, n).astype(np.uint32)
    ids[ids == 2] = 0                      # name 2 (when it exists) stays empty: no map entry
    v = rng.normal(0, 1e3, n) * 10.0 ** rng.integers(0, 12, n)
    want = oracle.histogram_pairs(ids, v, M)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_pairs(ids, v)
        with eng.flip() as snap:
            offsets, keys, counts = snap.buckets_all(count, first)
            per_name = [snap.buckets(first + i) for i in range(count)]
    assert offsets[0] == 0 and offsets[-1] == keys.size == counts.size
    for i in range(count):
        nz = np.nonzero(want[first + i])[0]
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        assert np.array_equal(keys[lo:hi], oracle.bin_to_key(nz)), i
        assert np.array_equal(counts[lo:hi], want[first + i][nz]), i
        assert np.array_equal(per_name[i][0], keys[lo:hi]) and np.array_equal(per_name[i][1], counts[lo:hi])
    if M > 2 and first <= 2 < first + count:
        assert offsets[2 - first] == offsets[3 - first]


def test_buckets_all_empty_snapshot(native_lib, torch_cuda):
    import loghisto_amd
    with loghisto_amd.Engine(max_metrics=8) as eng:
        with eng.flip() as snap:
            offsets, keys, counts = snap.buckets_all(8)
    assert not offsets.any() and keys.size == 0 and counts.size == 0
