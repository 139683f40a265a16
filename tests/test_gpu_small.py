"""Mixed (id, value) ingest with a handful of names: the single-pass LDS kernel
(lh_kernels_small.hip) against the oracle, bit-exact."""
import math

import numpy as np
import pytest

from tests.conftest import thresholds_until_round_6

import oracle

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("M,n,kind", [
    (1, 300_001, "lognormal"),
    (3, 1_000_003, "lognormal"),
    (4, 2_000_000, "constant"),          # wave-uniform path, one cell per name
    (16, 1_500_001, "lognormal"),        # 1 024-bin windows
    (16, 800_000, "signed_wide"),        # far more keys than the windows hold: global-atomic path
    (8, 65_536, "loguniform"),           # exactly the kernel's threshold
    (8, 65_535, "loguniform"),           # one below: direct-atomic kernel
    (5, 400_000, "shifting"),            # distribution moves after the first tile: windows misplaced, still exact
    (17, 1_200_001, "lognormal"),        # 128 KiB variant (1 024 threads): 1 024-bin windows
    (32, 900_000, "constant"),
    (32, 1_500_000, "signed_wide"),      # mostly outside the windows
    (31, 700_001, "shifting"),
    (33, 1_000_000, "lognormal"),        # one above the single-pass limit: partitioned path
])
def test_small_name_count_ingest(native_lib, torch_cuda, M, n, kind):
    import loghisto_amd
    rng = np.random.default_rng(M * 13 + n)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    if kind == "lognormal":
        v = rng.lognormal(math.log(1e5) + 0.5 * ids, 1.0)
    elif kind == "constant":
        v = 100.0 + ids.astype(np.float64)
    elif kind == "signed_wide":
        v = rng.normal(0, 1e3, n) * 10.0 ** rng.integers(0, 80, n)
        v[::997] = math.nan
    elif kind == "loguniform":
        v = 10.0 ** rng.uniform(-3, 18, n)
    else:
        v = rng.lognormal(math.log(1e3), 0.3, n)
        v[n // 50:] *= 1e9
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
            rows = [snap.dense_row(m) for m in range(M)]
    want = oracle.histogram_pairs(ids, v, M)
    assert int(got["count"].sum()) == n
    for m in range(M):
        assert np.array_equal(rows[m], want[m]), m
        ref = oracle.process_dense(want[m], PCTS)
        assert int(got["count"][m]) == ref["count"]
        if ref["count"]:
            assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m


def test_small_path_reports_bad_ids(native_lib, torch_cuda):
    import loghisto_amd
    rng = np.random.default_rng(2)
    M, n = 4, 500_000
    ids = rng.integers(0, M, n).astype(np.uint32)
    v = rng.lognormal(5, 1, n)
    bad = ids.copy()
    bad[[1, 250_000, n - 1]] = [M, 0xFFFFFFFF, 77]
    keep = np.ones(n, dtype=bool)
    keep[[1, 250_000, n - 1]] = False
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.submit_pairs_device(_dev(torch_cuda, bad), _dev(torch_cuda, v))
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        with e.flip() as snap:
            rows = np.stack([snap.dense_row(m) for m in range(M)])
    assert np.array_equal(rows, oracle.histogram_pairs(ids[keep], v[keep], M))


def test_adaptive_dispatch_and_counters(native_lib, torch_cuda):
    """16 names whose spans (4 146 buckets) dwarf the 1 024-bin windows: the first interval runs through the
    single-pass kernel and reports its window misses; from then on the engine uses the partitioned path.
    Results are exact either way."""
    import loghisto_amd
    rng = np.random.default_rng(7)
    M, n = 16, 600_000
    ids = rng.integers(0, M, n).astype(np.uint32)
    v = 10.0 ** rng.uniform(-3, 18, n)
    want = oracle.histogram_pairs(ids, v, M)
    d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        thresholds_until_round_6(e)
        for interval in range(3):
            e.submit_pairs_device(d_ids, d_v)
            with e.flip() as snap:
                got = snap.extract(PCTS, M)
                assert np.array_equal(snap.dense_row(3), want[3])
            assert int(got["count"].sum()) == n
            c = e.counters()
            assert c["flips"] == interval + 1 and c["extracts"] >= interval + 1
        assert c["samples_small"] == n                 # only the first interval
        assert c["samples_partitioned"] == 2 * n       # the two after it
        assert c["small_path_disabled"] == 1 and c["window_misses"] > n // 50
    # a well-behaved stream keeps the single-pass kernel
    v2 = rng.lognormal(math.log(1e5), 1.0, n)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        for _ in range(2):
            e.submit_pairs_device(d_ids, _dev(torch_cuda, v2))
            with e.flip() as snap:
                snap.extract(PCTS, M)
        c = e.counters()
        assert c["samples_small"] == 2 * n and c["small_path_disabled"] == 0
