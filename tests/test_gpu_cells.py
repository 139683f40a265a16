"""The direct path since round 6: launch_ingest_pairs_cells (lh_kernels_part2.h) -- whole tiles through a per-workgroup LDS
table of (name << 16 | bin) -> count (k_scatter_clustered on its own, three configurations by call size), the pairs behind the
last whole tile one global atomic each.  It takes every mixed call too small for a partitioned path (< 2^20 pairs up to 8 192
names, < 3 * 2^20 above), misaligned calls, and calls whose scratch block cannot be had.  Exact like every path
(metrics.go:273-295); what it exists for is the stream that falls into FEW cells, where one atomic per sample serialises
(12 ns per sample on one cell: profiles/r06_small_calls.txt)."""
import numpy as np
import pytest

from loghisto_amd import _native as N
from tests.test_gpu_part3 import PCTS, _dev, _ids, _values, check

pytestmark = pytest.mark.gpu

# names, pairs, values, id skew: sizes on both sides of the kernel's three configurations (1 024-pair tiles + 4 096 slots below
# 2^17 pairs, 1 024-pair tiles + 16 384 slots up to 2 * CUs tiles of 8 192, 8 192-pair tiles above) and of the tile sizes
CASES = [
    (1000, 999, "lognormal", 1.0),            # less than one tile: plain atomics
    (1000, 1024, "lognormal", 1.0),
    (1000, 5_001, "edge", 1.0),
    (65536, 70_000, "lognormal", 1.0),
    (300, (1 << 17) - 1, "kvalues2", 1.0),
    (300, (1 << 17) + 1025, "loguniform", 1.0),
    (8192, 900_001, "lognormal", 0.0),        # no skew: nearly every sample its own cell, the table empties every other tile
    (40000, 3_000_001, "sigma25", 1.0),       # the largest default direct call above 8 192 names
    (1024, (1 << 22) + 8191 + 1024, "lognormal", 1.0),   # 8 192-pair tiles (forced here: see the option below)
    (65536, (1 << 22) + 5, "huge", 1.5),
]


@pytest.mark.parametrize("M,n,kind,skew", CASES)
@pytest.mark.parametrize("id16", [False, True])
def test_cell_table_is_exact(native_lib, torch_cuda, M, n, kind, skew, id16):
    import loghisto_amd
    rng = np.random.default_rng(M + n)
    ids = _ids(rng, M, n, skew)
    v = _values(rng, kind, ids, n)
    d_ids = _dev(torch_cuda, ids.astype(np.uint16) if id16 else ids)
    d_v = _dev(torch_cuda, v)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        e.set_option(N.OPT_PART_MIN_PAIRS, 1 << 30)        # every call of this file: the direct path
        e.submit_pairs_device(d_ids, d_v)
        e.submit_pairs_device(d_ids[2:], d_v[2:])          # a second launch into the same interval (cells add up)
        e.sync()
        c = e.counters()
        assert c["samples_direct"] == 2 * n - 2 and c["samples_partitioned"] == 0 and c["scratch_bytes"] == 0, sorted(c.items())
        with e.flip() as snap:
            check(snap, np.concatenate([ids, ids[2:]]), np.concatenate([v, v[2:]]), M, snap.extract(PCTS, M))


def test_streams_that_fall_into_few_cells(native_lib, torch_cuda):
    """All pairs on one cell, on one name, on few cells per name: the counts are exact (up to 3e6 in one cell here), bad ids are
    reported and skipped, and a misaligned call (no vector loads: the direct path whatever its size) goes the same way."""
    import loghisto_amd
    rng = np.random.default_rng(4)
    M, n = 5000, 3_000_000
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        for which in range(4):
            if which == 0:
                ids, v = np.full(n, 77, np.uint32), np.full(n, 123.0)
            elif which == 1:
                ids, v = np.full(n, M - 1, np.uint32), rng.lognormal(10, 1.0, n)
            elif which == 2:
                ids, v = _ids(rng, M, n, 1.0), 1000.0 + (rng.integers(0, 3, n) * 100.0)
            else:
                ids, v = np.sort(_ids(rng, M, n, 1.0)), rng.lognormal(10, 0.3, n)
            bad = np.arange(1000, 1100)
            ids_in = ids.copy()
            ids_in[bad] = M + which
            keep = np.ones(n, bool)
            keep[bad] = False
            d_ids, d_v = _dev(torch_cuda, ids_in), _dev(torch_cuda, v)
            # [1:] of the ids only: the two arrays' alignments differ -> nothing to peel, the whole call is direct
            e.submit_pairs_device(d_ids[1:], d_v[:-1].clone())
            with pytest.raises(loghisto_amd.LhError) as ei:
                e.sync()
            assert ei.value.code == 6
            with e.flip() as snap:
                try:
                    got = snap.extract(PCTS, M)
                except loghisto_amd.LhError:
                    got = snap.extract(PCTS, M)
                check(snap, ids[1:][keep[1:]], v[:-1][keep[1:]], M, got)
        c = e.counters()
        assert c["samples_direct"] == 4 * (n - 1) and c["samples_partitioned"] == 0, sorted(c.items())
