"""lh_extract_rows_compact (round 6): the results of many names as count / sum / occupied buckets / selected keys / valid
bits -- 42 B per name at nine percentiles instead of 139 B -- and lh_expand_compact, which derives the full form on the
host.  processHistograms' other outputs are functions of those (metrics.go:349-356: avg = sum / float64(count);
metrics.go:374: uint64(sum); metrics.go:378-385: every percentile value is decompress(key)), so the expansion has to
equal what lh_extract_rows returns for the same snapshot BIT FOR BIT: values, keys, valid flags, uint64(sum), present.
Both extract kernels (wave per name from 2 048 names on, workgroup per name below), rows with totals up to 2^64 - 1,
wide spans, empty rows, invalid percentiles."""
import numpy as np
import pytest

from tests.test_gpu_extract_thresholds import P_A, _rows

pytestmark = pytest.mark.gpu

DEFAULT_P = [0.0, 0.5, 0.75, 0.9, 0.95, 0.99, 0.999, 0.9999, 1.0]   # metrics.go:145-155


def _same(a, b, what):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind == "f":   # bit for bit; two NaNs (avg of an empty row: 0 / 0) are the same result
        nan = np.isnan(a) & np.isnan(b)
        a, b = np.where(nan, 0.0, a).view(np.uint64), np.where(nan, 0.0, b).view(np.uint64)
    assert np.array_equal(a, b), what


@pytest.mark.parametrize("P", [DEFAULT_P, P_A, [0.5], []])
@pytest.mark.parametrize("M", [2400, 300])
def test_compact_plus_host_derivation_equals_extract_rows(native_lib, torch_cuda, P, M):
    torch = torch_cuda
    import loghisto_amd
    from loghisto_amd import merge
    rng = np.random.default_rng(17 + len(P) + M)
    rows = _rows(rng, M)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_device(0, torch.ones(8, dtype=torch.float64, device="cuda"))
        snap = eng.flip()
        t_rows, _ = merge.snapshot_tensors(snap, M)
        t_rows[0].zero_()
        for m, r in enumerate(rows):
            if not r:
                continue
            lo, hi = min(r), max(r)
            w = np.zeros(hi - lo + 1, dtype=np.uint64)
            for b, c in r.items():
                w[b - lo] = c
            t_rows[m, lo:hi + 1] = torch.from_numpy(w.view(np.int64)).cuda()
            snap.mark_dirty(m, 1, lo, hi)
        torch.cuda.synchronize()
        for first, n in ((0, M), (7, M - 20)):
            full = snap.extract(P, n, first=first)
            c = snap.extract_compact(P, n, first=first)
            # the compact arrays themselves
            _same(c["count"], full["count"], "count")
            _same(c["sum"], full["sum"], "sum")
            _same(c["nbuckets"], full["nbuckets"], "nbuckets")
            if P:
                bits = (c["pvalid_bits"][:, None] >> np.arange(len(P), dtype=np.uint32)[None, :]) & 1
                _same(bits.astype(np.uint8), full["pvalid"], "valid bits")
                _same(c["pkeys"], full["pkeys"], "keys")
            # ... and what the host derives from them
            ex = snap.expand_compact(c)
            for k in ("count", "sum", "avg", "agg_sum_add", "nbuckets", "present", "pvals", "pkeys", "pvalid"):
                _same(ex[k], full[k], k)
            assert np.isnan(ex["avg"][full["count"] == 0]).all()
        snap.release()


def test_compact_after_an_ingested_interval(native_lib, torch_cuda):
    """The same through the ingest path: 4 096 names, a mixed stream with negative values and an id that stays empty."""
    torch = torch_cuda
    import loghisto_amd
    M, n = 4096, 1 << 22
    g = torch.Generator(device="cuda").manual_seed(5)
    ids = torch.randint(0, M - 1, (n,), generator=g, device="cuda", dtype=torch.int32)   # name M - 1 never appears
    v = torch.randn(n, generator=g, device="cuda", dtype=torch.float64) * 1e4
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_pairs_device(ids, v, n)
        snap = eng.flip()
        full = snap.extract(DEFAULT_P, M)
        ex = snap.expand_compact(snap.extract_compact(DEFAULT_P, M))
        snap.release()
    assert int(full["count"].sum()) == n and full["present"][M - 1] == 0
    for k in ("count", "sum", "avg", "agg_sum_add", "nbuckets", "present", "pvals", "pkeys", "pvalid"):
        _same(ex[k], full[k], k)
