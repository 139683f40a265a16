"""k_extract_wave (>= 2 048 names) finds a percentile as the first bin whose prefix count reaches an INTEGER threshold
T = min{s : float64(s) / float64(total) >= p} (pct_threshold, lh_kernels.hip) instead of evaluating metrics.go:413's
quotient at every bucket.  Rows are written straight into a snapshot (lh_snapshot_rows: what a peer's merged cells would
be) so that totals the ingest cannot reach in a test -- 2^32, 2^53 +- 1, 2^63, 2^64 - 1 -- and percentiles that sit ON a
quotient k / total are covered; every value, key and "omitted" flag against the oracle's per-bucket loop
(metrics.go:406-418 restated), for the in-register path (spans <= 1 024 bins), the wide path, and the block-per-metric
kernel (< 2 048 names: the three must agree bit for bit)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

# <= 32 percentiles per call (K2_MAXP): on quotients of small totals, one ulp either side, the ends, invalid ones
P_A = [0.0, -1.0, 1.0, float(np.nextafter(1.0, 2.0)), float("nan"), 0.5, 1 / 3, 2 / 3, 0.1, 0.9, 0.99, 0.999, 1e-300,
       float(np.nextafter(1.0, 0.0)), 0.2, 0.3, 0.7, float(np.nextafter(0.3, 1.0)), float(np.nextafter(0.3, 0.0)),
       float(np.nextafter(0.5, 1.0)), float(np.nextafter(0.5, 0.0)), 0.25, 0.75, 0.9999, 1 / 7, 3 / 7, 6 / 7, 0.01,
       2.0, float("inf"), -float("inf"), 5e-324]


def _rows(rng, M):
    """M sparse rows: {bin: count} with the totals and window shapes that stress the threshold search."""
    rows = []
    big = [2 ** 32 - 1, 2 ** 32, 2 ** 32 + 1, 2 ** 53 - 1, 2 ** 53, 2 ** 53 + 1, 2 ** 63, 2 ** 64 - 1]
    for m in range(M):
        kind = m % 13
        lo = int(rng.integers(0, 65536 - 1100))
        if kind == 0:      # a handful of cells, total 1 .. 10: every p in P_A lands on or beside a quotient
            k = int(rng.integers(1, 6))
            bins = np.sort(rng.choice(np.arange(lo, lo + 40), size=k, replace=False))
            r = {int(b): int(rng.integers(1, 3)) for b in bins}
        elif kind == 1:    # total 100 / 1000 in ten equal cells: k / total == 0.1, 0.2, ... exactly representable or not
            step = int(rng.integers(1, 100))
            r = {lo + i * step: 10 ** int(rng.integers(1, 4)) for i in range(10)}
        elif kind == 2:    # one cell
            r = {lo: int(rng.integers(1, 2 ** 40))}
        elif kind == 3:    # huge totals, exact: cells that add up to one of `big`
            t = big[(m // 13) % len(big)]
            k = int(rng.integers(1, 5))
            parts = sorted(int(x) for x in rng.integers(1, max(2, min(t, 2 ** 62)), size=k - 1)) if k > 1 else []
            parts = [p for p in parts if 0 < p < t]
            cuts = [0] + sorted(set(parts)) + [t]
            cs = [b - a for a, b in zip(cuts, cuts[1:]) if b > a]
            bins = np.sort(rng.choice(np.arange(lo, lo + 900), size=len(cs), replace=False))
            r = {int(b): int(c) for b, c in zip(bins, cs)}
        elif kind == 4:    # dense window, < 1 024 bins (in-register path), small counts
            w = int(rng.integers(2, 1024))
            c = rng.integers(0, 4, w)
            r = {lo + i: int(c[i]) for i in range(w) if c[i]}
        elif kind == 5:    # dense window, 1 025 .. 5 000 bins (wide path)
            lo = int(rng.integers(0, 65536 - 5100))
            w = int(rng.integers(1025, 5000))
            c = rng.integers(0, 3, w)
            r = {lo + i: int(c[i]) for i in range(w) if c[i]}
        elif kind == 6:    # the top of the key space: the last lanes' 4-bin groups end beyond bin 65 535
            w = int(rng.integers(1, 700))
            c = rng.integers(0, 50, w)
            r = {65535 - i: int(c[i]) + (1 if i == 0 else 0) for i in range(w) if c[i] or i == 0}
        elif kind == 7:    # the bottom
            w = int(rng.integers(1, 700))
            c = rng.integers(0, 50, w)
            r = {i: int(c[i]) + (1 if i == 0 else 0) for i in range(w) if c[i] or i == 0}
        elif kind == 8:    # exactly 1 024 bins (the widest in-register span) and 1 025 (the narrowest wide one)
            w = 1024 + (m // 13) % 2
            r = {lo: 3, lo + w - 1: 5, lo + w // 2: 1}
        elif kind == 9:    # one dominant cell among many (prefix jumps over several thresholds at once)
            w = int(rng.integers(10, 1000))
            r = {lo + i: 1 for i in range(0, w, 7)}
            r[lo + w // 3] = 10 ** 9
        elif kind == 10:   # lognormal-like counts, up to 2^36 per cell
            w = int(rng.integers(50, 1000))
            c = (rng.lognormal(10, 4, w)).astype(np.uint64) % (2 ** 36)
            r = {lo + i: int(c[i]) for i in range(w) if c[i]}
        elif kind == 11:   # a WIDE span (the two-pass loop, 64-bit prefix counts) whose cells add up to one of `big`
            lo = int(rng.integers(0, 65536 - 9100))
            t = big[(m // 13) % len(big)]
            k = int(rng.integers(2, 40))
            cuts = [0] + sorted(set(int(x) for x in rng.integers(1, max(2, min(t, 2 ** 62)), size=k - 1) if 0 < x < t)) + [t]
            cs = [b - a for a, b in zip(cuts, cuts[1:]) if b > a]
            bins = np.sort(rng.choice(np.arange(lo, lo + int(rng.integers(1100, 9000))), size=len(cs), replace=False))
            bins[-1] = max(bins[-1], lo + 1100)   # (wider than the in-register path whatever was drawn)
            r = {int(b): int(c) for b, c in zip(np.sort(bins), cs)}
        else:              # empty row
            r = {}
        rows.append(r)
    return rows


@pytest.mark.parametrize("P", [P_A, [0.5], [0.0, 0.5, 0.9, 0.99, 0.999, 1.0, 0.75, 0.95, 0.9999]])
def test_percentile_thresholds_against_the_per_bucket_loop(native_lib, torch_cuda, P):
    torch = torch_cuda
    import loghisto_amd
    from loghisto_amd import merge
    rng = np.random.default_rng(len(P))
    M = 2400
    rows = _rows(rng, M)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_device(0, torch.ones(8, dtype=torch.float64, device="cuda"))   # (an interval has to hold something)
        snap = eng.flip()
        t_rows, _ = merge.snapshot_tensors(snap, M)
        t_rows[0].zero_()
        dense = np.zeros((M, 65536), dtype=np.uint64)
        for m, r in enumerate(rows):
            for b, c in r.items():
                dense[m, b] = c
        # upload the occupied windows only
        for m, r in enumerate(rows):
            if not r:
                continue
            lo, hi = min(r), max(r)
            t_rows[m, lo:hi + 1] = torch.from_numpy(dense[m, lo:hi + 1].view(np.int64)).cuda()
            snap.mark_dirty(m, 1, lo, hi)
        torch.cuda.synchronize()
        got = snap.extract(P, M)                                    # wave per metric
        sub = snap.extract(P, 1500, first=100)                      # < 2 048 names: block per metric
        snap.release()
    for k in ("count", "sum", "pvals", "pkeys", "pvalid", "nbuckets"):
        a, b = np.ascontiguousarray(got[k][100:1600]), np.ascontiguousarray(sub[k])
        if a.dtype.kind == "f":
            a, b = a.view(np.uint64), b.view(np.uint64)
        assert np.array_equal(a, b), k                              # the two kernels agree bit for bit
    for m in range(M):
        ref = oracle.process_dense(dense[m], P)
        assert int(got["count"][m]) == ref["count"] and int(got["nbuckets"][m]) == ref["nbuckets"], m
        assert np.array_equal(got["pvalid"][m], ref["pvalid"]), (m, rows[m] if len(rows[m]) < 8 else len(rows[m]))
        assert np.array_equal(got["pkeys"][m], ref["pkeys"]), (m, got["pkeys"][m], ref["pkeys"])
        assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m
