"""The arithmetic behind k_extract_wave's percentile search, on the CPU: metrics.go:413's test
float64(sofar) / float64(total) >= p is monotone in sofar, so a percentile is "the first bin whose inclusive prefix count
reaches T", T = min{s in [1, total] : float64(s) / float64(total) >= p}.  This file restates pct_threshold
(loghisto_amd/csrc/lh_kernels.hip: ceil(p * total) checked and corrected with IEEE divides, bisection when that does not
settle) in numpy float64 and holds it against the oracle's per-bucket loop on the rows of
tests/test_gpu_extract_thresholds.py -- totals of 1 .. 10, 2^32, 2^53 +- 1, 2^63, 2^64 - 1, percentiles on, beside and
outside the quotients.  The kernel itself is held against the same oracle by that GPU test."""
import importlib.util
import os

import numpy as np

import oracle

_spec = importlib.util.spec_from_file_location(
    "_thr_rows", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_extract_thresholds.py"))
_rows_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_rows_mod)


def _reached(s, ft, p):
    return np.float64(s) / ft >= p


def pct_threshold(p, total):
    """None: no prefix reaches p (p > 1 or NaN)."""
    if not (1.0 >= p):
        return None
    if p <= 0.0:
        return 1
    ft = np.float64(total)
    est = np.float64(p) * ft
    s = total if est >= 18446744073709549568.0 else int(est)
    if np.float64(s) < est:
        s += 1
    s = max(1, min(total, s))
    if _reached(s, ft, p) and (s == 1 or not _reached(s - 1, ft, p)):
        return s
    it = 0
    while it < 4 and s > 1 and _reached(s - 1, ft, p):
        s -= 1
        it += 1
    it = 0
    while it < 4 and s < total and not _reached(s, ft, p):
        s += 1
        it += 1
    if _reached(s, ft, p) and (s == 1 or not _reached(s - 1, ft, p)):
        return s
    lo, hi = 0, total
    while hi - lo > 1:
        mid = lo + (hi - lo) // 2
        if _reached(mid, ft, p):
            hi = mid
        else:
            lo = mid
    return hi


def test_threshold_search_equals_the_per_bucket_loop():
    rng = np.random.default_rng(32)
    rows = _rows_mod._rows(rng, 360)
    P = _rows_mod.P_A
    checked = quick = 0
    for r in rows:
        if not r:
            continue
        dense = np.zeros(65536, dtype=np.uint64)
        for b, c in r.items():
            dense[b] = c
        total = int(sum(r.values()))
        ref = oracle.process_dense(dense, P)
        bins = sorted(r)
        pre = np.cumsum([r[b] for b in bins], dtype=object)
        for i, p in enumerate(P):
            T = pct_threshold(p, total)
            if T is None:
                assert ref["pvalid"][i] == 0, (p, total)
            else:
                j = next(k for k, x in enumerate(pre) if x >= T)
                assert ref["pvalid"][i] == 1 and int(ref["pkeys"][i]) == oracle.bin_to_key(bins[j]), (p, total, T)
                # T is minimal: one less does not reach p
                assert _reached(T, np.float64(total), p) and (T == 1 or not _reached(T - 1, np.float64(total), p))
            checked += 1
    assert checked > 9000
