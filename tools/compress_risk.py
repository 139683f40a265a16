#!/usr/bin/env python3
"""How much of the input space is sensitive to WHICH few-ulp log computes compress()?  (the
oracle defines bucket parity as Go's math/log.go algorithm; no reference vector pins it at ulp granularity and Go cannot
run in this image.)  CPU only.  Compares the oracle's thresholds T[j] (x = 1 + |v| space) with the thresholds another
accurate log -- glibc's, through numpy -- would give, counts the float64 values whose bucket differs, and integrates the
C2 stream's density over them.  python tools/compress_risk.py"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402


def main():
    tx = oracle.thresholds()
    J = 32767
    T = tx[1:J + 1].copy()
    j = np.arange(1, J + 1)
    bits = T.view(np.uint64).astype(np.int64)
    first = np.full(J, 99, dtype=np.int64)           # offset (ulps of x) of the other log's threshold from the oracle's
    for d in range(-8, 9):
        x = (bits + d).astype(np.uint64).view(np.float64)
        ge = np.floor(100.0 * np.log(x) + 0.5).astype(np.int64) >= j
        first = np.where((first == 99) & ge, d, first)
    moved = first != 0
    off = np.abs(first[moved])
    Tm = T[moved]
    ulpx = np.spacing(Tm)
    v = Tm - 1.0
    nv = np.maximum(1.0, ulpx / np.spacing(np.maximum(v, 5e-324))) * off * 2
    mu, sg = math.log(1e5), 1.0
    pdf = np.exp(-(np.log(np.maximum(v, 1e-300)) - mu) ** 2 / (2 * sg * sg)) / (np.maximum(v, 1e-300) * sg * math.sqrt(2 * math.pi))
    p1 = float((pdf * ulpx * off).sum())
    width = np.diff(tx[1:J + 2])[moved]
    print(f"thresholds that move between the oracle (Go's log) and glibc's log: {int(moved.sum())} of {J}, by at most "
          f"{int(off.max())} ulp of x")
    print(f"float64 x = 1 + |v| whose bucket depends on the log: {int(off.sum())}; float64 v behind them (both signs): {nv.sum():.3g}")
    print(f"C2 stream, v ~ lognormal(ln 1e5, 1): P(a sample is sensitive) = {p1:.3g}; expected sensitive samples in 1e9: "
          f"{p1 * 1e9:.3g}; P(at least one) = {1 - math.exp(-p1 * 1e9):.3g}")
    print(f"with the survey's wider band (+-3 ulp at every one of them): P(at least one in 1e9) <= {3 * p1 * 1e9:.3g}")
    print(f"any stream that is smooth at the 1 % bucket scale: a sensitive bucket's sensitive share is <= {float((ulpx * off / width).max()):.3g} "
          f"of its width ({moved.mean():.3f} of the buckets are affected)")


if __name__ == "__main__":
    main()
