"""Writes tests/golden/thresholds_x.bin: the fixture that closes the `compress` pin on any box with Go.

`compress` (/root/reference/metrics.go:316-322) is pinned by the reference's own tests only to 1 %
(TestCompress, metrics_test.go:151-172); at threshold/ulp granularity the oracle DEFINES parity as
"math/log.go algorithm, amd64 non-fused evaluation" (DESIGN.md section 2).  No Go toolchain exists in this
image, so the claim "the oracle's compress == Go's compress at every threshold" is handed off as data:

  for every extended key j = 1 .. 70 978 (70 978 = floor(100*ln(MaxFloat64)+0.5)):
      V[j] = the smallest non-negative float64 v with floor(100*Log(1+v)+0.5) >= j   (bisection on the bits)
      records for prev(V[j]), V[j], next(V[j])    (one ulp of v either side of the threshold)

Record (12 bytes, little endian):  float64 v | int16 key = compress(v) per the oracle | uint16 flags
  flags bit 0: the extended key exceeds int16 (|v| > ~2.02e142): the expected key is amd64's wrapping
               conversion (CVTTSD2SL, keep the low 16 bits), implementation-defined in the Go spec --
               integration/compress_thresholds_test.go checks these on GOARCH=amd64 only.
Header (16 bytes): magic "LHTHRv1\\0", uint32 record count, uint32 reserved.

Consumers: integration/compress_thresholds_test.go (`go test` inside a checkout of the reference),
tests/test_oracle.py (freezes the oracle against the file), tests/test_gpu_parity.py (the HIP path
against the same file).  Run from the repo root:  python tests/golden/make_thresholds.py
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

OUT = os.path.join(HERE, "thresholds_x.bin")
MAGIC = b"LHTHRv1\0"


def f_of_bits(bits: np.ndarray) -> np.ndarray:
    v = bits.view(np.float64)
    return oracle.kext_many(1.0 + v)            # 1.0 + v: the same IEEE add as metrics.go:317


def build():
    jmax = oracle.KEXT_MAX
    j = np.arange(1, jmax + 1, dtype=np.int64)
    max_bits = np.float64(1.7976931348623157e308).view(np.uint64)
    lo = np.zeros(j.size, dtype=np.uint64)      # f(lo) < j   (f(0) == 0)
    hi = np.full(j.size, max_bits, dtype=np.uint64)   # f(hi) >= j
    assert int(f_of_bits(hi[:1])[0]) == jmax
    while True:
        gap = hi - lo
        if not (gap > 1).any():
            break
        mid = lo + gap // np.uint64(2)
        ge = f_of_bits(mid) >= j
        hi = np.where(ge & (gap > 1), mid, hi)
        lo = np.where(~ge & (gap > 1), mid, lo)
    V = hi                                       # bits of the smallest v with f(v) >= j
    assert (f_of_bits(V) >= j).all() and (f_of_bits(V - np.uint64(1)) < j).all()
    pts = np.stack([V - np.uint64(1), V, np.minimum(V + np.uint64(1), max_bits)], axis=1).reshape(-1)
    v = pts.view(np.float64)
    keys = oracle.compress_many(v)
    ext = oracle.kext_many(1.0 + v)
    flags = (ext > 32767).astype(np.uint16)
    # in-domain records: the int16 key IS the extended key
    assert (keys[flags == 0].astype(np.int32) == ext[flags == 0]).all()
    return v, keys, flags


def write(path=OUT):
    v, keys, flags = build()
    rec = np.zeros(v.size, dtype=np.dtype([("v", "<f8"), ("key", "<i2"), ("flags", "<u2")]))
    rec["v"], rec["key"], rec["flags"] = v, keys, flags
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", v.size, 0))
        f.write(rec.tobytes())
    return v.size


def read(path=OUT):
    raw = open(path, "rb").read()
    assert raw[:8] == MAGIC
    n, _ = struct.unpack("<II", raw[8:16])
    rec = np.frombuffer(raw, dtype=np.dtype([("v", "<f8"), ("key", "<i2"), ("flags", "<u2")]), count=n, offset=16)
    return rec["v"].copy(), rec["key"].copy(), rec["flags"].copy()


if __name__ == "__main__":
    n = write()
    print(f"{OUT}: {n} records, {os.path.getsize(OUT)} bytes")
