"""Property tests of the oracle (hypothesis): the structural facts the GPU design leans on, over
arbitrary float64 inputs rather than hand-picked ones."""
import math

import numpy as np
from hypothesis import given, settings, strategies as st

import oracle

IN_DOMAIN = 2.0196e142       # compress() is documented to fail at 1e142 and above (metrics.go:312-315)
finite = st.floats(allow_nan=False, allow_infinity=False, min_value=-IN_DOMAIN, max_value=IN_DOMAIN)
anyfloat = st.floats(allow_nan=True, allow_infinity=True)

_TX = oracle.thresholds()


@settings(max_examples=400, deadline=None)
@given(finite)
def test_compress_is_odd(v):
    assert oracle.compress(-v) == -oracle.compress(v)


@settings(max_examples=400, deadline=None)
@given(finite, finite)
def test_compress_is_monotone(a, b):
    lo, hi = (a, b) if a <= b else (b, a)
    assert oracle.compress(lo) <= oracle.compress(hi)


@settings(max_examples=400, deadline=None)
@given(st.floats(min_value=0.51, max_value=IN_DOMAIN))
def test_roundtrip_within_one_percent(v):
    # the reference's accuracy claim (readme.md:5, metrics.go:312-315), checked by TestCompress at 5 points
    r = oracle.decompress(oracle.compress(v))
    assert abs(v / r - 1) <= 0.01


@settings(max_examples=600, deadline=None)
@given(anyfloat)
def test_threshold_table_equals_compress_everywhere(v):
    # the table route the GPU uses (x = 1+|v| against Tx) equals the function for ANY float64,
    # including NaN, infinities, subnormals and the wrap-around region above 2.02e142
    x = 1.0 + abs(v)
    if math.isnan(x) or math.isinf(x):
        kext = 0
    else:
        kext = int(np.searchsorted(_TX, x, side="right")) - 1
    i = kext & 0xFFFF
    key = (-i & 0xFFFF) if v < 0 else i
    key = key - 65536 if key >= 32768 else key
    assert oracle.compress(v) == key


@settings(max_examples=200, deadline=None)
@given(st.lists(st.tuples(st.integers(-32768, 32767), st.integers(1, 10 ** 12)), min_size=1, max_size=40),
       st.lists(st.floats(min_value=0, max_value=1), min_size=2, max_size=6))
def test_percentiles_are_monotone_in_p_and_bracketed(cells, ps):
    row = np.zeros(oracle.NKEYS, dtype=np.uint64)
    for k, c in cells:
        row[(k & 0xFFFF) ^ 0x8000] += np.uint64(c)
    ps = sorted(ps)
    r = oracle.process_dense(row, [0.0] + ps + [1.0])
    assert r["pvalid"].all()
    keys = r["pkeys"].astype(int)
    assert all(a <= b for a, b in zip(keys, keys[1:]))            # non-decreasing in p
    occupied = np.nonzero(row)[0]
    assert keys[0] == int(oracle.bin_to_key(occupied[:1])[0])     # p=0 -> min occupied
    assert keys[-1] == int(oracle.bin_to_key(occupied[-1:])[0])   # p=1 -> max occupied
    assert r["count"] == int(row.sum())
