"""CPU tests of the oracle against the reference's own golden vectors.

The oracle (oracle/lh_oracle.c) is the checker for every GPU parity test, so it
is pinned first: every number the reference's tests and docs hold for this path
 is asserted here.  Each test cites the reference test it restates.
"""
import json
import math
import os
import struct
import sys

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "reference_vectors.json")) as f:
        return json.load(f)


def unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


# ---- reference goldens -----------------------------------------------------

def test_decompress_matches_reference_doc_values_bit_for_bit(golden):
    # readme.md:35-43, print_benchmark.go:34-39: 15 float64 printed with Go's
    # shortest round-trip formatting => exact bit patterns of decompress(k).
    for g in golden["reference"]["decompress_doc_values"]:
        assert oracle.decompress(g["key"]) == g["value"], g


def test_percentile_reference(golden):
    # TestPercentile, metrics_test.go:111-149
    t = golden["reference"]["test_percentile"]
    values = [float(k) for k in t["metrics"]]
    counts = [int(v) for v in t["metrics"].values()]
    total = sum(counts)
    for p, expected in t["expected"].items():
        result, err = oracle.percentile(total, values, counts, float(p))
        assert err is None
        assert abs(expected / result - 1) <= t["tolerance"]
        assert result == expected  # values are returned verbatim


def test_percentile_invalid():
    # metrics.go:417: error when no prefix reaches p
    for p in (1.0000001, 2.0, math.nan):
        _, err = oracle.percentile(10, [1.0, 2.0], [5, 5], p)
        assert err is not None


def test_compress_roundtrip_reference(golden):
    # TestCompress, metrics_test.go:151-172
    t = golden["reference"]["test_compress"]
    for f in t["values"]:
        result = oracle.decompress(oracle.compress(f))
        diff = abs(f - result) if result == 0 else abs(f / result - 1)
        assert diff <= t["tolerance"], (f, result)


def test_processed_broadcast_reference(golden):
    # TestProcessedBroadcast, metrics_test.go:289-319
    t = golden["reference"]["test_processed_broadcast"]
    row = oracle.histogram_dense(t["samples"])
    out = oracle.process_histograms("histogram1", row)
    assert int(out["histogram1_sum"]) == t["int_sum"]
    assert int(out["histogram1_count"]) == t["int_count"]
    # _agg_avg = float64(aggSum / aggCount), integer division (metrics.go:601-602)
    r = oracle.process_dense(row, [0.5])
    assert r["agg_sum_add"] // r["count"] == t["int_agg_avg"]


def test_default_percentile_labels(golden):
    assert oracle.DEFAULT_PERCENTILES == {k: float(v) for k, v in golden["reference"]["default_percentiles"].items()}
    out = oracle.process_histograms("t", oracle.histogram_dense([1.0, 2.0, 3.0]))
    for lab in golden["reference"]["default_percentiles"]:
        assert lab % "t" in out
    assert out["t_min"] <= out["t_50"] <= out["t_max"]  # TestTimer, metrics_test.go:196-199


# ---- frozen known answers ---------------------------------------------------

def test_compress_known_answers(golden):
    for k in golden["oracle_kat"]["compress"]:
        assert oracle.compress(unhex(k["bits"])) == k["key"], k


def test_survey_known_answers():
    # SURVEY.md 8(c), measured independently of this oracle during the survey
    kv = {33: 353, 59: 409, 330000: 1271, 123: 482, 1: 69, -1: -69, 0.0: 0, 0.005: 0, 0.00502: 1, 0.5: 41,
          0.51: 41, 1e9: 2072, 1e12: 2763, 9.2e18: 4367, -421408208120481: -3367, 214141241241241: 3300,
          1e142: 32697, 2.0196e142: 32767, 2.03e142: -32768, 3e142: -32729, 1e200: -19484,
          1.7976931348623157e308: 5442, math.inf: 0, -math.inf: 0, math.nan: 0, 4.9e-324: 0}
    for v, k in kv.items():
        assert oracle.compress(v) == k, (v, k)
    dk = {353: 33.123967614754356, 409: 58.739891704145194, 1271: 331040.82304912945, 69: 0.9937155332430823, 0: 0.0}
    for k, v in dk.items():
        assert oracle.decompress(k) == v
    r = oracle.process_dense(oracle.histogram_dense([33, 59, 330000]), [0.5])
    assert r["sum"] == 331132.68690844835


def test_decompress_known_answers(golden):
    for k in golden["oracle_kat"]["decompress"]:
        assert oracle.decompress(k["key"]) == unhex(k["bits"])


# ---- structural properties the GPU design relies on ----------------------------

@pytest.fixture(scope="module")
def tx():
    return oracle.thresholds()


def test_threshold_table_known_answers(golden, tx):
    for k in golden["oracle_kat"]["thresholds_x"]:
        assert tx[k["j"]] == unhex(k["bits"])
    assert tx[0] == 1.0 and math.isinf(tx[oracle.KEXT_MAX + 1])
    assert np.all(np.diff(tx[: oracle.KEXT_MAX + 1]) > 0)


def test_threshold_table_is_equivalent_to_the_function(tx):
    # kext must be monotone within +-64 ulp of every threshold, otherwise a table
    # compare is not the same function as floor(100*Log(x)+0.5).
    assert oracle.check_monotone(tx, 64) == 0
    for j in (1, 2, 3, 69, 70, 4367, 32767, 32768, 40000, 65535, 65536, oracle.KEXT_MAX):
        t = float(tx[j])
        assert oracle.kext(t) == j
        assert oracle.kext(np.nextafter(t, 0.0)) == j - 1


def test_thresholds_agree_with_compress_on_random_samples(tx):
    rng = np.random.default_rng(7)
    v = np.concatenate([rng.lognormal(math.log(1e5), 2.5, 200000), -rng.lognormal(0, 3, 100000),
                        10.0 ** rng.uniform(-3, 140, 100000), rng.uniform(-1, 1, 50000)])
    keys = oracle.compress_many(v)
    x = 1.0 + np.abs(v)
    kext = np.searchsorted(tx[: oracle.KEXT_MAX + 2], x, side="right") - 1
    expect = np.where(v < 0, -kext, kext).astype(np.int64)
    expect = ((expect + 32768) % 65536 - 32768).astype(np.int16)
    assert np.array_equal(keys, expect)


def test_roundtrip_every_key():
    # compress(decompress(k)) == k over the whole in-domain key space
    keys = np.arange(-32767, 32768, dtype=np.int16)
    d = oracle.decompress_table()
    vals = d[oracle.key_to_bin(keys)]
    assert np.array_equal(oracle.compress_many(vals), keys)
    assert np.all(np.diff(d[1:]) > 0)  # strictly monotone => sort by Value == sort by key (metrics.go:409)


def test_out_of_domain_wraps_like_amd64():
    # CVTTSD2SL then low 16 bits; -1*i wraps in int16
    assert oracle.compress(2.03e142) == -32768
    assert oracle.compress(-2.03e142) == -32768
    assert oracle.compress(-3e142) == 32729
    for v in (math.inf, -math.inf, math.nan):
        assert oracle.compress(v) == 0


def test_f64_to_u64_amd64():
    f = oracle.f64_to_u64_amd64
    assert f(331132.68690844835) == 331132
    assert f(-1.5) == 2 ** 64 - 1
    assert f(2.0 ** 63) == 2 ** 63
    assert f(2.0 ** 63 + 4096) == 2 ** 63 + 4096
    assert f(math.nan) == 2 ** 63
    assert f(1e30) == 0


def test_dense_and_pairs_histograms_agree():
    rng = np.random.default_rng(3)
    v = rng.lognormal(math.log(1e5), 1.0, 50000)
    ids = rng.integers(0, 7, v.size).astype(np.uint32)
    multi = oracle.histogram_pairs(ids, v, 7)
    for m in range(7):
        assert np.array_equal(multi[m], oracle.histogram_dense(v[ids == m]))
    assert int(multi.sum()) == v.size
    with pytest.raises(ValueError):
        oracle.histogram_pairs(np.array([9], dtype=np.uint32), np.array([1.0]), 7)


def test_process_dense_empty_and_p0_skips_empty_buckets():
    row = np.zeros(oracle.NKEYS, dtype=np.uint64)
    r = oracle.process_dense(row, [0.0, 0.5, 1.0])
    assert r["count"] == 0 and not r["pvalid"].any() and math.isnan(r["avg"])
    row[oracle.key_to_bin(409)] = 5
    r = oracle.process_dense(row, [0.0, 1.0])
    assert list(r["pkeys"]) == [409, 409]  # p=0 selects the first OCCUPIED bucket


def test_cpu_baseline_forms_are_exact():
    rng = np.random.default_rng(11)
    v = rng.lognormal(math.log(1e5), 1.0, 200000)
    ref = oracle.histogram_dense(v)
    for threads in (1, 4):
        _, a = oracle.bench_faithful(v, threads)
        _, b = oracle.bench_dense(v, threads)
        assert np.array_equal(a, ref) and np.array_equal(b, ref)


def test_format_f_is_go_percent_f():
    """fmt.Sprintf("%f") (graphite.go:40, opentsdb.go:48) = the exact decimal expansion rounded half-even to 6
    places: the oracle's formatter against Python's decimal module (an independent exact implementation)."""
    import decimal
    import math
    import random
    import struct
    from decimal import Decimal
    decimal.getcontext().prec = 2000

    def exact(v):
        if math.isnan(v):
            return "NaN"
        if math.isinf(v):
            return "-Inf" if v < 0 else "+Inf"
        s = format(Decimal(v).quantize(Decimal("0.000001"), rounding=decimal.ROUND_HALF_EVEN), "f")
        return "-0.000000" if v == 0 and math.copysign(1, v) < 0 else s

    rnd = random.Random(1)
    vals = [0.0, -0.0, 0.5, 5e-7, 1.5e-6, 2.5e-6, 0.9999995, 999999.9999995, 2.0 ** 53, 2.0 ** 64, 1e22, 1e23, 1e300,
            1.7976931348623157e308, 5e-324, 50.54, 10.21, 43.32, 12.3, float("nan"), float("inf"), -float("inf")]
    vals += [k / 128.0 for k in range(1, 2001, 2)]                   # exact ties
    vals += [struct.unpack("<d", struct.pack("<Q", rnd.getrandbits(64)))[0] for _ in range(20000)]
    vals += [rnd.uniform(0, 1e6) for _ in range(20000)]
    for v in vals:
        assert oracle.format_f(v) == exact(v), repr(v)
    # values the reference's docs print with %v, through %f
    assert oracle.format_f(58.739891704145194) == "58.739892"
    assert oracle.format_f(2.4642914167480484e+07) == "24642914.167480"
    assert oracle.format_f(-657.5233632152207) == "-657.523363"


def test_threaded_forms_equal_the_sequential_oracle():
    """The full-size parity checks (tests/test_gpu_fullsize.py, bench.py) run the oracle on every host core;
    the threaded forms must be the sequential restatement bit for bit."""
    rng = np.random.default_rng(11)
    n, M = 300_000, 37
    ids = rng.integers(0, M, n).astype(np.uint32)
    v = rng.lognormal(8, 2.0, n) * np.where(rng.random(n) < 0.2, -1.0, 1.0)
    v[:50] = [np.nan, np.inf, -np.inf, 0.0, 2.0196e142] * 10
    assert np.array_equal(oracle.histogram_pairs_mt(ids, v, M, threads=5), oracle.histogram_pairs(ids, v, M))
    assert np.array_equal(oracle.histogram_dense_mt(v, threads=3), oracle.histogram_dense(v))
    with pytest.raises(ValueError):
        oracle.histogram_pairs_mt(np.array([M], dtype=np.uint32), np.array([1.0]), M, threads=2)


def test_threshold_fixture_matches_the_oracle():
    """tests/golden/thresholds_x.bin is what integration/compress_thresholds_test.go feeds to real Go
    (`go test` in a checkout of the reference closes the compress pin).  The committed file must be exactly
    what the oracle produces today: 3 neighbours of each of the 70 978 thresholds, with the oracle's keys."""
    import zlib
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_thresholds as mt
    v, keys, flags = mt.read()
    assert v.size == 3 * oracle.KEXT_MAX
    assert np.array_equal(oracle.compress_many(v), keys)
    ext = oracle.kext_many(1.0 + v)
    assert np.array_equal(flags, (ext > 32767).astype(np.uint16))
    # each triple straddles its threshold: prev < j <= at <= next
    j = np.arange(1, oracle.KEXT_MAX + 1)
    assert (ext[0::3] < j).all() and (ext[1::3] >= j).all() and (ext[2::3] >= ext[1::3]).all()
    # regenerating gives the same bytes (the generator is deterministic)
    v2, k2, f2 = mt.build()
    assert np.array_equal(v2.view(np.uint64), v.view(np.uint64)) and np.array_equal(k2, keys) and np.array_equal(f2, flags)
    # the Go file carries the decompress checksum it compares against
    crc = zlib.crc32(oracle.decompress_table().tobytes())
    go = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "integration", "compress_thresholds_test.go")).read()
    assert f"uint32({crc})" in go
