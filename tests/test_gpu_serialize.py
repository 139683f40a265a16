"""K6 (lh_serialize / lh_format_f / lh_snapshot_accumulate): the histogram keys of a ProcessedMetricSet as
Graphite / OpenTSDB text formatted on the device (metrics.go:495-499, 590-608; graphite.go:37-48;
opentsdb.go:45-58), byte for byte against the oracle's restatement of fmt.Sprintf("%f")."""
import math
import struct

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

GRAPHITE = dict(prefix="cockroach.box-1_a.", sep=" ", suffix=" 1411104988\n", underscore_to_dot=True)
TSDB = dict(prefix="put ", sep=" 1411104988 ", suffix=" host=box-1_a\n", underscore_to_dot=False)


def special_values():
    v = [0.0, -0.0, 0.5, -0.5, 1e-7, 4.9e-7, 5e-7, 5.1e-7, 1.5e-6, 2.5e-6, 0.9999994, 0.9999995, 0.99999951,
         999999.9999995, 9.9999995, 1.0, 9.0, 10.0, 99.0, 1e15, 2.0 ** 52 + 0.5, 2.0 ** 53 - 1, 2.0 ** 53, 2.0 ** 53 + 2,
         2.0 ** 63, 2.0 ** 64 - 2048, 2.0 ** 64, 2.0 ** 64 + 4096, 1e19, 1e20, 1e22, 1e23, 1e100, 2.0196e142, 1e300,
         1.7976931348623157e308, 5e-324, 2.2250738585072014e-308, 50.54, 10.21, 43.32, 12.3, 33.123967614754356,
         58.739891704145194, 331040.82304912945, 2.4642914167480484e+07, -657.5233632152207,
         float("nan"), float("inf"), -float("inf")]
    v += [k / 128.0 for k in range(1, 4001, 2)]                    # exact ties at the 7th decimal: half-even
    v += [1048576.0 + k / 128.0 for k in range(1, 2001, 2)]
    v += [-(k / 64.0 + 1 / 128.0) for k in range(0, 500)]
    v += [float(10 ** k) for k in range(0, 23)] + [float(10 ** k - 1) for k in range(1, 16)]
    v += [(10 ** k - 1) / 1e6 for k in range(1, 16)]                # carries into the integer part
    return np.array(v, dtype=np.float64)


def test_format_f_matches_go_percent_f(native_lib, torch_cuda):
    import loghisto_amd
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 64, 200_000, dtype=np.uint64)
    v = np.concatenate([special_values(), bits.view(np.float64), rng.uniform(0, 1e6, 100_000),
                        np.round(rng.uniform(0, 100, 50_000), 6) + 5e-7,
                        oracle.decompress_table()[::7], rng.lognormal(11.5, 3.0, 50_000),
                        np.ldexp(rng.uniform(1, 2, 4000), rng.integers(60, 1023, 4000))])
    with loghisto_amd.Engine(max_metrics=1) as eng:
        got = eng.format_f(v)
    bad = [(float(x), g, oracle.format_f(float(x))) for x, g in zip(v, got) if g != oracle.format_f(float(x))]
    assert not bad, bad[:10]
    assert max(len(g) for g in got) <= 317


def _run(eng, names, ids, v, percentiles, wire, aggregates=False, accumulate=False, first=0, nmetrics=None):
    eng.submit_pairs(ids, v)
    with eng.flip() as snap:
        if accumulate:
            snap.accumulate()
        text = snap.serialize(percentiles, aggregates=aggregates, first=first, nmetrics=nmetrics, **wire)
        stats = snap.extract([percentiles[k] for k in percentiles], len(names))
    return text, stats


def _check(text, names, rows, percentiles, wire, stats, life=None, first=0, nmetrics=None):
    nmetrics = len(names) - first if nmetrics is None else nmetrics
    sel = slice(first, first + nmetrics)
    # _sum/_avg: the engine's own float64 (summation order is unpinned) ...
    want = oracle.wire_lines(names[sel], rows[sel], percentiles, life=None if life is None else
                             (life[0][sel], life[1][sel]), sums=stats["sum"][sel], **wire)
    got = text.decode().split("\n")
    assert got[-1] == "" and got[:-1] == [w.rstrip("\n") for w in want]
    # ... which itself is within 1e-12 of the oracle's ascending-key sum
    for i in range(first, first + nmetrics):
        r = oracle.process_dense(rows[i], [])
        if r["count"]:
            d = oracle.decompress_table()
            tol = 1e-12 * float(np.sum(np.abs(d) * rows[i].astype(np.float64)))
            assert abs(stats["sum"][i] - r["sum"]) <= tol


@pytest.mark.parametrize("wire", [GRAPHITE, TSDB], ids=["graphite", "opentsdb"])
def test_serialize_matches_reference_lines(native_lib, torch_cuda, wire):
    import loghisto_amd
    rng = np.random.default_rng(11)
    names = [f"svc_{i}.rpc_latency" for i in range(37)] + ["x", "a_b_c_", "_lead", "empty_one"]
    M = len(names)
    n = 400_000
    ids = rng.integers(0, M - 1, n).astype(np.uint32)          # the last name gets no sample
    v = rng.lognormal(11.5, 1.0, n) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
    v[ids == 37] = 123.0
    rows = oracle.histogram_pairs(ids, v, M)
    pct = dict(oracle.DEFAULT_PERCENTILES)
    pct["p%s_bad"] = 1.5                                        # "Invalid percentile": key omitted (metrics.go:379-384)
    pct["pre_%s"] = 0.25
    pct["100%%_%s"] = 0.5
    with loghisto_amd.Engine(max_metrics=M, num_lanes=1, lane_samples=1 << 16) as eng:
        for nm in names:
            eng.intern(nm)
        text, stats = _run(eng, names, ids, v, pct, wire)
    assert b"empty_one" not in text and b"empty.one" not in text and b"_bad" not in text and b".bad" not in text
    _check(text, names, rows, pct, wire, stats)
    if wire is GRAPHITE:
        assert b"cockroach.box-1_a.svc.0.rpc.latency.count " in text and b"_" not in text.replace(b"box-1_a", b"")
    else:
        assert b"put svc_0.rpc_latency_count 1411104988 " in text


def test_serialize_aggregates_and_lifetime(native_lib, torch_cuda):
    import loghisto_amd
    rng = np.random.default_rng(5)
    names = [f"h{i:03d}" for i in range(20)]
    M = len(names)
    pct = oracle.DEFAULT_PERCENTILES
    life_c = np.zeros(M, dtype=np.uint64)
    life_s = np.zeros(M, dtype=np.uint64)
    with loghisto_amd.Engine(max_metrics=M + 3, num_lanes=1, lane_samples=1 << 16) as eng:
        for nm in names:
            eng.intern(nm)
        for interval in range(3):
            n = 100_000
            ids = rng.integers(0, M if interval != 1 else M // 2, n).astype(np.uint32)
            v = rng.lognormal(8.0 + interval, 1.5, n)
            rows = oracle.histogram_pairs(ids, v, M)
            eng.submit_pairs(ids, v)
            with eng.flip() as snap:
                snap.accumulate()
                snap.accumulate()                                   # at most once per snapshot
                stats = snap.extract([], M)
                text = snap.serialize(pct, aggregates=True, **GRAPHITE)
                life_c += stats["count"]
                life_s += stats["agg_sum_add"]                      # uint64(totalSum), wrapping add
                got_c, got_s = eng.lifetime(M)
                assert np.array_equal(got_c, life_c) and np.array_equal(got_s, life_s)
                _check(text, names, rows, pct, GRAPHITE, stats, life=(life_c, life_s))
            for i in range(M):                                      # uint64(sum) as the oracle converts it
                if rows[i].any():
                    assert stats["agg_sum_add"][i] == oracle.f64_to_u64_amd64(float(stats["sum"][i]))
        assert b".agg.avg " in text and b".agg.count " in text and b".agg.sum " in text


def test_serialize_without_accumulate_has_no_agg_keys(native_lib, torch_cuda):
    import loghisto_amd
    with loghisto_amd.Engine(max_metrics=2) as eng:
        eng.intern("a")
        eng.submit(0, np.array([33.0, 59.0, 330000.0]))
        with eng.flip() as snap:
            text = snap.serialize(oracle.DEFAULT_PERCENTILES, aggregates=True, **TSDB)
    assert b"_agg_" not in text                                     # lifetime count is 0 (metrics.go:599)
    # TestProcessedBroadcast's three samples (metrics_test.go:289-319)
    assert b"put a_count 1411104988 3.000000 host=box-1_a\n" in text
    assert b"put a_sum 1411104988 331132.68690" in text
    assert b"put a_max 1411104988 331040.823049 host=box-1_a\n" in text


def test_serialize_row_ranges_and_sizing(native_lib, torch_cuda):
    import ctypes as C
    import loghisto_amd
    from loghisto_amd import _native as N
    rng = np.random.default_rng(3)
    names = [f"n{i}" for i in range(50)]
    ids = rng.integers(0, 50, 100_000).astype(np.uint32)
    v = rng.exponential(1e6, 100_000)
    rows = oracle.histogram_pairs(ids, v, 50)
    pct = {"%s_50": 0.5, "%s_99.9": 0.999}
    with loghisto_amd.Engine(max_metrics=64, num_lanes=1, lane_samples=1 << 16) as eng:
        for nm in names:
            eng.intern(nm)
        eng.submit_pairs(ids, v)
        with eng.flip() as snap:
            stats = snap.extract([0.5, 0.999], 50)
            whole = snap.serialize(pct, **TSDB)
            part = snap.serialize(pct, first=13, nmetrics=21, **TSDB)
            none = snap.serialize({}, first=13, nmetrics=21, **TSDB)
            _check(part, names, rows, pct, TSDB, stats, first=13, nmetrics=21)
            _check(none, names, rows, {}, TSDB, stats, first=13, nmetrics=21)
            assert part in whole
            # a buffer that is too small: *len reports the need, nothing is written
            L = N.lib()
            fmt = N.LhLineFormat(b"put ", b" 1 ", b"\n", 0, 0)
            need = C.c_size_t(0)
            buf = C.create_string_buffer(b"\xaa" * 64, 64)
            rc = L.lh_serialize(snap._h, 0, 50, None, None, 0, C.byref(fmt), 0, buf, 64, C.byref(need))
            assert rc == 0 and need.value > 64 and buf.raw == b"\xaa" * 64
            # names beyond the interned ones are an argument error
            rc = L.lh_serialize(snap._h, 40, 20, None, None, 0, C.byref(fmt), 0, None, 0, C.byref(need))
            assert rc == N.EINVAL
            # labels need exactly one %s
            lab = (C.c_char_p * 1)(b"nope")
            p = (C.c_double * 1)(0.5)
            rc = L.lh_serialize(snap._h, 0, 50, p, lab, 1, C.byref(fmt), 0, None, 0, C.byref(need))
            assert rc == N.EINVAL


def test_serialize_long_lines_and_huge_values(native_lib, torch_cuda):
    """Lines that do not fit the LDS staging area go straight to HBM; sums beyond 2^64 take the big-number path."""
    import loghisto_amd
    names = [("long_name_%03d_" % i) + "x" * 300 for i in range(300)] + ["huge"]
    M = len(names)
    rng = np.random.default_rng(9)
    n = 50_000
    ids = rng.integers(0, M - 1, n).astype(np.uint32)
    v = rng.lognormal(5, 2, n)
    big_ids = np.full(4096, M - 1, dtype=np.uint32)
    big_v = np.full(4096, 1.5e140)
    ids = np.concatenate([ids, big_ids])
    v = np.concatenate([v, big_v])
    rows = oracle.histogram_pairs(ids, v, M)
    pct = oracle.DEFAULT_PERCENTILES
    with loghisto_amd.Engine(max_metrics=M, num_lanes=1, lane_samples=1 << 16) as eng:
        for nm in names:
            eng.intern(nm)
        text, stats = _run(eng, names, ids, v, pct, GRAPHITE, aggregates=True, accumulate=True)
        life = eng.lifetime(M)
    _check(text, names, rows, pct, GRAPHITE, stats, life=life)
    huge = [ln for ln in text.decode().split("\n") if ".huge.sum " in ln]
    assert len(huge) == 1 and len(huge[0].split(" ")[1]) > 140


def test_serialize_many_names(native_lib, torch_cuda):
    import loghisto_amd
    M = 4096
    rng = np.random.default_rng(21)
    n = 2_000_000
    ids = rng.integers(0, M, n).astype(np.uint32)
    v = rng.lognormal(11.5, 1.0, n) * (1.0 + 0.002 * ids)
    names = [f"h{i:04d}" for i in range(M)]
    rows = oracle.histogram_pairs(ids, v, M)
    with loghisto_amd.Engine(max_metrics=M, num_lanes=1, lane_samples=1 << 20) as eng:
        for nm in names:
            eng.intern(nm)
        text, stats = _run(eng, names, ids, v, oracle.DEFAULT_PERCENTILES, GRAPHITE, aggregates=True, accumulate=True)
        life = eng.lifetime(M)
    assert text.count(b"\n") == M * 15
    _check(text, names, rows, oracle.DEFAULT_PERCENTILES, GRAPHITE, stats, life=life)
