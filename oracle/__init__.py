"""ctypes front-end of the CPU oracle (oracle/lh_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; nothing under loghisto_amd/ does.  See lh_oracle.h for what the
oracle restates (metrics.go:273-295, 316-332, 336-418) and its pinning status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liblh_oracle.so")
NKEYS = 65536
KEXT_MAX = 70978


def build(force: bool = False) -> str:
    srcs = [os.path.join(_DIR, f) for f in ("lh_oracle.c", "lh_oracle.h", "lh_cpu_baseline.cc", "Makefile")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _DIR, "liblh_oracle.so"])
    return _SO


class Stats(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_double), ("avg", C.c_double),
                ("agg_sum_add", C.c_uint64), ("nbuckets", C.c_uint32), ("reserved", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, u64p, u32p, i16p, u8p = (C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_int16), C.POINTER(C.c_uint8))
        L.lho_go_log.restype = C.c_double; L.lho_go_log.argtypes = [C.c_double]
        L.lho_go_exp.restype = C.c_double; L.lho_go_exp.argtypes = [C.c_double]
        L.lho_compress.restype = C.c_int16; L.lho_compress.argtypes = [C.c_double]
        L.lho_decompress.restype = C.c_double; L.lho_decompress.argtypes = [C.c_int16]
        L.lho_kext.restype = C.c_int32; L.lho_kext.argtypes = [C.c_double]
        L.lho_kext_many.restype = None; L.lho_kext_many.argtypes = [dp, C.c_size_t, C.POINTER(C.c_int32)]
        L.lho_compress_many.restype = None; L.lho_compress_many.argtypes = [dp, C.c_size_t, i16p]
        L.lho_histogram_dense.restype = None; L.lho_histogram_dense.argtypes = [dp, C.c_size_t, u64p]
        L.lho_histogram_pairs.restype = C.c_int
        L.lho_histogram_pairs.argtypes = [u32p, dp, C.c_size_t, u64p, C.c_uint32]
        L.lho_histogram_pairs_mt.restype = C.c_int
        L.lho_histogram_pairs_mt.argtypes = [u32p, dp, C.c_size_t, u64p, C.c_uint32, C.c_int]
        L.lho_thresholds.restype = None; L.lho_thresholds.argtypes = [dp, C.c_size_t]
        L.lho_check_monotone.restype = C.c_size_t; L.lho_check_monotone.argtypes = [dp, C.c_size_t, C.c_int]
        L.lho_decompress_table.restype = None; L.lho_decompress_table.argtypes = [dp]
        L.lho_process_dense.restype = None
        L.lho_process_dense.argtypes = [u64p, dp, C.c_size_t, C.POINTER(Stats), dp, i16p, u8p]
        L.lho_percentile.restype = C.c_int
        L.lho_percentile.argtypes = [C.c_uint64, dp, u64p, C.c_size_t, C.c_double, dp]
        L.lho_f64_to_u64_amd64.restype = C.c_uint64; L.lho_f64_to_u64_amd64.argtypes = [C.c_double]
        L.lho_bench_faithful.restype = C.c_double
        L.lho_bench_faithful.argtypes = [dp, C.c_size_t, C.c_int, u64p]
        L.lho_bench_dense.restype = C.c_double
        L.lho_bench_dense.argtypes = [dp, C.c_size_t, C.c_int, u64p]
        L.lho_bench_dense_reps.restype = C.c_double
        L.lho_bench_dense_reps.argtypes = [dp, C.c_size_t, C.c_int, C.c_int, u64p]
        L.lho_format_f.restype = C.c_int; L.lho_format_f.argtypes = [C.c_double, C.c_char_p]
        _lib = L
    return _lib


def _dp(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def _u64p(a): return a.ctypes.data_as(C.POINTER(C.c_uint64))
def _u32p(a): return a.ctypes.data_as(C.POINTER(C.c_uint32))


def go_log(x: float) -> float: return lib().lho_go_log(float(x))
def go_exp(x: float) -> float: return lib().lho_go_exp(float(x))
def compress(v: float) -> int: return int(lib().lho_compress(float(v)))
def decompress(c: int) -> float: return lib().lho_decompress(int(c))
def kext(x: float) -> int: return int(lib().lho_kext(float(x)))
def f64_to_u64_amd64(f: float) -> int: return int(lib().lho_f64_to_u64_amd64(float(f)))


def key_to_bin(k):
    return (np.asarray(k).astype(np.int64) & 0xFFFF) ^ 0x8000


def bin_to_key(b):
    return ((np.asarray(b).astype(np.int64) ^ 0x8000) & 0xFFFF).astype(np.uint16).view(np.int16)


def compress_many(v) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.empty(v.shape, dtype=np.int16)
    lib().lho_compress_many(_dp(v), v.size, out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def kext_many(x) -> np.ndarray:
    """floor(100*Log(x)+0.5) before the int16 wrap, for x >= 1 (-1 for NaN/Inf)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.shape, dtype=np.int32)
    lib().lho_kext_many(_dp(x), x.size, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def histogram_dense(v, counts: np.ndarray | None = None) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float64)
    if counts is None:
        counts = np.zeros(NKEYS, dtype=np.uint64)
    lib().lho_histogram_dense(_dp(v), v.size, _u64p(counts))
    return counts


def histogram_pairs(ids, v, nmetrics: int, counts: np.ndarray | None = None) -> np.ndarray:
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    v = np.ascontiguousarray(v, dtype=np.float64)
    assert ids.size == v.size
    if counts is None:
        counts = np.zeros((nmetrics, NKEYS), dtype=np.uint64)
    rc = lib().lho_histogram_pairs(_u32p(ids), _dp(v), v.size, _u64p(counts), nmetrics)
    if rc != 0:
        raise ValueError("metric id out of range")
    return counts


def granted_cores() -> int:
    """Cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    import math
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(math.ceil(q / p))))
        except (OSError, ValueError):
            pass
    return n


def histogram_dense_mt(v, threads: int | None = None) -> np.ndarray:
    """histogram_dense over `threads` slices (per-thread rows, summed): full-size parity checks."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    counts = np.zeros(NKEYS, dtype=np.uint64)
    lib().lho_bench_dense(_dp(v), v.size, threads or granted_cores(), _u64p(counts))
    return counts


def histogram_pairs_mt(ids, v, nmetrics: int, threads: int | None = None, counts: np.ndarray | None = None):
    """histogram_pairs over `threads` slices adding atomically into one matrix: full-size parity checks."""
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    v = np.ascontiguousarray(v, dtype=np.float64)
    assert ids.size == v.size
    if counts is None:
        counts = np.zeros((nmetrics, NKEYS), dtype=np.uint64)
    rc = lib().lho_histogram_pairs_mt(_u32p(ids), _dp(v), v.size, _u64p(counts), nmetrics,
                                      threads or granted_cores())
    if rc != 0:
        raise ValueError("metric id out of range")
    return counts


def thresholds(n: int = KEXT_MAX + 2) -> np.ndarray:
    t = np.empty(n, dtype=np.float64)
    lib().lho_thresholds(_dp(t), n)
    return t


def check_monotone(tx: np.ndarray, window: int = 64) -> int:
    tx = np.ascontiguousarray(tx, dtype=np.float64)
    return int(lib().lho_check_monotone(_dp(tx), tx.size, window))


def decompress_table() -> np.ndarray:
    d = np.empty(NKEYS, dtype=np.float64)
    lib().lho_decompress_table(_dp(d))
    return d


DEFAULT_PERCENTILES = {  # metrics.go:145-155
    "%s_min": 0.0, "%s_50": .5, "%s_75": .75, "%s_90": .9, "%s_95": .95,
    "%s_99": .99, "%s_99.9": .999, "%s_99.99": .9999, "%s_max": 1.0,
}


def process_dense(counts: np.ndarray, p):
    """processHistograms on one dense row -> dict(count,sum,avg,agg_sum_add,nbuckets,pvals,pkeys,pvalid)."""
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    assert counts.size == NKEYS
    p = np.ascontiguousarray(p, dtype=np.float64)
    st = Stats()
    pv = np.zeros(p.size, dtype=np.float64)
    pk = np.zeros(p.size, dtype=np.int16)
    ok = np.zeros(p.size, dtype=np.uint8)
    lib().lho_process_dense(_u64p(counts), _dp(p), p.size, C.byref(st), _dp(pv),
                            pk.ctypes.data_as(C.POINTER(C.c_int16)), ok.ctypes.data_as(C.POINTER(C.c_uint8)))
    return dict(count=int(st.count), sum=float(st.sum), avg=float(st.avg), agg_sum_add=int(st.agg_sum_add),
                nbuckets=int(st.nbuckets), pvals=pv, pkeys=pk, pvalid=ok)


def percentile(total: int, values, counts, p: float):
    values = np.ascontiguousarray(values, dtype=np.float64)
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    out = C.c_double(0)
    rc = lib().lho_percentile(int(total), _dp(values), _u64p(counts), values.size, float(p), C.byref(out))
    return (out.value, None) if rc == 0 else (0.0, "Invalid percentile.  Should be between 0 and 1.")


def process_histograms(name: str, counts: np.ndarray, percentiles=None) -> dict:
    """Key/value output of processHistograms (metrics.go:336-387) for one metric."""
    percentiles = DEFAULT_PERCENTILES if percentiles is None else percentiles
    labels = list(percentiles.keys())
    r = process_dense(counts, [percentiles[k] for k in labels])
    out = {f"{name}_count": float(r["count"]), f"{name}_sum": r["sum"], f"{name}_avg": r["avg"]}
    for i, lab in enumerate(labels):
        if r["pvalid"][i]:
            out[lab % name] = float(r["pvals"][i])
    return out


def format_f(v: float) -> str:
    """fmt.Sprintf("%f", v) (graphite.go:40, opentsdb.go:48)."""
    buf = C.create_string_buffer(336)
    n = lib().lho_format_f(float(v), buf)
    return buf.raw[:n].decode()


def fmt_label(label: str, name: str) -> str:
    """fmt.Sprintf(label, name) for labels holding one %s (metrics.go:383)."""
    return label.replace("%%", "\0").replace("%s", name).replace("\0", "%")


def wire_lines(names, counts_rows, percentiles, prefix: str, sep: str, suffix: str, underscore_to_dot: bool = False,
               life=None, sums=None) -> list:
    """The lines GraphiteProtocol / OpenTSDBProtocol emit for the histogram keys of one interval
    (metrics.go:495-499, 590-608; graphite.go:37-48; opentsdb.go:45-58), metric-major, keys in the order
    _count, _sum, _avg, labels, _agg_avg, _agg_count, _agg_sum.  counts_rows[i] is the dense row of
    names[i]; life = (count[], sum[]) lifetime stores AFTER this interval or None; sums (optional)
    overrides the oracle's ascending-key _sum/_avg with the caller's float64 sums (their order of
    summation is unpinned)."""
    labels = list(percentiles.keys())
    ps = [percentiles[k] for k in labels]
    out = []

    def line(key, value):
        if underscore_to_dot:
            key = key.replace("_", ".")
        out.append(f"{prefix}{key}{sep}{format_f(value)}{suffix}")

    for i, name in enumerate(names):
        r = process_dense(counts_rows[i], ps)
        if r["count"] == 0:
            continue
        s = r["sum"] if sums is None else float(sums[i])
        line(f"{name}_count", float(r["count"]))
        line(f"{name}_sum", s)
        line(f"{name}_avg", s / float(r["count"]))
        for j, lab in enumerate(labels):
            if r["pvalid"][j]:
                line(fmt_label(lab, name), float(r["pvals"][j]))
        if life is not None and int(life[0][i]) > 0:
            lc, ls = int(life[0][i]), int(life[1][i])
            line(f"{name}_agg_avg", float(ls // lc))
            line(f"{name}_agg_count", float(lc))
            line(f"{name}_agg_sum", float(ls))
    return out


def bench_faithful(v, threads: int):
    v = np.ascontiguousarray(v, dtype=np.float64)
    counts = np.zeros(NKEYS, dtype=np.uint64)
    s = lib().lho_bench_faithful(_dp(v), v.size, threads, _u64p(counts))
    return s, counts


def bench_dense_reps(v, threads: int, reps: int):
    """seconds, counts (== reps x the true row): every thread loops `reps` times over its slice."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    counts = np.zeros(NKEYS, dtype=np.uint64)
    s = lib().lho_bench_dense_reps(_dp(v), v.size, threads, reps, _u64p(counts))
    return s, counts


def bench_dense(v, threads: int):
    v = np.ascontiguousarray(v, dtype=np.float64)
    counts = np.zeros(NKEYS, dtype=np.uint64)
    s = lib().lho_bench_dense(_dp(v), v.size, threads, _u64p(counts))
    return s, counts
