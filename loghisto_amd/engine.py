"""Thin object wrapper over the C ABI: Engine (lh_engine) and Snapshot (lh_snapshot).

Nothing is computed here; every method is one or two calls into liblhgpu.so.
Reference counterparts are cited on the C declarations in include/loghisto_gpu.h.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _native as N


def _ptr(x) -> int:
    """Raw address of a torch tensor / numpy array / int."""
    if x is None:
        return 0
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    if isinstance(x, np.ndarray):
        return int(x.ctypes.data)
    raise TypeError(f"cannot take the address of {type(x)!r}")


def _stream_handle(stream) -> int:
    if stream is None:
        return 0
    if isinstance(stream, int):
        return stream
    if hasattr(stream, "cuda_stream"):  # torch.cuda.Stream
        return int(stream.cuda_stream)
    raise TypeError(f"not a stream: {type(stream)!r}")


class Snapshot:
    """One interval's cells, stolen at the epoch flip (metrics.go:460-463)."""

    def __init__(self, engine: "Engine", handle: int):
        self.engine = engine
        self._h = C.c_void_p(handle)

    def extract(self, percentiles: Sequence[float], nmetrics: Optional[int] = None, first: int = 0):
        """processHistograms for metrics [first, first+nmetrics) -> dict of numpy arrays."""
        L = N.lib()
        if nmetrics is None:
            nmetrics = self.engine.num_metrics() - first
        p = np.ascontiguousarray(percentiles, dtype=np.float64)
        np_ = int(p.size)
        stats = (N.LhStats * max(nmetrics, 1))()
        pvals = np.zeros((nmetrics, np_), dtype=np.float64)
        pkeys = np.zeros((nmetrics, np_), dtype=np.int16)
        pvalid = np.zeros((nmetrics, np_), dtype=np.uint8)
        N.check(L.lh_extract_rows(self._h, first, nmetrics, p.ctypes.data_as(C.POINTER(C.c_double)), np_, stats,
                                  pvals.ctypes.data_as(C.POINTER(C.c_double)),
                                  pkeys.ctypes.data_as(C.POINTER(C.c_int16)),
                                  pvalid.ctypes.data_as(C.POINTER(C.c_uint8))), "lh_extract_rows")
        raw = np.frombuffer(stats, dtype=np.dtype([("count", "<u8"), ("sum", "<f8"), ("avg", "<f8"),
                                                   ("agg_sum_add", "<u8"), ("nbuckets", "<u4"),
                                                   ("present", "<u4")]), count=nmetrics).copy()
        return dict(count=raw["count"], sum=raw["sum"], avg=raw["avg"], agg_sum_add=raw["agg_sum_add"],
                    nbuckets=raw["nbuckets"], present=raw["present"], pvals=pvals, pkeys=pkeys, pvalid=pvalid)

    def extract_view(self, percentiles: Sequence[float], nmetrics: Optional[int] = None, first: int = 0):
        """extract() without the last copy: numpy views of the engine's pinned result buffer (lh_extract_rows_view).
        Valid until the next call that produces results on this engine, or release()."""
        if nmetrics is None:
            nmetrics = self.engine.num_metrics() - first
        p = np.ascontiguousarray(percentiles, dtype=np.float64)
        v = N.LhExtractView()
        N.check(N.lib().lh_extract_rows_view(self._h, first, nmetrics, p.ctypes.data_as(C.POINTER(C.c_double)), int(p.size),
                                             C.byref(v)), "lh_extract_rows_view")
        dt = np.dtype([("count", "<u8"), ("sum", "<f8"), ("avg", "<f8"), ("agg_sum_add", "<u8"), ("nbuckets", "<u4"),
                       ("present", "<u4")])
        np_ = int(p.size)

        def arr(ptr, ctype, count, dtype):
            if not count:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((ctype * count).from_address(ptr), dtype=dtype, count=count)

        raw = arr(v.stats, C.c_uint8 * 40, nmetrics, dt)
        return dict(count=raw["count"], sum=raw["sum"], avg=raw["avg"], agg_sum_add=raw["agg_sum_add"],
                    nbuckets=raw["nbuckets"], present=raw["present"],
                    pvals=arr(v.pvals, C.c_double, nmetrics * np_, np.float64).reshape(nmetrics, np_),
                    pkeys=arr(v.pkeys, C.c_int16, nmetrics * np_, np.int16).reshape(nmetrics, np_),
                    pvalid=arr(v.pvalid, C.c_uint8, nmetrics * np_, np.uint8).reshape(nmetrics, np_))

    def extract_compact(self, percentiles: Sequence[float], nmetrics: Optional[int] = None, first: int = 0):
        """The compact form of extract_view (lh_extract_rows_compact): count, sum, nbuckets, the selected keys and one
        word of valid bits per metric -- 42 B per name at nine percentiles instead of 139 B.  numpy views of the engine's
        pinned result buffer, valid until the next call that produces results on this engine, or release().
        expand_compact() derives the full form on the host."""
        if nmetrics is None:
            nmetrics = self.engine.num_metrics() - first
        p = np.ascontiguousarray(percentiles, dtype=np.float64)
        v = N.LhExtractCompact()
        N.check(N.lib().lh_extract_rows_compact(self._h, first, nmetrics, p.ctypes.data_as(C.POINTER(C.c_double)),
                                                int(p.size), C.byref(v)), "lh_extract_rows_compact")
        np_ = int(p.size)

        def arr(ptr, ctype, count, dtype):
            if not count:
                return np.zeros(0, dtype=dtype)
            return np.frombuffer((ctype * count).from_address(ptr), dtype=dtype, count=count)

        return dict(count=arr(v.count, C.c_uint64, nmetrics, np.uint64), sum=arr(v.sum, C.c_double, nmetrics, np.float64),
                    nbuckets=arr(v.nbuckets, C.c_uint32, nmetrics, np.uint32),
                    pvalid_bits=arr(v.pvalid_bits, C.c_uint32, nmetrics, np.uint32),
                    pkeys=arr(v.pkeys, C.c_int16, nmetrics * np_, np.int16).reshape(nmetrics, np_), _view=v)

    def expand_compact(self, compact: dict):
        """extract()'s dict from extract_compact()'s, derived on the host by lh_expand_compact (bit for bit what
        lh_extract_rows returns for the same snapshot)."""
        v = compact["_view"]
        n, np_ = int(v.nmetrics), int(v.np)
        stats = (N.LhStats * max(n, 1))()
        pvals = np.zeros((n, np_), dtype=np.float64)
        pkeys = np.zeros((n, np_), dtype=np.int16)
        pvalid = np.zeros((n, np_), dtype=np.uint8)
        N.check(N.lib().lh_expand_compact(self.engine._h, C.byref(v), stats, pvals.ctypes.data_as(C.POINTER(C.c_double)),
                                          pkeys.ctypes.data_as(C.POINTER(C.c_int16)),
                                          pvalid.ctypes.data_as(C.POINTER(C.c_uint8))), "lh_expand_compact")
        raw = np.frombuffer(stats, dtype=np.dtype([("count", "<u8"), ("sum", "<f8"), ("avg", "<f8"),
                                                   ("agg_sum_add", "<u8"), ("nbuckets", "<u4"),
                                                   ("present", "<u4")]), count=n).copy()
        return dict(count=raw["count"], sum=raw["sum"], avg=raw["avg"], agg_sum_add=raw["agg_sum_add"],
                    nbuckets=raw["nbuckets"], present=raw["present"], pvals=pvals, pkeys=pkeys, pvalid=pvalid)

    def buckets(self, metric_id: int):
        """Occupied (key, count) cells of one metric, ascending key."""
        L = N.lib()
        n = C.c_size_t(0)
        N.check(L.lh_buckets(self._h, metric_id, None, None, 0, C.byref(n)), "lh_buckets")
        keys = np.zeros(n.value, dtype=np.int16)
        counts = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            N.check(L.lh_buckets(self._h, metric_id, keys.ctypes.data_as(C.POINTER(C.c_int16)),
                                 counts.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)), "lh_buckets")
        return keys, counts

    def buckets_all(self, nmetrics: int, first: int = 0):
        """CSR listing of the occupied cells of metrics [first, first+nmetrics): (offsets, keys, counts)."""
        L = N.lib()
        offsets = np.zeros(nmetrics + 1, dtype=np.uint64)
        total = C.c_size_t(0)
        op = offsets.ctypes.data_as(C.POINTER(C.c_uint64))
        N.check(L.lh_buckets_all(self._h, first, nmetrics, op, None, None, 0, C.byref(total)), "lh_buckets_all")
        keys = np.zeros(total.value, dtype=np.int16)
        counts = np.zeros(total.value, dtype=np.uint64)
        if total.value:
            N.check(L.lh_buckets_all(self._h, first, nmetrics, op, keys.ctypes.data_as(C.POINTER(C.c_int16)),
                                     counts.ctypes.data_as(C.POINTER(C.c_uint64)), total.value, C.byref(total)),
                    "lh_buckets_all")
        return offsets, keys, counts

    def serialize(self, percentiles, prefix: str, sep: str, suffix: str, underscore_to_dot: bool = False,
                  aggregates: bool = False, nmetrics: Optional[int] = None, first: int = 0) -> bytes:
        """K6: the interval's histogram keys as wire text, formatted on the device (lh_serialize).
        `percentiles` is the reference's label -> p mapping ({"%s_50": .5, ...}), in the order the keys
        should appear."""
        L = N.lib()
        if nmetrics is None:
            nmetrics = self.engine.num_metrics() - first
        labels = list(percentiles.keys())
        p = np.ascontiguousarray([percentiles[k] for k in labels], dtype=np.float64)
        lab = (C.c_char_p * max(1, len(labels)))(*[k.encode() for k in labels])
        fmt = N.LhLineFormat(prefix.encode(), sep.encode(), suffix.encode(),
                             N.FMT_UNDERSCORE_TO_DOT if underscore_to_dot else 0, 0)
        flags = N.SER_AGGREGATES if aggregates else 0
        n = C.c_size_t(0)
        pp = p.ctypes.data_as(C.POINTER(C.c_double))
        rc = L.lh_serialize(self._h, first, nmetrics, pp, lab, len(labels), C.byref(fmt), flags, None, 0, C.byref(n))
        if rc != N.ERANGE:
            N.check(rc, "lh_serialize")
        if n.value == 0:
            return b""
        buf = C.create_string_buffer(n.value)
        rc = L.lh_serialize(self._h, first, nmetrics, pp, lab, len(labels), C.byref(fmt), flags, buf, n.value,
                            C.byref(n))
        if rc != N.ERANGE:
            N.check(rc, "lh_serialize")
        return buf.raw[:n.value]

    def counter_values(self, n: Optional[int] = None, first: int = 0):
        """Counters [first, first+n) of the interval (lh_counters_collect): dict(rate, present, total, known).
        rate = the interval's amounts of the names touched (metrics.go:430-433), total = lifetime store
        after the fold (metrics.go:435-458)."""
        if n is None:
            n = self.engine.num_counters() - first
        rate, total = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        present, known = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        N.check(N.lib().lh_counters_collect(self._h, first, n, rate.ctypes.data_as(u64p), present.ctypes.data_as(u8p),
                                            total.ctypes.data_as(u64p), known.ctypes.data_as(u8p)), "lh_counters_collect")
        return dict(rate=rate, present=present.astype(bool), total=total, known=known.astype(bool))

    def serialize_counters(self, prefix: str, sep: str, suffix: str, underscore_to_dot: bool = False,
                           n: Optional[int] = None, first: int = 0) -> bytes:
        """"<name>" / "<name>_rate" wire lines of the counters, formatted on the device (lh_serialize_counters)."""
        L = N.lib()
        if n is None:
            n = self.engine.num_counters() - first
        fmt = N.LhLineFormat(prefix.encode(), sep.encode(), suffix.encode(),
                             N.FMT_UNDERSCORE_TO_DOT if underscore_to_dot else 0, 0)
        ln = C.c_size_t(0)
        N.check(L.lh_serialize_counters(self._h, first, n, C.byref(fmt), None, 0, C.byref(ln)), "lh_serialize_counters")
        if ln.value == 0:
            return b""
        buf = C.create_string_buffer(ln.value)
        N.check(L.lh_serialize_counters(self._h, first, n, C.byref(fmt), buf, ln.value, C.byref(ln)),
                "lh_serialize_counters")
        return buf.raw[:ln.value]

    def accumulate(self):
        """processHistograms' lifetime side effect (metrics.go:359-376), once per snapshot, in HBM."""
        N.check(N.lib().lh_snapshot_accumulate(self._h), "lh_snapshot_accumulate")

    def dense_row(self, metric_id: int) -> np.ndarray:
        """Dense uint64[65536] row (bin = key ^ 0x8000) rebuilt from lh_buckets."""
        keys, counts = self.buckets(metric_id)
        row = np.zeros(N.NKEYS, dtype=np.uint64)
        row[(keys.astype(np.int64) & 0xFFFF) ^ 0x8000] = counts
        return row

    @staticmethod
    def row_stride() -> int:
        """Cells from one device row to the next (lh_row_stride: more than 65 536)."""
        return int(N.lib().lh_row_stride())

    def device_rows(self):
        """(device pointer of row 0, nrows); row r starts row_stride() uint64 cells after row r - 1 and is 65 536 cells long."""
        p, n = C.c_void_p(0), C.c_uint32(0)
        N.check(N.lib().lh_snapshot_rows(self._h, C.byref(p), C.byref(n)), "lh_snapshot_rows")
        return int(p.value), int(n.value)

    def device_cells(self):
        """(device pointer of row 0, nrows, cell_bytes): the cells as they are (lh_snapshot_cells) -- uint32 on an engine of
        32-bit cells whose interval stayed below 2^32 samples; row r starts row_stride() * cell_bytes bytes after row r - 1."""
        p, n, cb = C.c_void_p(0), C.c_uint32(0), C.c_uint32(0)
        N.check(N.lib().lh_snapshot_cells(self._h, C.byref(p), C.byref(n), C.byref(cb)), "lh_snapshot_cells")
        return int(p.value), int(n.value), int(cb.value)

    def device_ranges(self) -> int:
        p = C.c_void_p(0)
        N.check(N.lib().lh_snapshot_ranges(self._h, C.byref(p)), "lh_snapshot_ranges")
        return int(p.value)

    def mark_dirty(self, first_row: int, nrows: int, lo_bin: int = 0, hi_bin: int = N.NKEYS - 1):
        N.check(N.lib().lh_snapshot_mark_dirty(self._h, first_row, nrows, lo_bin, hi_bin), "lh_snapshot_mark_dirty")

    def merge_rccl(self, comm: int, nranks: int, rank: int, nrows: int, plan: str = "allreduce"):
        """K4 through the C ABI: RCCL merge on the snapshot's stream (comm = ncclComm_t as int).
        Returns the [first, last) rows that hold merged data on this rank."""
        first, last = C.c_uint32(0), C.c_uint32(0)
        N.check(N.lib().lh_snapshot_merge(self._h, C.c_void_p(comm), nranks, rank,
                                          0 if plan == "allreduce" else 1, nrows, C.byref(first), C.byref(last)),
                "lh_snapshot_merge")
        return int(first.value), int(last.value)

    def merge_info(self) -> dict:
        """What the last merge moved (lh_snapshot_merge_info)."""
        mi = N.LhMergeInfo()
        N.check(N.lib().lh_snapshot_merge_info(self._h, C.byref(mi)), "lh_snapshot_merge_info")
        return {k: (float(getattr(mi, k)) if k.endswith("_ms") else int(getattr(mi, k)))
                for k, _ in N.LhMergeInfo._fields_ if k != "reserved"}

    def stream(self) -> int:
        p = C.c_void_p(0)
        N.check(N.lib().lh_snapshot_stream(self._h, C.byref(p)), "lh_snapshot_stream")
        return int(p.value or 0)

    def release(self):
        if self._h is not None and self._h.value:
            N.check(N.lib().lh_release(self._h), "lh_release")
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()


class Engine:
    def __init__(self, device: int = 0, max_metrics: int = 1024, num_buffers: int = 2, num_lanes: int = 4,
                 lane_samples: int = 1 << 20, max_counters: int = 1024, cell_bits: Optional[int] = None):
        """cell_bits: 0 = the library's default (uint64 cells up to 8 192 names, uint32 above), 32, 64.  None reads the TEST
        HARNESS's knobs -- this wrapper's, not the library's (liblhgpu.so never reads the environment): LH_TEST_CELL_BITS, and
        LH_TEST_WIDEN_AT (LH_OPT_WIDEN_AT_SAMPLES), so that the whole GPU suite can be run on engines of 32-bit cells, and on
        ones that move to uint64 cells in the middle of every test (tools/round.sh cells32)."""
        L = N.lib()
        cfg = N.LhConfig()
        N.check(L.lh_default_config(C.byref(cfg)), "lh_default_config")
        cfg.device, cfg.max_metrics, cfg.num_buffers = device, max_metrics, num_buffers
        cfg.num_lanes, cfg.lane_samples, cfg.max_counters = num_lanes, lane_samples, max_counters
        widen_at = 0
        if cell_bits is None:
            cell_bits = int(os.environ.get("LH_TEST_CELL_BITS", "0"))
            widen_at = int(os.environ.get("LH_TEST_WIDEN_AT", "0"))
        cfg.cell_bits = cell_bits
        h = C.c_void_p(0)
        N.check(L.lh_create(C.byref(cfg), C.byref(h)), "lh_create")
        self._h = h
        self.max_metrics = max_metrics
        if widen_at:
            N.check(L.lh_set_option(h, N.OPT_WIDEN_AT_SAMPLES, widen_at), "lh_set_option")

    # -- names -------------------------------------------------------------
    def intern(self, name: str) -> int:
        b = name.encode()
        out = C.c_uint32(0)
        N.check(N.lib().lh_intern(self._h, b, len(b), C.byref(out)), "lh_intern")
        return int(out.value)

    def lookup(self, name: str) -> Optional[int]:
        b = name.encode()
        out = C.c_uint32(0)
        rc = N.lib().lh_lookup(self._h, b, len(b), C.byref(out))
        if rc == N.ERANGE:
            return None
        N.check(rc, "lh_lookup")
        return int(out.value)

    def num_metrics(self) -> int:
        out = C.c_uint32(0)
        N.check(N.lib().lh_num_metrics(self._h, C.byref(out)), "lh_num_metrics")
        return int(out.value)

    def metric_name(self, metric_id: int) -> str:
        ln = C.c_size_t(0)
        buf = C.create_string_buffer(4096)
        N.check(N.lib().lh_metric_name(self._h, metric_id, buf, 4096, C.byref(ln)), "lh_metric_name")
        return buf.raw[:min(ln.value, 4096)].decode()

    # -- ingest ------------------------------------------------------------
    def submit(self, metric_id: int, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        N.check(N.lib().lh_submit(self._h, metric_id, v.ctypes.data, v.size), "lh_submit")

    def submit_pairs(self, ids, values):
        """(id, value) pairs from host arrays (copied before the call returns).  uint16 ids take the 10-byte form
        (lh_submit_pairs16: legal for <= 65 536 names), anything else is sent as uint32 (lh_submit_pairs)."""
        narrow = isinstance(ids, np.ndarray) and ids.dtype == np.uint16
        i = np.ascontiguousarray(ids, dtype=np.uint16 if narrow else np.uint32)
        v = np.ascontiguousarray(values, dtype=np.float64)
        if i.size != v.size:
            raise ValueError("ids and values differ in length")
        if narrow:
            N.check(N.lib().lh_submit_pairs16(self._h, i.ctypes.data, v.ctypes.data, v.size), "lh_submit_pairs16")
        else:
            N.check(N.lib().lh_submit_pairs(self._h, i.ctypes.data, v.ctypes.data, v.size), "lh_submit_pairs")

    def reserve_pairs(self, want: int, id_bits: int = 32):
        """(ids uint32[granted] or uint16[granted], values float64[granted], token): views of a pinned staging buffer
        to fill in place (lh_reserve_pairs / lh_reserve_pairs16); publish the first n with commit_pairs(token, n)."""
        pi, pv, g, tok = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_uint32(0)
        fn, ct = (N.lib().lh_reserve_pairs16, C.c_uint16) if id_bits == 16 else (N.lib().lh_reserve_pairs, C.c_uint32)
        N.check(fn(self._h, want, C.byref(pi), C.byref(pv), C.byref(g), C.byref(tok)), "lh_reserve_pairs")
        ids = np.ctypeslib.as_array(C.cast(pi, C.POINTER(ct)), shape=(g.value,))
        vals = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(g.value,))
        return ids, vals, tok.value

    def commit_pairs(self, token: int, n: int):
        N.check(N.lib().lh_commit_pairs(self._h, token, n), "lh_commit_pairs")

    def submit_pairs_in_place(self, ids, values):
        """The same stream as submit_pairs through reserve / fill / commit: one host-side copy, no id scan.  uint16 ids
        are staged as uint16 (10 bytes per pair over PCIe)."""
        narrow = isinstance(ids, np.ndarray) and ids.dtype == np.uint16
        i = np.ascontiguousarray(ids, dtype=np.uint16 if narrow else np.uint32)
        v = np.ascontiguousarray(values, dtype=np.float64)
        if i.size != v.size:
            raise ValueError("ids and values differ in length")
        done = 0
        while done < v.size:
            di, dv, tok = self.reserve_pairs(v.size - done, 16 if narrow else 32)
            k = di.size
            np.copyto(di, i[done:done + k])
            np.copyto(dv, v[done:done + k])
            self.commit_pairs(tok, k)
            done += k

    def submit_device(self, metric_id: int, d_values, n: Optional[int] = None, stream=None):
        n = int(d_values.numel()) if n is None else n
        N.check(N.lib().lh_submit_device(self._h, metric_id, _ptr(d_values), n, _stream_handle(stream)),
                "lh_submit_device")

    def submit_pairs_device(self, d_ids, d_values, n: Optional[int] = None, stream=None):
        """Device-resident (id, value) pairs.  A 2-byte id tensor (torch.int16 / torch.uint16: the bits of uint16 ids)
        takes lh_submit_pairs16_device, a 4-byte one lh_submit_pairs_device."""
        n = int(d_values.numel()) if n is None else n
        if getattr(d_ids, "element_size", lambda: 4)() == 2:
            N.check(N.lib().lh_submit_pairs16_device(self._h, _ptr(d_ids), _ptr(d_values), n, _stream_handle(stream)),
                    "lh_submit_pairs16_device")
            return
        N.check(N.lib().lh_submit_pairs_device(self._h, _ptr(d_ids), _ptr(d_values), n, _stream_handle(stream)),
                "lh_submit_pairs_device")

    # -- counters (metrics.go:251-269) -----------------------------------------
    def intern_counter(self, name: str) -> int:
        b = name.encode()
        out = C.c_uint32(0)
        N.check(N.lib().lh_intern_counter(self._h, b, len(b), C.byref(out)), "lh_intern_counter")
        return int(out.value)

    def num_counters(self) -> int:
        out = C.c_uint32(0)
        N.check(N.lib().lh_num_counters(self._h, C.byref(out)), "lh_num_counters")
        return int(out.value)

    def counter_name(self, cid: int) -> str:
        ln = C.c_size_t(0)
        buf = C.create_string_buffer(4096)
        N.check(N.lib().lh_counter_name(self._h, cid, buf, 4096, C.byref(ln)), "lh_counter_name")
        return buf.raw[:min(ln.value, 4096)].decode()

    def submit_counts(self, ids, amounts):
        i = np.ascontiguousarray(ids, dtype=np.uint32)
        a = np.ascontiguousarray(amounts, dtype=np.uint64)
        if i.size != a.size:
            raise ValueError("ids and amounts differ in length")
        N.check(N.lib().lh_submit_counts(self._h, i.ctypes.data, a.ctypes.data, a.size), "lh_submit_counts")

    def submit_counts_device(self, d_ids, d_amounts, n: Optional[int] = None, stream=None):
        n = int(d_amounts.numel()) if n is None else n
        N.check(N.lib().lh_submit_counts_device(self._h, _ptr(d_ids), _ptr(d_amounts), n, _stream_handle(stream)),
                "lh_submit_counts_device")

    def flush(self):
        N.check(N.lib().lh_flush(self._h), "lh_flush")

    def sync(self):
        N.check(N.lib().lh_sync(self._h), "lh_sync")

    # -- epoch -------------------------------------------------------------
    def flip(self) -> Snapshot:
        h = C.c_void_p(0)
        N.check(N.lib().lh_flip(self._h, C.byref(h)), "lh_flip")
        return Snapshot(self, h.value)

    # -- codec (parity tests) -----------------------------------------------
    def compress_device(self, d_values, d_keys, n: int, golog: bool = False, stream=None):
        fn = N.lib().lh_compress_device_golog if golog else N.lib().lh_compress_device
        N.check(fn(self._h, _ptr(d_values), _ptr(d_keys), n, _stream_handle(stream)), "lh_compress_device")

    def codec_tables(self):
        tx = np.zeros(N.NTHRESH, dtype=np.float64)
        d = np.zeros(N.NKEYS, dtype=np.float64)
        N.check(N.lib().lh_codec_tables(self._h, tx.ctypes.data_as(C.POINTER(C.c_double)),
                                        d.ctypes.data_as(C.POINTER(C.c_double))), "lh_codec_tables")
        return tx, d

    def selftest_vlog(self) -> float:
        out = C.c_double(0)
        N.check(N.lib().lh_selftest_vlog(self._h, C.byref(out)), "lh_selftest_vlog")
        return float(out.value)

    def lifetime(self, n: Optional[int] = None, first: int = 0):
        """(count, sum) lifetime stores of metrics [first, first+n) (histogramCountStore, metrics.go:127)."""
        if n is None:
            n = self.num_metrics() - first
        cnt = np.zeros(n, dtype=np.uint64)
        sm = np.zeros(n, dtype=np.uint64)
        if n:
            N.check(N.lib().lh_lifetime(self._h, first, n, cnt.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        sm.ctypes.data_as(C.POINTER(C.c_uint64))), "lh_lifetime")
        return cnt, sm

    def format_f(self, values) -> list:
        """Go's %f of each float64, formatted by the device formatter (lh_format_f)."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        out: list = []
        step = 1 << 16
        for i in range(0, v.size, step):
            part = v[i:i + step]
            buf = C.create_string_buffer(part.size * N.FMT_SLOT)
            lens = np.zeros(part.size, dtype=np.uint32)
            N.check(N.lib().lh_format_f(self._h, part.ctypes.data_as(C.POINTER(C.c_double)), part.size, buf,
                                        N.FMT_SLOT, lens.ctypes.data_as(C.POINTER(C.c_uint32))), "lh_format_f")
            raw = buf.raw
            out.extend(raw[k * N.FMT_SLOT:k * N.FMT_SLOT + int(lens[k])].decode() for k in range(part.size))
        return out

    def counters(self) -> dict:
        """Self-metrics of the engine (lh_get_counters)."""
        c = N.LhCounters()
        N.check(N.lib().lh_get_counters(self._h, C.byref(c)), "lh_get_counters")
        return {k: int(getattr(c, k)) for k, _ in N.LhCounters._fields_ if k != "reserved"}

    def set_option(self, option: int, value: int):
        """Dispatch settings (lh_set_option): they choose among exact kernel paths, never a result."""
        N.check(N.lib().lh_set_option(self._h, int(option), int(value)), "lh_set_option")

    def close(self):
        if self._h is not None and self._h.value:
            N.check(N.lib().lh_destroy(self._h), "lh_destroy")
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
