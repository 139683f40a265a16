"""Host-side mirror of the reference's MetricSystem API over the C ABI.

The reference is Go and no Go toolchain exists in this image, so the binding a
maintainer would add to metrics.go is mirrored here in Python
with the same method names, argument meaning, key naming and error behaviour, so
that tests/test_metric_system.py reads like metrics_test.go.

What runs where:
  * Histogram / TimerToken.Stop  -> per-thread staging of (id, value) pairs, shipped
    with lh_submit_pairs when a buffer fills and at every collection.  This is the
    batching the cgo binding needs (a crossing costs more than the Go fast path).
    compress + fan-in happen on the GPU (metrics.go:273-295, 316-322).
  * collectRawMetrics            -> lh_flip for the histogram half (metrics.go:460-463);
    counters / rates / gauges stay host side exactly as in the reference
    (metrics.go:425-458, 465-470): they carry no codec work.
  * processMetrics               -> lh_extract for processHistograms + percentile
    (metrics.go:336-418); key naming, lifetime `_agg_*` stores (uint64 truncation,
    integer division) follow metrics.go:349-376, 590-608.
  * reaper / subscriptions       -> a small thread + queue.Queue stand-in for the Go
    channels (metrics.go:530-639); glue, out of the accelerated scope.

No bucket arithmetic happens in this file.
"""
from __future__ import annotations

import queue
import sys
import threading
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional

import numpy as np

from . import _native as N
from .engine import Engine, Snapshot

_U64 = (1 << 64) - 1

# metrics.go:145-155
DEFAULT_PERCENTILES = {
    "%s_min": 0.0, "%s_50": .5, "%s_75": .75, "%s_90": .9, "%s_95": .95,
    "%s_99": .99, "%s_99.9": .999, "%s_99.99": .9999, "%s_max": 1.0,
}


def _hostname() -> str:
    import socket
    try:
        return socket.gethostname()
    except OSError:
        return "unknown"  # graphite.go:52-55


def _go_f(v: float) -> str:
    """fmt.Sprintf("%f", v) for the few host-side keys (counters, rates, gauges)."""
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "+Inf" if v > 0 else "-Inf"
    return "%f" % v


def GraphiteProtocol(ms: "ProcessedMetricSet") -> bytes:  # graphite.go:37-75
    if ms.wire_format == "graphite":
        return ms.wire
    host, ts = _hostname(), int(ms.Time)
    return "".join(f"cockroach.{host}.{k.replace('_', '.')} {_go_f(v)} {ts}\n" for k, v in ms.Metrics.items()).encode()


def OpenTSDBProtocol(ms: "ProcessedMetricSet") -> bytes:  # opentsdb.go:45-85
    if ms.wire_format == "opentsdb":
        return ms.wire
    host, ts = _hostname(), int(ms.Time)
    return "".join(f"put {k} {ts} {_go_f(v)} host={host}\n" for k, v in ms.Metrics.items()).encode()


@dataclass
class ProcessedMetricSet:  # metrics.go:47-50
    Time: float
    Metrics: Dict[str, float]
    # bulk wire path (MetricSystem.SetWireFormat): the whole request, histogram keys formatted on the device
    wire_format: Optional[str] = None
    wire: bytes = b""


@dataclass
class RawMetricSet:  # metrics.go:54-60
    Time: float
    Counters: Dict[str, int]
    Rates: Dict[str, int]
    Gauges: Dict[str, float]
    _snapshot: Optional[Snapshot] = None
    _names: List[str] = field(default_factory=list)
    _hist: Optional[Dict[str, Dict[int, int]]] = None

    @property
    def Histograms(self) -> Dict[str, Dict[int, int]]:
        """map[name]map[int16]count, occupied cells only (lh_buckets)."""
        if self._hist is None:
            self._hist = {}
            if self._snapshot is not None:
                offsets, keys, counts = self._snapshot.buckets_all(len(self._names))
                for mid, name in enumerate(self._names):
                    lo, hi = int(offsets[mid]), int(offsets[mid + 1])
                    if hi > lo:
                        self._hist[name] = {int(k): int(c) for k, c in zip(keys[lo:hi], counts[lo:hi])}
        return self._hist

    def release(self):
        if self._snapshot is not None:
            self._snapshot.release()
            self._snapshot = None


class TimerToken:  # metrics.go:63-67
    __slots__ = ("Name", "Start", "MetricSystem")

    def __init__(self, name: str, start_ns: int, ms: "MetricSystem"):
        self.Name, self.Start, self.MetricSystem = name, start_ns, ms

    def Stop(self) -> int:
        """metrics.go:242-246: submits float64(duration ns) as a histogram sample."""
        duration = time.perf_counter_ns() - self.Start
        self.MetricSystem.Histogram(self.Name, float(duration))
        return duration


class _Stage:
    """Per-thread staging buffer (the per-P buffer of the cgo binding)."""
    __slots__ = ("ids", "vals", "n", "lock")

    def __init__(self, cap: int):
        self.ids = np.empty(cap, dtype=np.uint32)
        self.vals = np.empty(cap, dtype=np.float64)
        self.n = 0
        self.lock = threading.Lock()


class MetricSystem:
    def __init__(self, interval: float, sysStats: bool, *, device: int = 0, max_metrics: int = 1024,
                 stage_samples: int = 4096, engine: Optional[Engine] = None):
        """NewMetricSystem(interval, sysStats), metrics.go:143.  `interval` in seconds."""
        self.percentiles = dict(DEFAULT_PERCENTILES)
        self._wire_format: Optional[str] = None
        self._wire_keep_map = True
        self.interval = float(interval)
        self._device, self._max_metrics, self._stage_cap = device, max_metrics, stage_samples
        self._engine = engine
        self._owns_engine = engine is None   # Stop() closes only an engine this system created
        self._engine_lock = threading.Lock()
        self._ids: Dict[str, int] = {}
        self._names: List[str] = []
        self._tls = threading.local()
        self._stages: List[_Stage] = []
        self._stages_lock = threading.Lock()
        self._hist_used = False
        # counters (host side, as in the reference)
        self.counterStore: Dict[str, int] = {}
        self.counterCache: Dict[str, int] = {}
        self.counterMu = threading.Lock()
        # (histogramCountStore, metrics.go:122-127, lives in HBM: Snapshot.accumulate / Engine.lifetime)
        self.gaugeFuncs: Dict[str, Callable[[], float]] = {}
        self.gaugeFuncsMu = threading.Lock()
        # subscriptions
        self._raw_subs: Dict[int, queue.Queue] = {}
        self._proc_subs: Dict[int, queue.Queue] = {}
        self._raw_bad: Dict[int, int] = {}
        self._proc_bad: Dict[int, int] = {}
        self._subs_mu = threading.Lock()
        self._pending_subs: "queue.Queue" = queue.Queue()
        self.reaping = False
        self._shutdown = threading.Event()
        self._reaper_thread: Optional[threading.Thread] = None
        if sysStats:  # metrics.go:172-193 (Go runtime stats; nearest Python analogues)
            import gc
            import resource
            self.gaugeFuncs["sys.Alloc"] = lambda: float(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024)
            self.gaugeFuncs["sys.NumGC"] = lambda: float(sum(s["collections"] for s in gc.get_stats()))
            self.gaugeFuncs["sys.PauseTotalNs"] = lambda: 0.0
            self.gaugeFuncs["sys.NumGoroutine"] = lambda: float(threading.active_count())

    # -- configuration -------------------------------------------------------------
    def SpecifyPercentiles(self, percentiles: Dict[str, float]):  # metrics.go:199
        self.percentiles = percentiles

    # -- engine ------------------------------------------------------------------
    def _eng(self) -> Engine:
        if self._engine is None:
            with self._engine_lock:
                if self._engine is None:
                    self._engine = Engine(device=self._device, max_metrics=self._max_metrics)
        return self._engine

    def _id(self, name: str) -> int:
        mid = self._ids.get(name)
        if mid is None:
            mid = self._eng().intern(name)  # idempotent, thread-safe in C
            with self._stages_lock:
                self._ids[name] = mid
                while len(self._names) <= mid:
                    self._names.append("")
                self._names[mid] = name
        return mid

    # -- ingest ------------------------------------------------------------------
    def StartTimer(self, name: str) -> TimerToken:  # metrics.go:232
        return TimerToken(name, time.perf_counter_ns(), self)

    def Counter(self, name: str, amount: int):  # metrics.go:251-269
        with self.counterMu:
            self.counterCache[name] = (self.counterCache.get(name, 0) + int(amount)) & _U64

    def Histogram(self, name: str, value: float):  # metrics.go:273-295
        st = getattr(self._tls, "stage", None)
        if st is None:
            st = _Stage(self._stage_cap)
            self._tls.stage = st
            with self._stages_lock:
                self._stages.append(st)
        mid = self._id(name)
        with st.lock:
            st.ids[st.n] = mid
            st.vals[st.n] = value
            st.n += 1
            self._hist_used = True
            if st.n == st.ids.size:
                self._ship(st)

    def HistogramBatch(self, name: str, values):
        """Bulk form of Histogram for array producers (one lh_submit)."""
        mid = self._id(name)
        self._hist_used = True
        self._eng().submit(mid, values)

    def _ship(self, st: _Stage):
        if st.n:
            self._eng().submit_pairs(st.ids[: st.n], st.vals[: st.n])
            st.n = 0

    def RegisterGaugeFunc(self, name: str, f: Callable[[], float]):  # metrics.go:299
        with self.gaugeFuncsMu:
            self.gaugeFuncs[name] = f

    def DeregisterGaugeFunc(self, name: str):  # metrics.go:306
        with self.gaugeFuncsMu:
            self.gaugeFuncs.pop(name, None)

    # -- collection ----------------------------------------------------------------
    def collectRawMetrics(self) -> RawMetricSet:  # metrics.go:420-479
        ivl_ns = max(1, int(self.interval * 1e9))
        normalized = (time.time_ns() // ivl_ns * ivl_ns) / 1e9

        with self.counterMu:
            fresh = self.counterCache
            self.counterCache = {}
        rates = dict(fresh)
        for name, count in fresh.items():
            self.counterStore[name] = (self.counterStore.get(name, 0) + count) & _U64
        counters = dict(self.counterStore)

        snap, names = None, []
        if self._hist_used:
            with self._stages_lock:
                stages = list(self._stages)
            for st in stages:       # every sample staged before the flip lands in this interval
                with st.lock:
                    self._ship(st)
            try:
                snap = self._eng().flip()  # epoch boundary, metrics.go:460-463
            except N.LhError as exc:
                if exc.code != N.EBUSY:
                    raise
                snap = None         # every epoch buffer is still being read: the interval keeps accumulating
            # AFTER the flip: _id() appends a name before its first sample is staged, so this copy covers every
            # row that can hold data in the snapshot (a copy taken earlier could miss a name interned in between,
            # and that name's first interval would be cleared without ever being reported)
            with self._stages_lock:
                names = list(self._names)

        with self.gaugeFuncsMu:
            gauges = {name: f() for name, f in self.gaugeFuncs.items()}
        return RawMetricSet(Time=normalized, Counters=counters, Rates=rates, Gauges=gauges,
                            _snapshot=snap, _names=names)

    def processHistograms(self, raw: RawMetricSet) -> Dict[str, float]:  # metrics.go:336-387, all names at once
        out: Dict[str, float] = {}
        if raw._snapshot is None or not raw._names:
            return out
        labels = list(self.percentiles.keys())
        res = raw._snapshot.extract([self.percentiles[k] for k in labels], len(raw._names))
        for mid, name in enumerate(raw._names):
            if not res["present"][mid]:
                continue  # the name is absent from this interval's histogramCache
            count = int(res["count"][mid])
            out[f"{name}_count"] = float(count)
            out[f"{name}_sum"] = float(res["sum"][mid])
            out[f"{name}_avg"] = float(res["avg"][mid])
            for i, label in enumerate(labels):
                if res["pvalid"][mid][i]:
                    out[label % name] = float(res["pvals"][mid][i])
                # else: the reference logs "unable to calculate percentile" and omits the key (metrics.go:379-384)
        return out

    def processMetrics(self, raw: RawMetricSet) -> ProcessedMetricSet:  # metrics.go:483-506
        metrics: Dict[str, float] = {}
        for name, count in raw.Counters.items():
            metrics[name] = float(count)
        for name, count in raw.Rates.items():
            metrics[f"{name}_rate"] = float(count)
        host_keys = dict(metrics)
        wf = self._wire_format
        wire = b""
        if raw._snapshot is not None and raw._names:
            # processHistograms' lifetime side effect (metrics.go:359-376: uint64(totalSum), wrapping adds), applied
            # once per snapshot to the ONE store there is -- in HBM -- whichever of the map / wire paths runs
            raw._snapshot.accumulate()
        if wf is not None and raw._snapshot is not None and raw._names:
            wire = self.serializeHistograms(raw, wf)
        if wf is None or self._wire_keep_map:
            metrics.update(self.processHistograms(raw))
        for name, value in raw.Gauges.items():
            metrics[name] = value
            host_keys[name] = value
        out = ProcessedMetricSet(Time=raw.Time, Metrics=metrics)
        if wf is not None and raw._snapshot is not None and raw._names:
            rest = ProcessedMetricSet(Time=raw.Time, Metrics=host_keys)
            out.wire_format = wf
            out.wire = wire + (GraphiteProtocol(rest) if wf == "graphite" else OpenTSDBProtocol(rest))
        return out

    def SetWireFormat(self, kind: Optional[str], histogram_keys_in_map: bool = True):
        """Opt-in bulk wire path (no counterpart in the reference): processMetrics also runs
        lh_snapshot_accumulate + lh_serialize (K6) and attaches the request GraphiteProtocol /
        OpenTSDBProtocol would build; with histogram_keys_in_map=False the histogram keys are not
        inserted into Metrics.  kind: None | "graphite" | "opentsdb"."""
        if kind not in (None, "graphite", "opentsdb"):
            raise ValueError(kind)
        self._wire_keep_map = histogram_keys_in_map
        self._wire_format = kind

    def serializeHistograms(self, raw: RawMetricSet, kind: str) -> bytes:
        """Histogram keys of the interval incl. _agg_* as wire text, formatted on the device."""
        host, ts = _hostname(), str(int(raw.Time))
        if kind == "graphite":      # graphite.go:40
            return raw._snapshot.serialize(self.percentiles, f"cockroach.{host}.", " ", f" {ts}\n",
                                           underscore_to_dot=True, aggregates=True, nmetrics=len(raw._names))
        return raw._snapshot.serialize(self.percentiles, "put ", f" {ts} ", f" host={host}\n",  # opentsdb.go:48
                                       aggregates=True, nmetrics=len(raw._names))

    def _add_aggregates(self, raw: RawMetricSet, processed: ProcessedMetricSet):  # metrics.go:590-608
        if raw._snapshot is None or not raw._names:
            return
        agg_count, agg_sum = self._eng().lifetime(len(raw._names))   # histogramCountStore lives in HBM (lh_lifetime)
        for mid, name in enumerate(raw._names):
            if f"{name}_count" not in processed.Metrics:
                continue
            c, sm = int(agg_count[mid]), int(agg_sum[mid])
            if c > 0:
                processed.Metrics[f"{name}_agg_avg"] = float(sm // c)  # integer division
                processed.Metrics[f"{name}_agg_count"] = float(c)
                processed.Metrics[f"{name}_agg_sum"] = float(sm)

    # -- subscriptions (glue; metrics.go:203-228, 508-525) -----------------------------
    def SubscribeToRawMetrics(self, q: "queue.Queue"):
        self._pending_subs.put(("raw+", q))

    def UnsubscribeFromRawMetrics(self, q: "queue.Queue"):
        self._pending_subs.put(("raw-", q))

    def SubscribeToProcessedMetrics(self, q: "queue.Queue"):
        self._pending_subs.put(("proc+", q))

    def UnsubscribeFromProcessedMetrics(self, q: "queue.Queue"):
        self._pending_subs.put(("proc-", q))

    def updateSubscribers(self):
        with self._subs_mu:
            while True:
                try:
                    op, q = self._pending_subs.get_nowait()
                except queue.Empty:
                    return
                table = self._raw_subs if op.startswith("raw") else self._proc_subs
                if op.endswith("+"):
                    table[id(q)] = q
                else:
                    table.pop(id(q), None)

    @staticmethod
    def _broadcast(subs, bad, item):
        # non-blocking send; a subscriber that is full on 2 consecutive intervals is dropped
        # (metrics.go:567-580, 613-626)
        for key, q in list(subs.items()):
            try:
                q.put_nowait(item)
                bad.pop(key, None)
            except queue.Full:
                bad[key] = bad.get(key, 0) + 1
                if bad[key] >= 2:
                    subs.pop(key, None)

    def _tick(self):
        raw = self.collectRawMetrics()
        try:
            self.updateSubscribers()
            with self._subs_mu:
                if self._raw_subs:
                    _ = raw.Histograms  # materialise before the snapshot is released
                self._broadcast(self._raw_subs, self._raw_bad, raw)
            processed = self.processMetrics(raw)
            self._add_aggregates(raw, processed)
        finally:
            raw.release()           # a leaked snapshot would pin an epoch buffer: every later flip LH_EBUSY
        with self._subs_mu:
            self._broadcast(self._proc_subs, self._proc_bad, processed)

    def reaper(self):  # metrics.go:530-639
        self.reaping = True
        ivl_ns = max(1, int(self.interval * 1e9))
        while True:
            tts = (ivl_ns - time.time_ns() % ivl_ns) / 1e9
            if self._shutdown.wait(tts):
                self.reaping = False
                return
            try:
                self._tick()
            except Exception as exc:  # noqa: BLE001 -- the reference logs and carries on (metrics.go:379-384, 572-576)
                print(f"loghisto: interval dropped: {exc!r}", file=sys.stderr)

    def Start(self):  # metrics.go:644
        if not self.reaping:
            self._shutdown.clear()  # Start after Stop
            self.reaping = True
            self._reaper_thread = threading.Thread(target=self.reaper, daemon=True)
            self._reaper_thread.start()

    def Stop(self):  # metrics.go:651
        self._shutdown.set()
        if self._reaper_thread is not None:
            self._reaper_thread.join(timeout=5)
            self._reaper_thread = None
        if self._engine is not None and self._owns_engine:
            self._engine.close()
            self._engine = None
            with self._stages_lock:   # ids of a destroyed engine must not reach its successor
                self._ids.clear()
                self._names.clear()
                self._hist_used = False
