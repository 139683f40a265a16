"""The reference's own tests (metrics_test.go), restated against the host mirror of
the MetricSystem API (loghisto_amd/metric_system.py) over the C ABI.

Counter / rate / gauge / subscription tests need no GPU (that logic stays on the
host in the reference too); anything that submits a histogram sample is marked gpu.
"""
import queue
import threading
import time

import numpy as np
import pytest

from loghisto_amd.metric_system import MetricSystem

US = 1e-6


def test_rate():
    # TestRate, metrics_test.go:202-223
    ms = MetricSystem(US, False)
    ms.Counter("rate1", 777)
    metrics = ms.processMetrics(ms.collectRawMetrics()).Metrics
    assert metrics["rate1_rate"] == 777
    ms.Counter("rate1", 1223)
    metrics = ms.processMetrics(ms.collectRawMetrics()).Metrics
    assert metrics["rate1_rate"] == 1223
    ms.Counter("rate1", 1223)
    ms.Counter("rate1", 1223)
    metrics = ms.processMetrics(ms.collectRawMetrics()).Metrics
    assert metrics["rate1_rate"] == 2446


def test_counter():
    # TestCounter, metrics_test.go:225-240
    ms = MetricSystem(US, False)
    ms.Counter("counter1", 3290)
    metrics = ms.processMetrics(ms.collectRawMetrics()).Metrics
    assert metrics["counter1"] == 3290
    ms.Counter("counter1", 10000)
    metrics = ms.processMetrics(ms.collectRawMetrics()).Metrics
    assert metrics["counter1"] == 13290


def test_sys_stats():
    # TestSysStats, metrics_test.go:174-181
    ms = MetricSystem(US, True)
    gauges = ms.collectRawMetrics().Gauges
    assert gauges["sys.Alloc"] > 0


def test_raw_broadcast():
    # TestRawBroadcast, metrics_test.go:321-346
    q = queue.Queue(128)
    ms = MetricSystem(1e-3, False)
    ms.SubscribeToRawMetrics(q)
    ms.Counter("counter2", 10)
    ms.Counter("counter2", 111)
    ms.Start()
    raw = q.get(timeout=2)
    assert raw.Counters["counter2"] == 121
    assert raw.Rates["counter2"] == 121
    ms.UnsubscribeFromRawMetrics(q)
    ms.Stop()


def test_update_subscribers():
    # TestUpdateSubscribers, metrics_test.go:242-287
    rq, pq = queue.Queue(1), queue.Queue(1)
    ms = MetricSystem(2e-3, False)
    ms.SubscribeToRawMetrics(rq)
    ms.SubscribeToProcessedMetrics(pq)
    ms.Counter("counter5", 33)
    ms.Start()
    assert rq.get(timeout=2) is not None
    ms.UnsubscribeFromRawMetrics(rq)
    assert pq.get(timeout=2) is not None
    ms.UnsubscribeFromProcessedMetrics(pq)
    time.sleep(0.05)
    for q_ in (rq, pq):     # drain anything sent before the unsubscribe was processed
        while not q_.empty():
            q_.get_nowait()
    time.sleep(0.05)
    assert rq.empty() and pq.empty()
    ms.Stop()


def test_slow_subscriber_is_dropped_not_blocking():
    # metrics.go:567-580: full on 2 consecutive intervals => forgotten; the reaper never blocks
    q = queue.Queue(1)
    ms = MetricSystem(1e-3, False)
    ms.SubscribeToProcessedMetrics(q)
    ms.Start()
    time.sleep(0.1)
    ms.Stop()
    assert q.qsize() == 1 and not ms._proc_subs


def test_metric_system_stop():
    # TestMetricSystemStop, metrics_test.go:348-363
    before = threading.active_count()
    ms = MetricSystem(US, False)
    ms.Start()
    ms.Stop()
    time.sleep(0.02)
    assert threading.active_count() <= before


# ---- histogram paths: GPU ---------------------------------------------------------

@pytest.mark.gpu
def test_timer(native_lib, torch_cuda):
    # TestTimer, metrics_test.go:183-200
    ms = MetricSystem(US, False)
    t1 = ms.StartTimer("timer1")
    t2 = ms.StartTimer("timer1")
    time.sleep(50e-6)
    t1.Stop()
    time.sleep(5e-6)
    t2.Stop()
    t3 = ms.StartTimer("timer1")
    time.sleep(10e-6)
    t3.Stop()
    raw = ms.collectRawMetrics()
    result = ms.processMetrics(raw).Metrics
    raw.release()
    assert result["timer1_min"] <= result["timer1_50"] <= result["timer1_max"]
    assert result["timer1_count"] == 3
    ms.Stop()


@pytest.mark.gpu
def test_processed_broadcast(native_lib, torch_cuda):
    # TestProcessedBroadcast, metrics_test.go:289-319
    q = queue.Queue(128)
    ms = MetricSystem(1e-3, False)
    ms.SubscribeToProcessedMetrics(q)
    ms.Histogram("histogram1", 33)
    ms.Histogram("histogram1", 59)
    ms.Histogram("histogram1", 330000)
    ms.Start()
    pm = q.get(timeout=5)
    assert int(pm.Metrics["histogram1_sum"]) == 331132
    assert int(pm.Metrics["histogram1_agg_avg"]) == 110377
    assert int(pm.Metrics["histogram1_count"]) == 3
    ms.UnsubscribeFromProcessedMetrics(q)
    ms.Stop()


@pytest.mark.gpu
def test_example_metric_system_keys(native_lib, torch_cuda):
    # ExampleMetricSystem, metrics_test.go:28-109: presence of the documented keys
    q = queue.Queue(2)
    ms = MetricSystem(2e-3, True)
    ms.SubscribeToProcessedMetrics(q)
    ms.RegisterGaugeFunc("gauge", lambda: 33.0)
    tok = ms.StartTimer("submit_metrics")
    ms.Counter("range_splits", 1)
    ms.Histogram("some_ipc", 123)
    tok.Stop()
    ms.Start()
    m = q.get(timeout=5).Metrics
    for key in ("range_splits", "range_splits_rate", "some_ipc_99.9", "some_ipc_max", "some_ipc_count",
                "some_ipc_agg_count", "some_ipc_sum", "some_ipc_avg", "some_ipc_agg_avg", "submit_metrics_sum",
                "sys.NumGoroutine", "sys.PauseTotalNs", "gauge"):
        assert key in m, key
    assert m["some_ipc_count"] == 1 and m["some_ipc_max"] == m["some_ipc_min"]
    assert abs(m["some_ipc_max"] / 123 - 1) < 0.01
    ms.Stop()


@pytest.mark.gpu
def test_raw_histograms_and_intervals(native_lib, torch_cuda):
    import numpy as np
    import oracle
    ms = MetricSystem(US, False, stage_samples=64)
    rng = np.random.default_rng(4)
    a = rng.lognormal(8, 1, 1000)
    for v in a:                      # crosses several 64-sample staging buffers
        ms.Histogram("lat", float(v))
    ms.HistogramBatch("bulk", a * 3)
    raw = ms.collectRawMetrics()
    h = raw.Histograms
    want = oracle.histogram_dense(a)
    nz = np.nonzero(want)[0]
    assert h["lat"] == {int(k): int(want[b]) for k, b in zip(oracle.bin_to_key(nz), nz)}
    assert sum(h["bulk"].values()) == 1000
    pm = ms.processMetrics(raw).Metrics
    raw.release()
    assert pm["lat_count"] == 1000 and pm["bulk_count"] == 1000
    # next interval: names without samples are absent, lifetime aggregates persist
    ms.Histogram("lat", 5.0)
    raw = ms.collectRawMetrics()
    pm = ms.processMetrics(raw)
    ms._add_aggregates(raw, pm)
    raw.release()
    assert "bulk_count" not in pm.Metrics and pm.Metrics["lat_count"] == 1
    assert pm.Metrics["lat_agg_count"] == 1001
    ms.Stop()


@pytest.mark.gpu
def test_invalid_percentile_is_omitted_like_the_reference(native_lib, torch_cuda):
    ms = MetricSystem(US, False)
    ms.SpecifyPercentiles({"%s_p50": 0.5, "%s_bad": 1.5})
    ms.Histogram("x", 10.0)
    raw = ms.collectRawMetrics()
    m = ms.processMetrics(raw).Metrics
    raw.release()
    assert "x_p50" in m and "x_bad" not in m   # metrics.go:379-384
    ms.Stop()


def test_serializers_per_key():
    """GraphiteProtocol / OpenTSDBProtocol (graphite.go:37-75, opentsdb.go:45-85) on the values the
    reference's own serializer tests submit (graphite_test.go, opentsdb_test.go)."""
    import socket
    from loghisto_amd.metric_system import GraphiteProtocol, OpenTSDBProtocol, ProcessedMetricSet
    host = socket.gethostname()
    ms = ProcessedMetricSet(Time=1411104988.0, Metrics={"test.3": 50.54, "test_4": 10.21, "nan": float("nan"),
                                                        "inf": float("inf")})
    g = GraphiteProtocol(ms).decode().split("\n")
    assert f"cockroach.{host}.test.3 50.540000 1411104988" in g and f"cockroach.{host}.test.4 10.210000 1411104988" in g
    assert f"cockroach.{host}.nan NaN 1411104988" in g and f"cockroach.{host}.inf +Inf 1411104988" in g
    t = OpenTSDBProtocol(ms).decode().split("\n")
    assert f"put test.3 1411104988 50.540000 host={host}" in t and f"put test_4 1411104988 10.210000 host={host}" in t


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["graphite", "opentsdb"])
def test_bulk_wire_matches_per_key_serializer(native_lib, torch_cuda, kind):
    """SetWireFormat: the request prepared on the GPU holds exactly the lines the per-key serializer builds
    from the map, _agg_* keys included, over several intervals."""
    from loghisto_amd.metric_system import (GraphiteProtocol, MetricSystem, OpenTSDBProtocol, ProcessedMetricSet)
    ser = GraphiteProtocol if kind == "graphite" else OpenTSDBProtocol
    ms = MetricSystem(1e-6, False)
    ms.SetWireFormat(kind, True)
    ms.RegisterGaugeFunc("some_gauge", lambda: 12.5)
    rng = np.random.default_rng(4)
    for interval in range(3):
        for i, v in enumerate(rng.lognormal(10, 1, 5000)):
            ms.Histogram(f"rpc_latency_{i % 7}", float(v))
            if interval != 1:
                ms.Histogram("only.sometimes", 0.25 * i)
        ms.Counter("requests_total", 5 + interval)
        raw = ms.collectRawMetrics()
        pm = ms.processMetrics(raw)
        ms._add_aggregates(raw, pm)
        raw.release()
        assert pm.wire_format == kind and ser(pm) is pm.wire
        per_key = ser(ProcessedMetricSet(Time=pm.Time, Metrics=pm.Metrics))
        assert sorted(per_key.split(b"\n")) == sorted(pm.wire.split(b"\n"))
        assert pm.wire.count(b"\n") >= 7 * 15 + 3
    ms.SetWireFormat(kind, False)
    ms.Histogram("rpc_latency_0", 42.0)
    raw = ms.collectRawMetrics()
    pm = ms.processMetrics(raw)
    raw.release()
    assert "rpc_latency_0_count" not in pm.Metrics and "some_gauge" in pm.Metrics
    assert (b"rpc.latency.0.count 1.000000" if kind == "graphite" else b"rpc_latency_0_count") in pm.wire
