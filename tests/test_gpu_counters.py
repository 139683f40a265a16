"""Counters on the device: lh_intern_counter / lh_submit_counts /
lh_counters_collect / lh_serialize_counters against the reference's semantics and its own test values.

  (*MetricSystem).Counter      /root/reference/metrics.go:251-269   counterCache[name] += amount
  collectRawMetrics            /root/reference/metrics.go:425-458   Rates = the interval's amounts of the names touched,
                                                                    Counters = lifetime totals of every name ever touched
  processMetrics               /root/reference/metrics.go:487-493   "<name>" = float64(total), "<name>_rate" = float64(rate)
  TestRate / TestCounter / TestRawBroadcast   /root/reference/metrics_test.go:202-240, 321-346
"""
import threading

import numpy as np
import pytest

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu


def _collect(e):
    with e.flip() as snap:
        c = snap.counter_values()
    names = [e.counter_name(i) for i in range(e.num_counters())]
    rates = {n: int(c["rate"][i]) for i, n in enumerate(names) if c["present"][i]}       # RawMetricSet.Rates
    totals = {n: int(c["total"][i]) for i, n in enumerate(names) if c["known"][i]}       # RawMetricSet.Counters
    return rates, totals


def test_reference_values_TestRate_TestCounter_TestRawBroadcast(native_lib, torch_cuda):
    import loghisto_amd
    with loghisto_amd.Engine(max_metrics=4, max_counters=16, num_lanes=1, lane_samples=1 << 12) as e:
        r1, c1, c2 = e.intern_counter("rate1"), e.intern_counter("counter1"), e.intern_counter("counter2")
        assert e.intern_counter("rate1") == r1 and (r1, c1, c2) == (0, 1, 2)
        # TestRate (metrics_test.go:202-223): rates reset every interval
        e.submit_counts([r1], [777])
        rates, totals = _collect(e)
        assert rates == {"rate1": 777} and totals == {"rate1": 777}
        e.submit_counts([r1], [1223])
        assert _collect(e)[0] == {"rate1": 1223}
        e.submit_counts([r1, r1], [1223, 1223])
        rates, totals = _collect(e)
        assert rates == {"rate1": 2446} and totals["rate1"] == 777 + 1223 + 2446
        # TestCounter (metrics_test.go:225-240): the lifetime counter accumulates across collections
        e.submit_counts([c1], [3290])
        assert _collect(e)[1]["counter1"] == 3290
        e.submit_counts([c1], [10000])
        rates, totals = _collect(e)
        assert totals["counter1"] == 13290 and rates == {"counter1": 10000}
        # TestRawBroadcast (metrics_test.go:321-346): Counters["counter2"] == Rates["counter2"] == 121
        e.submit_counts([c2], [10])
        e.submit_counts([c2], [111])
        rates, totals = _collect(e)
        assert rates == {"counter2": 121} and totals["counter2"] == 121
        # an interval without events: no rates, every known counter still exported with its total
        rates, totals = _collect(e)
        assert rates == {} and totals == {"rate1": 4446, "counter1": 13290, "counter2": 121}
        # Counter(name, 0) creates the entry: a rate of 0 is exported (metrics.go:263-267)
        z = e.intern_counter("zero")
        e.submit_counts([z], [0])
        rates, totals = _collect(e)
        assert rates == {"zero": 0} and totals["zero"] == 0


@pytest.mark.parametrize("ncounters", [256, 5000])      # LDS-aggregated kernel / one global atomic per event
def test_large_batches_host_threads_and_device_path(native_lib, torch_cuda, ncounters):
    import loghisto_amd
    torch = torch_cuda
    rng = np.random.default_rng(ncounters)
    n = 3_000_000
    w = 1.0 / np.arange(1, ncounters + 1)
    ids = rng.choice(ncounters, size=n, p=w / w.sum()).astype(np.uint32)
    amt = rng.integers(0, 1 << 40, n).astype(np.uint64)
    want = np.zeros(ncounters, dtype=np.uint64)
    np.add.at(want, ids, amt)
    touched = np.bincount(ids, minlength=ncounters) > 0
    with loghisto_amd.Engine(max_metrics=4, max_counters=ncounters, num_lanes=4, lane_samples=1 << 16) as e:
        for i in range(ncounters):
            e.intern_counter(f"c{i}")
        # half through 4 host threads (pinned lanes), half from device memory
        h = n // 2
        cuts = np.linspace(0, h, 5).astype(int)
        th = [threading.Thread(target=lambda a, b: e.submit_counts(ids[a:b], amt[a:b]), args=(cuts[k], cuts[k + 1]))
              for k in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        d_ids = torch.from_numpy(ids[h:].view(np.int32)).cuda()
        d_amt = torch.from_numpy(amt[h:].view(np.int64)).cuda()
        torch.cuda.synchronize()
        e.submit_counts_device(d_ids, d_amt)
        e.submit(0, np.array([1.0, 2.0]))                   # histograms and counters share the epoch
        e.sync()
        assert e.counters()["counter_events"] == n
        with e.flip() as snap:
            c = snap.counter_values()
            assert int(snap.extract([0.5], 1)["count"][0]) == 2
            assert np.array_equal(c["rate"], want) and np.array_equal(c["total"], want)
            assert np.array_equal(c["present"], touched) and np.array_equal(c["known"], touched)
            again = snap.counter_values()                   # the fold into the lifetime store happens once
            assert np.array_equal(again["total"], want)
        # second interval: lifetime keeps accumulating (uint64 wrap-around add), rates start from zero
        e.submit_counts(ids[:1000], amt[:1000])
        with e.flip() as snap:
            c = snap.counter_values()
        w2 = np.zeros(ncounters, dtype=np.uint64)
        np.add.at(w2, ids[:1000], amt[:1000])
        assert np.array_equal(c["rate"], w2) and np.array_equal(c["total"], want + w2)
        assert np.array_equal(c["known"], touched)


def test_bad_counter_ids(native_lib, torch_cuda):
    import loghisto_amd
    torch = torch_cuda
    with loghisto_amd.Engine(max_metrics=2, max_counters=8, num_lanes=1, lane_samples=1 << 12) as e:
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.submit_counts([8], [1])                       # host path: rejected before it is staged
        assert ei.value.code == N.ERANGE
        ids = torch.tensor([1, 8, 2, -1], dtype=torch.int32).cuda()
        amt = torch.tensor([5, 6, 7, 8], dtype=torch.int64).cuda()
        torch.cuda.synchronize()
        e.submit_counts_device(ids, amt)
        with pytest.raises(loghisto_amd.LhError):
            e.sync()                                        # device path: reported, the bad events are skipped
        with e.flip() as snap:
            c = snap.counter_values(8)
        assert c["rate"].tolist() == [0, 5, 7, 0, 0, 0, 0, 0]
    with loghisto_amd.Engine(max_metrics=2, max_counters=0, num_lanes=1, lane_samples=1 << 12) as e:
        with pytest.raises(loghisto_amd.LhError):
            e.intern_counter("none")                        # an engine without counters


def test_counter_wire_lines(native_lib, torch_cuda):
    """lh_serialize_counters: "<name>" for every known counter, "<name>_rate" for those touched this interval,
    values as Go's %f of float64(uint64) -- including counts beyond 2^53, which float64 rounds."""
    import loghisto_amd
    with loghisto_amd.Engine(max_metrics=2, max_counters=64, num_lanes=1, lane_samples=1 << 12) as e:
        names = ["requests_total", "bytes_in", "errors", "huge", "never"]
        ids = [e.intern_counter(n) for n in names]
        e.submit_counts(ids[:4], [3290, 1 << 33, 0, (1 << 63) + 12345])
        with e.flip() as snap:
            snap.counter_values()
        e.submit_counts([ids[0], ids[3]], [10000, 1])
        with e.flip() as snap:
            c = snap.counter_values()
            text = snap.serialize_counters("cockroach.host.", " ", " 1411104988\n", underscore_to_dot=True).decode()
            tsdb = snap.serialize_counters("put ", " 1411104988 ", " host=h\n").decode()
        want = []
        for i, n in enumerate(names):
            if c["known"][i]:
                want.append(f"cockroach.host.{n.replace('_', '.')} {oracle.format_f(float(int(c['total'][i])))} 1411104988\n")
            if c["present"][i]:
                want.append(f"cockroach.host.{n.replace('_', '.')}.rate {oracle.format_f(float(int(c['rate'][i])))} 1411104988\n")
        assert text == "".join(want)
        assert int(c["total"][0]) == 13290 and "cockroach.host.requests.total 13290.000000 1411104988\n" in text
        assert "put errors 1411104988 0.000000 host=h\n" in tsdb and "put errors_rate" not in tsdb
        assert "never" not in text
        assert f"put huge 1411104988 {oracle.format_f(float((1 << 63) + 12346))} host=h\n" in tsdb


def test_release_without_reading_counters_still_folds(native_lib, torch_cuda):
    """metrics.go:435-458 folds the interval's amounts into the lifetime totals at the epoch boundary whoever reads them:
    a snapshot released before lh_counters_collect / lh_serialize_counters (a consumer that skips an interval, an error
    path) must not lose that interval's amounts from the totals."""
    import loghisto_amd
    with loghisto_amd.Engine(max_metrics=4, max_counters=16, num_lanes=1, lane_samples=1 << 12) as e:
        c = e.intern_counter("skipped")
        e.submit_counts([c, c], [5, 7])
        e.flip().release()                       # nobody looks at this interval
        e.submit_counts([c], [100])
        rates, totals = _collect(e)
        assert rates == {"skipped": 100} and totals == {"skipped": 112}
        e.flip().release()
        rates, totals = _collect(e)
        assert rates == {} and totals == {"skipped": 112}
