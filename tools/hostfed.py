#!/usr/bin/env python3
"""Host-fed (PCIe-inclusive) ingest rate: T producer threads call lh_submit on host
arrays; the library memcpy's into pinned lanes, ships with hipMemcpyAsync and buckets
on the GPU.  This is the path the cgo binding uses; its roofline is PCIe Gen5 x16
(63 GB/s = 7.9 G float64 samples/s), never the HBM one.  Prints one JSON line per
thread count.  GPU only."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loghisto_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None, help="path of another build of liblhgpu.so (A/B runs)")
    ap.add_argument("--samples", type=float, default=4e8)
    ap.add_argument("--threads", default="1,4,16,64")
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--lane-samples", type=int, default=1 << 21)
    ap.add_argument("--pairs", type=int, default=0, help="if >0: lh_submit_pairs over this many Zipf(1.0) names (12 B/sample)")
    a = ap.parse_args()
    if a.lib:
        from loghisto_amd import _native
        _native.LIB_PATH = os.path.abspath(a.lib)
    n = int(a.samples)
    rng = np.random.default_rng(1)
    src = rng.lognormal(np.log(1e5), 1.0, 1 << 24)  # 128 MiB of host samples, reused
    M = max(1, a.pairs)
    ids = None
    if a.pairs:
        w = 1.0 / np.arange(1, M + 1)
        ids = rng.choice(M, size=src.size, p=w / w.sum()).astype(np.uint32)
    bps = 12 if a.pairs else 8
    for T in [int(x) for x in a.threads.split(",")]:
        eng = loghisto_amd.Engine(max_metrics=max(4, M), num_buffers=2, num_lanes=max(T, 1), lane_samples=a.lane_samples)
        per = n // T

        def work(t):
            done = 0
            off = (t * 7919 * a.batch) % (src.size - a.batch)
            while done < per:
                k = min(a.batch, per - done)
                if ids is None:
                    eng.submit(0, src[off:off + k])
                else:
                    eng.submit_pairs(ids[off:off + k], src[off:off + k])
                done += k
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        eng.sync()
        dt = time.perf_counter() - t0
        with eng.flip() as snap:
            cnt = int(snap.extract([0.5], M)["count"].sum())
        assert cnt == per * T, (cnt, per * T)
        print(json.dumps({"threads": T, "samples": per * T, "seconds": dt, "Gsamples_per_s": per * T / dt / 1e9,
                          "names": M, "bytes_per_sample": bps,
                          "GBps_over_pcie": per * T * bps / dt / 1e9, "frac_pcie_63GBps": per * T * bps / dt / 63e9,
                          "batch": a.batch, "lane_samples": a.lane_samples}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
