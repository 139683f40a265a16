"""K6 timing: lh_serialize (extract + lengths + scan + write + D2H of the text) for M names x 15 keys.
usage: python tools/serialize_bench.py [--names 65536] [--reps 10]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import loghisto_amd
from loghisto_amd import _native as N

DEFAULT_PERCENTILES = {"%s_min": 0.0, "%s_50": .5, "%s_75": .75, "%s_90": .9, "%s_95": .95, "%s_99": .99,
                       "%s_99.9": .999, "%s_99.99": .9999, "%s_max": 1.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", type=int, default=65536)
    ap.add_argument("--per", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    M = a.names
    rng = np.random.default_rng(1)
    ids = np.repeat(np.arange(M, dtype=np.uint32), a.per)
    v = rng.lognormal(11.5, 1.0, ids.size) * (1.0 + 1e-3 * ids)
    with loghisto_amd.Engine(max_metrics=M, num_lanes=1, lane_samples=1 << 20) as eng:
        for i in range(M):
            eng.intern(f"svc_{i:05d}_latency")
        eng.submit_pairs(ids, v)
        with eng.flip() as snap:
            snap.accumulate()
            L = N.lib()
            labels = list(DEFAULT_PERCENTILES)
            p = np.array([DEFAULT_PERCENTILES[k] for k in labels])
            lab = (C.c_char_p * len(labels))(*[k.encode() for k in labels])
            fmt = N.LhLineFormat(b"cockroach.host-1.", b" ", b" 1411104988\n", N.FMT_UNDERSCORE_TO_DOT, 0)
            need = C.c_size_t(0)
            pp = p.ctypes.data_as(C.POINTER(C.c_double))
            t0 = time.perf_counter()
            L.lh_serialize(snap._h, 0, M, pp, lab, len(labels), C.byref(fmt), 1, None, 0, C.byref(need))
            t_size = time.perf_counter() - t0
            buf = C.create_string_buffer(need.value)
            ts = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                rc = L.lh_serialize(snap._h, 0, M, pp, lab, len(labels), C.byref(fmt), 1, buf, need.value, C.byref(need))
                ts.append(time.perf_counter() - t0)
                assert rc == 0
            ts.sort()
            lines = buf.raw[:need.value].count(b"\n")
    print(json.dumps({"names": M, "lines": lines, "bytes": need.value, "size_only_ms": round(t_size * 1e3, 3),
                      "serialize_ms_p50": round(ts[len(ts) // 2] * 1e3, 3), "serialize_ms_min": round(ts[0] * 1e3, 3),
                      "lines_per_s": round(lines / ts[len(ts) // 2])}))


if __name__ == "__main__":
    main()
