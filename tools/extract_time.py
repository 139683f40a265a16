#!/usr/bin/env python3
"""K2 alone at config 4's name count: 65 536 Zipf(1) names, one rank's 1.25e8-pair slice, then `reps` extracts of the SAME
snapshot (lh_extract_rows_view) with the kernel and the result copy timed apart (lh_tool_last_extract_ms).  One JSON line.
--lib: a tuning build (tools/build_tuning.py -D... --name X) -- ablation builds give wrong results by construction."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:
    from loghisto_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import bench  # noqa: E402
import loghisto_amd  # noqa: E402
from loghisto_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", type=int, default=65536)
    ap.add_argument("--pairs", type=float, default=1.25e8)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--np", type=int, default=9, help="percentiles per name (the reference's table has 9)")
    ap.add_argument("--lib", default=None)
    ap.add_argument("--dist", default="lognormal", help="value distribution (bench.make_samples): loguniform = 21 decades, "
                                                        "spans > 1 024 bins: k_extract_wave's wide path")
    ap.add_argument("--form", default="view", choices=["view", "compact"], help="lh_extract_rows_view / lh_extract_rows_compact")
    a = ap.parse_args()
    n, M = int(a.pairs), a.names
    torch.cuda.set_device(0)
    P = [0.0, 0.5, 0.75, 0.9, 0.95, 0.99, 0.999, 0.9999, 1.0][:a.np]
    data = bench.make_samples(n, a.dist, 7)
    w = torch.arange(1, M + 1, dtype=torch.float64, device="cuda") ** -1.0
    ids = torch.multinomial(w / w.sum(), n, replacement=True).to(torch.int32)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_pairs_device(ids, data, n)
        eng.sync()
        snap = eng.flip()
        km, cm = C.c_float(0), C.c_float(0)
        ks, cs, wall = [], [], []
        for r in range(a.reps + 3):
            t0 = time.perf_counter()
            st = (snap.extract_compact if a.form == "compact" else snap.extract_view)(P if P else [0.5], M)
            if r >= 3:
                wall.append((time.perf_counter() - t0) * 1e6)
            if N.lib().lh_tool_last_extract_ms(eng._h, C.byref(km), C.byref(cm)) == 0 and r >= 3:
                ks.append(km.value)
                cs.append(cm.value)
        total = int(st["count"].sum())
        cells = int(st["nbuckets"].sum())
        snap.release()
    ks.sort()
    wall.sort()
    print(json.dumps({"tool": "extract_time", "names": M, "pairs": n, "np": len(P), "lib": a.lib, "dist": a.dist, "form": a.form,
                      "count_ok": total == n, "occupied_cells": cells,
                      "kernel_us_avg": 1e3 * sum(ks) / len(ks), "kernel_us_min": 1e3 * ks[0], "kernel_us_median": 1e3 * ks[len(ks) // 2],
                      "copy_us_avg": 1e3 * sum(cs) / len(cs),
                      "call_wall_us_median": wall[len(wall) // 2], "call_wall_us_p90": wall[int(len(wall) * 0.9)]}), flush=True)


if __name__ == "__main__":
    main()
