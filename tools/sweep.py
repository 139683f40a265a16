#!/usr/bin/env python3
"""Contention sweep of the ingest kernel: kernel-only time per
distribution at n samples, HIP events on the launch stream.  Prints one JSON line
per distribution.  GPU only."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:  # a -DLH_TUNING build of the library (ablation bits): tools only, never the product
    from loghisto_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    _native.ALLOW_OLDER_ABI = True  # (an A/B against the library of an earlier round)
import bench  # noqa: E402
import loghisto_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=float, default=1e8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=0, help="if >0: mixed stream over this many names, Zipf(1.0) ids")
    ap.add_argument("--dists", default="lognormal,constant,uniform,exponential,normal,loguniform,loguniform21,lognormal25,kvalues2,kvalues4,"
                                       "kvalues8,kvalues16,bimodal,far_1e30,negative_far,signed_wide,thin_far_tail")  # (+ lognormal50: sigma 5, 4 096-bin windows)
    ap.add_argument("--ids", default="zipf", choices=["zipf", "uniform", "zipf0.5", "sorted", "drift"],
                    help="name distribution of --pairs (sorted: Zipf(1.0) counts, the stream ordered by name; drift: the ranking of the names is reversed half way through the launch)")
    ap.add_argument("--lib", default=None, help="path of an alternative liblhgpu.so (a -DLH_TUNING build)")
    ap.add_argument("--opt", action="append", default=[], help="lh_set_option as ID=VALUE (repeatable), e.g. 9=0 turns "
                                                               "the survey + 2-byte-record path off")
    ap.add_argument("--nocheck", action="store_true", help="timing of an ablation build (tools/build_tuning.py -D...): counts are wrong")
    ap.add_argument("--survey-every", type=int, default=32, help="LH_OPT_SURVEY_EVERY; set again before every distribution, which "
                                                                 "ends the reuse of the previous distribution's survey")
    ap.add_argument("--warmup", type=int, default=0,
                    help="untimed single-metric launches of 1e9 lognormal samples IN THIS PROCESS before the first distribution "
                         "(a fresh process starts on idle clocks: its first line used to read 10 %% slow)")
    ap.add_argument("--keep-survey", action="store_true",
                    help="do NOT end the survey's reuse between distributions: every distribution after the first starts on a "
                         "STALE survey (round 5 measured its few-valued streams that way without knowing: profiles/r05_fewvalued.txt)")
    a = ap.parse_args()
    n = int(a.samples)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng = loghisto_amd.Engine(max_metrics=max(1, a.pairs), num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    for kv in a.opt:
        k, val = kv.split("=")
        eng.set_option(int(k), int(val))
    if a.warmup:
        wdata = bench.make_samples(int(1e9), "lognormal", 3)
        for r in range(a.warmup):
            eng.submit_device(0, wdata, int(1e9), stream=stream)
            if r % 8 == 7:
                torch.cuda.synchronize()
                eng.flip().release()
        torch.cuda.synchronize()
        eng.flip().release()
        del wdata
    for kind in a.dists.split(","):
        if a.pairs and not a.keep_survey:
            eng.set_option(16, a.survey_every)  # LH_OPT_SURVEY_EVERY: the tables of the previous distribution are not reused
        data = bench.make_samples(n, kind, 7)
        ids = None
        if a.pairs:
            w = torch.arange(1, a.pairs + 1, dtype=torch.float64, device="cuda") ** {"zipf": -1.0, "uniform": 0.0,
                                                                                      "zipf0.5": -0.5, "sorted": -1.0, "drift": -1.0}[a.ids]
            ids = torch.multinomial(w / w.sum(), n, replacement=True).to(torch.int32)
            if a.ids == "sorted":
                ids = torch.sort(ids).values.contiguous()
            if a.ids == "drift":
                ids[n // 2:] = (a.pairs - 1) - ids[n // 2:]
        ms = []
        for r in range(a.reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if ids is None:
                eng.submit_device(0, data, n, stream=stream)
            else:
                eng.submit_pairs_device(ids, data, n, stream=stream)
            e1.record(stream)
            torch.cuda.synchronize()
            if r >= 2:
                ms.append(e0.elapsed_time(e1))
            snap = eng.flip()
            st = snap.extract([0.5], max(1, a.pairs))
            snap.release()
            assert int(st["count"].sum()) == n or a.nocheck  # (ablation builds break counts)
        avg = sum(ms) / len(ms)
        bps = 12 if a.pairs else 8
        print(json.dumps({"dist": kind, "names": a.pairs or 1, "ids": a.ids if a.pairs else None, "n": n, "avg_ms": avg, "min_ms": min(ms),
                          "Gsamples_per_s": n / avg / 1e6, "GBps": n * bps / avg / 1e6,
                          "frac_hbm_peak": n * bps / avg / 1e6 / 8000.0,
                          "occupied_buckets": int(st["nbuckets"].sum()), "opts": a.opt,
                          "v2_samples": eng.counters()["samples_partitioned_v2"],
                          "region_overflows": eng.counters()["region_overflows"],
                          "regions_disabled": eng.counters()["regions_disabled"],
                          "surveys_reused": eng.counters()["surveys_reused"],
                          "survey_stale_pairs": eng.counters()["survey_stale_pairs"],
                          "v3": {k: eng.counters()[k] for k in ("samples_partitioned_v3", "window_log2", "records_level1",
                                                                "records_level2", "level2_overflows",
                                                                "reduce_window_misses")}}), flush=True)
        del data
    eng.close()


if __name__ == "__main__":
    main()
