"""BASELINE config 4 from ONE rank's point of view (gpurun exposes a single GPU): 65 536 names, this rank
ingests its 1/8 slice of a Zipf(1.0) stream over ALL names, and at the flip takes part in the reduce-scatter
merge, extracts and serialises the 8 192 names it owns.

Measured here (device work of one rank, HIP events): slice ingest, the merge's local steps exactly as
loghisto_amd/merge.py performs them around the collective (pack the occupied window into the send buffer,
add the 7 peer blocks that the reduce-scatter delivers -- emulated with a second buffer so that the
arithmetic runs at HBM speed --, scatter the result back into the snapshot), extract and K6 serialize of the
owned rows.  NOT measured: the xGMI transfer itself; its size is reported (`merge_bytes_*`) together with the
time it would take at the per-link rate of MI355X_MICROARCH.md, labelled "projected".

usage: python tools/c4_sim.py [--names 65536] [--world 8] [--stream 1e9]
"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import loghisto_amd
from loghisto_amd import merge

PCT = {"%s_min": 0.0, "%s_50": .5, "%s_75": .75, "%s_90": .9, "%s_95": .95, "%s_99": .99, "%s_99.9": .999,
       "%s_99.99": .9999, "%s_max": 1.0}
XGMI_LINK_GBPS = 153.0   # per direction per link, MI355X_MICROARCH.md


def timed(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", type=int, default=65536)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--stream", type=float, default=1e9, help="samples of the whole stream per interval")
    a = ap.parse_args()
    M, W = a.names, a.world
    n = int(a.stream) // W
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    # Zipf(1.0) ranks by inverse-CDF on a harmonic table; value ~ lognormal(ln 1e5 + 0.002*id/64, 1)
    w = 1.0 / torch.arange(1, M + 1, device=dev, dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0)
    ids = torch.searchsorted(cdf, torch.rand(n, device=dev, dtype=torch.float64, generator=g)).clamp_(max=M - 1)
    ids = ids.to(torch.int32)
    v = torch.exp(torch.randn(n, device=dev, dtype=torch.float64, generator=g) + math.log(1e5) + 3e-5 * ids)
    torch.cuda.synchronize()

    out = {"config": "C4, one rank of %d" % W, "names": M, "slice_samples": n}
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        for i in range(M):
            eng.intern(f"h{i:05d}")
        eng.submit_pairs_device(ids, v, stream=st)          # warm-up interval (scratch allocation, clocks)
        eng.sync()
        eng.flip().release()
        ms, _ = timed(lambda: eng.submit_pairs_device(ids, v, stream=st))
        out["ingest_ms"] = round(ms, 3)
        out["ingest_Gsamples_per_s"] = round(n / ms / 1e6, 1)
        eng.sync()
        snap = eng.flip()
        rows, ranges = merge.snapshot_tensors(snap, M)
        xs = torch.cuda.ExternalStream(snap.stream())
        st.wait_stream(xs)
        lo = int(ranges[:, 0].min().item())
        hi = int(ranges[:, 1].max().item())
        width = hi - lo + 1
        per = (M + W - 1) // W
        first, last = merge.name_blocks(M, 0, W)
        matrix = M * width * 8
        out.update(window_bins=width, merge_matrix_bytes=matrix,
                   merge_bytes_sent_per_rank=matrix * (W - 1) // W, merge_bytes_per_peer_block=per * width * 8)
        send = torch.zeros((W * per, width), dtype=torch.int64, device=dev)
        peers = torch.randint(0, 3, ((W - 1), per, width), dtype=torch.int64, device=dev, generator=g)
        recv = torch.empty((per, width), dtype=torch.int64, device=dev)

        def pack():
            send[:M] = rows[:, lo:hi + 1]

        def reduce_local():
            torch.add(send[first:first + per], peers[0], out=recv)
            for q in range(1, W - 1):
                recv.add_(peers[q])

        def unpack():
            rows[first:last, lo:hi + 1] = recv[:last - first]

        out["pack_ms"] = round(timed(pack)[0], 3)
        out["reduce_adds_ms"] = round(timed(reduce_local)[0], 3)
        out["unpack_ms"] = round(timed(unpack)[0], 3)
        xs.wait_stream(st)
        snap.mark_dirty(first, last - first, lo, hi)
        import time
        wire = dict(prefix="cockroach.host.", sep=" ", suffix=" 1411104988\n", underscore_to_dot=True, aggregates=True,
                    first=first, nmetrics=last - first)
        snap.accumulate()
        snap.extract(list(PCT.values()), last - first, first=first)   # first call sizes the pinned result buffers
        snap.serialize(PCT, **wire)
        t0 = time.perf_counter()
        got = snap.extract(list(PCT.values()), last - first, first=first)
        out["extract_owned_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        t0 = time.perf_counter()
        text = snap.serialize(PCT, **wire)
        out["serialize_owned_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out["owned_names_with_samples"] = int((got["count"] > 0).sum())
        out["wire_bytes"] = len(text)
        snap.release()
    # direct exchange on a fully connected hive: every peer block crosses its own link
    out["xgmi_direct_ms_projected"] = round(out["merge_bytes_per_peer_block"] / (XGMI_LINK_GBPS * 1e9) * 1e3, 3)
    out["xgmi_ring_ms_projected"] = round((W - 1) * out["merge_bytes_per_peer_block"] / (XGMI_LINK_GBPS * 1e9) * 1e3, 3)
    out["note"] = ("device work of one rank measured on one MI355X; the collective's wire time is projected from "
                   "the link rate, not measured")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
