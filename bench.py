#!/usr/bin/env python3
"""bench.py -- loghisto hot path on MI355X: float64 samples/s bucketed.

A "step" is one pass of the hot path over one resident batch: ingest, epoch flip, [N>1: merge], percentile /
sum / count scan, results on the host, buffer recycled.  Consecutive steps are software-pipelined as the reference
itself is (its reaper reduces interval i while producers fill interval i + 1): ingest of step i + 1 is enqueued right
after the flip of step i, before the host waits for step i's results (pipelined_steps).

N = 1 (the headline): BASELINE.json configs[1] "Single-metric 1B float64 samples, 1xMI355X, one ingest kernel
+ one percentile scan" (SURVEY.md 8d "C2": lognormal(mu = ln 1e5, sigma = 1), seeded, generated on device).
The same JSON line carries, under "secondary", the other configurations measured in the same process --
C3 (1e9 pairs over 1 024 Zipf names), one rank's slice of C4 (65 536 names, reduce-scatter merge through the
C ABI), the host-fed lh_submit_pairs rate and a short C5 burst -- each with its own roofline and parity flag
(VERDICT r1 next #4), and "parity": the GPU row against the oracle over ALL 1e9 samples of the step.

N > 1: one process per GPU (torch.distributed launches and rendezvous), WEAK SCALING OF THE SAME HEADLINE: every
rank buckets its own 1e9-sample slice of the ONE metric (K1), and at the flip lh_snapshot_merge all-reduces the
one row's merged window over RCCL/xGMI (LH_MERGE_ALLREDUCE, through the C ABI), then every rank extracts.  The
only collective is that merge.  value = N * n * K / max-over-ranks wall time; N = 1 of this code path IS the
N = 1 line (no merge).  BASELINE configs[3] ("65536 histogram names sharded across the GPUs, RCCL merge of bucket
arrays": every rank buckets ITS slice of a Zipf stream over ALL 65 536 names, lh_snapshot_merge reduce-scatters
the per-row merged windows so that rank r ends with the rows of the names it owns, which it extracts) runs on the
same ranks afterwards and is reported under secondary.c4 with its own one-rank reference.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PCTS = [0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0]   # metrics.go:145-155
HBM_PEAK_GBS = 8000.0                                    # MI355X_MICROARCH.md: 8.0 TB/s spec
PCIE_GBS = 63.0                                          # PCIe Gen5 x16 (spec)
BYTES_SINGLE, BYTES_PAIR, BYTES_PAIR16 = 8, 12, 10        # float64 / float64 + uint32 id / + uint16 id
METRIC = "float64 samples/sec bucketed (1 GPU) + % HBM roofline; p99 extract latency"
K1_PMC = os.path.join("profiles", "r06_k1_pmc.json")
C3_PMC = os.path.join("profiles", "r06_c3_pmc.json")
C4_PMC = os.path.join("profiles", "r06_c4_pmc.json")
C4_1E9_PMC = os.path.join("profiles", "r06_c4_names_1e9_pmc.json")
PREWARM = 25                                             # untimed K1 launches before the warm-up steps (run_c2)


# Kernel sources each committed PMC summary was measured on.  The summary records their hashes ("sources"); a summary
# whose hashes differ from the tree's is STALE -- the kernels changed after it was taken -- and its bytes are not
# reported as `traffic` (the constant must not go stale silently).
PMC_SOURCES = {
    "k1": ["lh_kernels.hip", "lh_codec.h", "lh_kernels.h"],
    "c3": ["lh_kernels_part.hip", "lh_kernels_part2.h", "lh_codec.h", "lh_windows.h", "lh_kernels.h"],
    "c4": ["lh_kernels_part.hip", "lh_kernels_part2.h", "lh_kernels_part3.h", "lh_codec.h", "lh_windows.h", "lh_kernels.h"],
}


def source_hashes(kind):
    import hashlib
    out = {}
    for f in PMC_SOURCES[kind]:
        with open(os.path.join(ROOT, "loghisto_amd", "csrc", f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


# Every measurement that claims to describe THIS tree carries one short stamp of the sources a result can depend on: the
# bench line has it (`tree_stamp`), tools/profile_round.sh writes it into every file it produces, and
# tests/test_profiles_fresh.py holds the committed profiles/r06_* evidence to it (a sweep committed
# before two later kernel changes was still quoted as "final").
STAMP_SOURCES = ["lh_kernels.hip", "lh_kernels_part.hip", "lh_kernels_part2.h", "lh_kernels_part3.h", "lh_kernels_small.hip",
                 "lh_kernels_fmt.hip", "lh_codec.h", "lh_windows.h", "lh_ids.h", "lh_kernels.h", "lh_dispatch.cc",
                 "lh_dispatch.h", "lh_engine.cc"]


def tree_stamp():
    import hashlib
    h = hashlib.sha256()
    for f in STAMP_SOURCES:
        with open(os.path.join(ROOT, "loghisto_amd", "csrc", f), "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    return h.hexdigest()[:16]


def pmc_stale(j, kind):
    """None when the summary j was measured on the kernel sources of this tree, else what differs."""
    rec = j.get("sources")
    if not rec:
        return "the summary records no source hashes"
    cur = source_hashes(kind)
    diff = sorted(f for f in cur if rec.get(f) != cur[f])
    return ("kernel sources changed since the summary was taken: " + ", ".join(diff)) if diff else None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=float, default=1e9, help="float64 samples per GPU per step (c2 / c3)")
    ap.add_argument("--dist", default="lognormal", choices=["lognormal", "constant", "uniform", "exponential",
                                                            "normal", "loguniform", "loguniform21", "lognormal25", "kvalues2",
                                                            "kvalues4", "kvalues8", "kvalues16", "bimodal"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: only the C2 headline")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-stream oracle checks (profiling runs)")
    ap.add_argument("--latency-flips", type=int, default=1000)
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "c4"],
                    help="auto = c2: the single-metric headline (+ secondary legs; on several GPUs weak scaling of it with "
                         "an all-reduce of the row, config 4 under secondary.c4); c3: 1024 Zipf names; c4: 65536 names + "
                         "reduce-scatter merge as the top-level line")
    ap.add_argument("--names", type=int, default=0, help="histogram names (default 1024 for c3, 65536 for c4)")
    ap.add_argument("--c4-slice", type=float, default=1.25e8, help="pairs per rank per step of the C4 stream")
    return ap.parse_args()


def make_samples(n: int, kind: str, seed: int) -> torch.Tensor:
    """Seeded synthetic stream, generated on device."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if kind == "constant":
        return torch.full((n,), 123.0, dtype=torch.float64, device="cuda")
    if kind == "kvalues3_skewed":
        # one dominant value: 90 % / 9 % / 1 % (a wave holds ~58 lanes of one bucket)
        u = torch.rand(n, device="cuda", generator=g)
        j = (u > 0.9).to(torch.int32) + (u > 0.99).to(torch.int32)
        return torch.pow(torch.tensor(1.5, dtype=torch.float64, device="cuda"), j).mul_(1e3)
    if kind.startswith("kvalues"):
        # few-valued stream (quantised timers, status codes, queue depths: what TimerToken.Stop produces): k distinct
        # values 1000 * 1.5^j, uniformly drawn
        j = torch.randint(0, int(kind[7:]), (n,), device="cuda", generator=g, dtype=torch.int32)
        return torch.pow(torch.tensor(1.5, dtype=torch.float64, device="cuda"), j).mul_(1e3)
    if kind == "bimodal":
        # two lognormal lobes 10x apart, 90 / 10
        v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g).add_(math.log(1e5)).exp_()
        step = 1 << 27
        for lo in range(0, n, step):
            u = torch.rand(min(step, n - lo), device="cuda", generator=g)
            v[lo:lo + step].mul_(torch.where(u < 0.1, 10.0, 1.0).to(torch.float64))
        return v
    if kind == "on_thresholds":
        # every sample within rounding of a bucket boundary (100 * log(1 + v) = k - 0.5, 500 boundaries around 1e5): the
        # bucket index takes its exact route (threshold table) for all of them -- the worst case of lh_bin_of
        k = torch.randint(900, 1400, (n,), device="cuda", generator=g, dtype=torch.int32).to(torch.float64)
        return k.sub_(0.5).div_(100.0).expm1_()
    if kind in ("far_1e30", "negative_far", "signed_wide", "thin_far_tail"):
        # streams that leave K1's main LDS window (|v| < 6.1e17): all of it far above / far below, a little beyond on
        # both sides, a 0.1 % tail up to 1e60 (tools/sweep.py; tests/test_gpu_parity.py holds the numpy twins)
        if kind == "far_1e30":
            return torch.randn(n, dtype=torch.float64, device="cuda", generator=g).add_(math.log(1e30)).exp_()
        if kind == "negative_far":
            return torch.randn(n, dtype=torch.float64, device="cuda", generator=g).mul_(0.5).add_(math.log(1e25)).exp_().neg_()
        v = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        if kind == "signed_wide":
            v = v.mul_(23.0).sub_(3.0).mul_(math.log(10.0)).exp_()
            step = 1 << 27
            for lo in range(0, n, step):
                u = torch.rand(min(step, n - lo), device="cuda", generator=g)
                v[lo:lo + step].mul_(torch.where(u < 0.5, -1.0, 1.0).to(torch.float64))
            return v
        far = v.mul_(41.0).add_(19.0).mul_(math.log(10.0)).exp_()
        step = 1 << 27
        for lo in range(0, n, step):
            m = min(step, n - lo)
            u = torch.rand(m, device="cuda", generator=g)
            body = torch.randn(m, dtype=torch.float64, device="cuda", generator=g).add_(math.log(1e5)).exp_()
            far[lo:lo + step] = torch.where(u < 1e-3, far[lo:lo + step], body)
        return far
    if kind in ("uniform", "loguniform", "loguniform21", "exponential"):
        v = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        if kind == "uniform":
            return v.mul_(1e9)
        if kind == "loguniform":
            return v.mul_(21.0).sub_(3.0).mul_(math.log(10.0)).exp_()
        if kind == "loguniform21":
            # a stream only slightly wider than K1's main window (|v| < 6.1e17): 10^U(-3, 21), 14 % of it up to 740 bins
            # above the window -- the bins a floating window must sit ADJACENT to the main window for
            return v.mul_(24.0).sub_(3.0).mul_(math.log(10.0)).exp_()
        return v.neg_().log1p_().neg_().mul_(1e6)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    if kind == "normal":
        return v.mul_(1e3)
    sigma = 2.5 if kind == "lognormal25" else 5.0 if kind == "lognormal50" else 1.0
    return v.mul_(sigma).add_(math.log(1e5)).exp_()


class OwnBuffer:
    """A bench input stream in plain hipMalloc'ed memory (lh_tool_device_alloc, include/loghisto_gpu_tuning.h) instead of
    torch's caching allocator: VERDICT r4 weak #7 -- one box of twelve ran every kernel that read the bench's
    torch-allocated inputs 20 - 60 % slow while kernels reading hipMalloc'ed memory ran normally.  `tensor` is a torch
    view of the block (for the launch wrappers and the parity legs); free() gives it back."""

    def __init__(self, src: torch.Tensor):
        import ctypes as C
        from loghisto_amd import _native as N
        from loghisto_amd.merge import _DeviceArray
        self._N, self.nbytes = N, src.numel() * src.element_size()
        p = C.c_void_p(0)
        N.check(N.lib().lh_tool_device_alloc(self.nbytes, C.byref(p)), "lh_tool_device_alloc")
        self.ptr = int(p.value)
        typestr = {torch.float64: "<f8", torch.int32: "<i4", torch.int16: "<i2", torch.int64: "<i8"}[src.dtype]
        self.tensor = torch.as_tensor(_DeviceArray(self.ptr, (src.numel(),), typestr), device=src.device)
        self.tensor.copy_(src)
        torch.cuda.synchronize()

    def read_ceiling_gbs(self, reps=10, stream=None):
        """GB/s a kernel that ONLY reads this block gets on this box, in this process (lh_tool_read_ceiling: K1's loop
        without the bucket work): (average, best launch)."""
        import ctypes as C
        avg, mn = C.c_float(0), C.c_float(0)
        h = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._N.check(self._N.lib().lh_tool_read_ceiling(C.c_void_p(self.ptr), self.nbytes, reps, h, C.byref(avg), C.byref(mn)),
                      "lh_tool_read_ceiling")
        return self.nbytes / (avg.value * 1e-3) / 1e9, self.nbytes / (mn.value * 1e-3) / 1e9

    def free(self):
        if self.ptr:
            import ctypes as C
            self.tensor = None
            self._N.check(self._N.lib().lh_tool_device_free(C.c_void_p(self.ptr)), "lh_tool_device_free")
            self.ptr = 0


def own(t: torch.Tensor) -> OwnBuffer:
    """the tensor's contents in hipMalloc'ed memory; torch's copy is dropped"""
    b = OwnBuffer(t)
    del t
    torch.cuda.empty_cache()
    return b


def zipf_ids(n: int, M: int, seed: int) -> torch.Tensor:
    """id ~ Zipf(1.0) over ranks by inverse CDF (chunked: searchsorted needs int64 scratch)."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    w = 1.0 / torch.arange(1, M + 1, dtype=torch.float64, device="cuda")
    cdf = torch.cumsum(w / w.sum(), 0)
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    step = 1 << 27
    for lo in range(0, n, step):
        u = torch.rand(min(step, n - lo), device="cuda", dtype=torch.float64, generator=g)
        ids[lo:lo + step] = torch.searchsorted(cdf, u).clamp_(max=M - 1).to(torch.int32)
    return ids


def effective_cores() -> int:
    import oracle
    return oracle.granted_cores()


def roofline(n_bytes: float, avg_ms: float, kernel: str, peak=HBM_PEAK_GBS, bound="hbm", traffic=None, **extra):
    achieved = n_bytes / (avg_ms * 1e-3) / 1e9
    r = {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "GB/s",
         "frac": achieved / peak, "traffic": traffic, "algorithmic_bytes_per_launch": n_bytes,
         "avg_launch_ms": avg_ms}
    r.update(extra)
    return r


def cpu_baseline(host: np.ndarray, target_s: float):
    """Oracle timed on this box's host cores (bounded sample of the same workload)."""
    import oracle
    cores = effective_cores()
    probe = host[: 1 << 21]
    t1, _ = oracle.bench_dense(probe, 1)
    rate1 = probe.size / max(t1, 1e-9)                       # one core
    n = int(min(host.size, 1 << 27))                         # 1 GiB of host samples
    part = host[:n]
    # every thread passes `reps` times over its slice so that thread start-up and the final merge do not
    # dominate on a many-core host: about target_s seconds of wall time if the cores scaled perfectly / 4
    reps = int(max(1, min(256, target_s * rate1 * cores / 2 / n)))
    t, counts = oracle.bench_dense_reps(part, cores, reps)
    assert int(counts.sum()) == n * reps and not (counts % reps).any()
    n_timed = n * reps
    # form A (the reference's cost shape: lock + 4 map probes + atomic per sample), small sample
    na = int(min(n, 1 << 22))
    ta, _ = oracle.bench_faithful(part[:na], cores)
    ta1, _ = oracle.bench_faithful(part[: na // 4], 1)
    return {
        "value": n_timed / t, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"first {n} samples of the step's stream, {reps} passes per thread ({t:.2f} s wall); C oracle "
                  f"(Go math.Log restated), per-thread dense uint64[65536] rows + merge (BASELINE.md form B), "
                  f"{cores} threads = the cores this process is granted (host reports {os.cpu_count()} CPUs); "
                  f"one thread alone: {rate1:.3g} samples/s",
        "faithful_form": {"value": na / ta, "cores": cores, "value_1thread": (na // 4) / ta1,
                          "sample": f"{na} samples; shared lock + map[name][int16] + atomic per call "
                                    "(cost shape of metrics.go:273-295, BASELINE.md form A)"},
    }


def timed_steps(step, steps, warmup, fence):
    for _ in range(warmup):
        out = step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step(True)
    fence()
    return time.perf_counter() - t0, out


def pipelined_steps(ingest, flip, finish, steps, warmup, fence):
    """K steps driven the way the engine is meant to be driven -- and the reference is built: its reaper reduces
    interval i on the worker pool while the producers already fill interval i + 1 (metrics.go:530-639).  Step i =
    ingest_i, flip_i, reduce_i (merge, extract, results on the host, release); ingest_{i+1} is enqueued right after
    flip_i, BEFORE the host waits for reduce_i's results, so that the small reduction runs beside the next ingest kernel
    instead of leaving the GPU idle for a launch round trip per step.  Exactly `steps` ingests, flips and reductions
    happen inside the timed region (the fences on both sides drain everything); nothing is skipped or carried over."""
    def run(k, timed):
        out = None
        if k:
            ingest(timed)
        for i in range(k):
            snap = flip()
            if i + 1 < k:
                ingest(timed)
            out = finish(snap, timed)
        return out

    run(warmup, False)
    fence()
    t0 = time.perf_counter()
    out = run(steps, True)
    fence()
    return time.perf_counter() - t0, out


def dense_from_csr(off, keys, counts, M):
    import oracle
    dense = np.zeros((M, 65536), dtype=np.uint64)
    rows = np.repeat(np.arange(M), np.diff(off.astype(np.int64)))
    dense[rows, oracle.key_to_bin(keys)] = counts
    return dense


# ---------------------------------------------------------------------------------------------------------
# the communicator of the C-ABI merge
# ---------------------------------------------------------------------------------------------------------
def make_comm(world, rank, dist, comm_override=None):
    """(ncclComm_t, front-end label, reason): the communicator lh_snapshot_merge runs on (what a cgo caller would hold;
    torch.distributed only carries the unique id).  If RCCL cannot be set up from here the run continues through the
    torch.distributed front-end (loghisto_amd.merge) -- NOT silently: the label goes into the line's top-level
    `config.merge` as "fallback: ..." together with the reason."""
    from loghisto_amd import rccl
    if comm_override is not None:
        return comm_override, "c-abi: lh_snapshot_merge -> RCCL", ""
    try:
        if world > 1:
            uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(rccl.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = rccl.comm_init_rank(world, bytes(uid.cpu().numpy().tobytes()), rank)
        else:
            comm = rccl.comm_init_rank(1, rccl.unique_id(), 0)
        return comm, "c-abi: lh_snapshot_merge -> RCCL", ""
    except Exception as exc:  # noqa: BLE001
        if world == 1:
            return 0, "none (one rank, no communicator)", repr(exc)
        return 0, "fallback: torch.distributed front-end (loghisto_amd.merge), the RCCL communicator for the C ABI " \
                  "could not be made", repr(exc)


# ---------------------------------------------------------------------------------------------------------
# C2: single metric (the headline); with several ranks: weak scaling of the same stream, one all-reduce per step
# ---------------------------------------------------------------------------------------------------------
def run_c2(args, la, stream, rank, world=1, dist=None, comm=0, frontend="none", why=""):
    """Every rank buckets its own slice (n samples of the one metric); with world > 1 the row's merged window is
    all-reduced at the flip (lh_snapshot_merge, LH_MERGE_ALLREDUCE) before every rank extracts.  world == 1 is the
    N = 1 headline: the same code with no merge."""
    from loghisto_amd import merge as tmerge
    n = int(args.samples)
    eng = la.Engine(device=torch.cuda.current_device(), max_metrics=1, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    buf = own(make_samples(n, args.dist, seed=2 + rank))          # plain hipMalloc'ed memory (OwnBuffer)
    data = buf.tensor
    torch.cuda.synchronize()
    events = []

    def ingest(timed):
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
        eng.submit_device(0, data, n, stream=stream)               # K1 on torch's current stream
        if timed:
            b.record(stream)
            events.append((a, b))

    def finish(snap, timed):
        if world > 1:                                              # the only collective: the one row's window
            if comm:
                snap.merge_rccl(comm, world, rank, 1, plan="allreduce")
            else:
                tmerge.merge_snapshot(snap, 1, plan="allreduce")
        out = snap.extract(PCTS, 1)                                # K2 + results on the host
        if timed and world > 1 and comm:
            minfo.update(snap.merge_info())
        snap.release()                                             # K3 (async)
        return out

    minfo = {}

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU's clocks take ~10 full-size launches to settle after an idle period (profiles/r04_kernel_trace.txt: the
    # first launches of a process run 1.76, 1.68, 1.55, 1.45, 1.46, 1.34, 1.34, 1.30 ... ms against 1.25 in steady state),
    # more than the W warm-up steps a caller may ask for: PREWARM untimed K1 launches into an interval that is thrown
    # away, before the W warm-up steps.  Setup, like generating the data; disclosed in config.prewarm.
    for _ in range(PREWARM):
        eng.submit_device(0, data, n, stream=stream)
    eng.flip().release()
    torch.cuda.synchronize()
    dt, out = pipelined_steps(ingest, eng.flip, finish, args.steps, args.warmup, fence)
    if dist is not None:                                           # the slowest rank's clock
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert int(out["count"].sum()) == n * world, (int(out["count"].sum()), n, world)
    k1_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    # The same K steps once more, SERIAL: every step drained (ingest, flip, reduce, results on the host) before the next
    # one is enqueued -- what a caller gets that does not overlap reduction with the next interval's ingest (ADVICE r4:
    # the pipelined figure alone is not comparable with rounds 1 - 3).  Reported beside the headline, never as `value`.
    def serial_step(timed):
        eng.submit_device(0, data, n, stream=stream)
        return finish(eng.flip(), False)

    dt_serial, _ = timed_steps(serial_step, args.steps, 1, fence)
    if dist is not None:
        tt = torch.tensor([dt_serial], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_serial = float(tt.item())
    # what a kernel that only reads gets from this box, now, on the same block of memory (K1's loop without the buckets)
    ceil_avg, ceil_best = buf.read_ceiling_gbs(reps=10, stream=stream)

    # p99 extract latency: flip -> stats on host (BASELINE.json metric, part 2); one rank's own interval
    lat = []
    small = data[: 1 << 20]
    for _ in range(args.latency_flips if world == 1 else 0):
        eng.submit_device(0, small, small.numel(), stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        snap = eng.flip()
        snap.extract(PCTS, 1)
        lat.append(time.perf_counter() - t1)
        snap.release()
    lat_us = np.array(lat) * 1e6 if lat else np.array([float("nan")])

    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, K1_PMC)
    if os.path.exists(pmc):
        try:
            # PMC passes cannot run inside this process: the committed rocprofv3 --pmc summary of this same
            # command supplies HBM bytes per K1 launch (read + write, corrected as the microarch guide
            # prescribes; see the JSON's "corrections").  It only applies to the workload it was measured on.
            j = json.load(open(pmc))
            if n * BYTES_SINGLE == int(j["algorithmic_bytes_per_launch"]) and args.dist == "lognormal":
                stale = pmc_stale(j, "k1")
                if stale:
                    traffic_source = f"stale: {K1_PMC} not used ({stale})"
                else:
                    traffic = j["hbm_read_bytes_per_launch"] + j["hbm_write_bytes_per_launch"]
                    traffic_source = f"{K1_PMC} (committed rocprofv3 --pmc summary of this command on these kernel " \
                                     "sources, not measured in this run)"
        except Exception:
            traffic = None
    res = {
        "value": world * n * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_serial": dt_serial / args.steps * 1e3, "value_serial": world * n * args.steps / dt_serial,
        "config": {"workload": "C2 single-metric float64 stream, one ingest kernel + one percentile scan per step"
                               + ("" if world == 1 else "; every rank buckets its own slice, the row's merged window is "
                                  "all-reduced at the flip (lh_snapshot_merge, LH_MERGE_ALLREDUCE), every rank extracts"),
                   "samples_per_gpu_per_step": n, "distribution": args.dist, "metrics": 1, "percentiles": PCTS,
                   "ranks": world, "merge": "none" if world == 1 else frontend,
                   "prewarm": f"{PREWARM} untimed K1 launches before the W warm-up steps (clock ramp after idle)",
                   "input_memory": "hipMalloc (lh_tool_device_alloc), not torch's caching allocator",
                   "steps": "software-pipelined: ingest of step i + 1 is enqueued after the flip of step i, before the host "
                            "waits for step i's results (as the reference's reaper overlaps reduction with ingest)"},
        "roofline": roofline(n * BYTES_SINGLE, k1_ms, "k_ingest_single", traffic=traffic,
                             traffic_source=traffic_source, frac_of_measured_copy_ceiling_6290=n * BYTES_SINGLE / (k1_ms * 1e-3) / 1e9 / 6290.0,
                             read_ceiling_same_box=ceil_avg, read_ceiling_same_box_best_launch=ceil_best,
                             frac_of_read_ceiling=n * BYTES_SINGLE / (k1_ms * 1e-3) / 1e9 / ceil_avg,
                             read_ceiling_source="lh_tool_read_ceiling: 10 launches of K1's loop without the bucket work over "
                                                 "the step's input block, in this process, right after the timed steps"),
        "extract_latency_us": {"p50": float(np.percentile(lat_us, 50)), "p99": float(np.percentile(lat_us, 99)),
                               "flips": len(lat), "names": 1},
    }
    if world > 1:
        res["value_per_gpu"] = res["value"] / world
        res["merge"] = {"device_ms": {k: minfo.get(k) for k in ("ranges_ms", "plan_ms", "pack_ms", "collective_ms",
                                                                "unpack_ms", "span_ms")},
                        "packed_cells": minfo.get("packed_cells"), "cell_bytes": minfo.get("cell_bytes"),
                        "note": "one row: its merged window (~1 000 cells) is all that crosses xGMI per step"}
        if why:
            res["config"]["merge_fallback_reason"] = why
    host = None
    if not args.no_parity and world > 1:
        # the merged row on every rank == the oracle over the CONCATENATED slices: every rank buckets its own slice
        # with the oracle, the expected merged row is the all-reduced sum of those
        import oracle
        eng.submit_device(0, data, n, stream=stream)
        snap = eng.flip()
        if comm:
            snap.merge_rccl(comm, world, rank, 1, plan="allreduce")
        else:
            tmerge.merge_snapshot(snap, 1, plan="allreduce")
        gpu_row = snap.dense_row(0)
        snap.release()
        mine = oracle.histogram_dense_mt(data.cpu().numpy())
        want = torch.from_numpy(mine.astype(np.int64)).cuda()
        dist.all_reduce(want)
        want = want.cpu().numpy().astype(np.uint64)
        ok = torch.tensor([int(np.array_equal(gpu_row, want))], dtype=torch.int64, device="cuda")
        dist.all_reduce(ok)
        res["parity"] = {"samples_checked": int(want.sum()), "exact": int(ok.item()) == world,
                         "ranks_with_the_exact_merged_row": int(ok.item()),
                         "checker": "oracle/ over every rank's slice, all-reduced, against the merged GPU row on every rank"}
        assert res["parity"]["exact"], "merged GPU row differs from the oracle over the concatenated slices"
    if not args.no_parity and world == 1:
        # the whole step's stream through the oracle on every granted host core: bit-exact bucket counts
        import oracle
        eng.submit_device(0, data, n, stream=stream)
        with eng.flip() as snap:
            gpu_row = snap.dense_row(0)
        t0 = time.perf_counter()
        host = data.cpu().numpy()
        t_copy = time.perf_counter() - t0
        t0 = time.perf_counter()
        want = oracle.histogram_dense_mt(host)
        t_cpu = time.perf_counter() - t0
        res["parity"] = {"samples_checked": int(want.sum()), "exact": bool(np.array_equal(gpu_row, want)),
                         "checker": f"oracle/ (C restatement of metrics.go:316-322 + Go math.Log), {effective_cores()} "
                                    f"threads, {t_cpu:.2f} s (+ {t_copy:.2f} s for the 8 GB device-to-host copy)"}
        assert res["parity"]["exact"], "GPU row differs from the oracle over the full stream"
    if not args.no_cpu_baseline and world == 1 and rank == 0:
        if host is None:
            host = data[: 1 << 27].cpu().numpy()
        res["cpu_baseline"] = cpu_baseline(host, args.cpu_seconds)
    del host
    eng.close()
    del data
    buf.free()
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------
# C3: 1 024 Zipf names, (id, value) stream
# ---------------------------------------------------------------------------------------------------------
def c3_traffic(n, names, which=None):
    """HBM bytes of one call from the committed rocprofv3 --pmc summary of the same stream (PMC passes cannot run
    inside this process); None when the run is not that configuration."""
    which = which or C3_PMC
    path = os.path.join(ROOT, which)
    try:
        j = json.load(open(path))
        if j["pairs_per_call"] == n and j["names"] == names:
            stale = pmc_stale(j, "c3" if names <= 8192 else "c4")
            if stale:
                return {"traffic": None, "traffic_source": f"stale: {which} not used ({stale})"}
            return {"traffic": j["hbm_bytes_per_call"],
                    "traffic_source": f"{which} (committed rocprofv3 --pmc summary of the same command on these kernel "
                                      "sources, every kernel of one call summed; not measured in this run)"}
    except (OSError, ValueError, KeyError):
        pass
    return {"traffic": None, "traffic_source": None}


def run_c3(args, la, stream, rank, steps, warmup, latency_flips=0):
    n = int(args.samples)
    M = args.names or 1024
    eng = la.Engine(device=torch.cuda.current_device(), max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    ids = zipf_ids(n, M, 1003 + rank)
    data = make_samples(n, args.dist, seed=3 + rank)
    chunk = 1 << 27
    for lo in range(0, n, chunk):      # value ~ lognormal(ln 1e5 + 0.002*id, 1): every name its own
        data[lo:lo + chunk].mul_(torch.exp(0.002 * ids[lo:lo + chunk].to(torch.float64)))
    bi, bd = OwnBuffer(ids), OwnBuffer(data)                       # plain hipMalloc'ed memory, as the headline's input
    ids, data = bi.tensor, bd.tensor
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    events = []

    def ingest(timed):
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
        eng.submit_pairs_device(ids, data, n, stream=stream)       # survey + scatter + plan + P2 on this stream
        if timed:
            b.record(stream)
            events.append((a, b))

    def finish(snap, timed):
        out = snap.extract(PCTS, M)
        snap.release()
        return out

    dt, out = pipelined_steps(ingest, eng.flip, finish, steps, warmup, torch.cuda.synchronize)
    assert int(out["count"].sum()) == n
    ing_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    c = eng.counters()
    res = {"value": n * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "config": {"workload": "C3 mixed (uint32 id, float64 value) stream over Zipf(1.0) names, one percentile "
                                  "scan per name per step", "samples_per_step": n, "metrics": M,
                      "distribution": args.dist, "path": "survey + 2-byte-record scatter + LDS reduce"
                      if c["samples_partitioned_v2"] else "partitioned (first generation)"},
           "roofline": roofline(n * BYTES_PAIR, ing_ms,
                                "k_survey_* + k_scatter3 + k_plan_* + k_part_hist2 (all launches of one lh_submit_pairs_device)",
                                **c3_traffic(n, M)),
           "scratch_bytes": c["scratch_bytes"], "sublaunches_per_step": c["sublaunches"] // max(1, steps + warmup)}
    if latency_flips:
        lat = []
        sl_i, sl_v = ids[: 1 << 22], data[: 1 << 22]
        for _ in range(latency_flips):
            eng.submit_pairs_device(sl_i, sl_v, sl_v.numel(), stream=stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            snap = eng.flip()
            snap.extract(PCTS, M)
            lat.append(time.perf_counter() - t1)
            snap.release()
        lu = np.array(lat) * 1e6
        res["extract_latency_us"] = {"p50": float(np.percentile(lu, 50)), "p99": float(np.percentile(lu, 99)),
                                     "flips": len(lat), "names": M}
    if not args.no_parity:
        import oracle
        eng.submit_pairs_device(ids, data, n, stream=stream)
        with eng.flip() as snap:
            off, keys, counts = snap.buckets_all(M)
        t0 = time.perf_counter()
        want = oracle.histogram_pairs_mt(ids.cpu().numpy().view(np.uint32), data.cpu().numpy(), M)
        t_cpu = time.perf_counter() - t0
        exact = bool(np.array_equal(dense_from_csr(off, keys, counts, M), want))
        res["parity"] = {"samples_checked": int(want.sum()), "rows_checked": M, "exact": exact,
                         "checker": f"oracle/ over every pair, {effective_cores()} threads, {t_cpu:.1f} s incl. the copy"}
        assert exact, "GPU cells differ from the oracle over the full C3 stream"
        del want
    eng.close()
    del ids, data
    bi.free()
    bd.free()
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------
# C4: 65 536 names, data-parallel ingest, reduce-scatter merge through the C ABI
# ---------------------------------------------------------------------------------------------------------
def run_c4(args, la, stream, rank, world, dist, steps, warmup, comm_override=None, frontend=None, why=""):
    """comm_override: the ncclComm_t to merge on -- the one main() made for this job (make_comm; 0 = the recorded
    torch.distributed fallback), or one the caller made (tests/_bench_ranks_driver.py: ranks as threads on one GPU over
    the stub RCCL, with a thread-rendezvous stand-in for torch.distributed as `dist`).  None: a one-rank communicator
    of its own (world == 1)."""
    import ctypes as C
    from loghisto_amd import _native as N
    from loghisto_amd import merge as tmerge
    from loghisto_amd import rccl
    M = args.names or 65536
    n = int(args.c4_slice)
    dev = torch.cuda.current_device()
    eng = la.Engine(device=dev, max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    ids = zipf_ids(n, M, 4000 + rank)                       # this rank's slice of a Zipf stream over ALL names
    data = make_samples(n, "lognormal", seed=40 + rank)
    data.mul_(torch.exp(3e-5 * ids.to(torch.float64)))
    bi, bd = OwnBuffer(ids), OwnBuffer(data)                # plain hipMalloc'ed memory, as the headline's input
    ids, data = bi.tensor, bd.tensor
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    own_comm = False
    if comm_override is None:
        comm, frontend, why = make_comm(world, rank, dist)
        own_comm = bool(comm)
    else:
        comm = comm_override
        frontend = frontend or "c-abi: lh_snapshot_merge -> RCCL"
    t_ing, t_merge, t_mwall, t_ext, t_k2, t_k2k, t_k2c = [], [], [], [], [], [], []
    info = {}

    def step(timed):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        eng.submit_pairs_device(ids, data, n, stream=stream)
        b.record(stream)
        snap = eng.flip()
        if timed:
            torch.cuda.synchronize()                                 # merge ms below = the merge alone, not the ingest tail
        t0 = time.perf_counter()
        if comm:
            first, last = snap.merge_rccl(comm, world, rank, M, plan="reduce_scatter")
        elif world > 1:
            first, last = tmerge.merge_snapshot(snap, M, plan="reduce_scatter")
        else:
            first, last = 0, M
        t1 = time.perf_counter()
        xs = torch.cuda.ExternalStream(snap.stream())            # the snapshot's own stream: K2 runs there
        if timed:
            # lh_snapshot_merge returns once its last kernel is enqueued: wait for the merge here, so that the extract
            # time below is the extract's (round 4 charged it the merge's 0.34 ms of device time: "extract_owned_ms 0.75")
            xs.synchronize()
        t1b = time.perf_counter()
        k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k0.record(xs)
        out = snap.extract_view(PCTS, last - first, first=first) # the names this rank owns, results in place (pinned)
        k1.record(xs)
        out = {k: v.copy() for k, v in out.items() if k in ("count", "nbuckets")}
        t2 = time.perf_counter()
        if timed:
            info.update(snap.merge_info() if comm else dict(tmerge.last_info))
            km, cm = C.c_float(0), C.c_float(0)
            if N.lib().lh_tool_last_extract_ms(eng._h, C.byref(km), C.byref(cm)) == 0:
                t_k2k.append(km.value)
                t_k2c.append(cm.value)
        snap.release()
        if timed:
            torch.cuda.synchronize()
            t_ing.append(a.elapsed_time(b))
            t_merge.append((t1 - t0) * 1e3)
            t_mwall.append((t1b - t0) * 1e3)
            t_ext.append((t2 - t1b) * 1e3)
            t_k2.append(k0.elapsed_time(k1))
        return out, (first, last)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # serial steps first: every phase drained before the next starts, so that the per-phase times (ingest, merge,
    # extract) are each phase alone
    dt_serial, (out, (first, last)) = timed_steps(step, steps, warmup, fence)

    # then the same K steps pipelined, as the headline's are (pipelined_steps): ingest of step i + 1 is enqueued right
    # after flip i, before the host waits for merge i and extract i -- the exchange and the small reduction kernels run
    # beside the next ingest on the snapshot's own stream.  `value` is this pass.
    def p_ingest(timed):
        eng.submit_pairs_device(ids, data, n, stream=stream)

    def p_finish(snap, timed):
        if comm:
            f, l = snap.merge_rccl(comm, world, rank, M, plan="reduce_scatter")
        elif world > 1:
            f, l = tmerge.merge_snapshot(snap, M, plan="reduce_scatter")
        else:
            f, l = 0, M
        o = snap.extract_compact(PCTS, l - f, first=f)   # the compact results (42 B per name): what the host layers take
        o = {k: v.copy() for k, v in o.items() if k in ("count", "nbuckets")}
        snap.release()
        return o, (f, l)

    dt, (out_p, fl_p) = pipelined_steps(p_ingest, eng.flip, p_finish, steps, min(warmup, 2), fence)
    assert fl_p == (first, last) and np.array_equal(out_p["count"], out["count"]), "pipelined steps: other results"
    if dist is not None:
        tt = torch.tensor([dt, dt_serial], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_serial = float(tt[0].item()), float(tt[1].item())
    # the owner blocks tile [0, M) in rank order (equal shares of the packed cells, not of the names)
    if dist is not None:
        fl = torch.tensor([first, last], dtype=torch.int64, device="cuda")
        allfl = [torch.zeros_like(fl) for _ in range(world)]
        dist.all_gather(allfl, fl)
        bounds = [tuple(int(x) for x in t.tolist()) for t in allfl]
    else:
        bounds = [(first, last)]
    assert bounds[0][0] == 0 and bounds[-1][1] == M and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1)), bounds
    # parity: every sample of every rank is in exactly one owned row (conservation), per-name counts of the owned
    # rows equal the all-reduced bincount of the ids
    per_name = torch.bincount(ids, minlength=M).to(torch.int64)
    if dist is not None:
        dist.all_reduce(per_name)
    counts_ok = bool(np.array_equal(out["count"].astype(np.int64), per_name[first:last].cpu().numpy()))
    owned_total = torch.tensor([int(out["count"].sum())], dtype=torch.int64, device="cuda")
    if dist is not None:
        dist.all_reduce(owned_total)
    conserved = int(owned_total.item()) == n * world
    parity = {"per_name_counts_exact": counts_ok, "samples_conserved": conserved, "exact": counts_ok and conserved}
    plan8 = None
    if not args.no_parity:
        # A probe set of rows cell by cell against the oracle AFTER the merge (dense matrices of 65 536 names do not
        # fit): every rank buckets ITS samples of every probe name with the oracle, the expected merged rows are the
        # all-reduced sum of those, and each rank compares the probes it owns with what the merge left in its rows.
        import oracle
        probe = sorted({0, 1, 2, 7, 63, 64, 255, 1023, 4095, M // 2, M - 2, M - 1} | set(range(100, 100 + 20))
                       | {int(x) for x in np.linspace(0, M - 1, 4 * world + 8)})
        probe = [m for m in probe if m < M]
        mine = np.stack([oracle.histogram_dense(data[ids == m].cpu().numpy()) for m in probe]).astype(np.int64)
        want = torch.from_numpy(mine).cuda()
        if dist is not None:
            dist.all_reduce(want)
        want = want.cpu().numpy().astype(np.uint64)
        eng.submit_pairs_device(ids, data, n, stream=stream)
        snap = eng.flip()
        if comm:
            f2, l2 = snap.merge_rccl(comm, world, rank, M, plan="reduce_scatter")
        elif world > 1:
            f2, l2 = tmerge.merge_snapshot(snap, M, plan="reduce_scatter")
        else:
            f2, l2 = 0, M
        owned = [i for i, m in enumerate(probe) if f2 <= m < l2]
        ok = all(np.array_equal(snap.dense_row(probe[i]), want[i]) for i in owned)
        if world == 1:
            # what an 8-rank reduce-scatter of this interval would pad: equal packed cells per block (this library)
            # against equal name counts per block (round 2), from the snapshot's merged ranges
            torch.cuda.synchronize()
            rg = tmerge.snapshot_ranges(snap, M).cpu().numpy().view(np.uint32).astype(np.int64)
            wd = np.where(rg[:, 0] <= rg[:, 1], rg[:, 1] - rg[:, 0] + 1, 0)
            P = np.concatenate([[0], np.cumsum(wd)])
            cut = [int(np.searchsorted(P, P[-1] * k // 8, side="left")) for k in range(9)]
            cut[0], cut[8] = 0, M
            bal = max(P[cut[k + 1]] - P[cut[k]] for k in range(8)) * 8 / max(1, P[-1])
            per8 = -(-M // 8)
            eq = max(P[min(M, (k + 1) * per8)] - P[min(M, k * per8)] for k in range(8)) * 8 / max(1, P[-1])
            plan8 = {"ranks": 8, "padding_ratio_equal_cells": float(bal), "padding_ratio_equal_names": float(eq)}
        snap.release()
        nchk = torch.tensor([len(owned), int(ok)], dtype=torch.int64, device="cuda")
        if dist is not None:
            dist.all_reduce(nchk)
        parity.update(rows_checked_cell_by_cell=int(nchk[0].item()), cells_after_merge_exact=int(nchk[1].item()) == world,
                      exact=parity["exact"] and int(nchk[1].item()) == world)
    assert parity["exact"], parity
    lat_c4 = None
    nlat = int(getattr(args, "latency_flips", 1000)) + 10          # BASELINE's metric: p99 over >= 1 000 flips (ten more to warm up)
    if world == 1 and steps and nlat > 20:
        # flip -> results of ALL 65 536 names on the host, small intervals, over >= 1 000 flips each:
        #   compact  lh_extract_rows_compact: count, sum, occupied buckets, the selected keys, valid bits (42 B per name;
        #            avg, uint64(sum) and the percentile VALUES = D[key] are functions of these: lh_expand_compact, or the
        #            host layer's own formatting) -- the headline of this leg
        #   view     lh_extract_rows_view: the full lh_stats + pvals + pkeys + pvalid (139 B per name), in place
        sl_i, sl_v = ids[: 1 << 22], data[: 1 << 22]
        forms = {}
        for form in ("compact", "view"):
            lat = []
            for _ in range(nlat):
                eng.submit_pairs_device(sl_i, sl_v, sl_v.numel(), stream=stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                snap = eng.flip()
                (snap.extract_compact if form == "compact" else snap.extract_view)(PCTS, M)
                lat.append(time.perf_counter() - t1)
                snap.release()
            lu = np.array(lat[10:]) * 1e6
            forms[form] = {"p50": float(np.percentile(lu, 50)), "p99": float(np.percentile(lu, 99)), "flips": len(lu)}
        # what the host-side derivation of the full form costs (not part of the latency above: a binding that formats keys
        # reads D[key] as it goes), and that it equals the full form bit for bit on this snapshot
        eng.submit_pairs_device(sl_i, sl_v, sl_v.numel(), stream=stream)
        with eng.flip() as snap:
            full = {k: np.array(v) for k, v in snap.extract_view(PCTS, M).items()}
            cpt = snap.extract_compact(PCTS, M)
            snap.expand_compact(cpt)                      # (the first call reads the decompress table back, once per engine)
            t1 = time.perf_counter()
            ex = snap.expand_compact(cpt)
            expand_ms = (time.perf_counter() - t1) * 1e3

            def same(a, b):
                a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
                if a.dtype.kind == "f":
                    nan = np.isnan(a) & np.isnan(b)
                    a, b = np.where(nan, 0.0, a).view(np.uint64), np.where(nan, 0.0, b).view(np.uint64)
                return bool(np.array_equal(a, b))
            expand_exact = all(same(ex[k], full[k]) for k in ("count", "sum", "avg", "agg_sum_add", "nbuckets", "present",
                                                              "pvals", "pkeys", "pvalid"))
        assert expand_exact
        lat_c4 = {"p50": forms["compact"]["p50"], "p99": forms["compact"]["p99"], "flips": forms["compact"]["flips"],
                  "names": M, "api": "lh_extract_rows_compact (count, sum, nbuckets, keys, valid bits: 42 B per name, in "
                                     "place in pinned memory)",
                  "full_form_view": dict(forms["view"], api="lh_extract_rows_view (139 B per name)"),
                  "expand_compact_ms_python_binding": expand_ms, "expand_equals_full_form_bit_for_bit": expand_exact}
    c = eng.counters()
    res = {
        "value": world * n * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
        "serial_ms_per_step": dt_serial / steps * 1e3,
        "config": {"workload": "C4 65536 histogram names, every rank ingests its slice of a Zipf(1.0) stream over ALL "
                               "names, reduce-scatter merge of the per-row windows at the flip, extract of the owned names",
                   "step_pipelining": "value / ms_per_step: ingest of step i + 1 enqueued after flip i, before the host "
                                      "waits for merge i and extract i (the reference's reaper overlaps reduction with "
                                      "ingest, metrics.go:530-639); serial_ms_per_step and the per-phase times: the same "
                                      "K steps with every phase drained before the next (the serial steps extract the "
                                      "full form, lh_extract_rows_view, so that their phase times compare with earlier "
                                      "rounds; the pipelined steps the compact one, lh_extract_rows_compact)",
                   "names": M, "pairs_per_gpu_per_step": n, "ranks": world, "owned_rows": [first, last],
                   "merge": frontend, "percentiles": PCTS},
        "roofline": roofline(n * BYTES_PAIR, sum(t_ing) / len(t_ing),
                             "k_survey_count_h .. k_scatter4 + k_split_waves + k_part_hist3 (third generation of the "
                             "partitioned ingest: every launch of one lh_submit_pairs_device)", **c3_traffic(n, M, C4_PMC)),
        "merge": {"host_call_ms": sum(t_merge) / len(t_merge), "wall_ms": sum(t_mwall) / len(t_mwall),
                  "device_ms": {k: info.get(k) for k in ("ranges_ms", "plan_ms", "pack_ms", "collective_ms",
                                                         "unpack_ms", "span_ms")},
                  "packed_cells": info.get("packed_cells"), "packed_words": info.get("packed_words"),
                  "padded_words": info.get("padded_words"),
                  "padding_ratio": (info["padded_words"] / info["packed_words"]
                                    if info.get("packed_words") and info.get("padded_words") else None),
                  "cell_bytes": info.get("cell_bytes"),
                  # a row travels at 8 / 16 / 32 bits per cell, the narrowest that holds ranks x its largest per-rank
                  # cell (LH_OPT_MERGE_NARROW_CELLS; one rank moves nothing between GPUs and skips the pass over the cells)
                  "rows_8bit": info.get("rows_8bit"), "rows_16bit": info.get("rows_16bit"),
                  "wire_bytes_per_cell": (info["packed_words"] * info["cell_bytes"] / info["packed_cells"]
                                          if info.get("packed_cells") and info.get("packed_words") else None),
                  "send_bytes": info.get("send_bytes"), "recv_bytes": info.get("recv_bytes"),
                  "widest_row": info.get("widest_row"), "occupied_rows": info.get("occupied_rows"),
                  # how dense the windows that travel are: occupied cells of the rows this rank owns
                  # after the merge over the cells of their windows; a (key, count) list for rows under ~2/3 density would
                  # move 6 B per occupied cell instead of 4 B per window cell
                  "owned_occupied_cells": int(out["nbuckets"].sum()),
                  "owned_window_density": (float(out["nbuckets"].sum()) * world / info["packed_cells"]
                                           if info.get("packed_cells") else None),
                  "owned_rows_by_rank": bounds if world <= 16 else None,
                  "note": "device_ms: HIP events on the snapshot stream around the merge's steps of the last timed "
                          "step (dirty-range all-reduce, window plan, pack, collective, unpack; span includes the host "
                          "round trip for the plan totals); host_call_ms: wall time of lh_snapshot_merge returning; "
                          "wall_ms: until its last kernel has finished (extract_owned_ms starts there)"},
        "extract_owned_ms": sum(t_ext) / len(t_ext), "parity": parity, "scratch_bytes": c["scratch_bytes"],
    }
    if info.get("packed_cells") and t_k2:
        # K2's roofline: every cell of every owned row's window read once (8 B x row_len per metric);
        # after a reduce-scatter the owned rows' windows are packed_cells / ranks on average
        # (ABI 7: above 8 192 names the store's cells are uint32 -- the bytes K2 has to read are the store's, not the reference's 8)
        owned_cells = info["packed_cells"] / world
        cell_b = float(N.lib().lh_cell_bytes(eng._h))
        res["store_cell_bytes"] = int(cell_b)
        res["extract_roofline"] = roofline(cell_b * owned_cells, sum(t_k2) / len(t_k2),
                                           "k_extract_wave over the owned rows + the device-to-host copy of the results "
                                           "(139 B per name) behind it: HIP events on the snapshot's stream around "
                                           "lh_extract_rows_view (the pipelined steps' form; the compact form's latency is "
                                           "in extract_latency_us).  The kernel alone: profiles/r06_c4_kernel_trace.txt",
                                           window_cells=owned_cells)
        if t_k2k:
            # the two parts apart (lh_tool_last_extract_ms: HIP events inside lh_extract_rows_view): K2's own roofline is
            # the kernel's; the copy is 139 B per name over PCIe
            kms, cms = sum(t_k2k) / len(t_k2k), sum(t_k2c) / len(t_k2c)
            res["extract_roofline"].update(kernel_ms=kms, copy_ms=cms, kernel_frac=cell_b * owned_cells / (kms * 1e-3) / 8e12,
                                           copy_GBps=139.0 * (last - first) / (cms * 1e-3) / 1e9 if cms > 0 else None)
    if plan8:
        res["merge"]["simulated_plan"] = plan8
    if lat_c4:
        res["extract_latency_us"] = lat_c4
    if why:
        res["config"]["merge_fallback_reason"] = why
    if own_comm:
        rccl.comm_destroy(comm)
    eng.close()
    del ids, data
    bi.free()
    bd.free()
    torch.cuda.empty_cache()
    return res


def run_c4_1e9(args, la, stream, steps=3, warmup=2):
    """config 4's name count at a full-size interval: 65 536 Zipf names, 1e9 pairs per step on one rank (the slice of
    c4_one_rank is 1.25e8 pairs = 2 ms of stream, where fixed work dominates; SURVEY.md 8d says "merge every 1 s of
    simulated stream").  ingest + flip + extract of all names (results in place) + release per step."""
    M, n = 65536, int(1e9)
    eng = la.Engine(device=torch.cuda.current_device(), max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    ids = zipf_ids(n, M, 4100)
    data = make_samples(n, "lognormal", seed=41)
    chunk = 1 << 27
    for lo in range(0, n, chunk):
        data[lo:lo + chunk].mul_(torch.exp(3e-5 * ids[lo:lo + chunk].to(torch.float64)))
    bi, bd = OwnBuffer(ids), OwnBuffer(data)                # plain hipMalloc'ed memory, as the headline's input
    ids, data = bi.tensor, bd.tensor
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    events = []

    def step(timed):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        eng.submit_pairs_device(ids, data, n, stream=stream)
        b.record(stream)
        if timed:
            events.append((a, b))
        snap = eng.flip()
        out = snap.extract_view(PCTS, M)
        cnt = out["count"].copy()
        snap.release()
        return cnt

    dt, cnt = timed_steps(step, steps, warmup, torch.cuda.synchronize)
    ing_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    per_name = torch.bincount(ids, minlength=M).cpu().numpy()
    counts_ok = bool(np.array_equal(cnt.astype(np.int64), per_name))
    parity = {"per_name_counts_exact": counts_ok, "exact": counts_ok}
    if not args.no_parity:
        import oracle
        probe = sorted({0, 1, 2, 7, 63, 255, 1023, 4095, M // 2, M - 2, M - 1} | {int(x) for x in np.linspace(0, M - 1, 12)})
        eng.submit_pairs_device(ids, data, n, stream=stream)
        with eng.flip() as snap:
            ok = all(np.array_equal(snap.dense_row(m), oracle.histogram_dense(data[ids == m].cpu().numpy())) for m in probe)
        parity.update(rows_checked_cell_by_cell=len(probe), exact=counts_ok and ok)
    assert parity["exact"], parity
    c = eng.counters()
    res = {"value": n * steps / dt, "unit": "samples/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
           "config": {"workload": "65536 histogram names (config 4's name count), Zipf(1.0), 1e9 pairs per step on one "
                                  "rank, extract of every name per step", "names": M, "pairs_per_step": n},
           "roofline": roofline(n * BYTES_PAIR, ing_ms,
                                "k_scatter4 + k_split_waves + k_part_hist3 (+ survey, plans): every launch of one "
                                "lh_submit_pairs_device", **c3_traffic(n, M, C4_1E9_PMC)),
           "parity": parity, "scratch_bytes": c["scratch_bytes"]}
    eng.close()
    del ids, data
    bi.free()
    bd.free()
    torch.cuda.empty_cache()
    return res


# ---------------------------------------------------------------------------------------------------------
# host-fed (PCIe-inclusive) and the C5 burst
# ---------------------------------------------------------------------------------------------------------
def run_hostfed(la, M=1024, total=int(8e8), forms=("in_place16", "in_place", "submit_pairs", "copy_engine")):
    """Host arrays -> pinned staging buffers -> PCIe -> buckets, from T producer threads.  Four forms of the same
    stream: lh_reserve_pairs16 / lh_commit_pairs16 (uint16 ids, 10 B per pair over the link, the producer's own store
    is the only host-side copy: what the binding uses for <= 65 536 names), lh_reserve_pairs / lh_commit_pairs (the
    same with uint32 ids, 12 B), lh_submit_pairs (the library copies the caller's batch into a pinned buffer; ids
    validated on the host), and lh_submit_pairs with LH_OPT_LANE_ZERO_COPY = 0 (round 2's path: hipMemcpyAsync into
    HBM before the kernel).  value = in place with uint16 ids."""
    import oracle
    from loghisto_amd import _native as N
    cores = effective_cores()
    T = max(1, min(16, cores))
    rng = np.random.default_rng(1)
    src = rng.lognormal(np.log(1e5), 1.0, 1 << 24)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=src.size, p=w / w.sum()).astype(np.uint32)
    ids16 = ids.astype(np.uint16)
    batch = 1 << 20
    per = total // T
    offs = [(t * 7919 * batch) % (src.size - batch) for t in range(T)]
    # every cell against the oracle: thread t submits the same slice [off_t, off_t + batch) per // batch times and
    # its first per % batch samples once more
    reps, rem = per // batch, per % batch
    dense = M <= 8192          # a dense oracle matrix of 65 536 names would be 32 GiB: sorted (name << 16 | bin, count) lists
    if dense:
        want = np.zeros((M, 65536), dtype=np.uint64)
        if reps:
            want += np.uint64(reps) * oracle.histogram_pairs_mt(np.concatenate([ids[o:o + batch] for o in offs]),
                                                                 np.concatenate([src[o:o + batch] for o in offs]), M)
        if rem:
            want += oracle.histogram_pairs_mt(np.concatenate([ids[o:o + rem] for o in offs]),
                                              np.concatenate([src[o:o + rem] for o in offs]), M)
    else:
        bins = oracle.key_to_bin(oracle.compress_many(src)).astype(np.uint64)
        cell = (ids.astype(np.uint64) << np.uint64(16)) | bins
        parts = [(cell[o:o + batch], reps) for o in offs if reps] + [(cell[o:o + rem], 1) for o in offs if rem]
        keys = np.concatenate([c for c, _ in parts])
        wts = np.concatenate([np.full(c.size, r, dtype=np.int64) for c, r in parts])
        wc, inv = np.unique(keys, return_inverse=True)
        want = (wc, np.bincount(inv, weights=wts).astype(np.int64))
        del bins, cell, keys, wts, inv

    def cells_exact(off_, keys_, counts_):
        if dense:
            return bool(np.array_equal(dense_from_csr(off_, keys_, counts_, M), want))
        rows = np.repeat(np.arange(M, dtype=np.uint64), np.diff(off_.astype(np.int64)))
        got = (rows << np.uint64(16)) | oracle.key_to_bin(keys_).astype(np.uint64)
        return bool(np.array_equal(got, want[0]) and np.array_equal(counts_.astype(np.int64), want[1]))

    ids32 = ids

    def one(form):
        eng = la.Engine(device=torch.cuda.current_device(), max_metrics=M, num_buffers=2, num_lanes=T,
                        lane_samples=1 << 21)
        if form == "copy_engine":
            eng.set_option(N.OPT_LANE_ZERO_COPY, 0)
        put = eng.submit_pairs_in_place if form.startswith("in_place") else eng.submit_pairs
        ids = ids16 if form == "in_place16" else ids32

        def work(t):
            done, off = 0, offs[t]
            while done < per:
                k = min(batch, per - done)
                put(ids[off:off + k], src[off:off + k])
                done += k

        # untimed: one batch per thread (first touch of the pinned buffers, the kernels' first launches), discarded
        wu = [threading.Thread(target=lambda t=t: put(ids[offs[t]:offs[t] + batch], src[offs[t]:offs[t] + batch]))
              for t in range(T)]
        [x.start() for x in wu]
        [x.join() for x in wu]
        eng.sync()
        eng.flip().release()
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        eng.sync()
        dt = time.perf_counter() - t0
        with eng.flip() as snap:
            cnt = int(snap.extract([0.5], M)["count"].sum())
            off_, keys_, counts_ = snap.buckets_all(M)
        eng.close()
        exact = cnt == per * T and cells_exact(off_, keys_, counts_)
        return per * T / dt, exact

    rates = {form: one(form) for form in forms}

    del want
    rate = rates["in_place16"][0]
    return {"value": rate, "unit": "samples/s", "threads": T, "samples": per * T, "names": M,
            "config": {"workload": "host arrays written into the engine's pinned staging buffers in place with uint16 ids "
                                   "(lh_reserve_pairs16 / lh_commit_pairs16: 10 B per pair), mixed ingest per half-buffer "
                                   "reading them over PCIe: the path a cgo binding uses for <= 65 536 names",
                       "id_bytes": 2},
            "roofline": {"bound": "pcie", "achieved": rate * BYTES_PAIR16 / 1e9, "peak": PCIE_GBS, "unit": "GB/s",
                         "frac": rate * BYTES_PAIR16 / 1e9 / PCIE_GBS, "bytes_per_sample": BYTES_PAIR16},
            "other_forms": {label: {"value": rates[form][0], "frac": rates[form][0] * BYTES_PAIR / 1e9 / PCIE_GBS}
                            for form, label in (("in_place", "lh_reserve_pairs / lh_commit_pairs (uint32 ids, 12 B per pair)"),
                                                ("submit_pairs", "lh_submit_pairs"),
                                                ("copy_engine", "lh_submit_pairs_through_the_copy_engine"))
                            if form in rates},
            "parity": {"rows_checked": M, "exact": all(r[1] for r in rates.values()),
                       "cells_exact_by_form": {k: r[1] for k, r in rates.items() if k in forms},
                       "checker": "oracle/ over every submitted pair (slices x repetitions), every cell of every row, "
                                  "for each of the four forms"}}


def run_c5(seconds=10.0):
    exe = os.path.join(ROOT, "loghisto_amd", "build", "c5_driver")
    if not os.path.exists(exe):
        return {"skipped": "loghisto_amd/build/c5_driver is not built"}
    r = subprocess.run([exe, "--seconds", str(seconds), "--rate", "1e8", "--bulk", "1", "--device-counters", "1"],
                       stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"failed": r.stdout[-400:]}
    j = json.loads(lines[-1])
    return {"value": j["events_per_s"], "unit": "events/s", "seconds": j["seconds"], "threads": j["threads"],
            "config": {"workload": j["workload"] + ", 1 s ProcessedMetricSet emit to a Graphite TCP sink (tools/c5_driver.cc)"},
            "target_events_per_s": j["target_events_per_s"], "dropped_intervals": j["dropped_intervals"],
            "emit_latency_ms_p50": j.get("emit_latency_ms_p50"), "device_counters": j.get("device_counters"),
            "parity": {"events_accounted": j["events_accounted"], "events_submitted": j["events_submitted"],
                       "exact": bool(j["lossless"])}}


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` started directly (no launcher in the environment): run this same command line under
    torch.distributed.run, one rank per GPU of this node, rendezvous on 127.0.0.1.  Rank 0 of the child job prints
    the JSON line; this process only relays the children's output and exit status.  LH_BENCH_LAUNCHER replaces the
    launcher module's command prefix (tests/test_bench_launch.py uses it to look at the command without GPUs)."""
    import shlex
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    launcher = os.environ.get("LH_BENCH_LAUNCHER")
    prefix = shlex.split(launcher) if launcher else [sys.executable, "-m", "torch.distributed.run"]
    cmd = prefix + ["--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1", "--master-port",
                    str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import loghisto_amd as la

    # a non-default stream: the default stream's handle is 0, which the C ABI reads as "use the engine's own
    # stream" and torch events would then not bracket the kernels
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    workload = args.workload
    if workload == "auto":
        workload = "c2"          # at any rank count: the N-rank line continues the N = 1 headline (weak scaling)
    if world > 1 and workload == "c3":
        raise SystemExit("several GPUs run the C2 headline (auto / c2) or config 4 (c4)")

    base = {"metric": METRIC, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "tree_stamp": tree_stamp()}
    comm, frontend, why = (0, "none", "")
    if world > 1:
        comm, frontend, why = make_comm(world, rank, dist)   # ONE communicator for the job: headline and secondary.c4

    def emit(res):
        res = dict(res)
        res.pop("unit", None)
        res.pop("steps", None)
        # RCCL prints its version banner through C stdio: push it out first so that the JSON line is the last one
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps({**base, **res}), flush=True)

    res, clean = run_job(args, la, stream, rank, world, dist, comm, frontend, why, workload,
                         on_headline=(lambda r: _Watchdog(r, emit)) if (world > 1 and rank == 0) else None)
    if not clean:
        # a secondary leg failed on this rank while the others may still sit in one of its collectives: no barrier, no
        # communicator teardown (either could hang) -- the headline is complete, print it and leave
        if rank == 0:
            emit(res)
        sys.stdout.flush()
        os._exit(0)
    if comm:
        from loghisto_amd import rccl
        rccl.comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(res)


class _Watchdog:
    """Several ranks, rank 0 only: the headline line is complete when the secondary config-4 leg starts; that leg is
    the first code of this repository to run real multi-rank RCCL collectives over 157 MB buffers.  If it has not come
    back after LIMIT seconds (a rank died inside a collective, a hang), print the headline with the leg marked
    "timed out" and leave, instead of losing the scaling point with it."""
    LIMIT = 300.0

    def __init__(self, headline, emit):
        self.t = threading.Timer(self.LIMIT, self.fire)
        self.headline, self.emit = headline, emit
        self.t.daemon = True
        self.t.start()

    def fire(self):
        r = dict(self.headline)
        r["secondary"] = {"c4": {"failed": f"no result after {self.LIMIT:.0f} s: abandoned so that the headline is not lost"}}
        self.emit(r)
        os._exit(0)

    def cancel(self):
        self.t.cancel()


def run_job(args, la, stream, rank, world, dist, comm, frontend, why, workload="c2", on_headline=None, ref_comm=None):
    """What main() runs between set-up and printing (also driven by tests/_bench_ranks_driver.py with ranks as threads):
    the headline, then the secondary legs.  Returns (result, clean); clean is False when a secondary leg raised on this
    rank of a multi-rank job.  on_headline(result) is called when the headline is complete and returns an object whose
    cancel() is called when the secondary legs are back.  ref_comm: a one-rank communicator for config 4's one-rank
    reference (tests: from the stub RCCL); None makes a real one."""
    clean = True

    def c4_with_reference():
        """config 4 on these ranks, with the same slice on this rank's GPU alone as its own one-rank reference (the
        curve's N = 1 line is the C2 headline -- a different workload from this one)."""
        saved = args.no_parity
        args.no_parity = True
        try:
            one = run_c4(args, la, stream, 0, 1, None, steps=5, warmup=2, comm_override=ref_comm)
            ref = {"value": one["value"], "ms_per_step": one["ms_per_step"], "ranks": 1,
                   "note": "same C4 slice on this rank's GPU alone, measured in this process before the N-rank steps"}
        except Exception as exc:  # noqa: BLE001
            ref = {"failed": repr(exc)[:300]}
        args.no_parity = saved
        dist.barrier()
        r4 = run_c4(args, la, stream, rank, world, dist, 5, 2, comm_override=comm, frontend=frontend, why=why)
        r4["value_per_gpu"] = r4["value"] / world
        r4["one_rank_reference"] = ref
        if ref.get("value"):
            r4["efficiency_vs_one_rank"] = r4["value"] / (world * ref["value"])
        return r4

    if workload == "c2":
        res = run_c2(args, la, stream, rank, world, dist, comm, frontend, why)
        if not args.no_secondary:
            sec = {}
            if world == 1:
                legs = (("c3", lambda: run_c3(args, la, stream, rank, steps=5, warmup=2, latency_flips=200)),
                        ("c4_one_rank", lambda: run_c4(args, la, stream, 0, 1, None, steps=5, warmup=2)),
                        ("c4_one_rank_1e9", lambda: run_c4_1e9(args, la, stream)),
                        ("hostfed_pairs", lambda: run_hostfed(la)),
                        # config 4's name count from the host (VERDICT r4 missing #4): a lane-sized launch over 65 536
                        # names is kernel-bound, not link-bound -- reported so that the driver sees it
                        ("hostfed_pairs_65536", lambda: run_hostfed(la, M=65536, total=int(4e8), forms=("in_place16", "in_place"))),
                        ("c5", run_c5))
            else:
                legs = (("c4", c4_with_reference),)
            guard = on_headline(res) if on_headline else None
            for name, fn in legs:
                try:
                    sec[name] = fn()
                except Exception as exc:  # noqa: BLE001 -- a secondary leg must not take the headline down
                    sec[name] = {"failed": repr(exc)[:300]}
                    if world > 1:
                        clean = False   # the other ranks may sit in a collective this rank left: see main()
                        break
            if guard:
                guard.cancel()
            res["secondary"] = sec
            # BASELINE's metric, part 2, at config 4's name count, where the headline's reader finds it (the C2 line's own
            # extract_latency_us is one name's)
            l4 = (sec.get("c4_one_rank") or {}).get("extract_latency_us")
            if l4:
                res["extract_latency_us_65536_names"] = {k: l4.get(k) for k in ("p50", "p99", "flips", "api")}
            c4 = sec.get("c4") or {}
            if world > 1 and c4.get("value"):
                # BOTH N-rank workloads at the top level, under names of their own: `value` continues the N = 1
                # headline (weak scaling of the single metric; its only collective is the all-reduce of one row), value_c4
                # is BASELINE config 4 on the same ranks -- 65 536 names, every rank ingests its slice over all of them,
                # reduce-scatter of the per-row windows, extract of the owned names: the exchange the north star names
                res.update(metric_c4="(uint32 id, float64 value) pairs/sec bucketed over 65536 names on N GPUs incl. the "
                                     "reduce-scatter merge and the extract of the owned names (BASELINE config 4)",
                           value_c4=c4["value"], value_c4_per_gpu=c4["value"] / world, ms_per_step_c4=c4["ms_per_step"],
                           ms_per_step_c4_serial=c4.get("serial_ms_per_step"),
                           c4_efficiency_vs_one_rank=c4.get("efficiency_vs_one_rank"))
    elif workload == "c3":
        res = run_c3(args, la, stream, rank, args.steps, args.warmup, latency_flips=min(args.latency_flips, 200))
    elif world > 1:
        res = c4_with_reference()
    else:
        res = run_c4(args, la, stream, rank, world, dist, args.steps, args.warmup)
        res["value_per_gpu"] = res["value"]
    return res, clean


if __name__ == "__main__":
    main()
