#!/usr/bin/env python3
"""bench.py -- loghisto hot path on MI355X: float64 samples/s bucketed.

A "step" is one pass of the hot path over one resident batch: ingest kernel over
n float64 samples of one metric (K1), epoch flip, [N>1: RCCL merge of the bucket
row], percentile/sum/count scan (K2), results on the host, buffer recycled (K3).
Workload at N=1 = BASELINE.json configs[1]: "Single-metric 1B float64 samples,
1xMI355X, one ingest kernel + one percentile scan" (SURVEY.md 8d "C2":
lognormal(mu=ln 1e5, sigma=1), seeded, generated on device).

N>1: one process per GPU (torch.distributed / RCCL), weak scaling: every rank
ingests its own n samples, the only exchange is the all-reduce of the uint64
bucket row at the flip.  value = N*n*K / max-over-ranks wall time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PCTS = [0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0]   # metrics.go:145-155
HBM_PEAK_GBS = 8000.0                                    # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_SAMPLE = 8                                     # one float64 read


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=float, default=1e9, help="float64 samples per GPU per step")
    ap.add_argument("--dist", default="lognormal", choices=["lognormal", "constant", "uniform", "exponential",
                                                            "normal", "loguniform", "lognormal25"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-flips", type=int, default=1000)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3"],
                    help="c2 (default, the headline): one metric; c3: 1024 Zipf names, (id, value) stream")
    ap.add_argument("--names", type=int, default=1024, help="histogram names for --workload c3")
    return ap.parse_args()


def make_samples(n: int, kind: str, seed: int) -> torch.Tensor:
    """Seeded synthetic stream, generated on device."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    if kind == "constant":
        return torch.full((n,), 123.0, dtype=torch.float64, device="cuda")
    if kind in ("uniform", "loguniform", "exponential"):
        v = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
        if kind == "uniform":
            return v.mul_(1e9)
        if kind == "loguniform":
            return v.mul_(21.0).sub_(3.0).mul_(math.log(10.0)).exp_()
        return v.neg_().log1p_().neg_().mul_(1e6)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    if kind == "normal":
        return v.mul_(1e3)
    sigma = 2.5 if kind == "lognormal25" else 1.0
    return v.mul_(sigma).add_(math.log(1e5)).exp_()


def effective_cores() -> int:
    """Cores this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota).
    (The MI355X box reports 256 CPUs but its cgroup grants 16: threads beyond that only time-slice.)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]       # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())             # cgroup v1
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, int(math.ceil(q / p))))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(samples: torch.Tensor, target_s: float):
    """Oracle timed on this box's host cores (bounded sample of the same workload)."""
    import oracle
    cores = effective_cores()
    probe = samples[: 1 << 21].cpu().numpy()
    t1, _ = oracle.bench_dense(probe, 1)
    rate1 = probe.size / max(t1, 1e-9)                       # one core
    n = int(min(samples.numel(), 1 << 27))                   # 1 GiB of host samples
    host = samples[:n].cpu().numpy()
    # every thread passes `reps` times over its slice so that thread start-up and the final merge do not
    # dominate on a many-core host: about target_s seconds of wall time if the cores scaled perfectly / 4
    reps = int(max(1, min(256, target_s * rate1 * cores / 2 / n)))
    t, counts = oracle.bench_dense_reps(host, cores, reps)
    assert int(counts.sum()) == n * reps and not (counts % reps).any()
    counts = counts // reps
    n_timed = n * reps
    # form A (the reference's cost shape: lock + 4 map probes + atomic per sample), small sample
    na = int(min(n, 1 << 22))
    ta, _ = oracle.bench_faithful(host[:na], cores)
    ta1, _ = oracle.bench_faithful(host[: na // 4], 1)
    return {
        "value": n_timed / t, "unit": "samples/s", "cores": cores, "kind": "port",
        "sample": f"first {n} samples of the step's stream, {reps} passes per thread ({t:.2f} s wall); C oracle "
                  f"(Go math.Log restated), per-thread dense uint64[65536] rows + merge (BASELINE.md form B), "
                  f"{cores} threads = the cores this process is granted (host reports {os.cpu_count()} CPUs); "
                  f"one thread alone: {rate1:.3g} samples/s",
        "faithful_form": {"value": na / ta, "cores": cores, "value_1thread": (na // 4) / ta1,
                          "sample": f"{na} samples; shared lock + map[name][int16] + atomic per call "
                                    "(cost shape of metrics.go:273-295, BASELINE.md form A)"},
    }, counts, n


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import loghisto_amd
    from loghisto_amd import merge

    n = int(args.samples)
    c3 = args.workload == "c3"
    M = args.names if c3 else 1
    bytes_per_sample = 12 if c3 else BYTES_PER_SAMPLE       # float64 + uint32 id for mixed streams
    eng = loghisto_amd.Engine(device=local_rank, max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
    data = make_samples(n, args.dist, seed=(3 if c3 else 2) + rank)
    ids = None
    if c3:  # id ~ Zipf(1.0) over ranks, value ~ lognormal(ln 1e5 + 0.002*id, 1)
        g = torch.Generator(device="cuda")
        g.manual_seed(1003 + rank)
        w = 1.0 / torch.arange(1, M + 1, dtype=torch.float64, device="cuda")
        ids = torch.multinomial(w / w.sum(), n, replacement=True, generator=g).to(torch.int32)
        data.mul_(torch.exp(0.002 * ids.to(torch.float64)))
    torch.cuda.synchronize()
    # a non-default stream: the default stream's handle is 0, which the C ABI reads as
    # "use the engine's own stream" and torch events would then not bracket the kernel
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    k1_events = []

    def step(timed: bool):
        if timed:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
        if c3:
            eng.submit_pairs_device(ids, data, n, stream=stream)   # P1 + plan + P2 on torch's current stream
        else:
            eng.submit_device(0, data, n, stream=stream)       # K1 on torch's current stream
        if timed:
            b.record(stream)
            k1_events.append((a, b))
        snap = eng.flip()
        if world > 1:
            merge.merge_snapshot(snap, M, plan="allreduce")
        out = snap.extract(PCTS, M)                            # K2 + D2H + sync
        snap.release()                                         # K3 (async)
        return out

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    assert int(out["count"].sum()) == n * world, (int(out["count"].sum()), n, world)
    k1_ms = [a.elapsed_time(b) for a, b in k1_events]
    k1_avg_ms = sum(k1_ms) / len(k1_ms)
    achieved = n * bytes_per_sample / (k1_avg_ms * 1e-3) / 1e9

    # p99 extract latency: flip -> stats on host (BASELINE.json metric, part 2)
    lat = []
    small = data[: 1 << 20]
    for _ in range(args.latency_flips):
        eng.submit_device(0, small, small.numel(), stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        snap = eng.flip()
        snap.extract(PCTS, M)
        lat.append(time.perf_counter() - t1)
        snap.release()
    lat_us = np.array(lat) * 1e6 if lat else np.array([float("nan")])

    result = None
    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_k1_pmc.json")
        if os.path.exists(pmc):
            try:
                # PMC passes cannot run inside this process: the committed rocprofv3 --pmc summary
                # of this same command supplies HBM bytes per K1 launch (read + write, corrected
                # as the microarch guide prescribes; see the JSON's "corrections").  It only
                # applies to the workload it was measured on.
                j = json.load(open(pmc))
                if not c3 and n * BYTES_PER_SAMPLE == int(j["algorithmic_bytes_per_launch"]) and args.dist == "lognormal":
                    traffic = j["hbm_read_bytes_per_launch"] + j["hbm_write_bytes_per_launch"]
            except Exception:
                traffic = None
        result = {
            "metric": "float64 samples/sec bucketed (1 GPU) + % HBM roofline; p99 extract latency",
            "value": world * n * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("C3 mixed (uint32 id, float64 value) stream over Zipf(1.0) names, partitioned ingest + "
                                    "one percentile scan per name per step") if c3 else
                                   "C2 single-metric float64 stream, one ingest kernel + one percentile scan per step",
                       "samples_per_gpu_per_step": n, "distribution": args.dist, "metrics": M,
                       "percentiles": PCTS, "merge": "allreduce(uint64 row) at flip" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": "k_scatter_samples+k_plan_*+k_part_hist" if c3 else "k_ingest_single",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": n * bytes_per_sample, "avg_launch_ms": k1_avg_ms,
                         "frac_of_measured_copy_ceiling_6290": achieved / 6290.0},
            "extract_latency_us": {"p50": float(np.percentile(lat_us, 50)), "p99": float(np.percentile(lat_us, 99)),
                                   "flips": len(lat)},
        }
        if world == 1 and not args.no_cpu_baseline and not c3:
            cb, cpu_counts, ncpu = cpu_baseline(data, args.cpu_seconds)
            # the same samples through the GPU path must give the same row
            eng.submit_device(0, data[:ncpu], ncpu, stream=stream)
            with eng.flip() as snap:
                gpu_row = snap.dense_row(0)
            assert np.array_equal(gpu_row, cpu_counts), "GPU row differs from the oracle on the baseline sample"
            result["cpu_baseline"] = cb
        print(json.dumps(result), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
