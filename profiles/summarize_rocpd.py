#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into the small text/JSON
files committed under profiles/.

  python profiles/summarize_rocpd.py stats <trace.db> [--min-ns N]      per-kernel stats table
  python profiles/summarize_rocpd.py pmc   <pmc.db> <kernel-substr>     per-launch counter values
  python profiles/summarize_rocpd.py list  <trace.db> <kernel-substr> [--min-ns N] [--skip K]
                                                                         every launch in start order, stats without the first K

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced streaming reads (MI355X_MICROARCH.md "HBM"), so the read
bytes of such a kernel are FETCH_SIZE * 1024 * 2; the correction is applied here and stated in
the output.
"""
import json
import sqlite3
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    return name if len(name) <= 70 else name[:67] + "..."


def stats(db_path: str, min_ns: int = 0):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, lds, vg, sg in rows:
        if dur < min_ns:
            continue
        a = agg.setdefault(short(name), dict(calls=0, total=0, mn=1 << 62, mx=0, grid=gx, wg=wx, lds=lds, vgpr=vg, sgpr=sg))
        a["calls"] += 1
        a["total"] += dur
        a["mn"] = min(a["mn"], dur)
        a["mx"] = max(a["mx"], dur)
    tot = sum(a["total"] for a in agg.values()) or 1
    out = [f"# kernel stats from {db_path} (durations in us; min duration filter {min_ns} ns)",
           f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'grid':>8s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'sgpr':>5s}"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["total"]):
        out.append(f"{k:70s} {a['calls']:6d} {a['total'] / 1e3:12.1f} {a['total'] / a['calls'] / 1e3:10.2f} "
                   f"{a['mn'] / 1e3:10.2f} {a['mx'] / 1e3:10.2f} {100 * a['total'] / tot:6.2f} {a['grid']:8d} {a['wg']:5d} "
                   f"{a['lds']:7d} {a['vgpr']:5d} {a['sgpr']:5d}")
    return "\n".join(out)


def pmc(db_path: str, kernel_substr: str, min_ns: int = 0):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    vals = {}
    for name, cname, value, dur in rows:
        if kernel_substr in name and dur >= min_ns:
            vals.setdefault(cname, []).append((value, dur))
    res = {"db": db_path, "kernel": kernel_substr, "min_ns": min_ns, "counters": {}}
    for cname, lst in vals.items():
        v = [x[0] for x in lst]
        d = [x[1] for x in lst]
        res["counters"][cname] = {"launches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v),
                                  "avg_duration_us_profiled": sum(d) / len(d) / 1e3}
    return res


def launches(db_path: str, kernel_substr: str, min_ns: int = 0, skip: int = 0):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, start, duration from kernels order by start").fetchall()
    d = [dur for name, _, dur in rows if kernel_substr in name and dur >= min_ns]
    out = [f"# launches of {kernel_substr} in {db_path}, in start order (us; min duration filter {min_ns} ns): {len(d)}",
           " ".join(f"{x / 1e3:.1f}" for x in d)]
    for label, sel in (("all", d), (f"without the first {skip}", d[skip:])):
        if sel:
            out.append(f"{label}: n {len(sel)} avg {sum(sel) / len(sel) / 1e3:.2f} min {min(sel) / 1e3:.2f} "
                       f"max {max(sel) / 1e3:.2f} median {sorted(sel)[len(sel) // 2] / 1e3:.2f}")
    return "\n".join(out)


if __name__ == "__main__":
    mode = sys.argv[1]
    min_ns = 0
    if "--min-ns" in sys.argv:
        i = sys.argv.index("--min-ns")
        min_ns = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    skip = 0
    if "--skip" in sys.argv:
        i = sys.argv.index("--skip")
        skip = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    if mode == "stats":
        print(stats(sys.argv[2], min_ns))
    elif mode == "list":
        print(launches(sys.argv[2], sys.argv[3], min_ns, skip))
    else:
        print(json.dumps(pmc(sys.argv[2], sys.argv[3], min_ns), indent=1))
