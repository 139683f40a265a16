"""In-tree build of liblhgpu.so (hipcc, gfx950 only).

`python -m loghisto_amd.build` or `__graft_entry__.build()`.  The shared object is
written next to this file so that it travels to the GPU box with the repository
snapshot; it is git-ignored.  hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB = os.path.join(_HERE, "liblhgpu.so")

# -ffp-contract=off: lh_codec.h restates Go's math.Log/Exp operation by operation.
_COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra",
           "-Wno-unused-parameter"]
_UNITS = [
    ("lh_kernels.hip", ["--offload-arch=gfx950"]),
    ("lh_kernels_part.hip", ["--offload-arch=gfx950"]),
    ("lh_kernels_small.hip", ["--offload-arch=gfx950"]),
    ("lh_kernels_fmt.hip", ["--offload-arch=gfx950"]),
    ("lh_tools.hip", ["--offload-arch=gfx950"]),   # measurement helpers (loghisto_gpu_tuning.h): read ceiling, hipMalloc'ed inputs
    ("lh_engine.cc", []),
    ("lh_dispatch.cc", []),         # the mixed ingest's path choice: pure functions (tests/test_dispatch.py)
    ("host/metric_system.cc", []),   # C++ host layer with the reference's MetricSystem API (include/loghisto.hpp)
]

# standalone C++ programs linked against liblhgpu.so: (source relative to the repo root, output name)
_PROGRAMS = [
    ("tests/cpp/metrics_test.cc", "metrics_test"),   # metrics_test.go restated
    ("tools/c5_driver.cc", "c5_driver"),             # BASELINE config 5 driver
    ("tools/wire_bench.cc", "wire_bench"),           # per-key vs bulk (K6) ProcessedMetricSet serialization
    ("tools/latency.cc", "latency"),                 # flip -> extract latency at the C ABI
    ("tools/hostfed_native.cc", "hostfed_native"),   # host-fed pairs from native threads through the in-place staging API
]


# standalone HIP measurement tools (no library): (source relative to the repo root, output name)
_HIP_PROGRAMS = [
    ("tools/read_ceiling.hip", "read_ceiling"),      # what a kernel that only reads gets from HBM (K1's ceiling)
    ("tools/valu_rates.hip", "valu_rates"),          # issue cost of the bucket index's VALU instructions
    ("tools/row_stride.hip", "row_stride"),          # the 65 536 rows' windows read / cleared / flushed at several row strides
]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found; liblhgpu.so cannot be built")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    bdir = os.path.join(_HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, "loghisto_gpu.h"), os.path.join(INCLUDE, "loghisto.hpp")]
    objs = []
    for src, extra in _UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, os.path.splitext(src)[0].replace("/", "_") + ".o")
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + _COMMON + extra + os.environ.get("LH_EXTRA_CXXFLAGS", "").split() + ["-I", INCLUDE, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    root = os.path.dirname(_HERE)
    for src, name in _PROGRAMS:
        s = os.path.join(root, src)
        if not os.path.exists(s):
            continue
        out = os.path.join(bdir, name)
        if force or _stale(out, [s, LIB] + headers):
            cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-I", INCLUDE, s, "-o", out,
                   "-L", _HERE, "-llhgpu", "-Wl,-rpath," + _HERE, "-Wl,-rpath,$ORIGIN/.."]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    # the path choice's pure functions, table-tested on the CPU box (tests/test_dispatch.py): needs the HIP headers for
    # the launch interface's types only
    dsrc = os.path.join(root, "tests", "cpp", "dispatch_test.cc")
    if os.path.exists(dsrc):
        out = os.path.join(bdir, "dispatch_test")
        if force or _stale(out, [dsrc, LIB, os.path.join(CSRC, "lh_dispatch.cc")] + headers):
            rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
            cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
                   dsrc, "-o", out, "-L", _HERE, "-llhgpu", "-Wl,-rpath," + _HERE, "-Wl,-rpath,$ORIGIN/..",
                   "-L", os.path.join(rocm, "lib"), "-Wl,-rpath," + os.path.join(rocm, "lib")]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    for src, name in _HIP_PROGRAMS:
        s = os.path.join(root, src)
        if not os.path.exists(s):
            continue
        out = os.path.join(bdir, name)
        if force or _stale(out, [s]):
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", s, "-o", out]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    # test infrastructure: in-process stand-in for librccl.so (tests/cpp/rccl_stub.cc) so that N ranks of the
    # C-ABI merge can run on the one reachable GPU
    stub_src = os.path.join(root, "tests", "cpp", "rccl_stub.cc")
    if os.path.exists(stub_src):
        out = os.path.join(bdir, "librccl_stub.so")
        if force or _stale(out, [stub_src]):
            rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
            cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall", "-D__HIP_PLATFORM_AMD__",
                   "-I", os.path.join(rocm, "include"), stub_src, "-o", out, "-L", os.path.join(rocm, "lib"),
                   "-lamdhip64"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
