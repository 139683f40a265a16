"""loghisto_amd -- MI355X (gfx950) engine for loghisto's histogram hot path.

compress -> atomic bucket fan-in -> percentile/sum/count reduction
(/root/reference/metrics.go:273-295, 316-332, 336-418), executed by hand-written
HIP kernels behind a C ABI (include/loghisto_gpu.h).  This package is the thin
host-side mirror of that ABI; it contains no CPU compute path and raises if
liblhgpu.so is not built.
"""
from ._native import (LIB_PATH, MAX_PERCENTILES, NKEYS, NTHRESH, LhError, NativeLibraryError)
from .engine import Engine, Snapshot

__all__ = ["Engine", "Snapshot", "LhError", "NativeLibraryError", "LIB_PATH", "NKEYS", "NTHRESH",
           "MAX_PERCENTILES"]
