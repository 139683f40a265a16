"""Minimal ctypes access to RCCL for the C-ABI merge (lh_snapshot_merge): communicator set-up only.

A cgo caller links RCCL itself; Python callers (bench.py, tests) use this.  The library is the one torch
already loaded (torch/lib/librccl.so), so that there is ONE RCCL and ONE HIP runtime in the process; its path
is also handed to lh_set_rccl_library so that liblhgpu.so resolves ncclAllReduce / ncclReduceScatter from the
same object.  Nothing here touches bucket data."""
from __future__ import annotations

import ctypes as C
import os

from . import _native as N

_lib = None
_path = None


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def library_path() -> str:
    import torch
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else "/opt/rocm/lib/librccl.so"


def lib():
    global _lib, _path
    if _lib is None:
        _path = library_path()
        rc = N.lib().lh_set_rccl_library(_path.encode())
        if rc not in (N.OK, N.ESTATE):          # ESTATE: already resolved (same process, earlier call)
            N.check(rc, "lh_set_rccl_library")
        L = C.CDLL(_path)
        L.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
        L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        L.ncclCommDestroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def unique_id() -> bytes:
    uid = UniqueId()
    rc = lib().ncclGetUniqueId(C.byref(uid))
    if rc != 0:
        raise RuntimeError(f"ncclGetUniqueId failed: {rc}")
    return bytes(C.string_at(C.byref(uid), 128))


def comm_init_rank(nranks: int, uid: bytes, rank: int) -> int:
    """ncclCommInitRank on the CURRENT HIP device; returns the ncclComm_t as an int."""
    u = UniqueId()
    C.memmove(C.byref(u), uid, 128)
    comm = C.c_void_p(0)
    rc = lib().ncclCommInitRank(C.byref(comm), nranks, u, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank failed: {rc}")
    return int(comm.value)


def comm_destroy(comm: int):
    lib().ncclCommDestroy(C.c_void_p(comm))
