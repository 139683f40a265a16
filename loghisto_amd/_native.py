"""ctypes binding of liblhgpu.so (include/loghisto_gpu.h).

There is no fallback: if the shared object is missing or does not export the ABI
this module raises.  All bucket arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblhgpu.so")

ABI_VERSION = 7
# tools/sweep.py --lib (A/B runs against a build of an earlier round): accept a library of an older ABI and bind
# what it exports; never set by the product or the tests
ALLOW_OLDER_ABI = False
NKEYS = 65536
NTHRESH = 70980
MAX_PERCENTILES = 32

OK, EINVAL, ENOMEM, EDEVICE, ENODEVICE, EBUSY, ERANGE, ESTATE = range(8)


class LhConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("max_metrics", C.c_uint32),
                ("num_buffers", C.c_uint32), ("num_lanes", C.c_uint32), ("max_counters", C.c_uint32),
                ("lane_samples", C.c_uint64), ("cell_bits", C.c_uint32), ("reserved0", C.c_uint32)]


class LhCounters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("samples_single", "samples_small", "samples_partitioned", "samples_direct",
                                           "launches", "flips", "flips_busy", "extracts", "backpressure_waits",
                                           "window_misses")] + [("small_path_disabled", C.c_uint32),
                                                                ("regions_disabled", C.c_uint32),
                                                                ("scratch_bytes", C.c_uint64),
                                                                ("sublaunches", C.c_uint64),
                                                                ("samples_partitioned_v2", C.c_uint64),
                                                                ("counter_events", C.c_uint64),
                                                                ("region_overflows", C.c_uint64),
                                                                ("samples_partitioned_v3", C.c_uint64),
                                                                ("window_log2", C.c_uint64),
                                                                ("records_level1", C.c_uint64),
                                                                ("records_level2", C.c_uint64),
                                                                ("level2_overflows", C.c_uint64),
                                                                ("reduce_window_misses", C.c_uint64),
                                                               ("surveys_reused", C.c_uint64),
                                                               ("scratch_alloc_failures", C.c_uint64),
                                                               ("samples_fallback", C.c_uint64),
                                                               ("survey_stale_pairs", C.c_uint64),
                                                               ("lane_scratch_bytes", C.c_uint64),
                                                               ("widenings", C.c_uint64),
                                                               ("store_bytes", C.c_uint64)]

# lh_set_option keys (include/loghisto_gpu.h; the path-steering ones and the fault hook: include/loghisto_gpu_tuning.h)
OPT_TWO_LEVEL_ABOVE, OPT_HOT_MIN_TILES, OPT_HOT_WINDOWS, OPT_NAMES_PER_PARTITION = 1, 2, 3, 4
OPT_EXTRACT_ZERO_COPY, OPT_SCRATCH_CAP_BYTES, OPT_SUBLAUNCH_PAIRS, OPT_SMALL_PATH = 5, 6, 7, 8
OPT_PART_V2, OPT_PART_V2_MIN_PAIRS, OPT_PART_V2_SHAPE = 9, 10, 11
OPT_PART_V3, OPT_PART_V3_MIN_PAIRS, OPT_PART_V3_LOG_W = 12, 13, 14
OPT_LANE_ZERO_COPY = 15
OPT_SURVEY_EVERY = 16
OPT_PART_MIN_PAIRS = 17
OPT_LANE_SCRATCH_BLOCKS = 18
OPT_FAIL_SCRATCH_ALLOCS = 19
OPT_LANE_GEN3 = 20
OPT_PART_V3_DIRECT_MAX_PAIRS = 21
OPT_MERGE_NARROW_CELLS = 22
OPT_WIDEN_AT_SAMPLES = 23


class LhDispatchQuery(C.Structure):
    """lh_dispatch_query (include/loghisto_gpu_tuning.h): the state the mixed ingest's path choice reads."""
    _fields_ = [("struct_size", C.c_uint32), ("max_metrics", C.c_uint32), ("n", C.c_uint64), ("ids_addr", C.c_uint64),
                ("vals_addr", C.c_uint64), ("id_width", C.c_uint32), ("host_fed", C.c_uint32), ("num_cus", C.c_uint32),
                ("lane_blocks", C.c_uint32), ("lane_samples", C.c_uint64), ("small_disabled", C.c_uint32),
                ("regions_disabled", C.c_uint32), ("v3_disabled", C.c_uint32), ("call_log_w", C.c_uint32),
                ("scratch_cap", C.c_uint64), ("sublaunch_pairs", C.c_uint64), ("part_min_pairs", C.c_uint64),
                ("v2_min_pairs", C.c_uint64), ("v3_min_pairs", C.c_uint64), ("v2_off", C.c_uint32), ("v3_off", C.c_uint32),
                ("hot_off", C.c_uint32), ("v2_shape_set", C.c_uint32), ("v2_shape", C.c_uint32), ("fail_allocs", C.c_uint32),
                ("lane_gen3_off", C.c_uint32)]


class LhDispatchStep(C.Structure):
    _fields_ = [("path", C.c_uint32), ("lane_block", C.c_uint32), ("take", C.c_uint64), ("scratch", C.c_uint64),
                ("fell_back", C.c_uint32), ("peeled", C.c_uint32)]


PATH_DIRECT, PATH_SMALL, PATH_GEN1, PATH_GEN2, PATH_GEN3 = range(5)


class LhExtractView(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("pvals", C.c_void_p), ("pkeys", C.c_void_p), ("pvalid", C.c_void_p),
                ("nmetrics", C.c_size_t), ("np", C.c_size_t)]


class LhExtractCompact(C.Structure):
    _fields_ = [("count", C.c_void_p), ("sum", C.c_void_p), ("nbuckets", C.c_void_p), ("pvalid_bits", C.c_void_p),
                ("pkeys", C.c_void_p), ("nmetrics", C.c_size_t), ("np", C.c_size_t)]


class LhMergeInfo(C.Structure):
    _fields_ = [("packed_cells", C.c_uint64), ("send_bytes", C.c_uint64), ("recv_bytes", C.c_uint64),
                ("widest_row", C.c_uint32), ("occupied_rows", C.c_uint32), ("padded_words", C.c_uint64),
                ("cell_bytes", C.c_uint32), ("rows_8bit", C.c_uint32), ("ranges_ms", C.c_float), ("plan_ms", C.c_float),
                ("pack_ms", C.c_float), ("collective_ms", C.c_float), ("unpack_ms", C.c_float), ("span_ms", C.c_float),
                ("packed_words", C.c_uint64), ("rows_16bit", C.c_uint32), ("reserved", C.c_uint32)]


class LhLineFormat(C.Structure):
    _fields_ = [("prefix", C.c_char_p), ("sep", C.c_char_p), ("suffix", C.c_char_p),
                ("flags", C.c_uint32), ("reserved", C.c_uint32)]


FMT_UNDERSCORE_TO_DOT = 1
SER_AGGREGATES = 1
FMT_SLOT = 336


class LhStats(C.Structure):
    _fields_ = [("count", C.c_uint64), ("sum", C.c_double), ("avg", C.c_double),
                ("agg_sum_add", C.c_uint64), ("nbuckets", C.c_uint32), ("present", C.c_uint32)]


_vp, _sz = C.c_void_p, C.c_size_t
_dp, _u32p, _u64p = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_i16p, _u8p = C.POINTER(C.c_int16), C.POINTER(C.c_uint8)

# name -> (restype, argtypes): the test / tuning hooks of include/loghisto_gpu_tuning.h
TUNING_SIGNATURES = {
    "lh_dispatch_probe": (C.c_int, [C.POINTER(LhDispatchQuery), C.POINTER(LhDispatchStep), _sz, C.POINTER(_sz)]),
    "lh_tool_device_alloc": (C.c_int, [_sz, C.POINTER(_vp)]),
    "lh_tool_device_free": (C.c_int, [_vp]),
    "lh_tool_read_ceiling": (C.c_int, [_vp, _sz, C.c_int, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "lh_tool_last_extract_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

# name -> (restype, argtypes): every symbol include/loghisto_gpu.h declares.
SIGNATURES = {
    "lh_abi_version": (C.c_int, []),
    "lh_strerror": (C.c_char_p, [C.c_int]),
    "lh_last_error": (C.c_char_p, []),
    "lh_default_config": (C.c_int, [C.POINTER(LhConfig)]),
    "lh_create": (C.c_int, [C.POINTER(LhConfig), C.POINTER(_vp)]),
    "lh_destroy": (C.c_int, [_vp]),
    "lh_intern": (C.c_int, [_vp, C.c_char_p, _sz, _u32p]),
    "lh_lookup": (C.c_int, [_vp, C.c_char_p, _sz, _u32p]),
    "lh_num_metrics": (C.c_int, [_vp, _u32p]),
    "lh_metric_name": (C.c_int, [_vp, C.c_uint32, C.c_char_p, _sz, C.POINTER(_sz)]),
    "lh_submit": (C.c_int, [_vp, C.c_uint32, _vp, _sz]),
    "lh_submit_pairs": (C.c_int, [_vp, _vp, _vp, _sz]),
    "lh_reserve_pairs": (C.c_int, [_vp, _sz, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint32)]),
    "lh_commit_pairs": (C.c_int, [_vp, C.c_uint32, _sz]),
    "lh_submit_pairs16": (C.c_int, [_vp, _vp, _vp, _sz]),
    "lh_reserve_pairs16": (C.c_int, [_vp, _sz, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                     C.POINTER(C.c_uint32)]),
    "lh_commit_pairs16": (C.c_int, [_vp, C.c_uint32, _sz]),
    "lh_submit_pairs16_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "lh_submit_device": (C.c_int, [_vp, C.c_uint32, _vp, _sz, _vp]),
    "lh_submit_pairs_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "lh_intern_counter": (C.c_int, [_vp, C.c_char_p, _sz, _u32p]),
    "lh_num_counters": (C.c_int, [_vp, _u32p]),
    "lh_counter_name": (C.c_int, [_vp, C.c_uint32, C.c_char_p, _sz, C.POINTER(_sz)]),
    "lh_submit_counts": (C.c_int, [_vp, _vp, _vp, _sz]),
    "lh_submit_counts_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "lh_counters_collect": (C.c_int, [_vp, C.c_uint32, _sz, _u64p, _u8p, _u64p, _u8p]),
    "lh_serialize_counters": (C.c_int, [_vp, C.c_uint32, _sz, C.POINTER(LhLineFormat), _vp, _sz, C.POINTER(_sz)]),
    "lh_flush": (C.c_int, [_vp]),
    "lh_sync": (C.c_int, [_vp]),
    "lh_flip": (C.c_int, [_vp, C.POINTER(_vp)]),
    "lh_extract": (C.c_int, [_vp, _dp, _sz, C.POINTER(LhStats), _dp, _i16p, _u8p, _sz]),
    "lh_extract_rows": (C.c_int, [_vp, C.c_uint32, _sz, _dp, _sz, C.POINTER(LhStats), _dp, _i16p, _u8p]),
    "lh_extract_rows_view": (C.c_int, [_vp, C.c_uint32, _sz, _dp, _sz, C.POINTER(LhExtractView)]),
    "lh_extract_rows_compact": (C.c_int, [_vp, C.c_uint32, _sz, _dp, _sz, C.POINTER(LhExtractCompact)]),
    "lh_expand_compact": (C.c_int, [_vp, C.POINTER(LhExtractCompact), C.POINTER(LhStats), _dp, _i16p, _u8p]),
    "lh_buckets": (C.c_int, [_vp, C.c_uint32, _i16p, _u64p, _sz, C.POINTER(_sz)]),
    "lh_buckets_all": (C.c_int, [_vp, C.c_uint32, _sz, _u64p, _i16p, _u64p, _sz, C.POINTER(_sz)]),
    "lh_snapshot_rows": (C.c_int, [_vp, C.POINTER(_vp), _u32p]),
    "lh_row_stride": (C.c_size_t, []),
    "lh_snapshot_cells": (C.c_int, [_vp, C.POINTER(_vp), _u32p, _u32p]),
    "lh_cell_bytes": (C.c_int, [_vp]),
    "lh_snapshot_ranges": (C.c_int, [_vp, C.POINTER(_vp)]),
    "lh_snapshot_mark_dirty": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "lh_snapshot_merge": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint32, _u32p, _u32p]),
    "lh_snapshot_merge_info": (C.c_int, [_vp, C.POINTER(LhMergeInfo)]),
    "lh_set_rccl_library": (C.c_int, [C.c_char_p]),
    "lh_serialize": (C.c_int, [_vp, C.c_uint32, _sz, _dp, C.POINTER(C.c_char_p), _sz, C.POINTER(LhLineFormat),
                               C.c_uint32, _vp, _sz, C.POINTER(_sz)]),
    "lh_snapshot_accumulate": (C.c_int, [_vp]),
    "lh_lifetime": (C.c_int, [_vp, C.c_uint32, _sz, _u64p, _u64p]),
    "lh_format_f": (C.c_int, [_vp, _dp, _sz, _vp, _sz, _u32p]),
    "lh_snapshot_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "lh_release": (C.c_int, [_vp]),
    "lh_get_counters": (C.c_int, [_vp, C.POINTER(LhCounters)]),
    "lh_set_option": (C.c_int, [_vp, C.c_int, C.c_uint64]),
    "lh_compress_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "lh_compress_device_golog": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "lh_codec_tables": (C.c_int, [_vp, _dp, _dp]),
    "lh_selftest_vlog": (C.c_int, [_vp, _dp]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


class LhError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        msg = f"{where}: {lib().lh_strerror(code).decode()} (code {code})"
        if detail:
            msg += f" [{detail}]"
        super().__init__(msg)


def lib():
    """Load liblhgpu.so; raises NativeLibraryError if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m loghisto_amd.build` "
            "(hipcc, gfx950). loghisto_amd has no CPU path.")
    # One HIP runtime per process: torch bundles its own libamdhip64, and streams /
    # device pointers are shared between torch (plumbing) and this library, so
    # torch's copy must be the one liblhgpu.so binds to.  Import torch first.
    import torch  # noqa: F401
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 not found
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in list(SIGNATURES.items()) + list(TUNING_SIGNATURES.items()):
        try:
            fn = getattr(L, name)
        except AttributeError as exc:
            if ALLOW_OLDER_ABI:
                continue
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    if L.lh_abi_version() != ABI_VERSION and not (ALLOW_OLDER_ABI and L.lh_abi_version() >= 5):
        raise NativeLibraryError(f"ABI mismatch: library {L.lh_abi_version()}, binding {ABI_VERSION}")
    _lib = L
    return L


def check(code: int, where: str):
    if code != OK:
        detail = lib().lh_last_error().decode() if code in (EDEVICE, ENODEVICE, ENOMEM) else ""
        raise LhError(code, where, detail)
