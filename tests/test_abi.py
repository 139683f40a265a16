"""CPU tests of the drop-in boundary: liblhgpu.so builds, loads, exports every
symbol include/loghisto_gpu.h declares, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(header="loghisto_gpu.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(native_lib):
    from loghisto_amd import _native
    declared = header_symbols()
    assert len(declared) >= 25
    raw = C.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in loghisto_gpu.h but not exported"
    assert sorted(_native.SIGNATURES) == declared, "ctypes binding and header disagree"
    # the test / tuning hooks are declared apart from the drop-in contract, and exported too
    tuning = header_symbols("loghisto_gpu_tuning.h")
    assert tuning == sorted(_native.TUNING_SIGNATURES) and not set(tuning) & set(declared)
    for name in tuning:
        assert hasattr(raw, name), f"{name} declared in loghisto_gpu_tuning.h but not exported"


def test_the_public_header_keeps_only_the_operational_options():
    """ABI 5: the keys that steer the mixed ingest's path choice (tests, tuning runs) and the allocation-failure hook are
    declared in loghisto_gpu_tuning.h; every key has one number, and no number two names."""
    def keys(header):
        src = open(os.path.join(ROOT, "include", header)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        return dict((k, int(v)) for k, v in re.findall(r"\b(LH_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", src))
    pub, tun = keys("loghisto_gpu.h"), keys("loghisto_gpu_tuning.h")
    assert sorted(pub) == ["LH_OPT_EXTRACT_ZERO_COPY", "LH_OPT_LANE_SCRATCH_BLOCKS", "LH_OPT_LANE_ZERO_COPY",
                           "LH_OPT_MERGE_NARROW_CELLS", "LH_OPT_SCRATCH_CAP_BYTES", "LH_OPT_SUBLAUNCH_PAIRS",
                           "LH_OPT_SURVEY_EVERY"]
    assert not set(pub) & set(tun) and len(set(pub.values()) | set(tun.values())) == len(pub) + len(tun)
    from loghisto_amd import _native
    for name, num in {**pub, **tun}.items():
        assert getattr(_native, name[3:]) == num, name


def test_abi_version_and_strerror(native_lib):
    assert native_lib.lh_abi_version() == 7
    msgs = {native_lib.lh_strerror(c).decode() for c in range(8)}
    assert len(msgs) == 8 and "ok" in msgs


def test_struct_layouts_match_header(native_lib):
    from loghisto_amd import _native
    assert C.sizeof(_native.LhConfig) == 40
    assert C.sizeof(_native.LhStats) == 40
    assert C.sizeof(_native.LhLineFormat) == 32 and C.sizeof(_native.LhCounters) == 232
    assert C.sizeof(_native.LhMergeInfo) == 88 and C.sizeof(_native.LhExtractView) == 48
    cfg = _native.LhConfig()
    assert native_lib.lh_default_config(C.byref(cfg)) == 0
    assert cfg.struct_size == 40 and cfg.max_metrics >= 1 and cfg.num_buffers >= 2 and cfg.cell_bits == 0


def test_argument_validation_needs_no_gpu(native_lib):
    from loghisto_amd import _native
    cfg = _native.LhConfig()
    native_lib.lh_default_config(C.byref(cfg))
    h = C.c_void_p(0)
    assert native_lib.lh_create(None, C.byref(h)) == _native.EINVAL
    cfg.num_buffers = 1
    assert native_lib.lh_create(C.byref(cfg), C.byref(h)) == _native.EINVAL
    cfg.num_buffers, cfg.cell_bits = 2, 16  # (ABI 7) 0, 32 or 64
    assert native_lib.lh_create(C.byref(cfg), C.byref(h)) == _native.EINVAL
    cfg.cell_bits, cfg.struct_size = 0, 36  # the ABI-6 struct (32 bytes) and this one are the sizes there are
    assert native_lib.lh_create(C.byref(cfg), C.byref(h)) == _native.EINVAL
    assert native_lib.lh_snapshot_cells(None, None, None, None) == _native.EINVAL and native_lib.lh_cell_bytes(None) == _native.EINVAL
    assert native_lib.lh_flush(None) == _native.EINVAL
    assert native_lib.lh_release(None) == _native.EINVAL
    # K6 entry points
    n = C.c_size_t(0)
    fmt = _native.LhLineFormat(b"put ", b" 1 ", b"\n", 0, 0)
    assert native_lib.lh_serialize(None, 0, 1, None, None, 0, C.byref(fmt), 0, None, 0, C.byref(n)) == _native.EINVAL
    assert native_lib.lh_snapshot_accumulate(None) == _native.EINVAL
    assert native_lib.lh_lifetime(None, 0, 1, None, None) == _native.EINVAL
    assert native_lib.lh_format_f(None, None, 1, None, 336, None) == _native.EINVAL
    assert native_lib.lh_set_option(None, 1, 0) == _native.EINVAL
    assert native_lib.lh_intern_counter(None, b"c", 1, None) == _native.EINVAL
    assert native_lib.lh_submit_counts(None, None, None, 1) == _native.EINVAL
    assert native_lib.lh_counters_collect(None, 0, 1, None, None, None, None) == _native.EINVAL
    assert native_lib.lh_serialize_counters(None, 0, 1, C.byref(fmt), None, 0, C.byref(n)) == _native.EINVAL


def test_product_build_never_reads_the_environment(native_lib):
    """VERDICT r1 weak #5 / ADVICE: tuning and ablation switches are compile-time (-DLH_TUNING, tools/ builds only);
    the shipped library has no getenv import and none of the old switch names."""
    import subprocess
    from loghisto_amd import _native
    blob = open(_native.LIB_PATH, "rb").read()
    for name in (b"LH_DEBUG_FLAGS", b"LH_PART_NAMES", b"LH_PART_HOT", b"LH_PART_TWO_LEVEL_ABOVE", b"LH_NO_ZERO_COPY"):
        assert name not in blob, name
    syms = subprocess.run(["nm", "-D", "--undefined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_no_cpu_fallback():
    """Without a gfx950 device the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import loghisto_amd
    with pytest.raises(loghisto_amd.LhError) as ei:
        loghisto_amd.Engine()
    assert ei.value.code == 4  # LH_ENODEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "loghisto_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cc", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "lh_oracle" not in text and "lho_" not in text, f
