"""The mixed ingest's path choice as a pure function (loghisto_amd/csrc/lh_dispatch.h): table-tested
on the CPU box -- no device is touched.

* tests/cpp/dispatch_test.cc (built against liblhgpu.so) holds the decision functions themselves: survey reuse for
  the second and third generation over every combination of its conditions, the skew-free-names switch, the peeled
  first sample, and choose_step's invariants over 12 768 engine states x launch sizes;
* lh_dispatch_probe (include/loghisto_gpu_tuning.h) answers "what would an engine in this state do with this call":
  the table below is DESIGN.md 5's dispatch rule written out, and the allocation-failure fallback (every pair still
  exactly once) is walked without a GPU.

Reference semantics (metrics.go:273-295) do not depend on the path: every one of them is exact; the GPU tests hold that."""
import ctypes as C
import os
import subprocess

import pytest

from loghisto_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIRECT, SMALL, GEN1, GEN2, GEN3 = N.PATH_DIRECT, N.PATH_SMALL, N.PATH_GEN1, N.PATH_GEN2, N.PATH_GEN3


def probe(native_lib, **kw):
    q = N.LhDispatchQuery()
    q.struct_size = C.sizeof(q)
    q.id_width, q.ids_addr, q.vals_addr, q.lane_samples = 4, 0x100000, 0x800000, 1 << 20
    for k, v in kw.items():
        setattr(q, k, v)
    steps = (N.LhDispatchStep * 4096)()
    ns = C.c_size_t(0)
    assert native_lib.lh_dispatch_probe(C.byref(q), steps, 4096, C.byref(ns)) == 0
    assert 1 <= ns.value <= 4096
    out = steps[:ns.value]
    assert sum(s.take for s in out) == q.n            # every pair of the call in exactly one sub-launch
    return out


def test_cpp_table_test_of_the_decision_functions(native_lib):
    exe = os.path.join(ROOT, "loghisto_amd", "build", "dispatch_test")
    assert os.path.exists(exe), "python -m loghisto_amd.build builds tests/cpp/dispatch_test.cc"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "0 failed" in r.stdout and "choose_step: 24624 states" in r.stdout, r.stdout


# DESIGN.md 5, "Dispatch": (names, pairs) -> the path of the call's first sub-launch, all options at their defaults,
# device-resident aligned arrays
# (round 6: DIRECT is the cell-table kernel -- no scratch, no survey -- and wins below 2^20 / 3 * 2^20 pairs; the first
# generation is no device-resident call's default any more: profiles/r06_small_calls.txt)
GOLDEN = [
    (1, 65535, DIRECT), (1, 65536, SMALL), (32, 10**9, SMALL),                 # <= 32 names: one streaming pass
    (33, 65536, DIRECT), (33, 131072, DIRECT), (33, (1 << 20) - 2, DIRECT), (33, 1 << 20, GEN2),   # 33 .. 8 192 names
    (1024, (1 << 20) - 2, DIRECT), (1024, 1 << 20, GEN2), (1024, 1 << 25, GEN2), (1024, 10**9, GEN2), (8192, 10**9, GEN2),
    (8193, 1 << 20, DIRECT), (8193, (3 << 20) - 2, DIRECT), (8193, 3 << 20, GEN3),   # 8 193 .. 65 536 names
    (65536, 1 << 18, DIRECT), (65536, 3 << 20, GEN3), (65536, 125_000_000, GEN3), (65536, 10**9, GEN3),
    (65537, 10**9, DIRECT), (1 << 20, 10**9, DIRECT),                           # beyond: no partitioned path
]


@pytest.mark.parametrize("names,n,path", GOLDEN)
def test_documented_dispatch_rule(native_lib, names, n, path):
    steps = probe(native_lib, max_metrics=names, n=n)
    assert steps[0].path == path, [(s.path, s.take) for s in steps[:4]]
    assert (steps[0].scratch > 0) == (path >= GEN1)


def test_config_3_and_4_as_the_bench_runs_them(native_lib):
    c3 = probe(native_lib, max_metrics=1024, n=10**9)          # two sub-launches of at most 2^29 pairs in a block < 1.5 GiB
    assert [(s.path, s.take) for s in c3] == [(GEN2, 1 << 29), (GEN2, 10**9 - (1 << 29))]
    assert max(s.scratch for s in c3) < (1536 << 20)
    c4 = probe(native_lib, max_metrics=65536, n=125_000_000)   # config 4's slice: ONE third-generation launch, not cut
    assert [(s.path, s.take) for s in c4] == [(GEN3, 125_000_000)]
    big = probe(native_lib, max_metrics=65536, n=10**9)        # above 8 192 names a call is only cut when the caller bounds it
    assert [(s.path, s.take) for s in big] == [(GEN3, 10**9)]
    cut = probe(native_lib, max_metrics=65536, n=10**9, scratch_cap=1 << 30)
    assert all(s.path == GEN3 and s.take >= 1 << 27 for s in cut) and len(cut) == 4


def test_adaptive_switches_and_options(native_lib):
    p = lambda **kw: probe(native_lib, **kw)[0].path           # noqa: E731
    assert p(max_metrics=16, n=10**8, small_disabled=1) == GEN2           # few wide names: the partitioned path
    assert p(max_metrics=1024, n=10**9, v2_off=1) == GEN1
    assert p(max_metrics=1024, n=10**9, regions_disabled=1) == GEN2        # still the second generation: its exact-layout scatter
    assert p(max_metrics=65536, n=10**9, v3_disabled=1) == GEN1            # names without skew
    assert p(max_metrics=65536, n=10**9, regions_disabled=1) == GEN1       # clustered stream: no region scatter
    assert p(max_metrics=65536, n=10**9, v3_off=1) == GEN1
    assert p(max_metrics=65536, n=1 << 17, v3_min_pairs=1 << 17) == GEN3   # what the parity tests do
    assert p(max_metrics=1024, n=1 << 17, v2_min_pairs=1 << 17) == GEN2
    assert p(max_metrics=1024, n=100_000, part_min_pairs=65536) == GEN1
    assert p(max_metrics=1024, n=10**6, part_min_pairs=1 << 21) == DIRECT


def test_alignment(native_lib):
    # both arrays one element short: ONE sample is peeled, the rest is aligned again
    s = probe(native_lib, max_metrics=1024, n=10**9, ids_addr=0x100004, vals_addr=0x800008)
    assert (s[0].path, s[0].take, s[0].peeled) == (DIRECT, 1, 1) and s[1].path == GEN2 and not s[1].peeled
    s = probe(native_lib, max_metrics=1024, n=10**9, ids_addr=0x100002, vals_addr=0x800008, id_width=2)
    assert (s[0].take, s[0].peeled) == (1, 1) and s[1].path == GEN2
    # only one of them misaligned: nothing to peel, no vector loads -- the direct kernel takes all of it
    s = probe(native_lib, max_metrics=1024, n=10**9, ids_addr=0x100004)
    assert [(x.path, x.take) for x in s] == [(DIRECT, 10**9)]
    s = probe(native_lib, max_metrics=4, n=10**9, vals_addr=0x800008)
    assert s[0].path == DIRECT


def test_host_fed_lane_launches(native_lib):
    for names in (1024, 20000, 65536):
        s = probe(native_lib, max_metrics=names, n=1 << 20, host_fed=1, lane_blocks=8)
        # a half-buffer in a lane's block: first generation up to 8 192 names, third above (tables shared by the lanes)
        assert [(x.path, x.lane_block) for x in s] == [(GEN1 if names <= 8192 else GEN3, 1)]
        assert s[0].scratch <= (64 << 20) or names > 8192
        s = probe(native_lib, max_metrics=names, n=1 << 20, host_fed=1, lane_blocks=8, lane_gen3_off=1)
        assert [(x.path, x.lane_block) for x in s] == [(GEN1, 1)]            # LH_OPT_LANE_GEN3 = 0: as up to ABI 4
        for off in ({"v3_disabled": 1}, {"regions_disabled": 1}, {"v3_off": 1}):
            s = probe(native_lib, max_metrics=names, n=1 << 20, host_fed=1, lane_blocks=8, **off)
            assert [(x.path, x.lane_block) for x in s] == [(GEN1, 1)]
        s = probe(native_lib, max_metrics=names, n=1 << 20, host_fed=1, lane_blocks=0)
        assert (s[0].path, s[0].lane_block, s[0].scratch) == (DIRECT, 0, 0)  # no lane blocks (the default since round 6): the direct path
        s = probe(native_lib, max_metrics=names, n=1 << 22, host_fed=1, lane_blocks=0)
        assert (s[0].path, s[0].scratch) == (DIRECT, 0)
        s = probe(native_lib, max_metrics=names, n=(1 << 22) + 2, host_fed=1, lane_blocks=0)
        assert s[0].path >= GEN1                                             # a buffer larger than that: the shared block's rules
        s = probe(native_lib, max_metrics=names, n=(1 << 22) + 2, host_fed=1, lane_blocks=8)
        assert not s[0].lane_block                                           # larger than a lane block serves
    s = probe(native_lib, max_metrics=8, n=1 << 20, host_fed=1, lane_blocks=8)
    assert s[0].path == SMALL
    s = probe(native_lib, max_metrics=1024, n=50_000, host_fed=1, lane_blocks=8)
    assert (s[0].path, s[0].lane_block) == (DIRECT, 0)                      # below the partitioned minimum: no block, no lock
    # the lanes keep the thresholds they were tuned with (2^17 / 2^18), whatever the device-resident calls' are
    s = probe(native_lib, max_metrics=1024, n=1 << 17, host_fed=1, lane_blocks=8)
    assert (s[0].path, s[0].lane_block) == (GEN1, 1)
    s = probe(native_lib, max_metrics=65536, n=1 << 18, host_fed=1, lane_blocks=8)
    assert (s[0].path, s[0].lane_block) == (GEN3, 1)
    s = probe(native_lib, max_metrics=65536, n=1 << 18, host_fed=0)
    assert s[0].path == DIRECT


@pytest.mark.parametrize("names,n,host_fed", [(1024, 10**9, 0), (65536, 125_000_000, 0), (65536, 3 * 10**9, 0),
                                               (1024, 1 << 20, 1), (300, 1 << 25, 0)])
@pytest.mark.parametrize("fail", [1, 2, 1000])
def test_a_block_that_cannot_be_had_never_fails_the_call(native_lib, names, n, host_fed, fail):
    """ingest never fails (metrics.go:251, 273): a sub-launch whose scratch cannot be allocated goes through the
    scratch-free kernel; the call still covers every pair exactly once (probe() asserts the sum), and once an allocation
    succeeds the following sub-launches run partitioned again."""
    ok = probe(native_lib, max_metrics=names, n=n, host_fed=host_fed, lane_blocks=8 * host_fed)
    got = probe(native_lib, max_metrics=names, n=n, host_fed=host_fed, lane_blocks=8 * host_fed, fail_allocs=fail)
    assert [s.take for s in got] == [s.take for s in ok]                     # the same cuts
    fell = [s for s in got if s.fell_back]
    assert all(s.path == DIRECT and s.scratch == 0 for s in fell)
    assert len(fell) == min(fail, len(ok))                                   # one allocation attempt per sub-launch until one succeeds
    for a, b in zip(ok[len(fell):], got[len(fell):]):
        assert (a.path, a.lane_block) == (b.path, b.lane_block)


def test_probe_rejects_bad_queries(native_lib):
    q = N.LhDispatchQuery()
    ns = C.c_size_t(0)
    steps = (N.LhDispatchStep * 4)()
    assert native_lib.lh_dispatch_probe(C.byref(q), steps, 4, C.byref(ns)) == N.EINVAL   # struct_size 0
    q.struct_size, q.max_metrics, q.n, q.id_width = C.sizeof(q), 8, 10, 3
    assert native_lib.lh_dispatch_probe(C.byref(q), steps, 4, C.byref(ns)) == N.EINVAL   # id width
    q.id_width = 4
    assert native_lib.lh_dispatch_probe(C.byref(q), None, 0, C.byref(ns)) == 0 and ns.value == 1
