"""uint16-id pairs (10 bytes per sample; SURVEY.md 8d "10 B if ids are uint16", legal for <= 65 536 names): every
mixed-ingest kernel is a template on the id width and reads the narrow ids directly.  Each path the engine can
dispatch to -- direct atomics, the single-pass kernel for few names, the first generation (and its two-level form), the
second generation in all four shapes, the third generation -- is driven with the same stream as uint32 and as uint16
ids, device-resident and through the three host staging forms; every cell of every row must equal the oracle's.
Reference semantics: Histogram(name, v) = histogramCache[name][compress(v)] += 1 (metrics.go:273-295, 316-322)."""
import math

import numpy as np
import pytest

import oracle
from loghisto_amd import _native as N
from tests.test_gpu_part3 import check, _ids, _values

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .9, .99, .999, 1.0]


def _dev16(torch, ids):
    return torch.from_numpy(np.ascontiguousarray(ids.astype(np.uint16)).view(np.int16)).cuda()


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


# names, pairs, values, options {id: value}, counter that must have moved (the path taken)
PATHS = [
    (700, 100_001, "lognormal", {}, "samples_direct"),                                    # small call: the direct path (cell table)
    (20, 1_500_001, "lognormal", {}, "samples_small"),                                     # <= 32 names: single-pass kernel
    (5, 400_000, "kvalues2", {}, "samples_small"),
    (1000, 600_001, "lognormal", {N.OPT_PART_MIN_PAIRS: 1 << 17, N.OPT_PART_V2: 0}, "samples_partitioned"),   # first generation
    (30000, 1_000_001, "edge", {N.OPT_PART_MIN_PAIRS: 1 << 17, N.OPT_PART_V3: 0}, "samples_partitioned"),     # ... with its second level
    (1024, 2_000_001, "lognormal", {N.OPT_PART_V2_MIN_PAIRS: 1 << 17, N.OPT_PART_V2_SHAPE: 2}, "samples_partitioned_v2"),
    (1024, 2_000_000, "edge", {N.OPT_PART_V2_MIN_PAIRS: 1 << 17, N.OPT_PART_V2_SHAPE: 3}, "samples_partitioned_v2"),
    (3000, 2_000_001, "signed", {N.OPT_PART_V2_MIN_PAIRS: 1 << 17, N.OPT_PART_V2_SHAPE: 0}, "samples_partitioned_v2"),
    (3000, 1_900_000, "loguniform", {N.OPT_PART_V2_MIN_PAIRS: 1 << 17, N.OPT_PART_V2_SHAPE: 1}, "samples_partitioned_v2"),
    (65536, 2_500_001, "lognormal", {N.OPT_PART_V3_MIN_PAIRS: 1 << 17}, "samples_partitioned_v3"),
    (65536, 2_000_000, "edge", {N.OPT_PART_V3_MIN_PAIRS: 1 << 17, N.OPT_PART_V3_LOG_W: 10}, "samples_partitioned_v3"),
    (40000, 2_200_000, "kvalues8", {N.OPT_PART_V3_MIN_PAIRS: 1 << 17}, "samples_partitioned_v3"),
]


@pytest.mark.parametrize("M,n,kind,opts,path", PATHS)
def test_uint16_ids_through_every_kernel_path(native_lib, torch_cuda, M, n, kind, opts, path):
    import loghisto_amd
    rng = np.random.default_rng(M * 3 + n)
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, kind, ids, n)
    d_v = _dev(torch_cuda, v)
    d16, d32 = _dev16(torch_cuda, ids), _dev(torch_cuda, ids)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        for k, val in opts.items():
            e.set_option(k, val)
        for rep, d_ids in enumerate((d16, d32, d16[1:], d16)):   # [1:] the odd-offset slice: alignment peel of 2-byte ids
            off = 1 if rep == 2 else 0
            before = e.counters()
            e.submit_pairs_device(d_ids, d_v[off:])
            e.sync()
            after = e.counters()
            moved = {k for k in after if isinstance(after[k], int) and k.startswith("samples_") and after[k] != before[k]}
            assert path in moved, (path, moved)
            with e.flip() as snap:
                check(snap, ids[off:], v[off:], M, snap.extract(PCTS, M))


def test_uint16_ids_with_bad_ids_are_reported(native_lib, torch_cuda):
    """An id >= max_metrics in a uint16 stream is skipped and reported exactly like a uint32 one."""
    import loghisto_amd
    rng = np.random.default_rng(5)
    M, n = 1000, 700_000
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, "lognormal", ids, n)
    bad = ids.copy()
    where = [7, 300_001, n - 1]
    bad[where] = [M, 65535, M + 3]
    keep = np.ones(n, dtype=bool)
    keep[where] = False
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        d_bad, d_v = _dev16(torch_cuda, bad), _dev(torch_cuda, v)
        e.submit_pairs_device(d_bad, d_v)
        with pytest.raises(loghisto_amd.LhError) as ei:
            e.sync()
        assert ei.value.code == 6
        with e.flip() as snap:
            try:
                got = snap.extract(PCTS, M)
            except loghisto_amd.LhError:
                got = snap.extract(PCTS, M)
            check(snap, ids[keep], v[keep], M, got)
        with pytest.raises(loghisto_amd.LhError):
            e.submit_pairs(bad.astype(np.uint16), v)        # the copying form validates on the host


@pytest.mark.parametrize("M", [1024, 65536])
@pytest.mark.parametrize("zero_copy", [1, 0])
def test_uint16_ids_through_the_host_staging_forms(native_lib, torch_cuda, M, zero_copy):
    """lh_submit_pairs16 (copy) and lh_reserve_pairs16 / lh_commit_pairs16 (in place), mixed with uint32 submissions on
    the same lanes (a staging buffer holds one width at a time), kernels reading the pinned buffers in place or after a
    copy; every cell against the oracle."""
    import loghisto_amd
    rng = np.random.default_rng(M + zero_copy)
    n = 1_300_003
    ids = _ids(rng, M, n, 1.0)
    v = _values(rng, "lognormal", ids, n)
    i16 = ids.astype(np.uint16)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=1 << 18) as e:
        e.set_option(N.OPT_LANE_ZERO_COPY, zero_copy)
        cuts = [0, 200_001, 500_000, 777_777, 1_000_001, n]
        forms = ["copy16", "inplace16", "copy32", "inplace16", "copy16"]
        for (a, b), form in zip(zip(cuts[:-1], cuts[1:]), forms):
            if form == "copy16":
                e.submit_pairs(i16[a:b], v[a:b])
            elif form == "inplace16":
                e.submit_pairs_in_place(i16[a:b], v[a:b])
            else:
                e.submit_pairs(ids[a:b], v[a:b])
        e.sync()
        with e.flip() as snap:
            check(snap, ids, v, M, snap.extract(PCTS, M))
        # a reservation of the narrow form hands out uint16 room
        di, dv, tok = e.reserve_pairs(100, 16)
        assert di.dtype == np.uint16 and di.size == dv.size and di.size >= 3
        di[:3] = [1, 2, M - 1]
        dv[:3] = [5.0, 6.0, 7.0]
        e.commit_pairs(tok, 3)
        e.sync()
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
        assert int(got["count"].sum()) == 3 and int(got["count"][M - 1]) == 1
