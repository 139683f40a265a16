"""1 025 .. 8 192 names: the second generation's cold window narrows with the name count (8 192 bins at 1 024 names, 1 024 at
8 192) and a sample outside it is a global atomic -- 1e9 pairs of normal(0, 1e3) over 8 192 names took 57 ms, of 21 decades
over 4 096 names 49 (lognormal: 3.0).  A launch of a stream that leaves more than 1/8 of its mass outside those windows (the
last survey's report) is left to the third generation, whose windows follow the stream.  Every cell exact on either side of the
hand-over and back (metrics.go:273-295)."""
import numpy as np
import pytest

import os
import sys

from loghisto_amd import _native as N

sys.path.insert(0, os.path.dirname(__file__))
from test_gpu_part3 import PCTS, _dev, _ids, _values, check  # noqa: E402

pytestmark = pytest.mark.gpu

CASES = [
    # names, pairs, wide stream, its width class, does the third generation take it?
    (8192, 3_500_000, "signed", 12, True),        # normal(0, 1e4): two lobes 1 840 bins apart against 1 024-bin cold windows
    (8192, 3_400_001, "loguniform", 13, True),
    (4096, 3_300_000, "loguniform", 13, True),    # 16 names per partition: 2 048-bin cold windows
    (4096, 3_300_000, "sigma25", 11, False),      # ... which hold lognormal sigma 2.5 (251 bins)
    (2048, 3_200_000, "signed_wide", 14, True),   # 8 names per partition
    (1025, 3_200_000, "signed_wide", 14, True),   # 5 names per level-1 partition of the third generation
    (2048, 3_200_000, "loguniform", 13, False),   # 4 147-bin spans in 4 096-bin cold windows: a percent outside
    (3000, 3_600_000, "huge", 14, True),
    (1024, 3_200_000, "loguniform", 13, False),   # <= 1 024 names: 8 192-bin cold windows (and the WIDE shape beyond)
]


@pytest.mark.parametrize("M,n,kind,cls,gen3", CASES)
def test_wide_streams_between_1025_and_8192_names(native_lib, torch_cuda, M, n, kind, cls, gen3):
    import loghisto_amd
    rng = np.random.default_rng(M + n)
    ids = _ids(rng, M, n, 1.0)
    wide = _values(rng, kind, ids, n)
    narrow = _values(rng, "lognormal", ids, n)
    d_ids, d_w, d_n = _dev(torch_cuda, ids), _dev(torch_cuda, wide), _dev(torch_cuda, narrow)
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        def call(d_v, v):
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
            return e.counters()

        c0 = call(d_w, wide)                       # the second generation, whose survey reports the stream's class
        assert c0["samples_partitioned_v2"] == n and c0["samples_partitioned_v3"] == 0 and c0["window_log2"] == cls, c0
        c1 = call(d_w, wide)
        c2 = call(d_w, wide)
        assert c2["samples_partitioned_v3"] == (2 * n if gen3 else 0), c2
        assert c2["samples_partitioned_v2"] == (n if gen3 else 3 * n), c2
        e.set_option(N.OPT_SURVEY_EVERY, 1)        # (a kept survey would keep the generation for up to 32 calls)
        c3 = call(d_n, narrow)                     # narrow values: the survey in charge reports them ...
        c4 = call(d_n, narrow)                     # ... and the launches are the second generation's again
        c5 = call(d_n, narrow)
        assert c5["samples_partitioned_v2"] - c4["samples_partitioned_v2"] == n, (c3, c4, c5)
        assert c5["samples_partitioned_v2"] + c5["samples_partitioned_v3"] == 6 * n, c5


def test_the_first_call_of_a_wide_stream_is_probed(native_lib, torch_cuda):
    """A fresh engine's first large call (>= 2^24 pairs) asks the survey before it chooses the path (probe_width, lh_engine.cc):
    normal(0, 1e4) over 8 192 names is the third generation's from the first call on (56 -> 10 ms per 1e9 pairs), a lognormal
    stream stays the second's.  Every cell exact."""
    import loghisto_amd
    M, n = 8192, (1 << 24) + 4096
    rng = np.random.default_rng(17)
    ids = _ids(rng, M, n, 1.0)
    for kind, gen3 in (("signed", True), ("lognormal", False)):
        v = _values(rng, kind, ids, n)
        d_ids, d_v = _dev(torch_cuda, ids), _dev(torch_cuda, v)
        with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
            e.submit_pairs_device(d_ids, d_v)
            e.sync()
            c = e.counters()
            assert c["samples_partitioned_v3"] == (n if gen3 else 0) and c["samples_partitioned_v2"] == (0 if gen3 else n), (kind, c)
            with e.flip() as snap:
                check(snap, ids, v, M, snap.extract(PCTS, M))
