"""Every value distribution of tools/sweep.py (SURVEY.md 8d's contention sweep + the round-4/5/6 additions) through the MIXED
paths, every cell of every row against the oracle: 1 024 names (second generation: survey, 16-bit hot windows in regions
sized by the plan, k_hot_reduce, 2-byte records) and 65 536 names (third generation).  tests/test_gpu_parity.py drives the
same list through K1; VERDICT r5 next #2 asked for the mixed paths.  Two-signed streams (readme.md:43: `_min -657.5` --
"time.Since(time.Now()) is often < 0") put two lobes of bins either side of key 0; +-10^U(-3, 20) spans 9 211 bins per name
(wider than any window of either path); thin_far_tail sends one sample in a thousand past every window."""
import numpy as np
import pytest

from tests.test_gpu_parity import dists
from tests.test_gpu_part3 import engine_cells, oracle_cells

pytestmark = pytest.mark.gpu

KINDS = ["lognormal_s1", "constant", "uniform", "exponential", "normal_signed", "loguniform", "lognormal_s2.5", "tiny",
         "kvalues2", "kvalues4", "kvalues8", "kvalues16", "kvalues3_skewed", "bimodal", "far_1e30", "negative_far",
         "signed_wide", "thin_far_tail"]


def _run(torch, M, n, opts):
    import loghisto_amd
    from loghisto_amd import _native as N
    rng = np.random.default_rng(1000 + M)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    d_ids = torch.from_numpy(ids.astype(np.int32)).cuda()
    all_v = dists(n, 77 + M)
    bad = {}
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        for k, val in opts.items():
            e.set_option(k, val)
        for kind in KINDS:
            v = all_v[kind]
            e.set_option(N.OPT_SURVEY_EVERY, 32)          # every distribution on its own survey ...
            d_v = torch.from_numpy(v).cuda()
            for rep in range(2):                           # ... and once more on the survey it left behind
                e.submit_pairs_device(d_ids, d_v)
                e.sync()
                with e.flip() as snap:
                    cells, counts = engine_cells(snap, M)
                if rep == 0:
                    want_cells, want_counts = oracle_cells(ids, v)   # sorted (name << 16 | bin) of every occupied cell
                if not (np.array_equal(cells, want_cells) and np.array_equal(counts, want_counts)):
                    a = dict(zip(cells.tolist(), counts.tolist()))
                    b = dict(zip(want_cells.tolist(), want_counts.tolist()))
                    wrong = sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))
                    bad[(kind, rep)] = [(k >> 16, k & 0xffff, a.get(k), b.get(k)) for k in wrong[:4]]
        c = e.counters()
    return bad, c


def test_every_sweep_distribution_through_the_second_generation(native_lib, torch_cuda):
    from loghisto_amd import _native as N
    n = 2_500_000
    bad, c = _run(torch_cuda, 1024, n, {N.OPT_PART_V2_MIN_PAIRS: 1 << 17})
    assert not bad, bad
    assert c["samples_partitioned_v2"] == 2 * len(KINDS) * n, c   # every call took the survey path


def test_every_sweep_distribution_through_the_third_generation(native_lib, torch_cuda):
    from loghisto_amd import _native as N
    n = 2_000_000
    bad, c = _run(torch_cuda, 65536, n, {N.OPT_PART_V3_MIN_PAIRS: 1 << 17, N.OPT_PART_V3_DIRECT_MAX_PAIRS: 1})
    assert not bad, bad
    # (a stream without skew in its forwarded share may be sent back to the first generation after it has reported: both are
    # partitioned paths)
    assert c["samples_partitioned"] == 2 * len(KINDS) * n and c["samples_partitioned_v3"] >= len(KINDS) * n, c
