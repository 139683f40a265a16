"""BASELINE.json configs[1] at full size (1e9 float64 samples, 8 GB resident):
size-independent properties, because the per-sample oracle would need minutes.

  * conservation: sum of all cells == n
  * linearity: hist(A ++ B) == hist(A) + hist(B) across two launches / two epochs
  * an independent device recomputation of the keys with torch's float64 log
    (differs from Go's log by <= a few ulp at ~10 % of thresholds, i.e. a sample
    would have to land within ~1e-15 relative of a threshold to differ: expected
    mismatches over 1e9 samples ~1e-5) must give the same row
  * the oracle, which works on the 65536-cell row and is therefore size
    independent, reproduces the extract output exactly
  * an oracle-exact check on a 4M-sample prefix of the same stream
"""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0]


def test_c2_one_billion_samples(native_lib, torch_cuda):
    torch = torch_cuda
    import loghisto_amd
    n = 1_000_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    data = torch.randn(n, dtype=torch.float64, device="cuda", generator=g).add_(math.log(1e5)).exp_()
    # the generator kernels run asynchronously on torch's stream; submit_device(stream=None) runs on the
    # engine's own non-blocking stream, so the producer must be finished first (lh_submit_device contract)
    torch.cuda.synchronize()
    with loghisto_amd.Engine(max_metrics=2, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_device(0, data)
        with eng.flip() as snap:
            got = snap.extract(PCTS, 1)
            row = snap.dense_row(0)
        assert int(row.sum()) == n and int(got["count"][0]) == n

        # size-independent oracle check of K2 on the full-size row
        want = oracle.process_dense(row, PCTS)
        assert np.array_equal(got["pvals"][0].view(np.uint64), want["pvals"].view(np.uint64))
        assert np.array_equal(got["pkeys"][0], want["pkeys"])
        assert abs(got["sum"][0] - want["sum"]) <= 1e-12 * abs(want["sum"])
        assert int(got["agg_sum_add"][0]) == oracle.f64_to_u64_amd64(float(got["sum"][0]))

        # linearity across launches and epochs
        h = n // 2 + 12345
        eng.submit_device(0, data[:h], h)
        eng.submit_device(1, data[h:], n - h)
        with eng.flip() as snap:
            ra, rb = snap.dense_row(0), snap.dense_row(1)
        assert np.array_equal(ra + rb, row)

        # oracle-exact on a prefix of the same stream
        m = 4_000_000
        eng.submit_device(0, data[:m], m)
        with eng.flip() as snap:
            rp = snap.dense_row(0)
        assert np.array_equal(rp, oracle.histogram_dense(data[:m].cpu().numpy()))

    # independent recomputation with torch's log (chunked to bound memory)
    indep = torch.zeros(65536, dtype=torch.int64, device="cuda")
    step = 1 << 27
    for lo in range(0, n, step):
        x = data[lo:lo + step]
        k = torch.log(1.0 + x.abs()).mul_(100.0).add_(0.5).floor_().to(torch.int64)
        k = torch.where(x < 0, -k, k)
        indep += torch.bincount(k + 32768, minlength=65536)[:65536]
    indep = indep.cpu().numpy().astype(np.uint64)
    mismatch = int(np.abs(indep.astype(np.int64) - row.astype(np.int64)).sum()) // 2
    assert mismatch <= 2, f"{mismatch} samples bucketed differently from an independent float64 log"
