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


def _zipf_stream(torch, n, M, seed, scale):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    w = 1.0 / torch.arange(1, M + 1, device="cuda", dtype=torch.float64)
    cdf = torch.cumsum(w / w.sum(), 0)
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    step = 1 << 27
    for lo in range(0, n, step):                                  # chunked: searchsorted needs int64 scratch
        u = torch.rand(min(step, n - lo), device="cuda", dtype=torch.float64, generator=g)
        ids[lo:lo + step] = torch.searchsorted(cdf, u).clamp_(max=M - 1).to(torch.int32)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    for lo in range(0, n, step):
        v[lo:lo + step].add_(math.log(1e5)).add_(ids[lo:lo + step].to(torch.float64), alpha=scale).exp_()
    torch.cuda.synchronize()
    return ids, v


@pytest.mark.parametrize("M,n,scale", [(1024, 1_000_000_000, 0.002), (65536, 125_000_000, 3e-5)],
                         ids=["c3-1e9-pairs-1024-names", "c4-rank-slice-65536-names"])
def test_mixed_stream_full_size(native_lib, torch_cuda, M, n, scale):
    """BASELINE configs[2] (1e9 pairs over 1 024 Zipf names) and one rank's slice of configs[3] (65 536 names)
    at full size, through size-independent properties:
      * per-name conservation against torch.bincount of the ids
      * the partitioned mixed kernels against the single-metric kernel (two different code paths) on the
        samples of a hot, a middle and two cold names: rows must be bit-identical
      * linearity across two launches
      * oracle-exact on a 4M-pair prefix"""
    torch = torch_cuda
    import loghisto_amd
    ids, v = _zipf_stream(torch, n, M, seed=3, scale=scale)
    per_name = torch.bincount(ids, minlength=M).cpu().numpy().astype(np.uint64)
    probe = [0, 1, M // 2, M - 1]
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng, \
            loghisto_amd.Engine(max_metrics=len(probe), num_buffers=2, num_lanes=1, lane_samples=1 << 16) as ref:
        eng.submit_pairs_device(ids, v)
        with eng.flip() as snap:
            got = snap.extract(PCTS, M)
            rows = {m: snap.dense_row(m) for m in probe}
        assert np.array_equal(got["count"], per_name) and int(got["count"].sum()) == n
        for i, m in enumerate(probe):
            sel = v[ids == m].contiguous()
            torch.cuda.synchronize()
            ref.submit_device(i, sel)
        with ref.flip() as rsnap:
            for i, m in enumerate(probe):
                assert np.array_equal(rsnap.dense_row(i), rows[m]), m
                want = oracle.process_dense(rows[m], PCTS)
                assert np.array_equal(got["pvals"][m].view(np.uint64), want["pvals"].view(np.uint64)), m
                assert abs(got["sum"][m] - want["sum"]) <= 1e-12 * abs(want["sum"]), m

        h = n // 3 + 4097                                           # odd split: unaligned second launch
        eng.submit_pairs_device(ids[:h], v[:h], h)
        eng.submit_pairs_device(ids[h:], v[h:], n - h)
        with eng.flip() as snap:
            again = snap.extract(PCTS, M)
            for m in probe:
                assert np.array_equal(snap.dense_row(m), rows[m]), m
        assert np.array_equal(again["count"], per_name)
        assert np.array_equal(again["pvals"].view(np.uint64), got["pvals"].view(np.uint64))

        k = 4_000_000
        eng.submit_pairs_device(ids[:k], v[:k], k)
        with eng.flip() as snap:
            off, keys, counts = snap.buckets_all(M)
        # oracle cells as a sorted (name, bin) list: a dense [M][65536] matrix would be 32 GiB at M = 65 536
        bins = oracle.key_to_bin(oracle.compress_many(v[:k].cpu().numpy())).astype(np.uint64)
        cell, cnt = np.unique((ids[:k].cpu().numpy().astype(np.uint64) << np.uint64(16)) | bins, return_counts=True)
        assert keys.size == cell.size
        assert np.array_equal(keys, oracle.bin_to_key(cell & np.uint64(0xFFFF)))
        assert np.array_equal(counts, cnt.astype(np.uint64))
        per = np.bincount((cell >> np.uint64(16)).astype(np.int64), minlength=M)
        assert np.array_equal(off, np.concatenate([[0], np.cumsum(per)]).astype(np.uint64))
