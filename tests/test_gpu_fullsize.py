"""BASELINE.json configs[1..3] at FULL size, oracle-exact over EVERY sample.

The threaded forms of the oracle (oracle/lh_cpu_baseline.cc: the same lho_compress per sample, the stream cut
into one slice per granted host core) bucket 1e9 samples in about a second, so "bit-exact bucket counts vs
metrics.go" no longer rests on a prefix of the stream:

  * C2  1e9 float64 samples, one metric: the GPU row == the oracle's row over all 1e9 samples
  * C3  1e9 (id, value) pairs over 1 024 Zipf names: all 1 024 rows == the oracle's matrix over all 1e9 pairs
  * C4  one rank's 1.25e8-pair slice over 65 536 names: every occupied (name, key) cell == the oracle's cells
  * plus the size-independent properties: conservation, linearity across launches / epochs, K2 against the
    oracle on the full-size rows, two different kernel paths agreeing on the same samples
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

        # oracle-exact over ALL 1e9 samples (8 GB to the host, one slice per granted core)
        host = data.cpu().numpy()
        want_row = oracle.histogram_dense_mt(host)
        del host
        assert int(want_row.sum()) == n
        assert np.array_equal(row, want_row), "GPU row differs from the oracle over the full 1e9-sample stream"

        # K2 against the oracle on the full-size row
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
    at full size:
      * every occupied cell of every name against the oracle over ALL pairs of the stream
      * per-name conservation against torch.bincount of the ids
      * the partitioned mixed kernels against the single-metric kernel (two different code paths) on the
        samples of a hot, a middle and two cold names: rows must be bit-identical
      * linearity across two launches"""
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
            off, keys, counts = snap.buckets_all(M)
        assert np.array_equal(got["count"], per_name) and int(got["count"].sum()) == n

        # oracle-exact over ALL n pairs: every occupied (name, key) cell
        h_ids = ids.cpu().numpy().view(np.uint32)
        h_v = v.cpu().numpy()
        if M <= 4096:
            # dense oracle matrix (512 MiB at 1 024 names), threaded over the granted cores
            want = oracle.histogram_pairs_mt(h_ids, h_v, M)
            assert int(want.sum()) == n
            got_m = np.zeros((M, 65536), dtype=np.uint64)
            rows_i = np.repeat(np.arange(M), np.diff(off.astype(np.int64)))
            got_m[rows_i, oracle.key_to_bin(keys)] = counts
            assert np.array_equal(got_m, want), "GPU cells differ from the oracle over the full stream"
            del want, got_m
        else:
            # oracle cells as a sorted (name, bin) list: a dense [M][65536] matrix would be 32 GiB at M = 65 536
            bins = oracle.key_to_bin(oracle.compress_many(h_v)).astype(np.uint64)
            cell, cnt = np.unique((h_ids.astype(np.uint64) << np.uint64(16)) | bins, return_counts=True)
            assert keys.size == cell.size
            assert np.array_equal(keys, oracle.bin_to_key(cell & np.uint64(0xFFFF)))
            assert np.array_equal(counts, cnt.astype(np.uint64))
            per = np.bincount((cell >> np.uint64(16)).astype(np.int64), minlength=M)
            assert np.array_equal(off, np.concatenate([[0], np.cumsum(per)]).astype(np.uint64))
        del h_ids, h_v

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
            off2, keys2, counts2 = snap.buckets_all(M)
        assert np.array_equal(again["count"], per_name)
        assert np.array_equal(again["pvals"].view(np.uint64), got["pvals"].view(np.uint64))
        assert np.array_equal(off2, off) and np.array_equal(keys2, keys) and np.array_equal(counts2, counts)


def test_calls_larger_than_one_launch(native_lib, torch_cuda):
    """One call above the engine's per-launch caps (2^31 samples single-metric, 2^30 pairs mixed): the call is cut into
    launches at 64-bit offsets.  Size-independent property (no 40 GB oracle pass): a stream made of k copies of a base
    stream gives k times the base stream's rows; the base rows are checked against the oracle by the tests above."""
    torch = torch_cuda
    import loghisto_amd
    base_n, k = 900_000_001, 5                                      # 4.5e9 samples = 36 GB: launches of 2^31, 2^31, rest
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    base = torch.randn(base_n, dtype=torch.float64, device="cuda", generator=g).add_(math.log(1e5)).exp_()
    big = base.repeat(k)
    torch.cuda.synchronize()
    with loghisto_amd.Engine(max_metrics=1, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_device(0, base)
        with eng.flip() as snap:
            row1 = snap.dense_row(0)
        assert int(row1.sum()) == base_n
        eng.submit_device(0, big)
        with eng.flip() as snap:
            rowk = snap.dense_row(0)
            got = snap.extract(PCTS, 1)
        assert np.array_equal(rowk, k * row1)
        assert int(got["count"][0]) == k * base_n                  # > 2^32 samples in one interval
        want = oracle.process_dense(rowk, PCTS)
        assert np.array_equal(got["pvals"][0].view(np.uint64), want["pvals"].view(np.uint64))
    del big, base
    torch.cuda.empty_cache()

    M, pn, pk = 1024, 500_000_003, 5                               # 2.5e9 pairs = 40 GB: launches of 2^30, 2^30, rest
    ids, v = _zipf_stream(torch, pn, M, 12, 0.002)
    bids, bv = ids.repeat(pk), v.repeat(pk)
    torch.cuda.synchronize()
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as eng:
        eng.submit_pairs_device(ids, v)
        with eng.flip() as snap:
            c1 = snap.extract([0.5], M)["count"].astype(np.int64).copy()
            probe = [0, 1, 7, 100, 511, 1023]
            r1 = [snap.dense_row(m).copy() for m in probe]
        eng.submit_pairs_device(bids, bv)
        with eng.flip() as snap:
            ck = snap.extract([0.5], M)["count"].astype(np.int64)
            assert np.array_equal(ck, pk * c1) and int(ck.sum()) == pk * pn
            for m, r in zip(probe, r1):
                assert np.array_equal(snap.dense_row(m), pk * r)
    del bids, bv, ids, v
    torch.cuda.empty_cache()                                       # 40 GB back to the device for the tests that follow
