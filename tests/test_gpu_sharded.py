"""BASELINE config 4 in miniature, on one device ("N logical shards on one device"):
S engines play S ranks.  Each ingests ITS slice of one Zipf stream over ALL names (data-parallel,
no per-sample communication); at the flip the uint64 bucket matrices are summed -- here with plain
device adds over the aliased snapshot rows, standing in for the reduce-scatter whose arithmetic
tests/test_merge_gloo.py checks across real processes -- and every "rank" extracts only the block
of names it owns (lh_extract_rows).  Checked against the oracle run on the whole stream."""
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .9, .99, .999, 1.0]


def test_sharded_ingest_merge_extract(native_lib, torch_cuda):
    torch = torch_cuda
    import loghisto_amd
    from loghisto_amd import merge

    S, M, n = 4, 8192, 6_000_000          # 32 names per partition: exercises narrow LDS windows
    rng = np.random.default_rng(44)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(math.log(1e5) + 0.0002 * ids, 1.0)
    engines = [loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) for _ in range(S)]
    try:
        snaps = []
        for r, eng in enumerate(engines):
            lo, hi = n * r // S, n * (r + 1) // S
            eng.submit_pairs_device(torch.from_numpy(ids[lo:hi].view(np.int32)).cuda(),
                                    torch.from_numpy(v[lo:hi]).cuda())
            eng.sync()
            snaps.append(eng.flip())
        views = [merge.snapshot_tensors(s, M) for s in snaps]
        torch.cuda.synchronize()
        # "reduce-scatter": rank r ends up with the sum of everybody's rows for the names it owns
        for r in range(S):
            first, last = merge.name_blocks(M, r, S)
            rows_r, ranges_r = views[r]
            for q in range(S):
                if q != r:
                    rows_r[first:last] += views[q][0][first:last]
            lo_bins = torch.stack([views[q][1][first:last, 0] for q in range(S)]).min(dim=0).values
            hi_bins = torch.stack([views[q][1][first:last, 1] for q in range(S)]).max(dim=0).values
            ranges_r[first:last, 0] = lo_bins
            ranges_r[first:last, 1] = hi_bins
        torch.cuda.synchronize()
        total = 0
        for r in range(S):
            first, last = merge.name_blocks(M, r, S)
            got = snaps[r].extract(PCTS, last - first, first=first)
            total += int(got["count"].sum())
            # oracle on the whole stream for a sample of the owned names (hot, middle and cold ones)
            for m in sorted({first, first + 1, (first + last) // 2, last - 1}):
                sel = ids == m
                want_row = oracle.histogram_dense(v[sel])
                want = oracle.process_dense(want_row, PCTS)
                i = m - first
                assert int(got["count"][i]) == want["count"] == int(sel.sum()), m
                if want["count"]:
                    assert np.array_equal(got["pvals"][i].view(np.uint64), want["pvals"].view(np.uint64)), m
                    assert abs(got["sum"][i] - want["sum"]) <= 1e-12 * abs(want["sum"]), m
                    keys, counts = snaps[r].buckets(m)
                    nz = np.nonzero(want_row)[0]
                    assert np.array_equal(keys, oracle.bin_to_key(nz)) and np.array_equal(counts, want_row[nz]), m
        assert total == n                     # every sample is owned by exactly one rank after the merge
        for s in snaps:
            s.release()
    finally:
        for e in engines:
            e.close()
