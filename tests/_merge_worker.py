"""Worker for tests/test_merge_gloo.py: one process per rank, gloo backend, CPU tensors.

Each rank buckets ITS slice of a seeded mixed stream for ALL names with the oracle
(standing in for the device rows, which need a GPU), merges with
loghisto_amd.merge.merge_rows, and checks the merged rows against the oracle run on
the whole stream."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NKEYS = 65536


def make_stream(n, nmetrics, seed=11, outliers=False):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, nmetrics + 1)
    ids = rng.choice(nmetrics, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(np.log(1e5) + 0.002 * ids, 1.0)
    if outliers:
        # ONE +1e140 sample in one name and ONE negative sample in another used to widen the
        # window of EVERY row (one global [wlo, whi]); with per-row windows they widen two rows only
        v[n // 3] = 1e140
        ids[n // 3] = 0
        v[2 * n // 3] = -5e6
        ids[2 * n // 3] = nmetrics - 1
    else:
        v[::97] *= -1.0   # signed keys too
    return ids, v


def rows_and_ranges(ids, v, nmetrics):
    import oracle
    rows = oracle.histogram_pairs(ids, v, nmetrics)
    ranges = np.zeros((nmetrics, 2), dtype=np.int32)
    for m in range(nmetrics):
        nz = np.nonzero(rows[m])[0]
        ranges[m] = (nz[0], nz[-1]) if nz.size else (NKEYS, 0)
    return rows, ranges


def run(rank, world, port, plan, nmetrics, n, out_dir, outliers=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from loghisto_amd import merge
        ids, v = make_stream(n, nmetrics, outliers=outliers)
        lo, hi = n * rank // world, n * (rank + 1) // world     # data-parallel slice of the stream
        rows, ranges = rows_and_ranges(ids[lo:hi], v[lo:hi], nmetrics)
        t_rows = torch.from_numpy(rows.view(np.int64))
        t_ranges = torch.from_numpy(ranges)
        first, last = merge.merge_rows(t_rows, t_ranges, plan=plan)
        info = dict(merge.last_info)
        want_rows, want_ranges = rows_and_ranges(ids, v, nmetrics)
        assert np.array_equal(t_ranges.numpy(), want_ranges), "merged ranges"
        got = t_rows.numpy().view(np.uint64)
        if plan == "allreduce":
            assert (first, last) == (0, nmetrics)
        else:
            # the ownership the C-ABI front-end returns for the same merged ranges (equal packed cells per block):
            # the rule is stated in tests/test_merge_gloo.py::expected_blocks, independently of merge.py
            # -- in WIRE WORDS: a row travels at the narrowest cell width that holds world x its largest per-rank cell
            from tests.test_merge_gloo import expected_blocks, expected_words
            wr0 = want_ranges.astype(np.int64)
            rowmax = np.zeros(nmetrics, dtype=np.int64)
            for r in range(world):
                a, b = n * r // world, n * (r + 1) // world
                rowmax = np.maximum(rowmax, rows_and_ranges(ids[a:b], v[a:b], nmetrics)[0].max(axis=1).astype(np.int64))
            words, _ = expected_words(np.clip(wr0[:, 1] - wr0[:, 0] + 1, 0, None), rowmax, world)
            assert info["packed_words"] == int(words.sum()) < info["packed_cells"], info
            brow = expected_blocks(words, world)
            assert (first, last) == (brow[rank], brow[rank + 1]), ((first, last), brow)
            assert info["owned_rows_by_rank"] == [(brow[k], brow[k + 1]) for k in range(world)]
        assert np.array_equal(got[first:last], want_rows[first:last]), "merged rows"
        # every rank's owned block together covers all names exactly once
        cover = torch.zeros(nmetrics, dtype=torch.int64)
        cover[first:last] = 1
        dist.all_reduce(cover)
        expect = world if plan == "allreduce" else 1
        assert bool((cover == expect).all())
        # what travelled: the sum of the per-row merged windows, nothing else
        wr = want_ranges.astype(np.int64)
        cells = int(np.clip(wr[:, 1] - wr[:, 0] + 1, 0, None).sum())
        assert info["packed_cells"] == cells, (info, cells)
        import json
        open(os.path.join(out_dir, f"ok_{plan}_{rank}"), "w").write(json.dumps(info))
    finally:
        dist.destroy_process_group()
