"""N>1 path on CPU: world-size-2 gloo runs of the multi-GPU merge."""
import os
import socket

import pytest
import torch.multiprocessing as mp

from tests import _merge_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("plan", ["allreduce", "reduce_scatter"])
@pytest.mark.parametrize("nmetrics", [1, 5, 8])
def test_merge_world2(tmp_path, plan, nmetrics):
    world = 2
    mp.spawn(_merge_worker.run, args=(world, _free_port(), plan, nmetrics, 60_000, str(tmp_path)),
             nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok_{plan}_{r}" for r in range(world)]


@pytest.mark.parametrize("plan", ["allreduce", "reduce_scatter"])
def test_outliers_widen_one_row_not_the_matrix(tmp_path, plan):
    """a single +1e140 sample and a single negative sample must not blow the merge up.
    Same stream with and without the two outliers: the bytes that travel stay within 2x of the clean case
    (per-row windows: the two affected rows grow, the other 62 do not)."""
    import json
    world, M, n = 2, 64, 120_000
    clean, dirty = tmp_path / "clean", tmp_path / "dirty"
    clean.mkdir()
    dirty.mkdir()
    mp.spawn(_merge_worker.run, args=(world, _free_port(), plan, M, n, str(clean), False), nprocs=world, join=True)
    mp.spawn(_merge_worker.run, args=(world, _free_port(), plan, M, n, str(dirty), True), nprocs=world, join=True)
    a = json.loads(open(clean / f"ok_{plan}_0").read())
    b = json.loads(open(dirty / f"ok_{plan}_0").read())
    assert b["widest_row"] > 10 * a["widest_row"]           # the outlier row really is wide ...
    assert b["send_bytes"] <= 2 * a["send_bytes"], (a, b)   # ... and the exchange does not care
    # under one global window the dirty case would move 64 rows x ~47 000 bins: > 30x the clean case
    assert b["packed_cells"] < 64 * b["widest_row"] / 8


def expected_blocks(width, world):
    """The reduce-scatter's owner blocks, stated independently of both front-ends: block
    k of `world` starts at the first row whose exclusive prefix of packed cells reaches k / world of the total;
    block 0 starts at row 0 and the last block ends at the last row.  Returns brow[world + 1]."""
    import numpy as np
    nrows = len(width)
    P = np.concatenate([[0], np.cumsum(width)]).astype(np.int64)
    total = int(P[-1])
    brow = [0]
    for k in range(1, world):
        target = total // world * k + total % world * k // world
        brow.append(int(np.nonzero(P >= target)[0][0]))
    return brow + [nrows]


def expected_words(width, rowmax, world):
    """Wire words of every row, stated independently of both front-ends: a row travels at 8, 16 or 32 bits
    per cell, the narrowest that holds world x (the largest cell any rank holds in it), in uint32 words."""
    import numpy as np
    bound = np.asarray(rowmax, dtype=np.int64) * world
    bits = np.where(bound <= 0xff, 8, np.where(bound <= 0xffff, 16, 32))
    return (np.asarray(width, dtype=np.int64) * bits + 31) // 32, bits


def test_window_plan_in_wire_words():
    """plan_windows with per-row cell widths: the prefix, the totals and the owner blocks are those of the words."""
    import numpy as np
    import torch
    from loghisto_amd import merge
    rng = np.random.default_rng(5)
    for nrows in (1, 6, 77, 1000):
        lo = rng.integers(0, 60000, nrows)
        width = rng.integers(0, 1500, nrows)
        width[rng.random(nrows) < 0.15] = 0
        hi = lo + width - 1
        lo[width == 0], hi[width == 0] = 65536, 0
        ranges = torch.from_numpy(np.stack([lo, hi], 1).astype(np.int32))
        rowmax = (10.0 ** rng.uniform(0, 6, nrows)).astype(np.int64)
        for world in (2, 3, 8):
            words, bits = expected_words(width, rowmax, world)
            got_bits = merge.row_bits(torch.from_numpy(rowmax), world, 1 << 20)
            assert np.array_equal(got_bits.numpy(), bits)
            assert set(np.unique(bits)) <= {8, 16, 32} and (nrows < 50 or len(set(np.unique(bits))) == 3)
            W = merge.plan_windows(ranges, world, "reduce_scatter", got_bits)
            assert np.array_equal(W["words"].numpy(), words) and W["total"] == int(words.sum())
            assert W["total_c"] == int(width.sum()) and np.array_equal(W["width"].numpy(), width)
            brow = expected_blocks(words, world)
            assert W["brow"].tolist() == brow
            assert W["bmax"] == max(int(words[brow[k]:brow[k + 1]].sum()) for k in range(world))
            assert W["bmax_c"] == max(int(width[brow[k]:brow[k + 1]].sum()) for k in range(world))
        # an interval whose largest per-rank sample count x ranks can reach 2^32: every cell a uint64 word
        b64 = merge.row_bits(torch.from_numpy(rowmax), 8, 1 << 30)
        assert bool((b64 == 64).all())
        W = merge.plan_windows(ranges, 8, "reduce_scatter", b64)
        assert np.array_equal(W["words"].numpy(), width)


def test_window_plan_for_2_4_8_ranks():
    """Pure plan arithmetic: loghisto_amd.merge.plan_windows must give per-row widths, their prefix and the owner
    blocks of EQUAL PACKED CELLS that k_merge_plan computes on the device for lh_snapshot_merge (the device side of
    the same comparison: tests/_stub_merge_driver.py); ragged name counts, empty rows, all-empty matrices."""
    import numpy as np
    import torch
    from loghisto_amd import merge
    rng = np.random.default_rng(3)
    for nrows in (1, 5, 8, 61, 1000):
        for variant in ("random", "zipf-like", "all-empty"):
            lo = rng.integers(0, 60000, nrows)
            hi = lo + rng.integers(0, 3000, nrows)
            if variant == "zipf-like":                       # names ranked by frequency: widest windows first
                hi = lo + (3000 / (1 + np.arange(nrows)) ** 0.3).astype(np.int64)
            empty = rng.random(nrows) < (1.0 if variant == "all-empty" else 0.2)
            lo[empty], hi[empty] = 65536, 0
            ranges = torch.from_numpy(np.stack([lo, hi], 1).astype(np.int32))
            width = np.where(empty, 0, hi - lo + 1)
            for world in (2, 4, 8):
                for plan in ("allreduce", "reduce_scatter"):
                    W = merge.plan_windows(ranges, world, plan)
                    assert np.array_equal(W["width"].numpy(), width)
                    assert np.array_equal(W["P"].numpy(), np.concatenate([[0], np.cumsum(width)]))
                    assert W["total"] == int(width.sum())
                    if plan == "allreduce":
                        assert W["nblocks"] == 1 and W["bmax"] == W["total"]
                        assert merge.owned_rows(W, 0) == (0, nrows) == merge.owned_rows(W, world - 1)
                        continue
                    brow = expected_blocks(width, world)
                    assert W["nblocks"] == world and W["brow"].tolist() == brow
                    blocks = [int(width[brow[k]:brow[k + 1]].sum()) for k in range(world)]
                    assert np.array_equal(np.diff(W["bstart"].numpy()), blocks) and W["bmax"] == max(blocks)
                    # equal shares: no block exceeds total / world by more than one row's window
                    assert W["bmax"] <= -(-W["total"] // world) + int(width.max(initial=0))
                    covered = []
                    for r in range(world):
                        a, b = merge.owned_rows(W, r)
                        assert (a, b) == (brow[r], brow[r + 1])
                        covered.extend(range(a, b))
                    assert covered == list(range(nrows))
                    rows = torch.arange(nrows)
                    blk = merge.block_of_rows(W, rows).numpy()
                    assert all(brow[blk[r]] <= r < brow[blk[r] + 1] for r in range(nrows))


def test_name_blocks_partition():
    from loghisto_amd import merge
    for nrows in (1, 7, 8, 65536):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = merge.name_blocks(nrows, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(nrows))
