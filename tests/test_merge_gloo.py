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


def test_window_plan_for_2_4_8_ranks():
    """Pure plan arithmetic (loghisto_amd.merge.plan_windows == k_merge_plan on the device): per-row widths,
    prefix, owner blocks of ceil(M/world) rows, padded block size; ragged name counts and empty rows."""
    import numpy as np
    import torch
    from loghisto_amd import merge
    rng = np.random.default_rng(3)
    for nrows in (1, 5, 8, 61, 1000):
        lo = rng.integers(0, 60000, nrows)
        hi = lo + rng.integers(0, 3000, nrows)
        empty = rng.random(nrows) < 0.2
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
                    assert W["per"] == nrows and W["bmax"] == W["total"]
                    continue
                per = -(-nrows // world)
                assert W["per"] == per and W["nblocks"] == world
                blocks = [int(width[k * per:(k + 1) * per].sum()) for k in range(world)]
                assert np.array_equal(np.diff(W["bstart"].numpy()), blocks) and W["bmax"] == max(blocks)
                covered = []
                for r in range(world):
                    a, b = merge.owned_rows(nrows, r, world)
                    assert (a, b) == (min(r * per, nrows), min((r + 1) * per, nrows))
                    covered.extend(range(a, b))
                assert covered == list(range(nrows))


def test_owned_rows_partition():
    from loghisto_amd import merge
    for nrows in (1, 7, 8, 65536):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = merge.owned_rows(nrows, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(nrows))
