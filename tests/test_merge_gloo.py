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


def test_owned_rows_partition():
    from loghisto_amd import merge
    for nrows in (1, 7, 8, 65536):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = merge.owned_rows(nrows, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(nrows))
