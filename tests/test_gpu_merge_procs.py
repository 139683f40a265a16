"""lh_snapshot_merge -- the C-ABI merge itself, not loghisto_amd/merge.py -- across PROCESS boundaries (VERDICT r5 next
#7): N processes on the one GPU, one engine each, joined by the process-shared mode of tests/cpp/rccl_stub.cc (staging
areas and barrier in a POSIX shared-memory segment).  Every rank buckets its slice of a seeded stream for ALL names
(data-parallel ingest), the merge leaves it the rows it owns, and every cell of every owned row has to
equal the oracle's histogram of the WHOLE stream; the owner blocks tile [0, M) in rank order."""
import json
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

import oracle
from tests._merge_proc_worker import stream

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nranks,M,plan", [(2, 2500, "reduce_scatter"), (2, 2500, "allreduce"), (3, 700, "reduce_scatter")])
def test_c_abi_merge_between_processes(native_lib, torch_cuda, tmp_path, nranks, M, plan):
    shm = f"/lh_merge_{uuid.uuid4().hex[:12]}"
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_merge_proc_worker.py"), str(r), str(nranks),
                               str(M), plan, shm, str(tmp_path)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), outs
    ids, v = stream(M, nranks)
    want = oracle.histogram_pairs(ids, v, M)
    per_name = np.bincount(ids, minlength=M)
    prev_last = 0
    for r in range(nranks):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        first, last = int(z["first"]), int(z["last"])
        if plan == "allreduce":
            assert (first, last) == (0, M)
        else:
            assert first == prev_last and last >= first, (r, first, last)   # the owner blocks tile [0, M) in rank order
            prev_last = last
        off, keys, counts = z["off"], z["keys"], z["counts"]
        dense = np.zeros((last - first, 65536), dtype=np.uint64)
        rows = np.repeat(np.arange(last - first), np.diff(off).astype(np.int64))
        dense[rows, oracle.key_to_bin(keys)] = counts
        assert np.array_equal(dense, want[first:last]), f"rank {r}: rows differ"
        assert np.array_equal(z["count"].astype(np.int64), per_name[first:last])
        info = json.loads(str(z["info"]))
        assert info["cell_bytes"] == 4 and info["packed_cells"] > 0, info
    if plan != "allreduce":
        assert prev_last == M
