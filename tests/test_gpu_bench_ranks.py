"""The N-rank path of bench.py's config-4 workload, N ranks as threads on one GPU (tests/_bench_ranks_driver.py):
what the driver's `python bench.py --gpus N` runs on a multi-GPU node, minus RCCL itself (the stub of
tests/cpp/rccl_stub.cc stands in) and torch.distributed's transport (a thread rendezvous stands in).
Reference semantics: cells are a commutative sum (metrics.go:278, 292); SURVEY.md 8(e)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nranks,names", [(2, 8192 + 1), (4, 65536)])
def test_c4_step_with_ranks_as_threads(native_lib, torch_cuda, nranks, names):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_bench_ranks_driver.py"), str(nranks), str(names),
                        "3e6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["parity"]["exact"] and res["parity"]["cells_after_merge_exact"]
    assert res["parity"]["rows_checked_cell_by_cell"] >= 40          # every probe row is owned by exactly one rank
    rows = res["owned_rows"]
    assert rows[0][0] == 0 and rows[-1][1] == names and all(rows[i][1] == rows[i + 1][0] for i in range(nranks - 1))
    m = res["merge"]
    assert m["cell_bytes"] == 4 and m["padded_words"] >= m["packed_words"] and m["padding_ratio"] <= 1.3
    # Zipf names, a few million pairs: nearly every row's counts fit 8 bits x the rank count -- packed bytes <= 0.6 x
    # of a whole word per cell
    assert m["rows_8bit"] > 0.5 * m["occupied_rows"] and m["wire_bytes_per_cell"] <= 0.6 * 4, m
    assert m["device_ms"]["span_ms"] > 0


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_headline_step_with_ranks_as_threads(native_lib, torch_cuda, nranks):
    """`bench.py --gpus N`'s top-level line: weak scaling of the C2 headline.  Every rank buckets its own slice of the
    one metric, lh_snapshot_merge all-reduces the row, every rank extracts; the merged row on every rank equals the
    oracle over the concatenated slices.  N = 1 runs the same function with no merge: it IS the N = 1 line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_bench_ranks_driver.py"), str(nranks), "c2", "3000001"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["parity"]["exact"]
    assert res["parity"]["samples_checked"] == 3000001 * nranks
    assert res["config"]["ranks"] == nranks and res["config"]["metrics"] == 1
    if nranks == 1:
        assert res["config"]["merge"] == "none" and res["merge"] is None
    else:
        assert res["config"]["merge"].startswith("c-abi") and res["parity"]["ranks_with_the_exact_merged_row"] == nranks
        assert res["merge"]["cell_bytes"] == 4 and 0 < res["merge"]["packed_cells"] < 4000
        assert len(set(res["values"])) == 1            # every rank reports the job's value (max-over-ranks clock)


def test_whole_job_with_ranks_as_threads(native_lib, torch_cuda):
    """bench.run_job, i.e. everything `python bench.py --gpus 2` does between set-up and printing: the headline on two
    ranks, then config 4 on the same ranks with its one-rank reference under secondary.c4 -- one communicator for both."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_bench_ranks_driver.py"), "2", "job", "3000001"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] and res["parity"]["exact"] and res["config"]["ranks"] == 2
    assert res["c4_parity"]["exact"] and res["c4_ranks"] == 2 and res["c4_merge"].startswith("c-abi")
    assert res["c4_one_rank_reference"].get("value", 0) > 0 and 0 < res["c4_efficiency"] < 10
