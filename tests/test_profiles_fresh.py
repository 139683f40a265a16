"""The committed rocprofv3 --pmc summaries that bench.py reports as `roofline.traffic` were taken on THIS tree's kernel
sources: every summary records the hashes of the sources it was measured on (tools/profile_round.sh) and bench.py refuses
one whose hashes differ (`traffic_source: "stale ..."`, traffic null).  This test is the same check on the CPU box, so that
a kernel edit without a fresh `tools/round.sh profile` run shows up here and not only as a null in the driver's line."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path,kind", [(bench.K1_PMC, "k1"), (bench.C3_PMC, "c3"), (bench.C4_PMC, "c4"),
                                       (bench.C4_1E9_PMC, "c4")])
def test_pmc_summary_was_taken_on_these_kernel_sources(path, kind):
    j = json.load(open(os.path.join(ROOT, path)))
    assert bench.pmc_stale(j, kind) is None, bench.pmc_stale(j, kind)
    ratio = j.get("read_over_algorithmic") or j.get("traffic_over_algorithmic")
    assert 0.99 < ratio < 2.5, ratio      # bytes moved over algorithmic bytes: K1 1.00, C3 1.18, C4 slice 2.02, 1e9 pairs 1.69
