"""The committed evidence that claims to describe THIS tree was measured on this tree.

Every file tools/profile_round.sh produces carries the tree stamp (bench.tree_stamp(): one digest of the sources a result
can depend on) -- JSON: "tree_stamp"; JSON lines: a first line {"tree_stamp": ...}; text: a first line "# tree_stamp: ..." --
and profiles/r06_MANIFEST.json lists which committed files are such final-tree evidence (bench copies, kernel traces, PMC
summaries, the sweep, the read ceiling) and which are experiment records of EARLIER trees of the round (ablations, A/B
runs: they name the commit they were measured on and are exempt).  A kernel edit without a fresh `tools/round.sh profile`
run shows up here, on the CPU box, and not only as a null `traffic` in the driver's line.  (Round 5's sweep was committed
before two later kernel changes and still quoted as "final": VERDICT r5 weak #8.)"""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MANIFEST = os.path.join(ROOT, "profiles", "r06_MANIFEST.json")


def _stamp_of(path):
    with open(path) as fh:
        first = fh.readline().strip()
        rest = fh.read()
    if path.endswith(".txt"):
        assert first.startswith("# tree_stamp: "), f"{path}: no stamp line"
        return first.split(": ", 1)[1]
    if path.endswith(".jsonl"):
        return json.loads(first)["tree_stamp"]
    return json.loads(first + rest)["tree_stamp"]


def _manifest():
    return json.load(open(MANIFEST)) if os.path.exists(MANIFEST) else {"final_tree": [], "experiments": {}}


def test_manifest_lists_every_r06_file():
    man = _manifest()
    listed = set(man["final_tree"]) | set(man["experiments"]) | {"r06_MANIFEST.json"}
    have = {f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r06_")}
    assert have == listed, (sorted(have - listed), sorted(listed - have))
    for f, why in man["experiments"].items():
        assert why, f   # every exempt file says what tree it was measured on


@pytest.mark.parametrize("name", _manifest()["final_tree"])
def test_final_tree_evidence_carries_this_trees_stamp(name):
    assert _stamp_of(os.path.join(ROOT, "profiles", name)) == bench.tree_stamp(), \
        f"{name} was measured on other sources: run tools/round.sh profile + tools/collect_profiles.py r06"


@pytest.mark.parametrize("path,kind", [(bench.K1_PMC, "k1"), (bench.C3_PMC, "c3"), (bench.C4_PMC, "c4"),
                                       (bench.C4_1E9_PMC, "c4")])
def test_pmc_summary_was_taken_on_these_kernel_sources(path, kind):
    j = json.load(open(os.path.join(ROOT, path)))
    assert bench.pmc_stale(j, kind) is None, bench.pmc_stale(j, kind)
    ratio = j.get("read_over_algorithmic") or j.get("traffic_over_algorithmic")
    assert 0.99 < ratio < 2.5, ratio      # bytes moved over algorithmic bytes: K1 1.00, C3 ~1.15, C4 slice ~2.0, 1e9 pairs ~1.6
