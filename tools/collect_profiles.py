"""Copies the evidence of a `tools/round.sh profile` run (gpurun_out/round/) into profiles/ as rNN_* and writes
profiles/rNN_MANIFEST.json: which files describe the FINAL tree (each carries the tree stamp;
tests/test_profiles_fresh.py holds them to the tree) and which are experiment records of earlier trees of the round.

    python tools/collect_profiles.py r06"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FILES = {  # gpurun_out/round/<src> -> profiles/<tag>_<dst>
    "bench.json": "bench.json", "bench_steps20_warmup5.json": "bench_steps20_warmup5.json", "c4_bench.json": "bench_c4_one_rank.json",
    "bench_under_trace.json": "bench_under_trace.json", "kernel_trace.txt": "kernel_trace.txt", "k1_pmc.json": "k1_pmc.json",
    "c3_kernel_trace.txt": "c3_kernel_trace.txt", "c3_pmc.json": "c3_pmc.json", "c4_kernel_trace.txt": "c4_kernel_trace.txt",
    "c4_pmc.json": "c4_pmc.json", "c4_names_1e9_kernel_trace.txt": "c4_names_1e9_kernel_trace.txt",
    "c4_names_1e9_pmc.json": "c4_names_1e9_pmc.json", "read_ceiling.jsonl": "read_ceiling.jsonl", "sweep_final.jsonl": "sweep_final.jsonl", "first_calls.txt": "first_calls.txt",
}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "round")
    final = []
    for s, d in FILES.items():
        sp = os.path.join(src, s)
        if not os.path.exists(sp) or os.path.getsize(sp) == 0:
            print("missing:", s)
            continue
        shutil.copyfile(sp, os.path.join(ROOT, "profiles", f"{tag}_{d}"))
        final.append(f"{tag}_{d}")
    man_path = os.path.join(ROOT, "profiles", f"{tag}_MANIFEST.json")
    man = json.load(open(man_path)) if os.path.exists(man_path) else {}
    man.update(tree_stamp=bench.tree_stamp(), final_tree=sorted(final))
    man.setdefault("experiments", {})
    json.dump(man, open(man_path, "w"), indent=1)
    print(json.dumps(man, indent=1))


if __name__ == "__main__":
    main()
