"""Builds loghisto_amd/build/liblhgpu_tuning.so: the library with -DLH_TUNING (ablation bits of the scatter kernels
reachable through lh_set_option(100, bits), dispatch steered by LH_* environment variables).  Tools only
(tools/sweep.py --lib ...): results are WRONG with any ablation bit set; the product build has none of this.

    python tools/build_tuning.py [-DNAME=VALUE ...] [--name SUFFIX]    ->  build/liblhgpu_tuning[_SUFFIX].so"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loghisto_amd import build as B  # noqa: E402


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    suffix = ""
    if "--name" in sys.argv:
        suffix = "_" + sys.argv[sys.argv.index("--name") + 1]
    bdir = os.path.join(B._HERE, "build", "tuning" + suffix)
    os.makedirs(bdir, exist_ok=True)
    objs = []
    for src, extra in B._UNITS:
        s = os.path.join(B.CSRC, src)
        o = os.path.join(bdir, os.path.splitext(src)[0].replace("/", "_") + ".o")
        subprocess.check_call([B._hipcc()] + B._COMMON + extra + defs + ["-DLH_TUNING", "-I", B.INCLUDE, "-c", s, "-o", o])
        objs.append(o)
    out = os.path.join(B._HERE, "build", "liblhgpu_tuning" + suffix + ".so")
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
