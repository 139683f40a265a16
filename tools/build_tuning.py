"""Builds loghisto_amd/build/liblhgpu_tuning.so: the library with -DLH_TUNING (ablation bits of the scatter kernels
reachable through lh_set_option(100, bits), dispatch steered by LH_* environment variables).  Tools only
(tools/sweep.py --lib ...): results are WRONG with any ablation bit set; the product build has none of this."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loghisto_amd import build as B  # noqa: E402


def main():
    bdir = os.path.join(B._HERE, "build", "tuning")
    os.makedirs(bdir, exist_ok=True)
    objs = []
    for src, extra in B._UNITS:
        s = os.path.join(B.CSRC, src)
        o = os.path.join(bdir, os.path.splitext(src)[0].replace("/", "_") + ".o")
        subprocess.check_call([B._hipcc()] + B._COMMON + extra + ["-DLH_TUNING", "-I", B.INCLUDE, "-c", s, "-o", o])
        objs.append(o)
    out = os.path.join(B._HERE, "build", "liblhgpu_tuning.so")
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
