"""Runs the survey-path parity tests against another build of the library (a tuning variant under evaluation):
    python tools/run_tests_with_lib.py loghisto_amd/build/liblhgpu_tuning_x.so [pytest args...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loghisto_amd import _native  # noqa: E402

_native.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest  # noqa: E402

args = sys.argv[2:]
if not any(a.startswith("tests/") for a in args):
    args = ["tests/test_gpu_part2.py", "-k", "direct or clustered or threshold"] + args
sys.exit(pytest.main(["-x", "-q"] + args))
