"""Runs the C++ restatement of the reference's metrics_test.go (tests/cpp/metrics_test.cc) against the
C++ host layer (include/loghisto.hpp).  Host-only tests run everywhere; the histogram tests and the
config-5 driver need the GPU."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "loghisto_amd", "build")


def run(args, timeout=300):
    return subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_cpp_host_layer_cpu_tests(native_lib):
    r = run([os.path.join(BUILD, "metrics_test"), "--cpu"])
    print(r.stdout)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("PASS ") >= 9 and "FAIL" not in r.stdout


@pytest.mark.gpu
def test_cpp_host_layer_all_tests(native_lib, torch_cuda):
    r = run([os.path.join(BUILD, "metrics_test")])
    print(r.stdout)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("PASS ") >= 14 and "FAIL" not in r.stdout


@pytest.mark.gpu
def test_config5_driver_is_lossless(native_lib, torch_cuda):
    # short, throttled run of BASELINE config 5: every event lands in exactly one interval
    r = run([os.path.join(BUILD, "c5_driver"), "--threads", "8", "--seconds", "2.5", "--rate", "2e7",
             "--interval-ms", "500"])
    print(r.stdout)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert r.returncode == 0 and res["lossless"] is True, res
    assert res["events_accounted"] == res["events_submitted"] > 1e7
    assert res["dropped_intervals"] == 0 and res["submit_failures"] == 0
    assert res["intervals_emitted"] >= 4 and res["graphite_lines"] == res["keys_emitted"]
