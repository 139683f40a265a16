"""Writes the golden fixtures under tests/golden/.

Two kinds of content, kept apart in the JSON:
  * "reference": numbers lifted verbatim from the reference's own tests and docs
    (file:line cited per entry).  These PIN the oracle.
  * "oracle_kat": known-answer vectors produced by the oracle itself (SURVEY.md
    8c lists the same values, measured independently during the survey).  These
    freeze the oracle against regressions; they are not reference evidence.

The reference is Go and cannot be executed in this image (no go toolchain), so
there is nothing to import from /root/reference; this script only needs the
oracle.  Run from the repo root:  python tests/golden/make_goldens.py
"""
import json
import math
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def hexd(x: float) -> str:
    return struct.pack(">d", x).hex()


def main():
    ref = {
        # readme.md:35-43 and print_benchmark.go:34-39: full-precision decompress outputs
        "decompress_doc_values": [
            {"key": 1702, "value": 2.4642914167480484e+07, "src": "readme.md:35"},
            {"key": 850, "value": 4913.768840299134, "src": "readme.md:36"},
            {"key": 691, "value": 1001.2472422902518, "src": "readme.md:37"},
            {"key": 428, "value": 71.24044000732538, "src": "readme.md:38"},
            {"key": 422, "value": 67.03348428941965, "src": "readme.md:39"},
            {"key": 420, "value": 65.68633104092515, "src": "readme.md:40"},
            {"key": 416, "value": 63.07152259993664, "src": "readme.md:41"},
            {"key": 409, "value": 58.739891704145194, "src": "readme.md:42"},
            {"key": -649, "value": -657.5233632152207, "src": "readme.md:43"},
            {"key": 1750, "value": 3.982478339757623e+07, "src": "print_benchmark.go:34"},
            {"key": 1747, "value": 3.864778314316012e+07, "src": "print_benchmark.go:35"},
            {"key": 1505, "value": 3.4366224772310276e+06, "src": "print_benchmark.go:36"},
            {"key": 1452, "value": 2.0228126576114902e+06, "src": "print_benchmark.go:37"},
            {"key": 1306, "value": 469769.7083161708, "src": "print_benchmark.go:38"},
            {"key": 1177, "value": 129313.15075081984, "src": "print_benchmark.go:39"},
        ],
        # metrics_test.go:111-149
        "test_percentile": {
            "metrics": {"10": 9000, "25": 900, "33": 90, "47": 9, "500": 1},
            "expected": {"0": 10, "0.99": 25, "0.999": 33, "0.9991": 47, "0.9999": 47, "1": 500},
            "tolerance": 0.01, "src": "metrics_test.go:111-149"},
        # metrics_test.go:151-172
        "test_compress": {"values": [-421408208120481, -1, 0, 1, 214141241241241], "tolerance": 0.01,
                          "src": "metrics_test.go:151-172"},
        # metrics_test.go:289-319
        "test_processed_broadcast": {"samples": [33, 59, 330000], "int_sum": 331132, "int_agg_avg": 110377,
                                     "int_count": 3, "src": "metrics_test.go:289-319"},
        # metrics.go:145-155
        "default_percentiles": {"%s_min": 0, "%s_50": .5, "%s_75": .75, "%s_90": .9, "%s_95": .95, "%s_99": .99,
                                "%s_99.9": .999, "%s_99.99": .9999, "%s_max": 1},
    }
    kat_in = [33, 59, 330000, 123, 1, -1, 0.0, -0.0, 0.005, 0.00502, 0.5, 0.51, 1e9, 1e12, 9.2e18,
              -421408208120481.0, 214141241241241.0, 1e142, 2.0196e142, 2.03e142, 3e142, 1e200,
              1.7976931348623157e308, -1.7976931348623157e308, math.inf, -math.inf, math.nan, 4.9e-324,
              -3e142, 2.2250738585072014e-308]
    tx = oracle.thresholds()
    kat = {
        "compress": [{"bits": hexd(v), "repr": repr(v), "key": oracle.compress(v)} for v in kat_in],
        "decompress": [{"key": k, "bits": hexd(oracle.decompress(k))}
                       for k in (0, 1, -1, 69, 353, 409, 1271, 4367, 32767, -32767, -32768, 12345, -20000)],
        "thresholds_x": [{"j": j, "bits": hexd(float(tx[j]))}
                         for j in (1, 2, 69, 70, 1000, 4367, 32767, 32768, 65536, 70978)],
        "broadcast_sum_bits": hexd(oracle.process_dense(oracle.histogram_dense([33, 59, 330000]), [0.5])["sum"]),
    }
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump({"reference": ref, "oracle_kat": kat}, f, indent=1)
    print("wrote reference_vectors.json")


if __name__ == "__main__":
    main()
