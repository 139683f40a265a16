"""Two-level partitioned ingest (more than 1 024 names: lh_kernels_part.hip P1b) against the oracle.
The dense oracle matrix would be gigabytes at these name counts, so rows are checked for a sample of
names (hot, middle, cold, first/last of partitions) and conservation is checked for all of them."""
import math

import numpy as np
import pytest

from tests.conftest import thresholds_until_round_6

import oracle
from loghisto_amd import _native as N

pytestmark = pytest.mark.gpu
PCTS = [0.0, .5, .9, .99, 1.0]


def _dev(torch, a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("M,n,kind", [
    (4096, 3_000_001, "lognormal"),      # 16 names per level-1 partition -> 4 sub-partitions
    (2049, 1_000_000, "signed"),         # odd name count: 9 names per partition -> 4 sub-partitions, ragged
    (65536, 4_000_000, "lognormal"),     # config 4's per-GPU name count: 256 -> 64 sub-partitions, 16 384 partitions
    (65536, 500_000, "edge"),            # includes the record value 0xffffffff and the last name
    (20000, 2_000_000, "constant"),
])
def test_two_level_partitioned_ingest(native_lib, torch_cuda, M, n, kind, monkeypatch):
    import loghisto_amd
    # the engine only uses the second level above 8 192 names (it is slower below); force it for every case
    rng = np.random.default_rng(M + n)
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    if kind == "lognormal":
        v = rng.lognormal(math.log(1e5) + 1e-4 * ids, 1.0)
    elif kind == "signed":
        v = rng.normal(0, 1e4, n)
    elif kind == "constant":
        v = 1000.0 + (ids % 7)
    else:
        v = rng.lognormal(math.log(1e5), 1.0, n)
        ids[:5000] = M - 1                # name 65535 = partition 255, local 255
        v[:2500] = 2.0196e142             # key +32767 -> bin 0xffff: the record is 0xffffffff
        v[2500:5000] = -2.0196e142
        ids[5000:6000] = M - 256          # same sub-partition pattern, partition 0
    sample = sorted({0, 1, 2, 255, 256, 257, 1023, 1024, M // 2, M // 2 + 1, M - 257, M - 256, M - 2, M - 1} & set(range(M)))
    with loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16) as e:
        thresholds_until_round_6(e)
        e.set_option(N.OPT_PART_V2, 0)
        e.set_option(N.OPT_TWO_LEVEL_ABOVE, 0)
        e.set_option(N.OPT_PART_V3, 0)      # the first generation's second level is what this file tests
        e.submit_pairs_device(_dev(torch_cuda, ids), _dev(torch_cuda, v))
        e.sync()
        c = e.counters()
        assert c["samples_partitioned"] == n
        with e.flip() as snap:
            got = snap.extract(PCTS, M)
            rows = {m: snap.dense_row(m) for m in sample}
    per_name = np.bincount(ids, minlength=M)
    assert np.array_equal(got["count"].astype(np.int64), per_name)          # every sample, in the right row
    assert int(got["count"].sum()) == n
    for m in sample:
        want = oracle.histogram_dense(v[ids == m])
        assert np.array_equal(rows[m], want), m
        ref = oracle.process_dense(want, PCTS)
        if ref["count"]:
            assert np.array_equal(got["pvals"][m].view(np.uint64), ref["pvals"].view(np.uint64)), m
            assert abs(got["sum"][m] - ref["sum"]) <= 1e-12 * max(abs(ref["sum"]), 1e-300), m
