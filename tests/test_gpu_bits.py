"""Compress parity over uniformly random float64 BIT PATTERNS (every exponent, NaN payloads, infinities,
subnormals, both signs, the int16 wrap-around region): both device routes against the oracle."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def test_compress_random_bit_patterns(native_lib, torch_cuda):
    torch = torch_cuda
    import loghisto_amd
    rng = np.random.default_rng(2024)
    bits = rng.integers(0, 2 ** 64, size=4_000_000, dtype=np.uint64)
    # make sure the special classes are all present
    bits[:8] = [0x7FF0000000000000, 0xFFF0000000000000, 0x7FF8000000000001, 0xFFF8000000000000,
                0x0000000000000001, 0x8000000000000001, 0x7FEFFFFFFFFFFFFF, 0xFFEFFFFFFFFFFFFF]
    v = bits.view(np.float64)
    want = oracle.compress_many(v)
    with loghisto_amd.Engine(max_metrics=1) as e:
        dv = torch.from_numpy(v).cuda()
        k1 = torch.empty(v.size, dtype=torch.int16, device="cuda")
        k2 = torch.empty(v.size, dtype=torch.int16, device="cuda")
        e.compress_device(dv, k1, v.size)
        e.compress_device(dv, k2, v.size, golog=True)
        e.sync()
        torch.cuda.synchronize()
        assert np.array_equal(k1.cpu().numpy(), want)
        assert np.array_equal(k2.cpu().numpy(), want)
        # and through the ingest kernel: bucket rows of the same values
        e.submit_device(0, dv)
        with e.flip() as snap:
            row = snap.dense_row(0)
    assert np.array_equal(row, oracle.histogram_dense(v))
