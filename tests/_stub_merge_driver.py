"""N ranks of lh_snapshot_merge on ONE GPU (run as a subprocess by tests/test_gpu_merge.py).

N engines on device 0 play N ranks; every rank buckets ITS slice of a seeded stream for ALL names on the GPU,
flips, and calls lh_snapshot_merge from its own thread with a communicator of tests/cpp/rccl_stub.cc (an
in-process stand-in for RCCL that reduces the threads' device buffers).  Checked against the oracle run on the
WHOLE stream: merged ranges, every cell of the rows a rank ends up owning, extract on the owned rows, the
bytes that travelled (per-row windows: outliers widen two rows, not the matrix).

The wire: a row travels at 8, 16 or 32 bits per cell, the narrowest that holds nranks x (its largest per-rank cell) --
the expected class of every row, the packed words and the owner blocks cut on them come from the oracle's per-rank rows.

usage: python tests/_stub_merge_driver.py NRANKS NROWS PLAN OUTLIERS(0/1) [NARROW: 1 on (default), 0 off, 2 off on rank 0 only,
       3 = uint64 wire words: rank 0's sample count is unknown (lh_snapshot_mark_dirty), the all-reduced bound makes every rank send
       whole uint64 cells -- and an engine of 32-bit cells moves its snapshot to uint64 cells first]
"""
import ctypes as C
import json
import math
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nranks, M, plan, outliers = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    narrow = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    import torch
    import loghisto_amd
    import oracle
    from loghisto_amd import _native as N
    from loghisto_amd import merge
    stub_path = os.path.join(ROOT, "loghisto_amd", "build", "librccl_stub.so")
    N.check(N.lib().lh_set_rccl_library(stub_path.encode()), "lh_set_rccl_library")
    stub = C.CDLL(stub_path)
    comms = (C.c_void_p * nranks)()
    assert stub.stub_comm_create(nranks, comms) == 0

    rng = np.random.default_rng(100 * nranks + M)
    n = 400_000
    w = 1.0 / np.arange(1, M + 1)
    ids = rng.choice(M, size=n, p=w / w.sum()).astype(np.uint32)
    v = rng.lognormal(math.log(1e5) + 0.002 * ids, 1.0)
    if M > 4:
        ids[ids == 3] = 2                                   # row 3 is empty on every rank
        v[ids == 1] = 1234.5                                # name 1 in ONE cell: the largest counts of the matrix
    if outliers:
        v[n // 3], ids[n // 3] = 1e140, 0                  # one +1e140 sample in name 0
        v[2 * n // 3], ids[2 * n // 3] = -5e6, M - 1       # one negative sample in the last name
    want = oracle.histogram_pairs(ids, v, M)
    want_ranges = np.zeros((M, 2), dtype=np.int64)
    for m in range(M):
        nz = np.nonzero(want[m])[0]
        want_ranges[m] = (nz[0], nz[-1]) if nz.size else (65536, 0)
    cells = int(np.clip(want_ranges[:, 1] - want_ranges[:, 0] + 1, 0, None).sum())

    engines = [loghisto_amd.Engine(max_metrics=M, num_buffers=2, num_lanes=1, lane_samples=1 << 16)
               for _ in range(nranks)]
    snaps = []
    rowmax = np.zeros(M, dtype=np.int64)                    # the largest cell any rank holds in a row
    largest_rank = 0
    for r, e in enumerate(engines):
        lo, hi = n * r // nranks, n * (r + 1) // nranks    # data-parallel slice, all names
        if hi > lo:
            e.submit_pairs(ids[lo:hi], v[lo:hi])
            rowmax = np.maximum(rowmax, oracle.histogram_pairs(ids[lo:hi], v[lo:hi], M).max(axis=1).astype(np.int64))
        largest_rank = max(largest_rank, hi - lo)
        if narrow == 0 or (narrow == 2 and r == 0):
            e.set_option(N.OPT_MERGE_NARROW_CELLS, 0)       # one rank is enough: the bound is all-reduced
        snaps.append(e.flip())
    wide64 = narrow == 3
    if wide64:  # (a cell the row already holds: the ranges stay as they are)
        snaps[0].mark_dirty(0, 1, int(want_ranges[0, 0]), int(want_ranges[0, 0]))
    results, errors = [None] * nranks, []

    def rank_main(r):
        try:
            results[r] = snaps[r].merge_rccl(comms[r], nranks, r, M, plan=plan)
        except Exception as exc:  # noqa: BLE001
            errors.append((r, repr(exc)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors
    assert all(not t.is_alive() for t in th), "a rank never left the collective"

    covered = []
    infos = []
    widths = np.where(want_ranges[:, 0] <= want_ranges[:, 1], want_ranges[:, 1] - want_ranges[:, 0] + 1, 0)
    # ONE merge plan: the torch front-end's plan (loghisto_amd.merge.plan_windows) from the same merged ranges must give
    # every rank the [first, last) that k_merge_plan gave it on the device, and the same padded block size
    bits = merge.row_bits(torch.from_numpy(rowmax), nranks, largest_rank) if narrow == 1 else None
    W = merge.plan_windows(torch.from_numpy(want_ranges.astype(np.int32)), nranks,
                           "allreduce" if plan == "allreduce" else "reduce_scatter", bits)
    words = W["words"].numpy()
    for r in range(nranks):
        assert tuple(results[r]) == merge.owned_rows(W, r), (r, results[r], merge.owned_rows(W, r))
    for r in range(nranks):
        first, last = results[r]
        if plan == "allreduce":
            assert (first, last) == (0, M)
        else:
            # owner blocks tile [0, M) in rank order and hold equal shares of the PACKED words (at most one row's
            # window more or less), not equal numbers of names
            assert first == (results[r - 1][1] if r else 0) and last >= first, (r, results)
            if r == nranks - 1:
                assert last == M
            share = int(words[first:last].sum())
            assert abs(share - W["total"] / nranks) <= int(words.max()) + 1, (r, share, W["total"] / nranks)
        covered.extend(range(first, last))
        snap = snaps[r]
        info = snap.merge_info()
        infos.append(info)
        assert info["packed_cells"] == cells, (info, cells)
        # every test stream is far below 2^32 samples: the wire word is uint32, and the equal-block collective pays
        # at most one row's window per block over the packed matrix (ratio <= 1.3)
        assert info["cell_bytes"] == (8 if wide64 else 4), info
        if wide64:
            assert snap.device_cells()[2] == 8
        assert info["packed_words"] == W["total"], (info, W["total"])
        assert info["padded_words"] >= info["packed_words"]
        assert info["padded_words"] == W["bmax"] * W["nblocks"], (info, W["bmax"], W["nblocks"])
        if plan != "allreduce" and W["total"] > 100 * int(words.max()):
            assert info["padded_words"] <= 1.3 * W["total"], info
        occupied = widths > 0
        if bits is None:
            assert info["rows_8bit"] == 0 and info["rows_16bit"] == 0 and info["packed_words"] == cells, info
        else:
            b = bits.numpy()
            assert info["rows_8bit"] == int((occupied & (b == 8)).sum()), (info, np.unique(b, return_counts=True))
            assert info["rows_16bit"] == int((occupied & (b == 16)).sum()), (info, np.unique(b, return_counts=True))
            assert info["rows_8bit"] + info["rows_16bit"] > 0 and info["packed_words"] <= 0.6 * cells, info
            if M >= 37:
                assert info["rows_8bit"] > 0.5 * M, info    # most names of a Zipf stream hold small counts
            if M > 4:
                assert b[1] > 8                             # the one-cell name travels wider than its neighbours
        assert info["send_bytes"] == (8 if wide64 else 4) * (info["padded_words"] if plan != "allreduce" else info["packed_words"]), info
        assert info["span_ms"] > 0 and info["collective_ms"] >= 0 and info["pack_ms"] >= 0, info
        assert info["occupied_rows"] == int((want_ranges[:, 0] <= want_ranges[:, 1]).sum())
        # merged ranges: identical on every rank, for every row
        torch.cuda.synchronize()
        got_ranges = merge.snapshot_ranges(snap, M).cpu().numpy().view(np.uint32)
        assert np.array_equal(got_ranges.astype(np.int64), want_ranges), r
        if last > first:
            off, keys, counts = snap.buckets_all(last - first, first=first)
            got = np.zeros((last - first, 65536), dtype=np.uint64)
            rows = np.repeat(np.arange(last - first), np.diff(off.astype(np.int64)))
            got[rows, oracle.key_to_bin(keys)] = counts
            assert np.array_equal(got, want[first:last]), f"rank {r}: merged rows differ"
            st = snap.extract([0.0, 0.5, 1.0], last - first, first=first)
            for m in range(first, last):
                ref = oracle.process_dense(want[m], [0.0, 0.5, 1.0])
                assert int(st["count"][m - first]) == ref["count"], m
                if ref["count"]:
                    assert np.array_equal(st["pkeys"][m - first], ref["pkeys"]), m
    expect = sorted(list(range(M)) * (nranks if plan == "allreduce" else 1))
    assert sorted(covered) == expect
    for s in snaps:
        s.release()
    for e in engines:
        e.close()
    stub.stub_comm_destroy(nranks, comms)
    print(json.dumps({"ok": True, "nranks": nranks, "rows": M, "plan": plan, "outliers": outliers, "narrow": narrow,
                      **infos[0]}))


if __name__ == "__main__":
    main()
