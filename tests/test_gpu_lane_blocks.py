"""Host-fed mixed launches in the lanes' own scratch blocks (LH_OPT_LANE_SCRATCH_BLOCKS): lanes of several threads launch
concurrently, each launch partitioned in its own block (first generation up to 8 192 names; above, the third generation
on survey tables the lanes share: LH_OPT_LANE_GEN3) -- every cell of every row against the oracle, with the blocks on
(default) and off (the engine's one shared block), for both name-count classes and both id widths.
Semantics: metrics.go:273-295 (lossless, one increment per sample whatever thread submitted it)."""
import os
import sys
import threading

import numpy as np
import pytest

from tests.conftest import thresholds_until_round_6

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la(native_lib, torch_cuda):
    import loghisto_amd
    return loghisto_amd


@pytest.mark.parametrize("M,width", [(1024, 4), (1024, 2), (20000, 4), (65536, 2)])
@pytest.mark.parametrize("blocks", [16, 2, 0])
def test_concurrent_lanes_in_their_own_blocks(la, M, width, blocks):
    from loghisto_amd import _native as N
    rng = np.random.default_rng(M + blocks)
    T, batches, per = 8, 6, 200_003                     # every launch above the partitioned minimum (131 072 pairs)
    eng = la.Engine(max_metrics=M, num_buffers=2, num_lanes=4, lane_samples=1 << 18)
    try:
        eng.set_option(N.OPT_LANE_SCRATCH_BLOCKS, blocks)
        w = 1.0 / np.arange(1, M + 1)
        ids = [rng.choice(M, per, p=w / w.sum()).astype(np.uint16 if width == 2 else np.uint32) for _ in range(T)]
        vals = [rng.lognormal(np.log(1e5), 1.0, per) * (1.0 + 1e-4 * ids[t]) for t in range(T)]
        errors = []

        def work(t):
            try:
                for _ in range(batches):
                    eng.submit_pairs(ids[t], vals[t])
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        assert not errors, errors
        with eng.flip() as snap:
            got = snap.extract([0.5], M)
            probe = sorted({0, 1, 2, 3, 17, 255, 256, M // 2, M - 1} | set(int(x) for x in rng.integers(0, M, 12)))
            rows = {m: snap.dense_row(m) for m in probe}
        want_count = np.zeros(M, dtype=np.int64)
        for t in range(T):
            want_count += batches * np.bincount(ids[t].astype(np.int64), minlength=M)
        assert np.array_equal(got["count"].astype(np.int64), want_count)
        for m in probe:
            want = np.zeros(oracle.NKEYS, dtype=np.uint64)
            for t in range(T):
                oracle.histogram_dense(vals[t][ids[t] == m], want)
            assert np.array_equal(rows[m], batches * want), m
        c = eng.counters()
        assert c["scratch_bytes"] == 0                                 # none of them used the shared block
        if blocks:
            assert c["samples_partitioned"] >= T * batches * per * 0.8 # the full half-buffers were partitioned launches
            assert c["lane_scratch_bytes"] > 0
        else:                                                          # the default since round 6: the direct path, no scratch at all
            assert c["samples_partitioned"] == 0 and c["samples_direct"] == T * batches * per and c["lane_scratch_bytes"] == 0
    finally:
        eng.close()


def test_device_resident_launches_keep_the_shared_block(la, torch_cuda):
    """lh_submit_pairs_device is not host-fed: its launches (and the survey tables the later generations keep between
    calls) stay in the engine's one block whatever the option says; lanes running beside it do not disturb them."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    M, n = 20000, 600_000
    eng = la.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=1 << 18)
    try:
        from loghisto_amd import _native as N
        thresholds_until_round_6(eng)            # (600 000 pairs per call: a third-generation launch under them)
        eng.set_option(N.OPT_LANE_SCRATCH_BLOCKS, 16) # (the lanes beside it: partitioned launches in blocks of their own)
        w = 1.0 / np.arange(1, M + 1)
        ids = rng.choice(M, n, p=w / w.sum()).astype(np.uint32)
        v = rng.lognormal(np.log(1e5), 1.0, n)
        d_ids, d_v = torch.from_numpy(ids.view(np.int32)).cuda(), torch.from_numpy(v).cuda()
        stop, errors = threading.Event(), []

        def lanes():
            try:
                while not stop.is_set():
                    eng.submit_pairs(ids[:200_000], v[:200_000])
                    lanes.count += 1
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)
        lanes.count = 0
        th = threading.Thread(target=lanes)
        th.start()
        calls = 12
        for _ in range(calls):
            eng.submit_pairs_device(d_ids, d_v)
        stop.set()
        th.join()
        assert not errors, errors
        with eng.flip() as snap:
            got = snap.extract([0.5], M)["count"].astype(np.int64)
        want = calls * np.bincount(ids, minlength=M) + lanes.count * np.bincount(ids[:200_000], minlength=M)
        assert np.array_equal(got, want)
        c = eng.counters()
        # (the lanes' own launches over this many names are third-generation launches too, on table sets of their own)
        assert c["scratch_bytes"] > 0 and c["samples_partitioned_v3"] >= calls * n
        assert c["surveys_reused"] >= calls - 4                        # the lanes did not end the survey's reuse
    finally:
        eng.close()


@pytest.mark.parametrize("gen3", [1, 0])
def test_lanes_over_many_names_share_survey_tables(la, gen3):
    """65 536 names from four producer threads: every half-buffer (262 144 pairs) is a third-generation launch whose
    records live in the lane's own block while the survey's tables are shared, read-only, by all lanes -- two sets: a
    launch that finds the active set used up (LH_OPT_SURVEY_EVERY) surveys its own pairs into the OTHER set while the
    launches in flight keep reading the old one.  Half way the stream changes (other names frequent, values a million
    times larger): the launches that still run on the old survey are slower, never wrong.  Every occupied cell of every
    row against the oracle; LH_OPT_LANE_GEN3 = 0 is the first generation as before."""
    from loghisto_amd import _native as N
    M, T, batches, per = 65536, 4, 12, 1 << 18
    rng = np.random.default_rng(77 + gen3)
    w = 1.0 / np.arange(1, M + 1)
    ids, vals = [], []
    for t in range(T):
        a = rng.choice(M, per * batches, p=w / w.sum()).astype(np.uint16)
        v = rng.lognormal(np.log(1e5), 1.0, per * batches)
        half = per * batches // 2
        a[half:] = (M - 1) - a[half:]                    # the ranking reversed ...
        v[half:] *= 1e6                                   # ... and every window somewhere else
        ids.append(a)
        vals.append(v)
    eng = la.Engine(max_metrics=M, num_buffers=2, num_lanes=4, lane_samples=per)
    try:
        eng.set_option(N.OPT_LANE_SCRATCH_BLOCKS, 16)     # (0, the default since round 6, is the direct path: test above)
        eng.set_option(N.OPT_LANE_GEN3, gen3)
        eng.set_option(N.OPT_SURVEY_EVERY, 8)             # several surveys, into alternating sets, during the run
        errors = []

        def work(t):
            try:
                for k in range(batches):
                    eng.submit_pairs_in_place(ids[t][k * per:(k + 1) * per], vals[t][k * per:(k + 1) * per])
            except Exception as exc:  # noqa: BLE001
                errors.append(exc)

        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        [x.start() for x in th]
        [x.join() for x in th]
        assert not errors, errors
        eng.sync()
        c = eng.counters()
        total = T * batches * per
        if gen3:
            assert c["samples_partitioned_v3"] == total and c["surveys_reused"] >= T * batches // 2, c
        else:
            assert c["samples_partitioned_v3"] == 0 and c["samples_partitioned"] == total, c
        assert c["scratch_bytes"] == 0 and c["samples_fallback"] == 0, c   # the shared block was never needed
        all_ids, all_v = np.concatenate(ids).astype(np.uint64), np.concatenate(vals)
        bins = oracle.key_to_bin(oracle.compress_many(all_v)).astype(np.uint64)
        want_cells, want_counts = np.unique((all_ids << np.uint64(16)) | bins, return_counts=True)
        with eng.flip() as snap:
            off, keys, counts = snap.buckets_all(M)
        rows = np.repeat(np.arange(M, dtype=np.uint64), np.diff(off.astype(np.int64)))
        cells = (rows << np.uint64(16)) | oracle.key_to_bin(keys).astype(np.uint64)
        assert np.array_equal(cells, want_cells) and np.array_equal(counts.astype(np.int64), want_counts)
    finally:
        eng.close()


def test_lanes_that_carry_different_streams_do_not_take_turns_surveying(la):
    """The launches report what a stale survey costs their hot windows (lh_counters.survey_stale_pairs) and the next call
    surveys again -- but lanes share ONE set of survey tables, and two producers may carry streams whose values lie apart
    for good (one thread times requests, another sizes payloads).  Each lane's launch would find the other's survey
    stale; a survey is a large part of a lane-sized launch.  The lanes heed such a report only after a set has served
    kLaneStaleMinAge = 8 launches: exact, and most launches still run on a kept survey."""
    M, per, batches = 65536, 1 << 18, 20
    rng = np.random.default_rng(123)
    w = 1.0 / np.arange(1, M + 1)
    ids = [rng.choice(M, per, p=w / w.sum()).astype(np.uint16) for _ in range(2)]
    # 300 bins apart: outside each other's hot windows (~260 bins wide), inside the 1 024-bin windows of levels 2 and 3 --
    # nothing overflows or misses (that WOULD be a reason to survey again at once), only the hot windows' share moves
    vals = [rng.lognormal(np.log(1e5), 0.5, per), rng.lognormal(np.log(5e3), 0.5, per)]
    eng = la.Engine(max_metrics=M, num_buffers=2, num_lanes=2, lane_samples=per)
    try:
        from loghisto_amd import _native as N
        eng.set_option(N.OPT_LANE_SCRATCH_BLOCKS, 16)     # (0, the default since round 6, is the direct path: no survey at all)
        for k in range(batches):                          # one producer after the other: the worst case, strict alternation
            for t in range(2):
                eng.submit_pairs_in_place(ids[t], vals[t])
        eng.sync()
        c = eng.counters()
        launches = 2 * batches
        assert c["samples_partitioned_v3"] == launches * per, c
        assert c["survey_stale_pairs"] > 0, c             # the reports are there ...
        assert c["surveys_reused"] >= launches - 2 - launches // 8, c   # ... and end a set's reuse once in eight launches at most
        with eng.flip() as snap:
            got = snap.extract([0.5], M)["count"].astype(np.int64)
            rows = {m: snap.dense_row(m) for m in (0, 1, 7, 100, 4097, 65535)}
        assert np.array_equal(got, batches * (np.bincount(ids[0], minlength=M) + np.bincount(ids[1], minlength=M)))
        for m, row in rows.items():
            want = np.zeros(65536, dtype=np.uint64)
            for t in range(2):
                oracle.histogram_dense(vals[t][ids[t] == m], want)
            assert np.array_equal(row, batches * want), m
    finally:
        eng.close()
