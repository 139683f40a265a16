// hostfed_native.cc -- host-fed (PCIe-inclusive) mixed ingest from NATIVE producer threads, at the C ABI.
//
// bench.py's secondary.hostfed_pairs drives the staging API from 16 Python threads (numpy copies into the pinned
// buffers).  This is the same stream from plain C++ threads, the way a compiled binding (the C++ host layer's stage
// flush, the cgo binding's ship()) produces it -- it showed that the producers were not what held the path at 44 GB/s
// (the lanes' launches sharing one scratch block were: profiles/r04_hostfed_native.jsonl, 4.4 -> 5.3 G pairs/s): every thread reserves the tail of a pinned staging buffer
// (lh_reserve_pairs16 / lh_reserve_pairs), stores its pairs there -- the one host-side copy -- and commits.
// Checked: per-name counts of the interval against the generator's own counts.  Prints one JSON line per form.
//
//   usage: hostfed_native [threads=16] [pairs=8e8] [names=1024] [batch=1048576] [lane_gen3=1] [survey_every=0] [lane_blocks=-1] [direct_max=0] [part_min=0]
//   (lane_gen3 = 0: the lanes' launches over more than 8 192 names take the first generation, as up to ABI 4)
#include "loghisto_gpu_tuning.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

static void die(const char *what, int rc)
{
    std::fprintf(stderr, "%s: %s (%s)\n", what, lh_strerror(rc), lh_last_error());
    std::exit(1);
}

static int g_lane_gen3 = 1, g_survey_every = 0, g_blocks = -1;
static long long g_direct_max = 0, g_part_min = 0;

template <typename IDT> static double run(uint32_t T, size_t total, uint32_t M, size_t batch, const std::vector<uint32_t> &ids,
                                          const std::vector<double> &vals, bool *exact)
{
    lh_config cfg;
    lh_default_config(&cfg);
    cfg.max_metrics = M;
    cfg.num_buffers = 2;
    cfg.num_lanes = T;
    cfg.lane_samples = 1 << 21;
    lh_engine *e = nullptr;
    int rc = lh_create(&cfg, &e);
    if (rc) die("lh_create", rc);
    if ((rc = lh_set_option(e, LH_OPT_LANE_GEN3, (uint64_t)g_lane_gen3))) die("lh_set_option", rc);
    if (g_survey_every && (rc = lh_set_option(e, LH_OPT_SURVEY_EVERY, (uint64_t)g_survey_every))) die("lh_set_option", rc);
    if (g_blocks >= 0 && (rc = lh_set_option(e, LH_OPT_LANE_SCRATCH_BLOCKS, (uint64_t)g_blocks))) die("lh_set_option", rc);
    if (g_direct_max && (rc = lh_set_option(e, LH_OPT_PART_V3_DIRECT_MAX_PAIRS, (uint64_t)g_direct_max))) die("lh_set_option", rc);
    if (g_part_min && (rc = lh_set_option(e, LH_OPT_PART_MIN_PAIRS, (uint64_t)g_part_min))) die("lh_set_option", rc);
    std::vector<IDT> nid(ids.begin(), ids.end()); // the producer's own id array in the width it ships
    const size_t per = total / T;
    auto put = [&](size_t off, size_t n) {
        size_t done = 0;
        while (done < n) {
            IDT *pi = nullptr;
            double *pv = nullptr;
            size_t granted = 0;
            uint32_t tok = 0;
            int r;
            if constexpr (sizeof(IDT) == 2) r = lh_reserve_pairs16(e, n - done, &pi, &pv, &granted, &tok);
            else r = lh_reserve_pairs(e, n - done, &pi, &pv, &granted, &tok);
            if (r) die("lh_reserve_pairs", r);
            std::memcpy(pi, nid.data() + off + done, granted * sizeof(IDT));
            std::memcpy(pv, vals.data() + off + done, granted * sizeof(double));
            r = lh_commit_pairs(e, tok, granted);
            if (r) die("lh_commit_pairs", r);
            done += granted;
        }
    };
    auto work = [&](uint32_t t, size_t n) {
        const size_t off = ((size_t)t * 7919 * batch) % (ids.size() - batch);
        for (size_t done = 0; done < n;) {
            const size_t k = std::min(batch, n - done);
            put(off, k);
            done += k;
        }
    };
    {   // untimed: first touch of the pinned buffers, the kernels' first launches; discarded
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; t++) th.emplace_back(work, t, batch);
        for (auto &x : th) x.join();
        lh_sync(e);
        lh_snapshot *s = nullptr;
        if ((rc = lh_flip(e, &s))) die("lh_flip", rc);
        lh_release(s);
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; t++) th.emplace_back(work, t, per);
    for (auto &x : th) x.join();
    if ((rc = lh_sync(e))) die("lh_sync", rc);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // per-name counts against the generator's
    std::vector<uint64_t> want(M, 0);
    for (uint32_t t = 0; t < T; t++) {
        const size_t off = ((size_t)t * 7919 * batch) % (ids.size() - batch);
        const size_t reps = per / batch, rem = per % batch;
        for (size_t i = 0; i < batch; i++) want[ids[off + i]] += reps + (i < rem ? 1 : 0);
    }
    lh_snapshot *s = nullptr;
    if ((rc = lh_flip(e, &s))) die("lh_flip", rc);
    std::vector<lh_stats> st(M);
    const double p50 = 0.5;
    std::vector<double> pv(M);
    std::vector<uint8_t> ok(M);
    if ((rc = lh_extract(s, &p50, 1, st.data(), pv.data(), nullptr, ok.data(), M))) die("lh_extract", rc);
    *exact = true;
    for (uint32_t m = 0; m < M; m++)
        if (st[m].count != want[m]) *exact = false;
    lh_release(s);
    lh_counters c;
    if (lh_get_counters(e, &c) == LH_OK)
        std::fprintf(stderr, "counters: partitioned %llu (v3 %llu) direct %llu fallback %llu | launches %llu surveys_reused %llu | level-1 records %llu "
                             "forwarded %llu | region overflows %llu level-2 overflows %llu reduce misses %llu | waits %llu\n",
                     (unsigned long long)c.samples_partitioned, (unsigned long long)c.samples_partitioned_v3,
                     (unsigned long long)c.samples_direct, (unsigned long long)c.samples_fallback, (unsigned long long)c.launches,
                     (unsigned long long)c.surveys_reused, (unsigned long long)c.records_level1, (unsigned long long)c.records_level2,
                     (unsigned long long)c.region_overflows, (unsigned long long)c.level2_overflows,
                     (unsigned long long)c.reduce_window_misses, (unsigned long long)c.backpressure_waits);
    lh_destroy(e);
    return (double)(per * T) / dt;
}

int main(int argc, char **argv)
{
    const uint32_t T = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 16u;
    const size_t total = argc > 2 ? (size_t)std::atof(argv[2]) : (size_t)8e8;
    const uint32_t M = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 1024u;
    const size_t batch = argc > 4 ? (size_t)std::atoll(argv[4]) : (size_t)1 << 20;
    g_lane_gen3 = argc > 5 ? std::atoi(argv[5]) : 1;
    g_survey_every = argc > 6 ? std::atoi(argv[6]) : 0;   // LH_OPT_SURVEY_EVERY (0: default)
    g_blocks = argc > 7 ? std::atoi(argv[7]) : -1;        // LH_OPT_LANE_SCRATCH_BLOCKS (-1: default)
    g_part_min = argc > 9 ? std::atoll(argv[9]) : 0;      // LH_OPT_PART_MIN_PAIRS (0: default; 2^30: every lane launch takes the direct path)
    g_direct_max = argc > 8 ? std::atoll(argv[8]) : 0;    // LH_OPT_PART_V3_DIRECT_MAX_PAIRS (0: default, 1: the windowed reduce pass always)
    const size_t N = (size_t)1 << 24;
    std::vector<uint32_t> ids(N);
    std::vector<double> vals(N);
    std::mt19937_64 rng(1);
    std::vector<double> cdf(M);
    double acc = 0;
    for (uint32_t m = 0; m < M; m++) { acc += 1.0 / (m + 1); cdf[m] = acc; }
    std::uniform_real_distribution<double> U(0.0, acc);
    std::lognormal_distribution<double> LN(std::log(1e5), 1.0);
    for (size_t i = 0; i < N; i++) {
        ids[i] = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin());
        if (ids[i] >= M) ids[i] = M - 1;
        vals[i] = LN(rng);
    }
    for (int form = 0; form < 2; form++) {
        bool exact = false;
        const double rate = form == 0 ? run<uint16_t>(T, total, M, batch, ids, vals, &exact)
                                      : run<uint32_t>(T, total, M, batch, ids, vals, &exact);
        const double bytes = form == 0 ? 10.0 : 12.0;
        std::printf("{\"what\": \"host-fed pairs from native threads, in place (%s)\", \"threads\": %u, \"pairs\": %zu, \"names\": %u, "
                    "\"lane_gen3\": %d, \"pairs_per_s\": %.4g, \"GBps_over_pcie\": %.2f, \"frac_of_63GBps\": %.4f, \"per_name_counts_exact\": %s}\n",
                    form == 0 ? "lh_reserve_pairs16, uint16 ids, 10 B per pair" : "lh_reserve_pairs, uint32 ids, 12 B per pair", T,
                    total / T * T, M, g_lane_gen3, rate, rate * bytes / 1e9, rate * bytes / 1e9 / 63.0, exact ? "true" : "false");
        std::fflush(stdout);
    }
    return 0;
}
