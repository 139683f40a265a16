// latency.cc -- BASELINE metric, part 2, at the C ABI: wall time from the lh_flip call to the lh_extract results
// on the host (one metric, the default nine percentiles), p50 / p99 over N flips.  bench.py reports the same
// interval through the Python binding (ctypes + numpy allocation add a few microseconds).
//
// With names > 1 the interval holds `samples` (id, value) pairs spread over that many histogram names and the
// timed region ends when the results of ALL names are on the host (latency at scale).
//
// usage: latency [flips=2000] [samples_per_interval=1048576] [names=1] [view=0]
#include "loghisto_gpu.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
    const int flips = argc > 1 ? std::atoi(argv[1]) : 2000;
    const size_t n = argc > 2 ? (size_t)std::atoll(argv[2]) : (size_t)1 << 20;
    const uint32_t names = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 1u;
    const bool use_view = argc > 4 && std::atoi(argv[4]) != 0; // lh_extract_rows_view instead of lh_extract
    const long long zc = argc > 5 ? std::atoll(argv[5]) : -1;  // LH_OPT_EXTRACT_ZERO_COPY (0 off, 1 default, >= 4096: limit)
    lh_config cfg;
    lh_default_config(&cfg);
    cfg.max_metrics = names;
    cfg.num_buffers = 2;
    cfg.num_lanes = 1;
    lh_engine *e = nullptr;
    int rc = lh_create(&cfg, &e);
    if (rc != LH_OK) { std::fprintf(stderr, "lh_create: %s [%s]\n", lh_strerror(rc), lh_last_error()); return 2; }
    if (zc >= 0 && lh_set_option(e, LH_OPT_EXTRACT_ZERO_COPY, (uint64_t)zc) != LH_OK) { std::fprintf(stderr, "bad zero-copy value\n"); return 2; }
    uint32_t id = 0;
    for (uint32_t m = 0; m < names; m++) {
        char nm[32];
        const int len = std::snprintf(nm, sizeof(nm), "m%u", m);
        lh_intern(e, nm, (size_t)len, &id);
    }
    id = 0;
    std::vector<uint32_t> ids(names > 1 ? n : 0);
    for (size_t i = 0; i < ids.size(); i++) ids[i] = (uint32_t)((i * 2654435761ull) % names); // every name gets samples
    std::vector<double> v(n);
    uint64_t x = 88172645463325252ull;
    for (auto &d : v) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        d = std::exp(11.5 + ((double)(x >> 11) / 9007199254740992.0 - 0.5) * 4.0);
    }
    const double p[9] = {0.0, .5, .75, .9, .95, .99, .999, .9999, 1.0};
    std::vector<double> lat;
    lat.reserve((size_t)flips);
    uint64_t total = 0;
    for (int i = 0; i < flips + 20; i++) {
        if (names > 1) lh_submit_pairs(e, ids.data(), v.data(), n);
        else lh_submit(e, id, v.data(), n);
        lh_sync(e); // every sample is in the bucket arrays: the timed region is flip -> results only
        static std::vector<lh_stats> st;
        static std::vector<double> pv;
        static std::vector<uint8_t> ok;
        st.resize(names); pv.resize((size_t)names * 9); ok.resize((size_t)names * 9);
        const auto t0 = std::chrono::steady_clock::now();
        lh_snapshot *s = nullptr;
        rc = lh_flip(e, &s);
        lh_extract_view view;
        if (rc == LH_OK) {
            if (use_view) rc = lh_extract_rows_view(s, 0, names, p, 9, &view);        // results in place (pinned)
            else rc = lh_extract(s, p, 9, st.data(), pv.data(), nullptr, ok.data(), names);
        }
        const auto t1 = std::chrono::steady_clock::now();
        if (rc == LH_OK && use_view) {
            std::copy(view.stats, view.stats + names, st.begin());
            std::copy(view.pvals, view.pvals + (size_t)names * 9, pv.begin());
            std::copy(view.pvalid, view.pvalid + (size_t)names * 9, ok.begin());
        }
        if (rc != LH_OK) { std::fprintf(stderr, "flip/extract: %s\n", lh_strerror(rc)); return 3; }
        lh_release(s);
        uint64_t cnt = 0;
        for (uint32_t m = 0; m < names; m++) cnt += st[m].count;
        if (cnt != n || !ok[8] || !(pv[1] > 0)) { std::fprintf(stderr, "bad result\n"); return 4; }
        total += cnt;
        if (i >= 20) lat.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(lat.begin(), lat.end());
    std::printf("{\"what\": \"lh_flip -> lh_extract results on host, C ABI\", \"names\": %u, \"view\": %d, \"zero_copy\": %lld, \"flips\": %d, \"samples_per_interval\": %zu, "
                "\"p50_us\": %.2f, \"p90_us\": %.2f, \"p99_us\": %.2f, \"max_us\": %.2f, \"samples_total\": %llu}\n",
                names, (int)use_view, zc, flips, n, lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[lat.size() * 99 / 100], lat.back(),
                (unsigned long long)total);
    lh_destroy(e);
    return 0;
}
