// latency.cc -- BASELINE metric, part 2, at the C ABI: wall time from the lh_flip call to the lh_extract results
// on the host (one metric, the default nine percentiles), p50 / p99 over N flips.  bench.py reports the same
// interval through the Python binding (ctypes + numpy allocation add a few microseconds).
//
// usage: latency [flips=2000] [samples_per_interval=1048576]
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
    lh_config cfg;
    lh_default_config(&cfg);
    cfg.max_metrics = 1;
    cfg.num_buffers = 2;
    cfg.num_lanes = 1;
    lh_engine *e = nullptr;
    int rc = lh_create(&cfg, &e);
    if (rc != LH_OK) { std::fprintf(stderr, "lh_create: %s [%s]\n", lh_strerror(rc), lh_last_error()); return 2; }
    uint32_t id = 0;
    lh_intern(e, "m0", 2, &id);
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
        lh_submit(e, id, v.data(), n);
        lh_sync(e); // every sample is in the bucket arrays: the timed region is flip -> results only
        lh_stats st;
        double pv[9];
        uint8_t ok[9];
        const auto t0 = std::chrono::steady_clock::now();
        lh_snapshot *s = nullptr;
        rc = lh_flip(e, &s);
        if (rc == LH_OK) rc = lh_extract(s, p, 9, &st, pv, nullptr, ok, 1);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc != LH_OK) { std::fprintf(stderr, "flip/extract: %s\n", lh_strerror(rc)); return 3; }
        lh_release(s);
        if (st.count != n || !ok[8] || !(pv[1] > 0)) { std::fprintf(stderr, "bad result\n"); return 4; }
        total += st.count;
        if (i >= 20) lat.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(lat.begin(), lat.end());
    std::printf("{\"what\": \"lh_flip -> lh_extract results on host, C ABI\", \"flips\": %d, \"samples_per_interval\": %zu, "
                "\"p50_us\": %.2f, \"p90_us\": %.2f, \"p99_us\": %.2f, \"max_us\": %.2f, \"samples_total\": %llu}\n",
                flips, n, lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat[lat.size() * 99 / 100], lat.back(),
                (unsigned long long)total);
    lh_destroy(e);
    return 0;
}
