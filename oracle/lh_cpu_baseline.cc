// lh_cpu_baseline.cc -- timed CPU forms of the oracle.  TEST INFRASTRUCTURE ONLY.
//
// Used by bench.py's `cpu_baseline` leg and by tests.  No Go toolchain exists in
// this image, so the reference's pure-Go path cannot be timed; these are C++
// restatements of its cost SHAPE around the oracle's exact arithmetic
// (BASELINE.md section 2):
//
//   form A "faithful": one call per sample: shared read lock -> map[name] ->
//          map[int16] -> atomic add, with the reference's double lookup
//          (presence probe, then the add) and its lock-promotion slow path.
//          Mirrors (*MetricSystem).Histogram, /root/reference/metrics.go:273-295.
//   form B "dense":    per-thread uint64[65536] rows, same exact compress,
//          merged at the end.  A fair upper bound for host cores.
//
// Both return elapsed seconds and fill counts_out (dense, bin = key ^ 0x8000) so
// the caller can check them against lho_histogram_dense.
#include "lh_oracle.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct FaithfulSystem {
    // histogramCache map[string]map[int16]*uint64  (metrics.go:119)
    std::unordered_map<std::string,
                       std::unordered_map<int16_t, std::unique_ptr<std::atomic<uint64_t>>>>
        cache;
    std::shared_mutex mu; // histogramMu (metrics.go:121)

    void histogram(const std::string &name, double value)
    {
        int16_t c = lho_compress(value);
        mu.lock_shared();
        bool present = false;
        {
            auto it = cache.find(name);                       // probe 1
            if (it != cache.end()) present = it->second.find(c) != it->second.end(); // probe 2
        }
        if (present) {
            cache.find(name)->second.find(c)->second->fetch_add(1); // probes 3+4
            mu.unlock_shared();
        } else {
            mu.unlock_shared();
            mu.lock();
            auto &inner = cache[name];
            auto it = inner.find(c);
            if (it == inner.end())
                it = inner.emplace(c, std::make_unique<std::atomic<uint64_t>>(0)).first;
            it->second->fetch_add(1);
            mu.unlock();
        }
    }
};

double now_s()
{
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

double lho_bench_faithful(const double *v, size_t n, int threads, uint64_t *counts_out)
{
    if (threads < 1) threads = 1;
    FaithfulSystem ms;
    const std::string name = "m0";
    std::vector<std::thread> th;
    double t0 = now_s();
    for (int t = 0; t < threads; t++) {
        size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
        th.emplace_back([&, lo, hi] {
            for (size_t i = lo; i < hi; i++) ms.histogram(name, v[i]);
        });
    }
    for (auto &x : th) x.join();
    double t1 = now_s();
    if (counts_out) {
        std::memset(counts_out, 0, sizeof(uint64_t) * LHO_NKEYS);
        auto it = ms.cache.find(name);
        if (it != ms.cache.end())
            for (auto &kv : it->second)
                counts_out[(uint16_t)kv.first ^ 0x8000u] = kv.second->load();
    }
    return t1 - t0;
}

// Same as lho_bench_dense but every thread passes `reps` times over its slice inside the timed region, so
// that thread start-up and the final merge do not dominate on many-core hosts.  counts_out receives
// reps x the true row.
double lho_bench_dense_reps(const double *v, size_t n, int threads, int reps, uint64_t *counts_out)
{
    if (threads < 1) threads = 1;
    if (reps < 1) reps = 1;
    std::vector<std::vector<uint64_t>> rows((size_t)threads, std::vector<uint64_t>(LHO_NKEYS, 0));
    std::vector<std::thread> th;
    double t0 = now_s();
    for (int t = 0; t < threads; t++) {
        size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
        th.emplace_back([&, t, lo, hi] {
            for (int r = 0; r < reps; r++) lho_histogram_dense(v + lo, hi - lo, rows[(size_t)t].data());
        });
    }
    for (auto &x : th) x.join();
    if (counts_out) {
        std::memset(counts_out, 0, sizeof(uint64_t) * LHO_NKEYS);
        for (auto &r : rows)
            for (size_t b = 0; b < LHO_NKEYS; b++) counts_out[b] += r[b];
    }
    double t1 = now_s();
    return t1 - t0;
}

// Threaded mixed-stream fan-in for FULL-SIZE parity checks (1e9 pairs): the same arithmetic as
// lho_histogram_pairs (lho_compress per sample, +1 per (name, key) cell, metrics.go:273-295), the stream cut
// into `threads` slices that add into ONE shared matrix with relaxed atomic adds -- the reference's own
// fan-in is an atomic add per sample (metrics.go:278), and integer sums commute, so the result is the
// sequential one bit for bit.  Returns -1 if any id is >= nmetrics (that sample is skipped), else 0.
int lho_histogram_pairs_mt(const uint32_t *ids, const double *v, size_t n, uint64_t *counts, uint32_t nmetrics,
                           int threads)
{
    if (threads < 1) threads = 1;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) {
        size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
        th.emplace_back([&, lo, hi] {
            for (size_t i = lo; i < hi; i++) {
                if (ids[i] >= nmetrics) { bad.store(1, std::memory_order_relaxed); continue; }
                const uint32_t bin = (uint16_t)lho_compress(v[i]) ^ 0x8000u;
                __atomic_fetch_add(&counts[(size_t)ids[i] * LHO_NKEYS + bin], 1ull, __ATOMIC_RELAXED);
            }
        });
    }
    for (auto &x : th) x.join();
    return bad.load() ? -1 : 0;
}

double lho_bench_dense(const double *v, size_t n, int threads, uint64_t *counts_out)
{
    if (threads < 1) threads = 1;
    std::vector<std::vector<uint64_t>> rows((size_t)threads, std::vector<uint64_t>(LHO_NKEYS, 0));
    std::vector<std::thread> th;
    double t0 = now_s();
    for (int t = 0; t < threads; t++) {
        size_t lo = n * (size_t)t / (size_t)threads, hi = n * (size_t)(t + 1) / (size_t)threads;
        th.emplace_back([&, t, lo, hi] { lho_histogram_dense(v + lo, hi - lo, rows[(size_t)t].data()); });
    }
    for (auto &x : th) x.join();
    if (counts_out) {
        std::memset(counts_out, 0, sizeof(uint64_t) * LHO_NKEYS);
        for (auto &r : rows)
            for (size_t b = 0; b < LHO_NKEYS; b++) counts_out[b] += r[b];
    }
    double t1 = now_s();
    return t1 - t0;
}

} // extern "C"
