// c5_driver.cc -- BASELINE.json config 5: mixed Counter + Histogram + Timer events from many producer
// threads, 1 s ProcessedMetricSet emission serialised with GraphiteProtocol to a TCP sink
// (SURVEY.md 8d "C5": 50 % Histogram, 25 % Timer, 25 % Counter over 1 024 histogram + 256 timer +
// 256 counter names, Zipf(1.0) popularity).  Uses the C++ host layer (include/loghisto.hpp) exactly as
// an application would; reports sustained events/s, dropped intervals, emit latency and checks that
// every submitted event is accounted for in exactly one interval.  Prints one JSON line.
#include "loghisto.hpp"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>

using namespace loghisto;
using Clock = std::chrono::steady_clock;

// cores this process may actually use: the cgroup quota can be far below hardware_concurrency()
// (the MI355X box reports 256 CPUs and grants 16; threads beyond the quota only time-slice)
static unsigned granted_cores()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64];
        long period = 0;
        if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max(1L, (std::atol(q) + period - 1) / period));
        std::fclose(f);
    }
    return n;
}

static double arg_d(int argc, char **argv, const char *key, double def)
{
    for (int i = 1; i + 1 < argc; i++)
        if (!std::strcmp(argv[i], key)) return std::atof(argv[i + 1]);
    return def;
}

struct Sink { // local Graphite stand-in: accepts connections, counts bytes and lines
    int fd = -1, port = 0;
    std::atomic<uint64_t> bytes{0}, lines{0}, conns{0};
    std::atomic<bool> stop{false};
    std::thread th;
    bool start()
    {
        fd = socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
        a.sin_port = 0;
        if (bind(fd, (sockaddr *)&a, sizeof a) || listen(fd, 64)) return false;
        socklen_t l = sizeof a;
        getsockname(fd, (sockaddr *)&a, &l);
        port = ntohs(a.sin_port);
        th = std::thread([this] {
            std::vector<char> buf(1 << 20);
            while (!stop.load()) {
                timeval tv{0, 100000};
                fd_set rs;
                FD_ZERO(&rs);
                FD_SET(fd, &rs);
                if (select(fd + 1, &rs, nullptr, nullptr, &tv) <= 0) continue;
                int c = accept(fd, nullptr, nullptr);
                if (c < 0) continue;
                conns++;
                ssize_t n;
                while ((n = read(c, buf.data(), buf.size())) > 0) {
                    bytes += (uint64_t)n;
                    lines += (uint64_t)std::count(buf.begin(), buf.begin() + n, '\n');
                }
                close(c);
            }
        });
        return true;
    }
    void shutdown()
    {
        stop.store(true);
        if (th.joinable()) th.join();
        if (fd >= 0) close(fd);
    }
};

static bool submit_tcp(int port, const std::string &req) // submitter.go:106-116: dial, write, close
{
    int c = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = htons((uint16_t)port);
    if (connect(c, (sockaddr *)&a, sizeof a)) { close(c); return false; }
    size_t off = 0;
    while (off < req.size()) {
        ssize_t n = write(c, req.data() + off, req.size() - off);
        if (n <= 0) { close(c); return false; }
        off += (size_t)n;
    }
    close(c);
    return true;
}

int main(int argc, char **argv)
{
    const int T = (int)arg_d(argc, argv, "--threads", std::max(1u, granted_cores() - 2)); // leave room for reaper, submitter, sink
    const double seconds = arg_d(argc, argv, "--seconds", 10);
    const double target = arg_d(argc, argv, "--rate", 100e6); // events/s over all threads; 0 = unthrottled
    const int NH = (int)arg_d(argc, argv, "--hist-names", 1024), NT = 256, NC = 256;
    const int interval_ms = (int)arg_d(argc, argv, "--interval-ms", 1000);

    Options opt;
    opt.max_metrics = 2048;
    opt.num_lanes = (uint32_t)std::min(64, std::max(8, T / 2));
    opt.stage_samples = 8192;
    // --device-counters 1: Counter() events are summed on the GPU (lh_submit_counts) instead of per-thread host maps
    opt.device_counters = arg_d(argc, argv, "--device-counters", 0) != 0;
    MetricSystem ms(std::chrono::milliseconds(interval_ms), false, opt);
    // --bulk 1: the Graphite request is prepared on the GPU (K6, lh_serialize) instead of one %f per key; the map
    // is still filled because the event accounting below reads it
    const bool bulk = arg_d(argc, argv, "--bulk", 0) != 0;
    if (bulk) ms.SetWireFormat(WireFormat::Graphite, true);

    std::vector<std::string> hn, tn, cn;
    char b[32];
    for (int i = 0; i < NH; i++) { std::snprintf(b, sizeof b, "h%04d", i); hn.push_back(b); }
    for (int i = 0; i < NT; i++) { std::snprintf(b, sizeof b, "t%04d", i); tn.push_back(b); }
    for (int i = 0; i < NC; i++) { std::snprintf(b, sizeof b, "c%04d", i); cn.push_back(b); }
    // Zipf(1.0) over ranks through a 64K-entry inverse-CDF table
    auto zipf_table = [](int n) {
        std::vector<uint16_t> t(65536);
        double H = 0;
        for (int r = 1; r <= n; r++) H += 1.0 / r;
        double acc = 0;
        int r = 1;
        for (int i = 0; i < 65536; i++) {
            const double u = (i + 0.5) / 65536.0;
            while (r < n && acc + 1.0 / r / H < u) { acc += 1.0 / r / H; r++; }
            t[i] = (uint16_t)(r - 1);
        }
        return t;
    };
    const auto zh = zipf_table(NH), zt = zipf_table(NT), zc = zipf_table(NC);
    std::vector<double> values(4096);
    {
        std::mt19937_64 g(5);
        std::lognormal_distribution<double> d(std::log(1e5), 1.0);
        for (auto &v : values) v = d(g);
    }

    Sink sink;
    if (!sink.start()) { std::fprintf(stderr, "sink failed\n"); return 2; }

    auto ch = std::make_shared<Channel<std::shared_ptr<ProcessedMetricSet>>>(60); // submitter.go:55
    ms.SubscribeToProcessedMetrics(ch);
    std::atomic<bool> sub_stop{false};
    std::atomic<uint64_t> accounted{0}, intervals{0}, keys_emitted{0}, submit_fail{0};
    std::vector<double> emit_ms;
    std::mutex emit_mu;
    std::thread submitter([&] {
        std::shared_ptr<ProcessedMetricSet> pm;
        while (!sub_stop.load() || ch->Len()) {
            if (!ch->Receive(pm, std::chrono::milliseconds(50))) continue;
            uint64_t ev = 0;
            for (auto &kv : pm->Metrics) {
                const std::string &k = kv.first;
                if (k.size() > 6 && !k.compare(k.size() - 6, 6, "_count") && k.find("_agg_") == std::string::npos)
                    ev += (uint64_t)kv.second;
                else if (k.size() > 5 && !k.compare(k.size() - 5, 5, "_rate"))
                    ev += (uint64_t)kv.second;
            }
            accounted += ev;
            keys_emitted += pm->Metrics.size();
            const std::string req = GraphiteProtocol(*pm);
            if (!submit_tcp(sink.port, req)) submit_fail++;
            // emit latency: end of the interval (Time is the truncated collection time) -> bytes handed to the sink
            const auto lat = std::chrono::system_clock::now() - pm->Time;
            {
                std::lock_guard<std::mutex> g(emit_mu);
                emit_ms.push_back(std::chrono::duration<double, std::milli>(lat).count());
            }
            intervals++;
        }
    });

    ms.Start();
    std::vector<uint64_t> sent((size_t)T, 0);
    std::atomic<bool> stop{false};
    const auto t0 = Clock::now();
    std::vector<std::thread> prod;
    for (int t = 0; t < T; t++) {
        prod.emplace_back([&, t] {
            uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1), n = 0;
            const double per_thread = target > 0 ? target / T : 0;
            while (!stop.load(std::memory_order_relaxed)) {
                for (int k = 0; k < 1024; k++, n++) {
                    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                    const uint32_t r = (uint32_t)(x >> 32);
                    switch (n & 3) {
                    case 0:
                    case 1: ms.Histogram(hn[zh[r & 0xffff]], values[(r >> 16) & 4095]); break;
                    case 2: { auto tok = ms.StartTimer(tn[zt[r & 0xffff]]); tok.Stop(); break; }
                    default: ms.Counter(cn[zc[r & 0xffff]], 1); break;
                    }
                }
                if (per_thread > 0) {
                    const double should = std::chrono::duration<double>(Clock::now() - t0).count() * per_thread;
                    if ((double)n > should) {
                        const double ahead_s = ((double)n - should) / per_thread;
                        std::this_thread::sleep_for(std::chrono::duration<double>(std::min(ahead_s, 0.01)));
                    }
                }
            }
            sent[(size_t)t] = n;
        });
    }
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto &p : prod) p.join();
    const double elapsed = std::chrono::duration<double>(Clock::now() - t0).count();
    std::this_thread::sleep_for(std::chrono::milliseconds(2 * interval_ms + 200)); // let the last intervals emit
    ms.Stop();
    // whatever arrived after the last reaper tick
    {
        auto raw = ms.collectRawMetrics();
        auto pm = ms.processMetrics(raw);
        raw->Release();
        ch->TrySend(pm);
    }
    sub_stop.store(true);
    submitter.join();
    sink.shutdown();

    uint64_t total = 0;
    for (auto v : sent) total += v;
    std::sort(emit_ms.begin(), emit_ms.end());
    const double p50 = emit_ms.empty() ? 0 : emit_ms[emit_ms.size() / 2];
    const double pmax = emit_ms.empty() ? 0 : emit_ms.back();
    std::printf("{\"workload\": \"C5 mixed Counter+Histogram+Timer, %d hist + %d timer + %d counter names, Zipf(1.0)\", "
                "\"threads\": %d, \"seconds\": %.3f, \"target_events_per_s\": %.4g, \"events_submitted\": %llu, "
                "\"events_per_s\": %.4g, \"events_accounted\": %llu, \"lossless\": %s, \"intervals_emitted\": %llu, "
                "\"dropped_intervals\": %llu, \"interval_ms\": %d, \"emit_latency_ms_p50\": %.2f, \"emit_latency_ms_max\": %.2f, "
                "\"graphite_bytes\": %llu, \"graphite_lines\": %llu, \"keys_emitted\": %llu, \"submit_failures\": %llu, "
                "\"last_status\": %d, \"device_counters\": %s, \"bulk_wire\": %s}\n",
                NH, NT, NC, T, elapsed, target, (unsigned long long)total, (double)total / elapsed,
                (unsigned long long)accounted.load(), accounted.load() == total ? "true" : "false",
                (unsigned long long)intervals.load(), (unsigned long long)ms.dropped_intervals(), interval_ms, p50, pmax,
                (unsigned long long)sink.bytes.load(), (unsigned long long)sink.lines.load(),
                (unsigned long long)keys_emitted.load(), (unsigned long long)submit_fail.load(), ms.last_status(),
                opt.device_counters ? "true" : "false", bulk ? "true" : "false");
    return accounted.load() == total ? 0 : 1;
}
