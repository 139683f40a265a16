// metrics_test.cc -- the reference's metrics_test.go restated against the C++ host layer
// (include/loghisto.hpp) over the C ABI.  `--cpu` runs the tests that need no GPU (counters, rates,
// gauges, subscriptions, serializers: that logic is host-side in the reference too); without it the
// histogram tests run as well and need an MI355X.  Exit code = number of failed tests.
#include "loghisto.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

using namespace loghisto;
using namespace std::chrono_literals;

static int g_failed = 0, g_checks = 0;
#define CHECK(cond)                                                                                      \
    do {                                                                                                 \
        g_checks++;                                                                                      \
        if (!(cond)) { std::printf("    CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ok = false; } \
    } while (0)
#define TEST(name) static void name(bool &ok)
#define RUN(name)                                                  \
    do {                                                           \
        bool ok = true;                                            \
        name(ok);                                                  \
        std::printf("%s %s\n", ok ? "PASS" : "FAIL", #name);       \
        if (!ok) g_failed++;                                       \
    } while (0)

using RawCh = Channel<std::shared_ptr<RawMetricSet>>;
using ProcCh = Channel<std::shared_ptr<ProcessedMetricSet>>;

// ---- host-only -------------------------------------------------------------------------------
TEST(TestRate) // metrics_test.go:202-223
{
    MetricSystem ms(1us, false);
    ms.Counter("rate1", 777);
    auto m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 777);
    ms.Counter("rate1", 1223);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 1223);
    ms.Counter("rate1", 1223);
    ms.Counter("rate1", 1223);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 2446);
}

TEST(TestCounter) // metrics_test.go:225-240
{
    MetricSystem ms(1us, false);
    ms.Counter("counter1", 3290);
    auto m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["counter1"] == 3290);
    ms.Counter("counter1", 10000);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["counter1"] == 13290);
}

TEST(TestCounterManyThreads)
{
    MetricSystem ms(1us, false);
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++)
        th.emplace_back([&] { for (int i = 0; i < 10000; i++) ms.Counter("c", 3); });
    for (auto &x : th) x.join();
    auto raw = ms.collectRawMetrics();
    CHECK(raw->Counters["c"] == 240000 && raw->Rates["c"] == 240000);
}

TEST(TestSysStats) // metrics_test.go:174-181
{
    MetricSystem ms(1us, true);
    auto g = ms.collectRawMetrics()->Gauges;
    CHECK(g.count("sys.Alloc") && g["sys.Alloc"] > 0);
}

TEST(TestRawBroadcast) // metrics_test.go:321-346
{
    auto ch = std::make_shared<RawCh>(128);
    MetricSystem ms(1ms, false);
    ms.SubscribeToRawMetrics(ch);
    ms.Counter("counter2", 10);
    ms.Counter("counter2", 111);
    ms.Start();
    std::shared_ptr<RawMetricSet> raw;
    CHECK(ch->Receive(raw, 2s));
    if (raw) {
        CHECK(raw->Counters["counter2"] == 121);
        CHECK(raw->Rates["counter2"] == 121);
    }
    ms.UnsubscribeFromRawMetrics(ch);
    ms.Stop();
}

// ---- the same three tests with the counters on the device (Options::device_counters): every Counter() call is an
// (id, amount) event summed by the GPU, Rates / Counters come back from the snapshot (lh_counters_collect)
static Options device_counter_options()
{
    Options o;
    o.device_counters = true;
    o.max_counters = 64;
    o.max_metrics = 8;
    return o;
}

TEST(TestRateOnDevice) // metrics_test.go:202-223
{
    MetricSystem ms(1us, false, device_counter_options());
    ms.Counter("rate1", 777);
    auto m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 777);
    ms.Counter("rate1", 1223);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 1223);
    ms.Counter("rate1", 1223);
    ms.Counter("rate1", 1223);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["rate1_rate"] == 2446 && m["rate1"] == 777 + 1223 + 2446);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics; // an idle interval: no rate key, the total stays
    CHECK(m.count("rate1_rate") == 0 && m["rate1"] == 4446);
}

TEST(TestCounterOnDevice) // metrics_test.go:225-240
{
    MetricSystem ms(1us, false, device_counter_options());
    ms.Counter("counter1", 3290);
    auto m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["counter1"] == 3290);
    ms.Counter("counter1", 10000);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(m["counter1"] == 13290);
}

TEST(TestCounterManyThreadsOnDevice)
{
    MetricSystem ms(1us, false, device_counter_options());
    std::vector<std::thread> th;
    for (int t = 0; t < 8; t++)
        th.emplace_back([&, t] {
            for (int i = 0; i < 10000; i++) {
                ms.Counter("c", 3);
                if ((i & 1023) == 0) ms.Histogram("h", 1.0 + t); // counters and histograms share the epoch
            }
        });
    for (auto &x : th) x.join();
    auto raw = ms.collectRawMetrics();
    CHECK(raw->Counters["c"] == 240000 && raw->Rates["c"] == 240000);
    auto m = ms.processMetrics(raw)->Metrics;
    CHECK(m["h_count"] == 80);
}

TEST(TestRawBroadcastOnDevice) // metrics_test.go:321-346
{
    auto ch = std::make_shared<RawCh>(128);
    MetricSystem ms(1ms, false, device_counter_options());
    ms.SubscribeToRawMetrics(ch);
    ms.Counter("counter2", 10);
    ms.Counter("counter2", 111);
    ms.Start();
    std::shared_ptr<RawMetricSet> raw;
    CHECK(ch->Receive(raw, 2s));
    if (raw) {
        CHECK(raw->Counters["counter2"] == 121);
        CHECK(raw->Rates["counter2"] == 121);
    }
    ms.UnsubscribeFromRawMetrics(ch);
    ms.Stop();
}

TEST(TestUpdateSubscribers) // metrics_test.go:242-287
{
    auto rc = std::make_shared<RawCh>(1);
    auto pc = std::make_shared<ProcCh>(1);
    MetricSystem ms(2ms, false);
    ms.SubscribeToRawMetrics(rc);
    ms.SubscribeToProcessedMetrics(pc);
    ms.Counter("counter5", 33);
    ms.Start();
    std::shared_ptr<RawMetricSet> r;
    std::shared_ptr<ProcessedMetricSet> p;
    CHECK(rc->Receive(r, 2s));
    ms.UnsubscribeFromRawMetrics(rc);
    CHECK(pc->Receive(p, 2s));
    ms.UnsubscribeFromProcessedMetrics(pc);
    std::this_thread::sleep_for(50ms);
    while (rc->TryReceive(r)) {}
    while (pc->TryReceive(p)) {}
    std::this_thread::sleep_for(50ms);
    CHECK(!rc->TryReceive(r) && !pc->TryReceive(p)); // nothing after unsubscribing
    ms.Stop();
}

TEST(TestSlowSubscriberIsDropped) // metrics.go:567-580
{
    auto pc = std::make_shared<ProcCh>(1);
    MetricSystem ms(1ms, false);
    ms.SubscribeToProcessedMetrics(pc);
    ms.Start();
    std::this_thread::sleep_for(100ms);
    ms.Stop();
    CHECK(pc->Closed());   // full twice in a row: forgotten and closed, the reaper never blocked
}

TEST(TestMetricSystemStop) // metrics_test.go:348-363
{
    auto threads = [] { std::FILE *f = std::fopen("/proc/self/status", "r"); char l[256]; int n = 0;
                        while (std::fgets(l, sizeof l, f)) if (!std::strncmp(l, "Threads:", 8)) n = std::atoi(l + 8);
                        std::fclose(f); return n; };
    const int before = threads();
    { MetricSystem ms(1us, false); ms.Start(); ms.Stop(); }
    std::this_thread::sleep_for(20ms);
    CHECK(threads() <= before);
}

TEST(TestSerializers) // graphite_test.go / opentsdb_test.go shape + exact line format
{
    ProcessedMetricSet pm;
    pm.Time = std::chrono::system_clock::time_point(std::chrono::seconds(1418352105));
    pm.Metrics["some_ipc_99.9"] = 1001.25;
    const std::string g = GraphiteProtocol(pm), o = OpenTSDBProtocol(pm);
    CHECK(g.rfind("cockroach.", 0) == 0 && g.find(".some.ipc.99.9 1001.250000 1418352105\n") != std::string::npos);
    CHECK(o.rfind("put some_ipc_99.9 1418352105 1001.250000 host=", 0) == 0 && o.back() == '\n');
}

// graphite_test.go:8-23 / opentsdb_test.go:8-23 run the serializer + submit against localhost:7777 without
// asserting anything; here a local sink is listening and the bytes are checked.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
TEST(TestSubmitterGraphite)
{
    int lfd = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    CHECK(bind(lfd, (sockaddr *)&a, sizeof a) == 0 && listen(lfd, 16) == 0);
    socklen_t l = sizeof a;
    getsockname(lfd, (sockaddr *)&a, &l);
    const int port = ntohs(a.sin_port);
    std::string received;
    std::atomic<bool> stop{false};
    std::thread sink([&] {
        char buf[65536];
        while (!stop.load()) {
            timeval tv{0, 50000};
            fd_set rs;
            FD_ZERO(&rs);
            FD_SET(lfd, &rs);
            if (select(lfd + 1, &rs, nullptr, nullptr, &tv) <= 0) continue;
            int c = accept(lfd, nullptr, nullptr);
            ssize_t n;
            while ((n = read(c, buf, sizeof buf)) > 0) received.append(buf, (size_t)n);
            close(c);
        }
    });
    {
        MetricSystem ms(5ms, false);
        Submitter s(&ms, GraphiteProtocol, "tcp", "127.0.0.1:" + std::to_string(port), 5ms);
        s.Start();
        ms.Counter("requests_total", 7);
        ms.Start();
        std::this_thread::sleep_for(150ms);
        s.Shutdown();
        ms.Stop();
        CHECK(s.sent_requests() >= 2);
        CHECK(s.connections() == 1); // persistent: every interval's requests over the one connection
        CHECK(s.DestinationNetwork == "tcp");
    }
    stop.store(true);
    sink.join();
    close(lfd);
    CHECK(received.find(".requests.total 7.000000 ") != std::string::npos);      // lifetime counter, '_' -> '.'
    CHECK(received.find(".requests.total.rate 7.000000 ") != std::string::npos); // first interval's rate
    CHECK(received.rfind("cockroach.", 0) == 0);
}


// submitter.go:70-104: requests that cannot be delivered wait in the backlog and leave, oldest first, once the
// destination answers -- here in one batch over one connection (no reference test covers the backlog).
TEST(TestSubmitterBacklogDrainsAfterTheSinkComesUp)
{
    // reserve a port, then close the listener: nothing answers at first
    int lfd = socket(AF_INET, SOCK_STREAM, 0);
    int one = 1;
    setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    CHECK(bind(lfd, (sockaddr *)&a, sizeof a) == 0);
    socklen_t l = sizeof a;
    getsockname(lfd, (sockaddr *)&a, &l);
    close(lfd);
    MetricSystem ms(5ms, false);
    Submitter s(&ms, GraphiteProtocol, "tcp", "127.0.0.1:" + std::to_string(ntohs(a.sin_port)), 5ms);
    s.Start();
    ms.Counter("queued_total", 3);
    ms.Start();
    std::this_thread::sleep_for(80ms);
    CHECK(s.sent_requests() == 0 && s.connections() == 0); // every interval's request is waiting
    lfd = socket(AF_INET, SOCK_STREAM, 0);
    setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    CHECK(bind(lfd, (sockaddr *)&a, sizeof a) == 0 && listen(lfd, 16) == 0);
    std::string received;
    std::thread sink([&] {
        char buf[65536];
        int c = accept(lfd, nullptr, nullptr);
        ssize_t n;
        while ((n = read(c, buf, sizeof buf)) > 0) received.append(buf, (size_t)n);
        close(c);
    });
    std::this_thread::sleep_for(80ms);
    s.Shutdown(); // closes the connection: the sink sees EOF
    ms.Stop();
    sink.join();
    close(lfd);
    CHECK(s.connections() == 1 && s.sent_requests() >= 10 && s.evicted_requests() == 0);
    // the first interval's request (the only one with a rate line for the counter) arrived, and arrived first
    const size_t first_rate = received.find(".queued.total.rate 3.000000 ");
    CHECK(first_rate != std::string::npos && first_rate < 200);
}

// The reference dials per request (submitter.go:106-116); the kept connection must notice a peer that closed or
// restarted BEFORE it writes a batch into the dead socket (the first write into a half-closed socket succeeds and the
// batch would be counted as sent and lost), and must not re-send what was delivered.
TEST(TestSubmitterRedialsAfterThePeerRestarts)
{
    int lfd = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    CHECK(bind(lfd, (sockaddr *)&a, sizeof a) == 0 && listen(lfd, 16) == 0);
    socklen_t l = sizeof a;
    getsockname(lfd, (sockaddr *)&a, &l);
    std::string received;
    std::atomic<int> accepted{0};
    std::thread sink([&] {
        char buf[65536];
        // first connection: take whatever the first batch holds, then "restart"
        int c = accept(lfd, nullptr, nullptr);
        accepted.fetch_add(1);
        ssize_t n = read(c, buf, sizeof buf);
        if (n > 0) received.append(buf, (size_t)n);
        std::this_thread::sleep_for(2ms); // the rest of a batch that is on the wire
        while ((n = recv(c, buf, sizeof buf, MSG_DONTWAIT)) > 0) received.append(buf, (size_t)n);
        close(c);
        // second connection: until the submitter shuts down
        c = accept(lfd, nullptr, nullptr);
        accepted.fetch_add(1);
        while ((n = read(c, buf, sizeof buf)) > 0) received.append(buf, (size_t)n);
        close(c);
    });
    uint64_t sent = 0;
    {
        MetricSystem ms(20ms, false);
        Submitter s(&ms, GraphiteProtocol, "tcp", "127.0.0.1:" + std::to_string(ntohs(a.sin_port)), 20ms);
        s.Start();
        ms.Counter("restart_total", 1);
        ms.Start();
        std::this_thread::sleep_for(400ms);
        s.Shutdown();
        ms.Stop();
        sent = s.sent_requests();
        CHECK(s.connections() == 2);
    }
    sink.join();
    close(lfd);
    // every request carries the lifetime counter once: as many of its lines arrived as requests were counted as sent
    size_t lines = 0;
    for (size_t at = 0; (at = received.find(".restart.total 1.000000 ", at)) != std::string::npos; at++) lines++;
    CHECK(accepted.load() == 2 && sent >= 5);
    if (lines != sent) std::printf("    lines %zu sent %llu\n", lines, (unsigned long long)sent);
    CHECK(lines == sent);
}

// ---- histogram paths: need the GPU ---------------------------------------------------------------
TEST(TestTimer) // metrics_test.go:183-200
{
    MetricSystem ms(1us, false);
    auto t1 = ms.StartTimer("timer1");
    auto t2 = ms.StartTimer("timer1");
    std::this_thread::sleep_for(50us);
    t1.Stop();
    std::this_thread::sleep_for(5us);
    t2.Stop();
    auto t3 = ms.StartTimer("timer1");
    std::this_thread::sleep_for(10us);
    t3.Stop();
    auto r = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(ms.engine_ready());
    CHECK(r["timer1_count"] == 3);
    CHECK(!(r["timer1_min"] > r["timer1_50"] || r["timer1_50"] > r["timer1_max"]));
}

TEST(TestProcessedBroadcast) // metrics_test.go:289-319
{
    auto ch = std::make_shared<ProcCh>(128);
    MetricSystem ms(1ms, false);
    ms.SubscribeToProcessedMetrics(ch);
    ms.Histogram("histogram1", 33);
    ms.Histogram("histogram1", 59);
    ms.Histogram("histogram1", 330000);
    ms.Start();
    std::shared_ptr<ProcessedMetricSet> pm;
    CHECK(ch->Receive(pm, 5s));
    if (pm) {
        CHECK((int)pm->Metrics["histogram1_sum"] == 331132);
        CHECK((int)pm->Metrics["histogram1_agg_avg"] == 110377);
        CHECK((int)pm->Metrics["histogram1_count"] == 3);
    }
    ms.UnsubscribeFromProcessedMetrics(ch);
    ms.Stop();
}

TEST(ExampleMetricSystem) // metrics_test.go:28-109: presence of the documented keys
{
    auto ch = std::make_shared<ProcCh>(2);
    MetricSystem ms(2ms, true);
    ms.SubscribeToProcessedMetrics(ch);
    ms.RegisterGaugeFunc("gauge", [] { return 33.0; });
    auto tok = ms.StartTimer("submit_metrics");
    ms.Counter("range_splits", 1);
    ms.Histogram("some_ipc", 123);
    tok.Stop();
    ms.Start();
    std::shared_ptr<ProcessedMetricSet> pm;
    CHECK(ch->Receive(pm, 5s));
    if (pm) {
        for (const char *k : {"range_splits", "range_splits_rate", "some_ipc_99.9", "some_ipc_max", "some_ipc_count",
                              "some_ipc_agg_count", "some_ipc_sum", "some_ipc_avg", "some_ipc_agg_avg",
                              "submit_metrics_sum", "sys.NumGoroutine", "sys.PauseTotalNs", "gauge"})
            CHECK(pm->Metrics.count(k));
        CHECK(pm->Metrics["some_ipc_count"] == 1);
        CHECK(std::fabs(pm->Metrics["some_ipc_max"] / 123 - 1) < 0.01);
        CHECK(pm->Metrics["some_ipc_max"] == 122.96509077982394); // decompress(482), bit-exact
    }
    ms.Stop();
}

TEST(TestRawHistogramsAndInvalidPercentile)
{
    MetricSystem ms(1us, false);
    ms.SpecifyPercentiles({{"%s_p50", 0.5}, {"%s_bad", 1.5}});
    for (double v : {33.0, 59.0, 330000.0, 33.0}) ms.Histogram("h", v);
    auto raw = ms.collectRawMetrics();
    auto &h = raw->Histograms();
    CHECK(h.count("h") && h.at("h").size() == 3);
    if (h.count("h")) {
        CHECK(h.at("h").at(353) == 2 && h.at("h").at(409) == 1 && h.at("h").at(1271) == 1); // compress keys
    }
    auto m = ms.processMetrics(raw)->Metrics;
    CHECK(m.count("h_p50") && !m.count("h_bad")); // metrics.go:379-384: error logged, key omitted
    CHECK(m["h_p50"] == 33.123967614754356);      // decompress(353)
    // next interval: "h" received nothing -> absent, like a name missing from histogramCache
    ms.Histogram("other", 1.0);
    m = ms.processMetrics(ms.collectRawMetrics())->Metrics;
    CHECK(!m.count("h_count") && m["other_count"] == 1);
}

TEST(TestProducersAreLosslessAcrossIntervals)
{
    // a sample belongs to exactly one interval (metrics.go:460-463): many producers, the collector
    // flipping concurrently; the per-interval counts must add up to exactly what was submitted
    MetricSystem ms(1us, false);
    const int T = 8, N = 200000;
    std::atomic<bool> done{false};
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
        th.emplace_back([&] {
            for (int i = 0; i < N; i++) {
                ms.Histogram(i % 3 ? "a" : "b", 100.0 + i % 977);
                if (i % 5 == 0) ms.Counter("ev", 1);
            }
        });
    double total = 0, ev = 0;
    std::thread collector([&] {
        while (!done.load()) {
            auto raw = ms.collectRawMetrics();
            auto m = ms.processMetrics(raw)->Metrics;
            raw->Release();
            total += (m.count("a_count") ? m["a_count"] : 0) + (m.count("b_count") ? m["b_count"] : 0);
            ev += m.count("ev_rate") ? m["ev_rate"] : 0;
            std::this_thread::sleep_for(2ms);
        }
    });
    for (auto &x : th) x.join();
    done.store(true);
    collector.join();
    auto raw = ms.collectRawMetrics();
    auto m = ms.processMetrics(raw)->Metrics;
    total += (m.count("a_count") ? m["a_count"] : 0) + (m.count("b_count") ? m["b_count"] : 0);
    ev += m.count("ev_rate") ? m["ev_rate"] : 0;
    CHECK(total == (double)T * N);
    CHECK(ev == (double)T * (N / 5));
    CHECK(m["ev"] == (double)T * (N / 5)); // lifetime counter
    CHECK(ms.last_status() == 0 || ms.last_status() == 5 /* LH_EBUSY is benign here */);
    // the engine's self-metrics, published as gauges
    ms.RegisterEngineGauges();
    auto g = ms.collectRawMetrics()->Gauges;
    const double seen = g["loghisto.gpu.samples_small"] + g["loghisto.gpu.samples_partitioned"] +
                        g["loghisto.gpu.samples_direct"] + g["loghisto.gpu.samples_single"];
    CHECK(seen == (double)T * N && g["loghisto.gpu.launches"] > 0);
}

TEST(TestFormatGoV)
{
    // fmt.Println of a float64, on the values the reference's own docs print (readme.md:35-43,
    // print_benchmark.go:31-48): shortest round-trip digits, %e from 1e6 upwards and below 1e-4
    CHECK(FormatGoV(2.4642914167480484e+07) == "2.4642914167480484e+07");
    CHECK(FormatGoV(4913.768840299134) == "4913.768840299134");
    CHECK(FormatGoV(58.739891704145194) == "58.739891704145194");
    CHECK(FormatGoV(-657.5233632152207) == "-657.5233632152207");
    CHECK(FormatGoV(3.982478339757623e+07) == "3.982478339757623e+07");
    CHECK(FormatGoV(3.4366224772310276e+06) == "3.4366224772310276e+06");
    CHECK(FormatGoV(469769.7083161708) == "469769.7083161708");
    CHECK(FormatGoV(129313.15075081984) == "129313.15075081984");
    CHECK(FormatGoV(9.975892639594093e+09) == "9.975892639594093e+09");
    CHECK(FormatGoV(605039.5827022133) == "605039.5827022133");
    CHECK(FormatGoV(16488) == "16488" && FormatGoV(618937) == "618937" && FormatGoV(121095) == "121095");
    CHECK(FormatGoV(7.4950269894e+10) == "7.4950269894e+10" && FormatGoV(2.94946542e+08) == "2.94946542e+08");
    CHECK(FormatGoV(997328) == "997328" && FormatGoV(1e6) == "1e+06" && FormatGoV(100000) == "100000");
    CHECK(FormatGoV(0) == "0" && FormatGoV(0.0001) == "0.0001" && FormatGoV(0.00001) == "1e-05");
    CHECK(FormatGoV(0.5) == "0.5" && FormatGoV(1e100) == "1e+100" && FormatGoV(-2.5e-7) == "-2.5e-07");
    CHECK(FormatGoV(NAN) == "NaN" && FormatGoV(INFINITY) == "+Inf" && FormatGoV(-INFINITY) == "-Inf");
}

TEST(TestPrintBenchmark) // print_benchmark.go:49-106
{
    std::FILE *f = std::tmpfile();
    std::atomic<uint64_t> calls{0};
    PrintBenchmark("raft_AppendLogEntries", 3, [&] {
        calls.fetch_add(1);
        std::this_thread::sleep_for(200us);
    }, 2500ms, f);
    std::rewind(f);
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    CHECK(calls.load() > 1000);
    // at least one full interval: a time line, the 19 keys in the reference's order, a blank line
    const char *order[] = {"_count:", "_max:", "_99.99:", "_99.9:", "_99:", "_95:", "_90:", "_75:", "_50:", "_min:",
                           "_sum:", "_avg:", "_agg_avg:", "_agg_count:", "_agg_sum:"};
    size_t pos = 0;
    for (const char *k : order) {
        const size_t at = text.find(std::string("raft_AppendLogEntries") + k, pos);
        CHECK(at != std::string::npos);
        if (at == std::string::npos) break;
        pos = at;
    }
    for (const char *k : {"sys.Alloc:", "sys.NumGC:", "sys.PauseTotalNs:", "sys.NumGoroutine:"})
        CHECK(text.find(k, pos) != std::string::npos);
    // tabwriter shape: the widest cell "raft_AppendLogEntries_agg_count:" is 32 wide -> no tab, the others pad
    // with tabs to column 32
    CHECK(text.find("raft_AppendLogEntries_agg_count: ") != std::string::npos);
    CHECK(text.find("raft_AppendLogEntries_max:\t ") != std::string::npos);
    CHECK(text.find("sys.NumGC:\t\t\t 0\n") != std::string::npos);
    // a sleep of 200 us is timed at >= 200 000 ns, reported as its bucket's value (within 1 % below or above); a loaded
    // box may stretch the sleeps, hence the loose upper bound: the median prints in Go's %v form
    // (of an interval that saw the workers for its whole length: intervals are aligned to the wall clock, so the first
    // one printed may have lasted a few milliseconds and hold no call at all -- every key of it prints 0)
    size_t busy = std::string::npos;
    for (size_t at = text.find("raft_AppendLogEntries_count:"); at != std::string::npos;
         at = text.find("raft_AppendLogEntries_count:", at + 1)) {
        if (std::atof(text.c_str() + text.find(' ', at)) >= 500.0) { busy = at; break; }
    }
    CHECK(busy != std::string::npos);
    const size_t p50 = busy == std::string::npos ? busy : text.find("raft_AppendLogEntries_50:", busy);
    CHECK(p50 != std::string::npos);
    if (p50 != std::string::npos) {
        const double v = std::atof(text.c_str() + text.find(' ', p50));
        // lower bound: a guarantee (steady_clock around a sleep of 200 us); upper bound: a sanity check only -- on a box
        // that is still paging the image in, sleeps of the first interval have been seen stretched a lot
        CHECK(v >= 197000.0);
        CHECK(v < 6e10);
        if (!(v >= 197000.0 && v < 6e10)) std::printf("    p50 = %.17g in\n%s\n", v, text.substr(0, 3000).c_str());
    }
}

static std::vector<std::string> sorted_lines(const std::string &text)
{
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos < text.size()) {
        const size_t nl = text.find('\n', pos);
        out.push_back(text.substr(pos, nl - pos));
        pos = nl == std::string::npos ? text.size() : nl + 1;
    }
    std::sort(out.begin(), out.end());
    return out;
}

TEST(TestBulkWireMatchesPerKeySerializer)
{
    // SetWireFormat: the request prepared on the GPU (lh_serialize) holds exactly the lines the per-key
    // serializer builds from the map (graphite.go:37-48, opentsdb.go:45-58), _agg_* keys included
    for (WireFormat wf : {WireFormat::Graphite, WireFormat::OpenTSDB}) {
        MetricSystem ms(1us, false);
        ms.SetWireFormat(wf, true);
        ms.RegisterGaugeFunc("some_gauge", [] { return 12.5; });
        for (int interval = 0; interval < 3; interval++) {
            for (int i = 0; i < 20000; i++) {
                ms.Histogram("rpc_latency_" + std::to_string(i % 7), 1000.0 + (i * 37) % 90001 + interval);
                if (interval != 1) ms.Histogram("only.sometimes", 0.25 * i);
            }
            ms.Counter("requests_total", 5 + interval);
            auto raw = ms.collectRawMetrics();
            auto pm = ms.processMetrics(raw);
            ms.addAggregates(raw, *pm);
            raw->Release();
            CHECK(pm->wire_format == wf && !pm->wire.empty());
            ProcessedMetricSet plain;
            plain.Time = pm->Time;
            plain.Metrics = pm->Metrics;
            const std::string per_key = wf == WireFormat::Graphite ? GraphiteProtocol(plain) : OpenTSDBProtocol(plain);
            const std::string bulk = wf == WireFormat::Graphite ? GraphiteProtocol(*pm) : OpenTSDBProtocol(*pm);
            CHECK(bulk == pm->wire);
            const auto a = sorted_lines(per_key), b = sorted_lines(bulk);
            CHECK(a.size() == b.size() && a.size() >= 7 * 15 + 3);
            CHECK(a == b);
            if (a != b)
                for (size_t i = 0; i < std::min(a.size(), b.size()); i++)
                    if (a[i] != b[i]) { std::printf("    per-key: %s\n    bulk:    %s\n", a[i].c_str(), b[i].c_str()); break; }
        }
        // without the map: the histogram keys exist only as text
        ms.SetWireFormat(wf, false);
        ms.Histogram("rpc_latency_0", 42.0);
        ms.Counter("requests_total", 1);
        auto raw = ms.collectRawMetrics();
        auto pm = ms.processMetrics(raw);
        raw->Release();
        CHECK(!pm->Metrics.count("rpc_latency_0_count") && pm->Metrics.count("requests_total"));
        CHECK(pm->wire.find(wf == WireFormat::Graphite ? "rpc.latency.0.count 1.000000" : "rpc_latency_0_count") !=
              std::string::npos);
        CHECK(pm->wire.find(wf == WireFormat::Graphite ? "rpc.latency.0.agg.count 8575.000000"
                                                       : "rpc_latency_0_agg_count") != std::string::npos);
        CHECK(ms.last_status() == 0);
    }
}

int main(int argc, char **argv)
{
    const bool cpu_only = argc > 1 && !std::strcmp(argv[1], "--cpu");
    RUN(TestRate);
    RUN(TestCounter);
    RUN(TestCounterManyThreads);
    RUN(TestSysStats);
    RUN(TestRawBroadcast);
    RUN(TestUpdateSubscribers);
    RUN(TestSlowSubscriberIsDropped);
    RUN(TestMetricSystemStop);
    RUN(TestSerializers);
    RUN(TestSubmitterGraphite);
    RUN(TestSubmitterBacklogDrainsAfterTheSinkComesUp);
    RUN(TestSubmitterRedialsAfterThePeerRestarts);
    RUN(TestFormatGoV);
    if (!cpu_only) {
        RUN(TestTimer);
        RUN(TestProcessedBroadcast);
        RUN(ExampleMetricSystem);
        RUN(TestRawHistogramsAndInvalidPercentile);
        RUN(TestProducersAreLosslessAcrossIntervals);
        RUN(TestBulkWireMatchesPerKeySerializer);
        RUN(TestPrintBenchmark);
        RUN(TestRateOnDevice);
        RUN(TestCounterOnDevice);
        RUN(TestCounterManyThreadsOnDevice);
        RUN(TestRawBroadcastOnDevice);
    }
    std::printf("%d checks, %d failed tests\n", g_checks, g_failed);
    return g_failed;
}
