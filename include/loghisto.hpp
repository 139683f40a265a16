// loghisto.hpp -- C++ host layer with the reference's MetricSystem API over the C ABI
// (include/loghisto_gpu.h).
//
// The reference is Go; no Go toolchain exists in the build image, so the host side a cgo
// binding would provide is written in C++ with the same names, argument
// meaning, key naming and error behaviour as /root/reference/metrics.go:
//
//   NewMetricSystem(interval, sysStats)            metrics.go:143   -> MetricSystem(...)
//   SpecifyPercentiles                             metrics.go:199
//   Subscribe/UnsubscribeTo{Raw,Processed}Metrics  metrics.go:205-228  (Channel<T> ~ buffered Go chan)
//   StartTimer / TimerToken.Stop                   metrics.go:232-246
//   Counter / Histogram                            metrics.go:251-295
//   RegisterGaugeFunc / DeregisterGaugeFunc        metrics.go:299-310
//   collectRawMetrics / processMetrics             metrics.go:420-506 (unexported in Go; public here
//                                                   because the reference's tests call them)
//   Start / Stop, reaper                           metrics.go:530-653
//   GraphiteProtocol / OpenTSDBProtocol            graphite.go:73, opentsdb.go:83
//   NewSubmitter / Start / Shutdown                submitter.go:52-159 -> Submitter
//   PrintBenchmark                                 print_benchmark.go:49
//
// What moved to the GPU: compress + the per-(name,bucket) fan-in (Histogram), the epoch flip of the
// histogram cells (collectRawMetrics) and processHistograms/percentile (processMetrics).  Counters,
// rates, gauges, subscriptions and the reaper stay on the host exactly as in the reference.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

struct lh_engine;
struct lh_snapshot;

namespace loghisto {

// Buffered channel with Go's non-blocking-send semantics (select { case ch <- v: default: }).
template <class T> class Channel {
public:
    explicit Channel(size_t capacity) : cap_(capacity) {}
    bool TrySend(T v)
    {
        std::lock_guard<std::mutex> g(mu_);
        if (closed_ || q_.size() >= cap_) return false;
        q_.push_back(std::move(v));
        cv_.notify_one();
        return true;
    }
    // false on timeout or when closed and drained
    bool Receive(T &out, std::chrono::nanoseconds timeout)
    {
        std::unique_lock<std::mutex> g(mu_);
        // system_clock deadline -> pthread_cond_timedwait (the steady-clock form, pthread_cond_clockwait,
        // is not intercepted by GCC 11's ThreadSanitizer and floods it with false positives)
        if (!cv_.wait_until(g, std::chrono::system_clock::now() + timeout, [&] { return !q_.empty() || closed_; }))
            return false;
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
    bool TryReceive(T &out) { return Receive(out, std::chrono::nanoseconds(0)); }
    void Close()
    {
        std::lock_guard<std::mutex> g(mu_);
        closed_ = true;
        cv_.notify_all();
    }
    bool Closed()
    {
        std::lock_guard<std::mutex> g(mu_);
        return closed_;
    }
    size_t Len()
    {
        std::lock_guard<std::mutex> g(mu_);
        return q_.size();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

// Bulk wire path (no counterpart in the reference): which serializer's text the reaper
// prepares on the GPU while it still holds the interval's snapshot (MetricSystem::SetWireFormat).
enum class WireFormat { None = 0, Graphite = 1, OpenTSDB = 2 };

// metrics.go:47-50
struct ProcessedMetricSet {
    std::chrono::system_clock::time_point Time;
    std::unordered_map<std::string, double> Metrics;

    // Filled only when a wire format is set: the complete request GraphiteProtocol / OpenTSDBProtocol would
    // build from this interval -- histogram keys formatted on the device by lh_serialize, counters / rates /
    // gauges appended by the host -- so the serializer returns it instead of one snprintf per key.
    WireFormat wire_format = WireFormat::None;
    std::string wire;
};

// metrics.go:54-60.  Histograms holds the occupied (key -> count) cells per name; it is
// materialised from the device only when somebody asks (raw subscribers, tests).
struct RawMetricSet {
    std::chrono::system_clock::time_point Time;
    std::unordered_map<std::string, uint64_t> Counters;
    std::unordered_map<std::string, uint64_t> Rates;
    std::unordered_map<std::string, double> Gauges;
    const std::unordered_map<std::string, std::map<int16_t, uint64_t>> &Histograms();
    ~RawMetricSet();

    // implementation state
    lh_snapshot *snapshot = nullptr;
    std::vector<std::string> names; // dense id -> name at flip time
    std::mutex mu;
    bool hist_ready = false;
    std::unordered_map<std::string, std::map<int16_t, uint64_t>> hist;
    void Release(); // return the epoch buffer (called by processMetrics' caller / destructor)
};

class MetricSystem;

// metrics.go:63-67
struct TimerToken {
    std::string Name;
    std::chrono::steady_clock::time_point Start;
    MetricSystem *System;
    // metrics.go:242-246: submits float64(duration in ns) as a histogram sample, returns the duration
    std::chrono::nanoseconds Stop();
};

struct Options {
    int device = 0;
    uint32_t max_metrics = 1024;  // histogram names
    uint32_t num_buffers = 3;     // intervals that may be in flight (flip .. processed)
    uint32_t num_lanes = 8;       // staging lanes inside the engine
    uint32_t stage_samples = 4096; // per-thread (id,value) pairs per crossing into the library
    uint64_t lane_samples = 1u << 20;
    // Counters on the device (lh_submit_counts / lh_counters_collect): every Counter() call is staged as an
    // (id, amount) event next to the histogram samples and summed on the GPU; Rates / Counters of the interval
    // come back from the snapshot.  Off: per-thread host maps merged at the flip (the reference's own shape,
    // metrics.go:251-269, 425-458; also what runs when no device is present).
    bool device_counters = false;
    uint32_t max_counters = 4096;
};

class MetricSystem {
public:
    // NewMetricSystem, metrics.go:143
    MetricSystem(std::chrono::nanoseconds interval, bool sysStats, const Options &opt = Options());
    ~MetricSystem();
    MetricSystem(const MetricSystem &) = delete;
    MetricSystem &operator=(const MetricSystem &) = delete;

    void SpecifyPercentiles(const std::map<std::string, double> &percentiles); // label is a "%s" format
    void SubscribeToRawMetrics(std::shared_ptr<Channel<std::shared_ptr<RawMetricSet>>> ch);
    void UnsubscribeFromRawMetrics(std::shared_ptr<Channel<std::shared_ptr<RawMetricSet>>> ch);
    void SubscribeToProcessedMetrics(std::shared_ptr<Channel<std::shared_ptr<ProcessedMetricSet>>> ch);
    void UnsubscribeFromProcessedMetrics(std::shared_ptr<Channel<std::shared_ptr<ProcessedMetricSet>>> ch);

    TimerToken StartTimer(const std::string &name);
    void Counter(const std::string &name, uint64_t amount);
    void Histogram(const std::string &name, double value);
    void RegisterGaugeFunc(const std::string &name, std::function<double()> f);
    // opt-in: publishes the engine's self-metrics (lh_get_counters) as gauges "<prefix>samples_small", ...
    void RegisterEngineGauges(const std::string &prefix = "loghisto.gpu.");
    void DeregisterGaugeFunc(const std::string &name);

    std::shared_ptr<RawMetricSet> collectRawMetrics();
    std::shared_ptr<ProcessedMetricSet> processMetrics(const std::shared_ptr<RawMetricSet> &raw);
    // the `_agg_*` keys the reaper adds after processMetrics, metrics.go:590-608
    void addAggregates(const std::shared_ptr<RawMetricSet> &raw, ProcessedMetricSet &processed);

    void Start();
    void Stop();

    // Opt-in bulk wire path: processMetrics also calls lh_snapshot_accumulate + lh_serialize and attaches the
    // text to the ProcessedMetricSet.  With histogram_keys_in_map = false the histogram keys are NOT inserted
    // into ProcessedMetricSet::Metrics (65 536 names x 15 keys is ~1e6 string-keyed inserts per interval);
    // counters, rates and gauges always are.
    void SetWireFormat(WireFormat f, bool histogram_keys_in_map = true);

    // diagnostics
    int last_status() const { return last_status_.load(); } // last non-zero lh_* code (0 if none)
    uint64_t dropped_intervals() const { return dropped_intervals_.load(); }
    bool engine_ready() const { return engine_ != nullptr; }

private:
    struct Stage;
    Stage *stage();
    void ship(Stage &s);
    void ship_counts(Stage &s);
    bool ensure_engine();
    uint32_t intern(const std::string &name);
    void updateSubscribers();
    void reaper();
    int note(int rc, const char *where); // returns rc
    void serializeHistograms(RawMetricSet &raw, const std::vector<std::string> &labels, const std::vector<double> &ps,
                             WireFormat wf, std::string &text);
    const double *decompressTable(); // decompress(key) for every key, by dense bin (lh_codec_tables, read back once)

    std::chrono::nanoseconds interval_;
    Options opt_;
    std::map<std::string, double> percentiles_;
    std::mutex percentiles_mu_;

    lh_engine *engine_ = nullptr;
    bool narrow_ids_ = false;      // max_metrics <= 65 536: stages ship uint16 ids (lh_reserve_pairs16)
    std::mutex engine_mu_;

    std::atomic<int> wire_format_{0};
    std::atomic<bool> wire_keep_map_{true};
    std::atomic<size_t> wire_bytes_hint_{0};
    std::once_flag dtable_once_;
    std::vector<double> dtable_;

    // histogramMu (metrics.go:121): submitters shared, the flip exclusive
    std::shared_mutex histogram_mu_;
    std::mutex stages_mu_;
    std::vector<std::unique_ptr<Stage>> stages_;
    alignas(64) std::atomic<bool> hist_used_{false}; // own cache line: written once, read by every producer
    alignas(64) std::atomic<bool> counters_used_{false}; // device counters received an event
    alignas(64) uint64_t instance_id_;

    std::mutex names_mu_;
    std::vector<std::string> names_;

    // counters: host side as in the reference (metrics.go:112-117), or -- Options::device_counters -- a mirror of
    // the lifetime store that lives in HBM (lh_counters_collect)
    std::mutex counter_store_mu_;
    std::unordered_map<std::string, uint64_t> counter_store_;        // lifetime totals as exported: device + host-staged
    std::unordered_map<std::string, uint64_t> host_counter_store_;   // lifetime amounts that never reached the device
    std::unordered_map<std::string, uint64_t> device_counter_total_; // last totals the device reported
    std::vector<std::string> counter_names_; // device counter id -> name (counter_store_mu_)

    // lifetime histogram aggregates (metrics.go:122-125)
    // (histogramCountStore, metrics.go:122-127, lives in HBM: lh_snapshot_accumulate / lh_lifetime)

    std::mutex gauge_mu_;
    std::unordered_map<std::string, std::function<double()>> gauge_funcs_;

    using RawCh = std::shared_ptr<Channel<std::shared_ptr<RawMetricSet>>>;
    using ProcCh = std::shared_ptr<Channel<std::shared_ptr<ProcessedMetricSet>>>;
    struct SubOp { int kind; RawCh raw; ProcCh proc; };
    std::mutex pending_mu_;
    std::vector<SubOp> pending_;
    std::mutex subs_mu_;
    std::vector<RawCh> raw_subs_;
    std::vector<ProcCh> proc_subs_;
    std::unordered_map<void *, int> raw_bad_, proc_bad_;

    std::atomic<bool> reaping_{false};
    std::atomic<bool> shutdown_{false};
    std::mutex shutdown_mu_;
    std::condition_variable shutdown_cv_;
    std::thread reaper_thread_;
    alignas(64) std::atomic<int> last_status_{0};
    std::atomic<uint64_t> dropped_intervals_{0};
};

// submitter.go:27-159: subscribes a 60-deep channel, serialises every ProcessedMetricSet, keeps the last 60
// requests in an evicting ring and (re)sends the backlog once per interval over a fresh TCP/UDP connection
// with a 5 s deadline.  Pure host I/O downstream of the hot path; kept so that the API surface is whole.
class Submitter {
public:
    using Serializer = std::function<std::string(const ProcessedMetricSet &)>;
    // NewSubmitter(metricSystem, serializer, destinationNetwork ("tcp" | "udp"), destinationAddress "host:port")
    Submitter(MetricSystem *ms, Serializer serializer, std::string network, std::string address,
              std::chrono::nanoseconds interval);
    ~Submitter();
    void Start();
    void Shutdown();
    std::string DestinationNetwork, DestinationAddress;
    uint64_t sent_requests() const { return sent_.load(); }
    uint64_t evicted_requests() const { return evicted_.load(); }
    uint64_t connections() const { return connects_.load(); } // dials so far (tcp: one while the peer stays up)

private:
    bool connectIfNeeded();
    void disconnect();
    bool submitBatch(const std::shared_ptr<const std::string> *requests, size_t n, size_t *done);
    bool retryBacklog();
    void appendToBacklog(std::string request);
    MetricSystem *ms_;
    Serializer serializer_;
    std::chrono::nanoseconds interval_;
    std::shared_ptr<Channel<std::shared_ptr<ProcessedMetricSet>>> chan_;
    std::mutex backlog_mu_;
    std::shared_ptr<const std::string> backlog_[60];
    int head_ = 0, tail_ = 0;
    uint64_t head_seq_ = 0; // sequence number of the entry at head_ (evictions and sends both advance it)
    std::atomic<bool> shutdown_{false};
    std::atomic<uint64_t> sent_{0}, evicted_{0}, connects_{0};
    int fd_ = -1; // send thread only
    std::thread recv_thread_, send_thread_;
};

// graphite.go:73 / opentsdb.go:83: "cockroach.<host>.<metric with _ -> .> %f %d\n" and
// "put <metric> <unix> %f host=<host>\n".  Downstream text formatting; kept for config 5.
std::string GraphiteProtocol(const ProcessedMetricSet &ms);
std::string OpenTSDBProtocol(const ProcessedMetricSet &ms);

// print_benchmark.go:49-106: runs `op` on `concurrency` threads, timing every call into the histogram `name`
// of a 1 s MetricSystem, and once per interval prints the same 19 keys in the same order through a
// tabwriter-shaped table (values as Go's %v prints a float64).  The reference never returns; `run_for` == 0
// keeps that behaviour, a positive duration stops the workers and returns (tests, scripted runs).
void PrintBenchmark(const std::string &name, unsigned concurrency, std::function<void()> op,
                    std::chrono::nanoseconds run_for = std::chrono::nanoseconds(0), std::FILE *out = stdout,
                    const Options &opt = Options());
// fmt.Println's rendering of a float64 (%v = %g with the shortest round-trip digits; exponent form when the
// decimal exponent is < -4 or >= 6: "2.4642914167480484e+07" but "469769.7083161708", readme.md:35-43)
std::string FormatGoV(double v);

} // namespace loghisto
