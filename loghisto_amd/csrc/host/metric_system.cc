// metric_system.cc -- C++ host layer: the reference's MetricSystem API over the C ABI.
// See include/loghisto.hpp for the mapping to /root/reference/metrics.go.
//
// Concurrency model.  The reference serialises every Histogram call on one RWMutex word plus
// four map probes (metrics.go:275-279) -- that, not the logarithm, is what limits it.  Here each
// producer thread owns a staging buffer guarded by its own (uncontended) mutex; the epoch flip
// takes every stage mutex, drains the buffers, calls lh_flip and releases them, which gives the
// same "a sample belongs to exactly one interval" cut (metrics.go:460-463) without any shared
// cache line on the submit path.  One crossing into the library moves `stage_samples` samples.
#include "../../../include/loghisto.hpp"
#include "../../../include/loghisto_gpu.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cerrno>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace loghisto {

namespace {

std::atomic<uint64_t> g_next_instance{1};

std::string fmt_label(const std::string &label, const std::string &name)
{
    // labels are Go format strings with one %s (metrics.go:145-155, 383)
    std::string out;
    out.reserve(label.size() + name.size());
    for (size_t i = 0; i < label.size(); i++) {
        if (label[i] == '%' && i + 1 < label.size() && label[i + 1] == 's') {
            out += name;
            i++;
        } else if (label[i] == '%' && i + 1 < label.size() && label[i + 1] == '%') {
            out += '%';
            i++;
        } else {
            out += label[i];
        }
    }
    return out;
}

std::chrono::system_clock::time_point truncate_to(std::chrono::nanoseconds interval)
{
    using namespace std::chrono;
    const int64_t ivl = std::max<int64_t>(1, interval.count());
    const int64_t now = duration_cast<nanoseconds>(system_clock::now().time_since_epoch()).count();
    return system_clock::time_point(duration_cast<system_clock::duration>(nanoseconds(now / ivl * ivl)));
}

double read_proc_status_kb(const char *key)
{
    std::ifstream f("/proc/self/status");
    std::string line;
    const size_t klen = std::strlen(key);
    while (std::getline(f, line))
        if (line.compare(0, klen, key) == 0) return std::atof(line.c_str() + klen + 1);
    return 0;
}

std::string hostname()
{
    char buf[256];
    if (gethostname(buf, sizeof(buf)) != 0) return "unknown";
    buf[sizeof(buf) - 1] = 0;
    return buf;
}

} // namespace

struct MetricSystem::Stage {
    std::mutex mu;
    std::vector<uint32_t> ids;
    std::vector<double> vals;
    size_t n = 0;
    std::unordered_map<std::string, uint32_t> idcache;  // name -> dense id, thread private
    std::unordered_map<std::string, uint64_t> counters; // counterCache share of this thread (host counters)
    // device counters: (counter id, amount) events, shipped with lh_submit_counts
    std::vector<uint32_t> cids;
    std::vector<uint64_t> camts;
    size_t cn = 0;
    std::unordered_map<std::string, uint32_t> cidcache;
};

// ---------------------------------------------------------------------------
// RawMetricSet
// ---------------------------------------------------------------------------
const std::unordered_map<std::string, std::map<int16_t, uint64_t>> &RawMetricSet::Histograms()
{
    std::lock_guard<std::mutex> g(mu);
    if (hist_ready) return hist;
    hist_ready = true;
    if (!snapshot || names.empty()) return hist;
    // one crossing for every name: device-compacted CSR listing of the occupied cells
    const size_t n = names.size();
    std::vector<uint64_t> offsets(n + 1);
    size_t total = 0;
    if (lh_buckets_all(snapshot, 0, n, offsets.data(), nullptr, nullptr, 0, &total) != LH_OK) return hist;
    std::vector<int16_t> keys(total);
    std::vector<uint64_t> counts(total);
    if (total && lh_buckets_all(snapshot, 0, n, offsets.data(), keys.data(), counts.data(), total, &total) != LH_OK)
        return hist;
    for (size_t id = 0; id < n; id++) {
        if (offsets[id] == offsets[id + 1]) continue; // the name has no map entry this interval
        auto &m = hist[names[id]];
        for (uint64_t i = offsets[id]; i < offsets[id + 1]; i++) m.emplace_hint(m.end(), keys[i], counts[i]);
    }
    return hist;
}

void RawMetricSet::Release()
{
    std::lock_guard<std::mutex> g(mu);
    if (snapshot) {
        lh_release(snapshot);
        snapshot = nullptr;
    }
}

RawMetricSet::~RawMetricSet() { Release(); }

// ---------------------------------------------------------------------------
// MetricSystem
// ---------------------------------------------------------------------------
MetricSystem::MetricSystem(std::chrono::nanoseconds interval, bool sysStats, const Options &opt)
    : interval_(interval), opt_(opt), instance_id_(g_next_instance.fetch_add(1))
{
    // metrics.go:145-155
    percentiles_ = {{"%s_min", 0},    {"%s_50", .5},    {"%s_75", .75},     {"%s_90", .9}, {"%s_95", .95},
                    {"%s_99", .99},   {"%s_99.9", .999}, {"%s_99.99", .9999}, {"%s_max", 1}};
    if (sysStats) { // metrics.go:172-193: Go runtime statistics; nearest process-level analogues
        gauge_funcs_["sys.Alloc"] = [] { return read_proc_status_kb("VmRSS:") * 1024.0; };
        gauge_funcs_["sys.NumGC"] = [] { return 0.0; };
        gauge_funcs_["sys.PauseTotalNs"] = [] { return 0.0; };
        gauge_funcs_["sys.NumGoroutine"] = [] { return read_proc_status_kb("Threads:"); };
    }
}

MetricSystem::~MetricSystem()
{
    Stop();
    if (engine_) {
        lh_destroy(engine_);
        engine_ = nullptr;
    }
}

int MetricSystem::note(int rc, const char *where)
{
    if (rc == LH_OK) return rc;
    const int prev = last_status_.exchange(rc);
    if (prev != rc) // the reference logs through glog and carries on (metrics.go:379-384)
        std::fprintf(stderr, "loghisto: %s: %s [%s]\n", where, lh_strerror(rc), lh_last_error());
    return rc;
}

bool MetricSystem::ensure_engine()
{
    if (engine_) return true;
    std::lock_guard<std::mutex> g(engine_mu_);
    if (engine_) return true;
    lh_config cfg;
    lh_default_config(&cfg);
    cfg.device = opt_.device;
    cfg.max_metrics = opt_.max_metrics;
    cfg.num_buffers = opt_.num_buffers;
    cfg.num_lanes = opt_.num_lanes;
    cfg.lane_samples = opt_.lane_samples;
    cfg.max_counters = opt_.device_counters ? opt_.max_counters : 0;
    lh_engine *e = nullptr;
    const int rc = lh_create(&cfg, &e);
    if (rc != LH_OK) {
        note(rc, "lh_create");
        return false;
    }
    narrow_ids_ = opt_.max_metrics <= 65536u;
    engine_ = e;
    return true;
}

void MetricSystem::SpecifyPercentiles(const std::map<std::string, double> &percentiles)
{
    std::lock_guard<std::mutex> g(percentiles_mu_);
    percentiles_ = percentiles;
}

MetricSystem::Stage *MetricSystem::stage()
{
    // one-entry cache in front of a per-thread map keyed by the system's instance id
    thread_local uint64_t last_id = 0;
    thread_local Stage *last_stage = nullptr;
    thread_local std::unordered_map<uint64_t, Stage *> mine;
    if (last_id == instance_id_) return last_stage;
    auto it = mine.find(instance_id_);
    if (it == mine.end()) {
        auto st = std::make_unique<Stage>();
        st->ids.resize(opt_.stage_samples);
        st->vals.resize(opt_.stage_samples);
        if (opt_.device_counters) {
            st->cids.resize(opt_.stage_samples);
            st->camts.resize(opt_.stage_samples);
        }
        Stage *raw = st.get();
        {
            std::lock_guard<std::mutex> g(stages_mu_);
            stages_.push_back(std::move(st));
        }
        it = mine.emplace(instance_id_, raw).first;
    }
    last_id = instance_id_;
    last_stage = it->second;
    return last_stage;
}

uint32_t MetricSystem::intern(const std::string &name)
{
    uint32_t id = 0;
    const int rc = lh_intern(engine_, name.data(), name.size(), &id);
    if (rc != LH_OK) {
        note(rc, "lh_intern");
        return UINT32_MAX;
    }
    std::lock_guard<std::mutex> g(names_mu_);
    if (names_.size() <= id) names_.resize(id + 1);
    names_[id] = name;
    return id;
}

void MetricSystem::ship(Stage &s)
{
    if (s.n == 0) return;
    // in-place staging: the stage's pairs go straight into the tail of a pinned staging buffer (their ids come from
    // lh_intern, so the host-side validation scan of lh_submit_pairs has nothing to find); the reservation is held
    // for the two memcpys only, never across a call that can block
    size_t done = 0;
    if (narrow_ids_) {
        // at most 65 536 names: the ids are staged as uint16 -- 10 bytes per sample over PCIe instead of 12
        // (lh_reserve_pairs16; the narrowing store is the stage's one copy of the id)
        while (done < s.n) {
            uint16_t *ids = nullptr;
            uint32_t token = 0;
            double *vals = nullptr;
            size_t granted = 0;
            if (note(lh_reserve_pairs16(engine_, s.n - done, &ids, &vals, &granted, &token), "lh_reserve_pairs16") != LH_OK) {
                note(lh_submit_pairs(engine_, s.ids.data() + done, s.vals.data() + done, s.n - done), "lh_submit_pairs");
                break;
            }
            const uint32_t *src = s.ids.data() + done;
            for (size_t i = 0; i < granted; i++) ids[i] = (uint16_t)src[i];
            std::memcpy(vals, s.vals.data() + done, granted * sizeof(double));
            note(lh_commit_pairs16(engine_, token, granted), "lh_commit_pairs16");
            done += granted;
        }
        s.n = 0;
        return;
    }
    while (done < s.n) {
        uint32_t *ids = nullptr, token = 0;
        double *vals = nullptr;
        size_t granted = 0;
        if (note(lh_reserve_pairs(engine_, s.n - done, &ids, &vals, &granted, &token), "lh_reserve_pairs") != LH_OK) {
            note(lh_submit_pairs(engine_, s.ids.data() + done, s.vals.data() + done, s.n - done), "lh_submit_pairs");
            break;
        }
        std::memcpy(ids, s.ids.data() + done, granted * sizeof(uint32_t));
        std::memcpy(vals, s.vals.data() + done, granted * sizeof(double));
        note(lh_commit_pairs(engine_, token, granted), "lh_commit_pairs");
        done += granted;
    }
    s.n = 0;
}

TimerToken MetricSystem::StartTimer(const std::string &name)
{
    return TimerToken{name, std::chrono::steady_clock::now(), this};
}

std::chrono::nanoseconds TimerToken::Stop()
{
    const auto d = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - Start);
    System->Histogram(Name, (double)d.count());
    return d;
}

void MetricSystem::ship_counts(Stage &s)
{
    if (s.cn == 0) return;
    note(lh_submit_counts(engine_, s.cids.data(), s.camts.data(), s.cn), "lh_submit_counts");
    s.cn = 0;
}

void MetricSystem::Counter(const std::string &name, uint64_t amount)
{
    if (opt_.device_counters && (engine_ || ensure_engine())) {
        Stage *st = stage();
        std::lock_guard<std::mutex> g(st->mu);
        uint32_t id;
        auto it = st->cidcache.find(name);
        if (it != st->cidcache.end()) {
            id = it->second;
        } else {
            const int rc = lh_intern_counter(engine_, name.data(), name.size(), &id);
            if (rc != LH_OK) {
                note(rc, "lh_intern_counter");
                return;
            }
            st->cidcache.emplace(name, id);
        }
        st->cids[st->cn] = id;
        st->camts[st->cn] = amount; // the sum happens on the GPU
        if (!counters_used_.load(std::memory_order_relaxed)) counters_used_.store(true, std::memory_order_relaxed);
        if (++st->cn == st->cids.size()) ship_counts(*st);
        return;
    }
    Stage *st = stage();
    std::lock_guard<std::mutex> g(st->mu);
    st->counters[name] += amount;
}

void MetricSystem::Histogram(const std::string &name, double value)
{
    if (!engine_ && !ensure_engine()) return; // no device: reported through last_status(), nothing to compute on
    Stage *st = stage();
    std::lock_guard<std::mutex> g(st->mu);
    uint32_t id;
    auto it = st->idcache.find(name);
    if (it != st->idcache.end()) {
        id = it->second;
    } else {
        id = intern(name);
        if (id == UINT32_MAX) return;
        st->idcache.emplace(name, id);
    }
    st->ids[st->n] = id;
    st->vals[st->n] = value; // compress() happens on the GPU
    // read-mostly: an unconditional store here would bounce one cache line between every producer core
    if (!hist_used_.load(std::memory_order_relaxed)) hist_used_.store(true, std::memory_order_relaxed);
    if (++st->n == st->ids.size()) ship(*st);
}

void MetricSystem::RegisterGaugeFunc(const std::string &name, std::function<double()> f)
{
    std::lock_guard<std::mutex> g(gauge_mu_);
    gauge_funcs_[name] = std::move(f);
}

void MetricSystem::RegisterEngineGauges(const std::string &prefix)
{
    auto read = [this](uint64_t lh_counters::*field) {
        return [this, field]() -> double {
            lh_counters c{};
            if (!engine_ || lh_get_counters(engine_, &c) != LH_OK) return 0.0;
            return (double)(c.*field);
        };
    };
    RegisterGaugeFunc(prefix + "samples_single", read(&lh_counters::samples_single));
    RegisterGaugeFunc(prefix + "samples_small", read(&lh_counters::samples_small));
    RegisterGaugeFunc(prefix + "samples_partitioned", read(&lh_counters::samples_partitioned));
    RegisterGaugeFunc(prefix + "samples_direct", read(&lh_counters::samples_direct));
    RegisterGaugeFunc(prefix + "launches", read(&lh_counters::launches));
    RegisterGaugeFunc(prefix + "flips_busy", read(&lh_counters::flips_busy));
    RegisterGaugeFunc(prefix + "backpressure_waits", read(&lh_counters::backpressure_waits));
    RegisterGaugeFunc(prefix + "window_misses", read(&lh_counters::window_misses));
}

void MetricSystem::DeregisterGaugeFunc(const std::string &name)
{
    std::lock_guard<std::mutex> g(gauge_mu_);
    gauge_funcs_.erase(name);
}

std::shared_ptr<RawMetricSet> MetricSystem::collectRawMetrics()
{
    auto raw = std::make_shared<RawMetricSet>();
    raw->Time = truncate_to(interval_);

    std::unordered_map<std::string, uint64_t> fresh;
    {
        // the epoch boundary: every stage is held while its counters are stolen, its samples shipped and
        // the device buffers flipped (counterMu / histogramMu write locks, metrics.go:425-428, 460-463)
        std::lock_guard<std::mutex> sg(stages_mu_);
        for (auto &st : stages_) st->mu.lock();
        for (auto &st : stages_) {
            for (auto &kv : st->counters) fresh[kv.first] += kv.second;
            st->counters.clear();
        }
        if ((hist_used_.load() || counters_used_.load()) && engine_) {
            for (auto &st : stages_) ship(*st);
            if (counters_used_.load())
                for (auto &st : stages_) ship_counts(*st);
            lh_snapshot *snap = nullptr;
            const int rc = lh_flip(engine_, &snap);
            if (rc == LH_OK) raw->snapshot = snap;
            else note(rc, "lh_flip"); // LH_EBUSY: the epoch keeps accumulating, nothing is lost
        }
        for (auto &st : stages_) st->mu.unlock();
    }
    {
        std::lock_guard<std::mutex> g(names_mu_);
        raw->names = names_;
    }
    if (counters_used_.load() && raw->snapshot) {
        // device counters: the interval's amounts and the lifetime store come back from the snapshot
        // (metrics.go:430-458 ran on the GPU: sum per name, fold into the store at the epoch boundary)
        uint32_t nc = 0;
        note(lh_num_counters(engine_, &nc), "lh_num_counters");
        std::vector<uint64_t> rate(nc), total(nc);
        std::vector<uint8_t> present(nc), known(nc);
        if (nc && note(lh_counters_collect(raw->snapshot, 0, nc, rate.data(), present.data(), total.data(), known.data()),
                       "lh_counters_collect") == LH_OK) {
            std::lock_guard<std::mutex> g(counter_store_mu_);
            std::vector<char> buf(256);
            while (counter_names_.size() < nc) {
                size_t len = 0;
                const uint32_t id = (uint32_t)counter_names_.size();
                if (lh_counter_name(engine_, id, buf.data(), buf.size(), &len) != LH_OK) break;
                if (len > buf.size()) { // the whole name, whatever its length: a truncated name is a different key
                    buf.resize(len);
                    if (lh_counter_name(engine_, id, buf.data(), buf.size(), &len) != LH_OK) break;
                }
                counter_names_.emplace_back(buf.data(), len);
            }
            // amounts staged on the host (Counter() calls made before the engine existed) are part of this interval's
            // rates AND of the lifetime totals (metrics.go:435-458); the device only knows what was shipped to it
            for (auto &kv : fresh) host_counter_store_[kv.first] += kv.second;
            for (uint32_t i = 0; i < nc && i < counter_names_.size(); i++) {
                if (present[i]) fresh[counter_names_[i]] += rate[i];
                if (known[i]) device_counter_total_[counter_names_[i]] = total[i];
            }
            counter_store_ = device_counter_total_;
            for (auto &kv : host_counter_store_) counter_store_[kv.first] += kv.second;
        } else {
            std::lock_guard<std::mutex> g(counter_store_mu_);
            for (auto &kv : fresh) {
                host_counter_store_[kv.first] += kv.second;
                counter_store_[kv.first] += kv.second;
            }
        }
        raw->Rates = fresh;
        std::lock_guard<std::mutex> g(counter_store_mu_);
        raw->Counters = counter_store_;
    } else {
        raw->Rates = fresh; // metrics.go:430-433
        std::lock_guard<std::mutex> g(counter_store_mu_); // metrics.go:435-458
        for (auto &kv : fresh) {
            host_counter_store_[kv.first] += kv.second;
            counter_store_[kv.first] += kv.second;
        }
        raw->Counters = counter_store_;
    }
    {
        std::lock_guard<std::mutex> g(gauge_mu_); // metrics.go:465-470
        for (auto &kv : gauge_funcs_) raw->Gauges[kv.first] = kv.second();
    }
    return raw;
}

std::shared_ptr<ProcessedMetricSet> MetricSystem::processMetrics(const std::shared_ptr<RawMetricSet> &raw)
{
    auto out = std::make_shared<ProcessedMetricSet>();
    out->Time = raw->Time;
    auto &m = out->Metrics;
    for (auto &kv : raw->Counters) m[kv.first] = (double)kv.second;       // metrics.go:487-489
    for (auto &kv : raw->Rates) m[kv.first + "_rate"] = (double)kv.second; // metrics.go:491-493

    // processHistograms for every name in one extract (metrics.go:336-387, 495-499)
    size_t n_map = 0;
    if (raw->snapshot && !raw->names.empty()) {
        std::vector<std::string> labels;
        std::vector<double> ps;
        {
            std::lock_guard<std::mutex> g(percentiles_mu_);
            for (auto &kv : percentiles_) { labels.push_back(kv.first); ps.push_back(kv.second); }
        }
        const size_t n = raw->names.size();
        const WireFormat wf = (WireFormat)wire_format_.load();
        if (wf != WireFormat::None) {
            out->wire_format = wf;
            serializeHistograms(*raw, labels, ps, wf, out->wire);
        }
        if (wf != WireFormat::None && !wire_keep_map_.load()) n_map = 0;
        else n_map = n;
    }
    if (n_map) {
        {   // processHistograms' lifetime side effect (metrics.go:359-376: uint64(totalSum), wrapping adds): applied
            // once per snapshot to the ONE store there is -- in HBM -- whether the map path, the wire path or both run
            std::lock_guard<std::mutex> g(raw->mu);
            if (raw->snapshot) note(lh_snapshot_accumulate(raw->snapshot), "lh_snapshot_accumulate");
        }
        std::vector<std::string> labels;
        std::vector<double> ps;
        {
            std::lock_guard<std::mutex> g(percentiles_mu_);
            for (auto &kv : percentiles_) { labels.push_back(kv.first); ps.push_back(kv.second); }
        }
        // The COMPACT results (lh_extract_rows_compact, round 6): count, sum and the selected keys are all that crosses
        // PCIe -- 42 B per name instead of 139 -- and the keys of the map are derived here exactly as processHistograms
        // derives them: _avg = sum / float64(count) (metrics.go:356), a percentile's value = decompress(key)
        // (metrics.go:326-332, 413-415: the device-generated table, read back once per MetricSystem).
        const size_t n = n_map, np = ps.size();
        std::vector<uint64_t> cnt(n);
        std::vector<double> sum(n);
        std::vector<int16_t> keys(n * np);
        std::vector<uint32_t> bits(n);
        int rc;
        {
            std::lock_guard<std::mutex> g(raw->mu);
            lh_extract_compact c{};
            rc = raw->snapshot ? lh_extract_rows_compact(raw->snapshot, 0, n, ps.data(), np, &c) : LH_ESTATE;
            if (rc == LH_OK || rc == LH_ERANGE) { // the view lives in the engine's pinned block: copy out under the lock
                std::memcpy(cnt.data(), c.count, n * sizeof(uint64_t));
                std::memcpy(sum.data(), c.sum, n * sizeof(double));
                if (np) {
                    std::memcpy(keys.data(), c.pkeys, n * np * sizeof(int16_t));
                    std::memcpy(bits.data(), c.pvalid_bits, n * sizeof(uint32_t));
                }
            }
        }
        note(rc, "lh_extract_rows_compact");
        if (rc == LH_OK || rc == LH_ERANGE) {
            const double *D = decompressTable();
            for (size_t id = 0; id < n; id++) {
                if (!cnt[id]) continue;
                const std::string &name = raw->names[id];
                m[name + "_count"] = (double)cnt[id];
                m[name + "_sum"] = sum[id];
                m[name + "_avg"] = sum[id] / (double)cnt[id];
                for (size_t i = 0; i < np; i++) {
                    const bool ok = (bits[id] >> i) & 1u;
                    if (ok && D) m[fmt_label(labels[i], name)] = D[(uint16_t)keys[id * np + i] ^ 0x8000u];
                    else if (!ok) std::fprintf(stderr, "loghisto: unable to calculate percentile: Invalid percentile.  "
                                              "Should be between 0 and 1.\n"); // metrics.go:379-384
                }
            }
        }
    }
    for (auto &kv : raw->Gauges) m[kv.first] = kv.second; // metrics.go:501-503
    if (out->wire_format != WireFormat::None) {
        // the keys that never were on the device: counters, rates, gauges (a few hundred at most)
        ProcessedMetricSet rest;
        rest.Time = out->Time;
        for (auto &kv : raw->Counters) rest.Metrics[kv.first] = (double)kv.second;
        for (auto &kv : raw->Rates) rest.Metrics[kv.first + "_rate"] = (double)kv.second;
        for (auto &kv : raw->Gauges) rest.Metrics[kv.first] = kv.second;
        out->wire += out->wire_format == WireFormat::Graphite ? GraphiteProtocol(rest) : OpenTSDBProtocol(rest);
    }
    return out;
}

// decompress (metrics.go:326-332) of every int16 key as the DEVICE generated it (lh_codec.h: d_go_exp), indexed by the
// dense bin key ^ 0x8000: what a compact extract's keys are turned into values with.  nullptr if it cannot be read.
const double *MetricSystem::decompressTable()
{
    std::call_once(dtable_once_, [this] {
        if (!engine_) return;
        dtable_.resize(LH_NKEYS);
        if (note(lh_codec_tables(engine_, nullptr, dtable_.data()), "lh_codec_tables") != LH_OK) dtable_.clear();
    });
    return dtable_.empty() ? nullptr : dtable_.data();
}

void MetricSystem::SetWireFormat(WireFormat f, bool histogram_keys_in_map)
{
    wire_keep_map_.store(histogram_keys_in_map);
    wire_format_.store((int)f);
}

// The histogram keys of the interval as wire text, formatted on the device (K6): processHistograms' lifetime
// side effect first (metrics.go:359-376), then every key incl. _agg_* in one lh_serialize.
void MetricSystem::serializeHistograms(RawMetricSet &raw, const std::vector<std::string> &labels,
                                       const std::vector<double> &ps, WireFormat wf, std::string &text)
{
    const std::string host = hostname();
    const long long t =
        (long long)std::chrono::duration_cast<std::chrono::seconds>(raw.Time.time_since_epoch()).count();
    const std::string ts = std::to_string(t);
    std::string prefix, sep, suffix;
    lh_line_format fmt{};
    if (wf == WireFormat::Graphite) { // graphite.go:40
        prefix = "cockroach." + host + ".";
        sep = " ";
        suffix = " " + ts + "\n";
        fmt.flags = LH_FMT_UNDERSCORE_TO_DOT;
    } else {                          // opentsdb.go:48
        prefix = "put ";
        sep = " " + ts + " ";
        suffix = " host=" + host + "\n";
    }
    fmt.prefix = prefix.c_str();
    fmt.sep = sep.c_str();
    fmt.suffix = suffix.c_str();
    std::vector<const char *> lab;
    for (auto &l : labels) lab.push_back(l.c_str());

    std::lock_guard<std::mutex> g(raw.mu);
    if (!raw.snapshot) return;
    note(lh_snapshot_accumulate(raw.snapshot), "lh_snapshot_accumulate");
    const size_t n = raw.names.size();
    size_t need = 0;
    text.resize(std::max<size_t>(wire_bytes_hint_.load(), 4096));
    for (int attempt = 0; attempt < 2; attempt++) {
        const int rc = lh_serialize(raw.snapshot, 0, n, ps.data(), lab.data(), ps.size(), &fmt, LH_SER_AGGREGATES,
                                    &text[0], text.size(), &need);
        if (rc != LH_OK && rc != LH_ERANGE) {
            note(rc, "lh_serialize");
            text.clear();
            return;
        }
        if (need <= text.size()) break;
        text.resize(need + need / 8);
    }
    text.resize(need);
    wire_bytes_hint_.store(need + need / 8);
}

void MetricSystem::addAggregates(const std::shared_ptr<RawMetricSet> &raw, ProcessedMetricSet &processed)
{
    // metrics.go:590-608.  histogramCountStore lives in HBM (lh_snapshot_accumulate / lh_lifetime): the map keys
    // and the wire text's _agg_* lines read the same store.
    const size_t n = raw->names.size();
    if (!n || !engine_) return;
    std::vector<uint64_t> agg_count(n), agg_sum(n);
    if (note(lh_lifetime(engine_, 0, n, agg_count.data(), agg_sum.data()), "lh_lifetime") != LH_OK) return;
    for (size_t id = 0; id < n; id++) {
        const std::string &name = raw->names[id];
        if (!processed.Metrics.count(name + "_count") || agg_count[id] == 0) continue;
        processed.Metrics[name + "_agg_avg"] = (double)(agg_sum[id] / agg_count[id]); // integer division
        processed.Metrics[name + "_agg_count"] = (double)agg_count[id];
        processed.Metrics[name + "_agg_sum"] = (double)agg_sum[id];
    }
}

// ---------------------------------------------------------------------------
// subscriptions + reaper (glue; metrics.go:203-228, 508-639)
// ---------------------------------------------------------------------------
void MetricSystem::SubscribeToRawMetrics(RawCh ch)
{
    std::lock_guard<std::mutex> g(pending_mu_);
    pending_.push_back({0, std::move(ch), nullptr});
}
void MetricSystem::UnsubscribeFromRawMetrics(RawCh ch)
{
    std::lock_guard<std::mutex> g(pending_mu_);
    pending_.push_back({1, std::move(ch), nullptr});
}
void MetricSystem::SubscribeToProcessedMetrics(ProcCh ch)
{
    std::lock_guard<std::mutex> g(pending_mu_);
    pending_.push_back({2, nullptr, std::move(ch)});
}
void MetricSystem::UnsubscribeFromProcessedMetrics(ProcCh ch)
{
    std::lock_guard<std::mutex> g(pending_mu_);
    pending_.push_back({3, nullptr, std::move(ch)});
}

void MetricSystem::updateSubscribers()
{
    std::vector<SubOp> ops;
    {
        std::lock_guard<std::mutex> g(pending_mu_);
        ops.swap(pending_);
    }
    std::lock_guard<std::mutex> g(subs_mu_);
    for (auto &op : ops) {
        switch (op.kind) {
        case 0: raw_subs_.push_back(op.raw); break;
        case 1: raw_subs_.erase(std::remove(raw_subs_.begin(), raw_subs_.end(), op.raw), raw_subs_.end()); break;
        case 2: proc_subs_.push_back(op.proc); break;
        case 3: proc_subs_.erase(std::remove(proc_subs_.begin(), proc_subs_.end(), op.proc), proc_subs_.end()); break;
        }
    }
}

namespace {
template <class Ch, class Item>
void broadcast(std::vector<Ch> &subs, std::unordered_map<void *, int> &bad, const Item &item)
{
    // non-blocking send; a subscriber that is full on 2 consecutive intervals is forgotten and its
    // channel closed (metrics.go:567-580, 613-626)
    for (size_t i = 0; i < subs.size();) {
        void *key = subs[i].get();
        if (subs[i]->TrySend(item)) {
            bad.erase(key);
            i++;
        } else if (++bad[key] >= 2) {
            subs[i]->Close();
            bad.erase(key);
            subs.erase(subs.begin() + (long)i);
        } else {
            i++;
        }
    }
}
} // namespace

void MetricSystem::reaper()
{
    // processing pool: a 16-deep hand-off, the interval is dropped when it is full (metrics.go:533-545, 630-637)
    // heap-allocated so that ThreadSanitizer sees a fresh mutex per reaper run (std::mutex has a trivial
    // destructor; a stack slot reused by the next run looks like a double lock to it)
    auto work_p = std::make_shared<Channel<std::shared_ptr<RawMetricSet>>>(16);
    auto &work = *work_p;
    std::vector<std::thread> pool;
    const unsigned nworkers = std::max(4u, std::thread::hardware_concurrency() / 4);
    for (unsigned w = 0; w < std::min(nworkers, 8u); w++) {
        pool.emplace_back([&] {
            std::shared_ptr<RawMetricSet> raw;
            while (!work.Closed() || work.Len()) {
                if (!work.Receive(raw, std::chrono::milliseconds(50))) continue;
                auto processed = processMetrics(raw);
                addAggregates(raw, *processed);
                raw->Release();
                std::lock_guard<std::mutex> g(subs_mu_);
                broadcast(proc_subs_, proc_bad_, processed);
            }
        });
    }
    const int64_t ivl = std::max<int64_t>(1, interval_.count());
    while (true) {
        using namespace std::chrono;
        const int64_t now = duration_cast<nanoseconds>(system_clock::now().time_since_epoch()).count();
        const nanoseconds tts(ivl - now % ivl);
        {
            std::unique_lock<std::mutex> g(shutdown_mu_);
            if (shutdown_cv_.wait_until(g, system_clock::now() + tts, [&] { return shutdown_.load(); })) break;
        }
        auto raw = collectRawMetrics();
        updateSubscribers();
        {
            std::lock_guard<std::mutex> g(subs_mu_);
            if (!raw_subs_.empty()) {
                raw->Histograms(); // materialise while the snapshot is alive
                broadcast(raw_subs_, raw_bad_, raw);
            }
        }
        if (!work.TrySend(raw)) {
            dropped_intervals_.fetch_add(1);
            std::fprintf(stderr, "loghisto: processing of metrics is taking longer than this node can handle; "
                                 "dropping this entire interval\n");
            raw->Release();
        }
    }
    work.Close();
    for (auto &t : pool) t.join();
    reaping_.store(false);
}

void MetricSystem::Start()
{
    bool expected = false;
    if (!reaping_.compare_exchange_strong(expected, true)) return;
    shutdown_.store(false);
    reaper_thread_ = std::thread([this] { reaper(); });
}

void MetricSystem::Stop()
{
    {
        std::lock_guard<std::mutex> g(shutdown_mu_);
        shutdown_.store(true);
    }
    shutdown_cv_.notify_all();
    if (reaper_thread_.joinable()) reaper_thread_.join();
}

// ---------------------------------------------------------------------------
// Submitter (submitter.go:27-159)
// ---------------------------------------------------------------------------
Submitter::Submitter(MetricSystem *ms, Serializer serializer, std::string network, std::string address,
                     std::chrono::nanoseconds interval)
    : DestinationNetwork(std::move(network)), DestinationAddress(std::move(address)), ms_(ms),
      serializer_(std::move(serializer)), interval_(interval),
      chan_(std::make_shared<Channel<std::shared_ptr<ProcessedMetricSet>>>(60)) // submitter.go:55
{
    ms_->SubscribeToProcessedMetrics(chan_);
}

Submitter::~Submitter() { Shutdown(); }

// submitter.go:106-116 dials, writes one request and closes, once per backlog entry per interval.  Here the
// connection is kept (re-dialled only after an error) and everything the backlog holds leaves in ONE writev per
// interval: same bytes in the same order, same 5 s deadline, same "retry next interval" on failure.  UDP keeps one
// datagram per request.
bool Submitter::connectIfNeeded()
{
    if (fd_ >= 0) return true;
    const size_t colon = DestinationAddress.rfind(':');
    if (colon == std::string::npos) return false;
    const std::string host = DestinationAddress.substr(0, colon), port = DestinationAddress.substr(colon + 1);
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_UNSPEC;
    hints.ai_socktype = DestinationNetwork == "udp" ? SOCK_DGRAM : SOCK_STREAM;
    if (getaddrinfo(host.c_str(), port.c_str(), &hints, &res) != 0 || !res) return false;
    const int fd = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
    if (fd >= 0) {
        timeval tv{5, 0}; // 5 s deadline
        setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            fd_ = fd;
            connects_.fetch_add(1);
        } else {
            close(fd);
        }
    }
    freeaddrinfo(res);
    return fd_ >= 0;
}

void Submitter::disconnect()
{
    if (fd_ >= 0) close(fd_);
    fd_ = -1;
}

// Sends requests[0 .. n) in order; *done receives how many were written completely (they must not be sent again).
bool Submitter::submitBatch(const std::shared_ptr<const std::string> *requests, size_t n, size_t *done)
{
    *done = 0;
    if (fd_ >= 0 && DestinationNetwork != "udp") {
        // A kept connection whose peer has closed or restarted still accepts one write (the reference dials per
        // request, submitter.go:106-116, and never meets this): look for the EOF before trusting it with a batch.
        char c;
        const ssize_t r = ::recv(fd_, &c, 1, MSG_PEEK | MSG_DONTWAIT);
        if (r == 0 || (r < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)) disconnect();
    }
    if (!connectIfNeeded()) return false;
    if (DestinationNetwork == "udp") {
        for (size_t i = 0; i < n; i++) {
            if (::send(fd_, requests[i]->data(), requests[i]->size(), MSG_NOSIGNAL) != (ssize_t)requests[i]->size()) {
                disconnect();
                return false;
            }
            *done = i + 1;
        }
        return true;
    }
    iovec iov[60];
    size_t req_of[60];
    size_t cnt = 0, lead = 0; // lead: empty requests before the first non-empty one count as written
    for (size_t i = 0; i < n && cnt < 60; i++) {
        if (!requests[i]->empty()) {
            req_of[cnt] = i;
            iov[cnt++] = iovec{const_cast<char *>(requests[i]->data()), requests[i]->size()};
        } else if (cnt == 0) {
            lead = i + 1;
        }
    }
    *done = cnt ? lead : n;
    size_t first = 0;
    while (first < cnt) {
        msghdr mh{};
        mh.msg_iov = iov + first;
        mh.msg_iovlen = cnt - first;
        ssize_t w = ::sendmsg(fd_, &mh, MSG_NOSIGNAL); // writev with MSG_NOSIGNAL
        if (w <= 0) {
            disconnect(); // the peer went away or the deadline passed: re-dial at the next interval
            return false; // (a request that was written in part is sent again whole: *done stops before it)
        }
        while (w > 0 && first < cnt) { // partial write: drop what has left
            if ((size_t)w >= iov[first].iov_len) {
                w -= (ssize_t)iov[first].iov_len;
                first++;
                // everything up to the next non-empty request is done
                *done = first < cnt ? req_of[first] : n;
            } else {
                iov[first].iov_base = static_cast<char *>(iov[first].iov_base) + w;
                iov[first].iov_len -= (size_t)w;
                w = 0;
            }
        }
    }
    return true;
}

bool Submitter::retryBacklog() // submitter.go:70-93
{
    std::shared_ptr<const std::string> batch[60];
    size_t n = 0;
    uint64_t seq0;
    {
        std::lock_guard<std::mutex> g(backlog_mu_);
        seq0 = head_seq_;
        for (int i = head_; i != tail_; i = (i + 1) % 60) batch[n++] = backlog_[i];
    }
    if (n == 0) return true;
    size_t done = 0;
    const bool ok = submitBatch(batch, n, &done);
    sent_.fetch_add(done);
    std::lock_guard<std::mutex> g(backlog_mu_);
    // only what was written completely leaves the backlog (a failure after a partial write must not duplicate the
    // requests already delivered); entries evicted while the batch was on the wire already moved the head
    for (uint64_t end = seq0 + done; head_seq_ < end; head_seq_++) {
        backlog_[head_].reset();
        head_ = (head_ + 1) % 60;
    }
    return ok;
}

void Submitter::appendToBacklog(std::string request) // submitter.go:95-104
{
    auto entry = std::make_shared<const std::string>(std::move(request));
    std::lock_guard<std::mutex> g(backlog_mu_);
    backlog_[tail_] = std::move(entry);
    tail_ = (tail_ + 1) % 60;
    if (head_ == tail_) { // ran into the head: evict it
        backlog_[head_].reset();
        head_ = (head_ + 1) % 60;
        head_seq_++;
        evicted_.fetch_add(1);
    }
}

void Submitter::Start() // submitter.go:119-149
{
    recv_thread_ = std::thread([this] {
        std::shared_ptr<ProcessedMetricSet> pm;
        while (!shutdown_.load()) {
            if (chan_->Receive(pm, std::chrono::milliseconds(20))) appendToBacklog(serializer_(*pm));
            else if (chan_->Closed()) return; // we can no longer make progress
        }
    });
    send_thread_ = std::thread([this] {
        const int64_t ivl = std::max<int64_t>(1, interval_.count());
        while (!shutdown_.load()) {
            retryBacklog();
            using namespace std::chrono;
            const int64_t now = duration_cast<nanoseconds>(system_clock::now().time_since_epoch()).count();
            int64_t tts = ivl - now % ivl;
            while (tts > 0 && !shutdown_.load()) { // interruptible sleep to the next interval boundary
                const int64_t step = std::min<int64_t>(tts, 20000000);
                std::this_thread::sleep_for(nanoseconds(step));
                tts -= step;
            }
        }
    });
}

void Submitter::Shutdown() // submitter.go:152-159
{
    if (shutdown_.exchange(true)) return;
    if (recv_thread_.joinable()) recv_thread_.join();
    if (send_thread_.joinable()) send_thread_.join();
    disconnect();
}

// ---------------------------------------------------------------------------
// serializers (graphite.go:37-75, opentsdb.go:45-85)
// ---------------------------------------------------------------------------
std::string GraphiteProtocol(const ProcessedMetricSet &ms)
{
    if (ms.wire_format == WireFormat::Graphite) return ms.wire; // prepared in bulk by processMetrics
    const std::string host = hostname();
    const long long t = (long long)std::chrono::duration_cast<std::chrono::seconds>(ms.Time.time_since_epoch()).count();
    std::string out;
    out.reserve(ms.Metrics.size() * 64);
    char num[400];
    for (auto &kv : ms.Metrics) {
        std::string metric = kv.first;
        std::replace(metric.begin(), metric.end(), '_', '.');
        out += "cockroach.";
        out += host;
        out += '.';
        out += metric;
        std::snprintf(num, sizeof(num), " %f %lld\n", kv.second, t);
        out += num;
    }
    return out;
}

std::string OpenTSDBProtocol(const ProcessedMetricSet &ms)
{
    if (ms.wire_format == WireFormat::OpenTSDB) return ms.wire;
    const std::string host = hostname();
    const long long t = (long long)std::chrono::duration_cast<std::chrono::seconds>(ms.Time.time_since_epoch()).count();
    std::string out;
    out.reserve(ms.Metrics.size() * 64);
    char num[400];
    for (auto &kv : ms.Metrics) {
        out += "put ";
        out += kv.first;
        std::snprintf(num, sizeof(num), " %lld %f host=", t, kv.second);
        out += num;
        out += host;
        out += '\n';
    }
    return out;
}

// ---------------------------------------------------------------------------
// PrintBenchmark (print_benchmark.go:49-106)
// ---------------------------------------------------------------------------
std::string FormatGoV(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-Inf" : "+Inf";
    // shortest digits that round-trip, as strconv.FormatFloat(v, 'g', -1, 64) picks them
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr); // [-]d[.ddd]e[+-]XX
    std::string out;
    size_t i = 0;
    if (sci[0] == '-') { out = "-"; i = 1; }
    const size_t epos = sci.find('e');
    std::string digits;
    for (size_t k = i; k < epos; k++)
        if (sci[k] != '.') digits += sci[k];
    const int exp = std::atoi(sci.c_str() + epos + 1);
    if (digits == "0") return out + "0";
    if (exp < -4 || exp >= 6) { // strconv %g: %e when exp < -4 || exp >= eprec, eprec = 6 for the shortest form
        out += digits[0];
        if (digits.size() > 1) { out += '.'; out += digits.substr(1); }
        char e[16]; // |exp| <= 324
        std::snprintf(e, sizeof e, "e%c%02d", exp < 0 ? '-' : '+', exp < 0 ? -exp : exp);
        return out + e;
    }
    if (exp < 0) {
        out += "0.";
        out.append((size_t)(-exp - 1), '0');
        return out + digits;
    }
    if ((size_t)exp + 1 >= digits.size()) {
        out += digits;
        out.append((size_t)exp + 1 - digits.size(), '0');
        return out;
    }
    return out + digits.substr(0, (size_t)exp + 1) + "." + digits.substr((size_t)exp + 1);
}

namespace {
void print_interval(std::FILE *out, const std::string &name, const ProcessedMetricSet &m)
{
    static const char *suffixes[] = {"_count", "_max", "_99.99", "_99.9", "_99", "_95", "_90", "_75", "_50", "_min",
                                     "_sum", "_avg", "_agg_avg", "_agg_count", "_agg_sum"}; // print_benchmark.go:76-97
    std::vector<std::string> keys;
    for (const char *sfx : suffixes) keys.push_back(name + sfx);
    for (const char *k : {"sys.Alloc", "sys.NumGC", "sys.PauseTotalNs", "sys.NumGoroutine"}) keys.push_back(k);
    // time.Time's default rendering: 2014-08-09 17:44:57 -0400 EDT
    const std::time_t t = std::chrono::system_clock::to_time_t(m.Time);
    std::tm tm{};
    localtime_r(&t, &tm);
    char ts[64];
    std::strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S %z %Z", &tm);
    std::fprintf(out, "%s\n", ts);
    // text/tabwriter with Init(out, 0, 8, 0, '\t', 0): the "<key>:" cells form one column, padded with tabs
    // to the smallest multiple of 8 that holds the widest cell
    size_t width = 0;
    for (auto &k : keys) width = std::max(width, k.size() + 1);
    width = (width + 7) / 8 * 8;
    for (auto &k : keys) {
        auto it = m.Metrics.find(k);
        const double v = it == m.Metrics.end() ? 0.0 : it->second; // a missing map key reads as 0 in Go
        const size_t textw = k.size() + 1;
        std::string line = k + ":";
        line.append((width - textw + 7) / 8, '\t');
        std::fprintf(out, "%s %s\n", line.c_str(), FormatGoV(v).c_str());
    }
    std::fprintf(out, "\n");
    std::fflush(out);
}
} // namespace

void PrintBenchmark(const std::string &name, unsigned concurrency, std::function<void()> op,
                    std::chrono::nanoseconds run_for, std::FILE *out, const Options &opt)
{
    MetricSystem ms(std::chrono::seconds(1), true, opt);
    auto mc = std::make_shared<Channel<std::shared_ptr<ProcessedMetricSet>>>(1);
    ms.SubscribeToProcessedMetrics(mc);
    ms.Start();
    std::atomic<bool> stop{false};
    std::thread receiver([&] {
        std::shared_ptr<ProcessedMetricSet> m;
        while (!stop.load() || mc->Len())
            if (mc->Receive(m, std::chrono::milliseconds(50))) print_interval(out, name, *m);
    });
    std::vector<std::thread> workers;
    for (unsigned i = 0; i < concurrency; i++)
        workers.emplace_back([&] {
            while (!stop.load(std::memory_order_relaxed)) {
                TimerToken timer = ms.StartTimer(name);
                op();
                timer.Stop();
            }
        });
    if (run_for.count() > 0) std::this_thread::sleep_for(run_for);
    else
        for (;;) std::this_thread::sleep_for(std::chrono::hours(1)); // <-make(chan struct{})
    stop.store(true);
    for (auto &w : workers) w.join();
    ms.Stop();
    receiver.join();
}

} // namespace loghisto
