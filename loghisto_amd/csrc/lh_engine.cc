// lh_engine.cc -- host runtime and C ABI of liblhgpu.so (include/loghisto_gpu.h).
//
// What lives here, and what it replaces in the reference:
//   * name -> id intern table        : outer map of histogramCache (metrics.go:119)
//   * staging lanes (pinned double buffers + one HIP stream each)
//                                    : the per-call path of Histogram (metrics.go:273-295),
//                                      batched because a cgo crossing costs more than the
//                                      whole Go fast path (SURVEY.md 8b "cost model")
//   * epoch buffers + flip           : the map swap in collectRawMetrics (metrics.go:460-463)
//   * extract                        : processHistograms / percentile (metrics.go:336-418)
//
// All bucket arithmetic runs in the kernels of lh_kernels.hip.  There is no CPU
// compute path: without a gfx950 device lh_create fails with LH_ENODEVICE.
#include "../../include/loghisto_gpu_tuning.h"
#include "lh_dispatch.h"
#include "lh_kernels.h"

#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local char g_last_error[512] = "";

void set_last_error(const char *what, hipError_t e)
{
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    // the runtime keeps the failure as this thread's "last error"; the launch wrappers report hipGetLastError(),
    // so leaving it in place would fail the next, unrelated launch (e.g. an lh_create after an out-of-memory one)
    (void)hipGetLastError();
}

#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            set_last_error(#expr, _e);                                                                      \
            return _e == hipErrorOutOfMemory ? LH_ENOMEM : LH_EDEVICE;                                      \
        }                                                                                                   \
    } while (0)

enum BufState { BUF_FREE = 0, BUF_CURRENT = 1, BUF_SNAPSHOT = 2 };

struct EpochBuffer {
    // [max_metrics][LH_ROW_STRIDE] cells.  `counts` is what the kernels take: the store in use, its width in bit 0
    // (lh_cells.h).  A wide engine has store64 only.  A narrow engine (above 8 192 names, or lh_config.cell_bits = 32)
    // counts in store32 while the interval holds fewer than 2^32 samples -- no cell can wrap -- and moves to store64
    // (allocated the first time it is needed, kept) before an enqueue could pass that: widen_buffer.
    uint64_t *counts = nullptr;
    uint32_t *store32 = nullptr;
    uint64_t *store64 = nullptr;
    uint64_t reserved = 0;       // narrow store: samples of the launches enqueued or about to be (atomic builtins)
    // A buffer whose intervals keep passing 2^32 samples stays on its wide store and gives the narrow one back (lh_release:
    // after kStayWide consecutive intervals widened by their ingest), so that a high-rate engine holds what a 64-bit engine
    // holds, not both stores; kBackToNarrow consecutive intervals below 2^31 samples bring the narrow store back.
    bool ingest_widened = false; // this interval's ingest moved the buffer to its wide store (widen_buffer, quiesce)
    uint32_t wide_streak = 0, quiet_streak = 0;
    uint32_t *ranges = nullptr;  // [max_metrics][2]
    uint64_t *ccur = nullptr;    // [max_counters] the interval's counter amounts (metrics.go:425-433)
    uint32_t *cflag = nullptr;   // [max_counters] touched this interval
    hipEvent_t cleared = nullptr; // recorded on xstream after the last clear
    BufState state = BUF_FREE;
    uint64_t nsamples = 0;        // samples enqueued into this buffer since its last clear (atomic builtins); ~0 = unknown
};

// LANE_PAIRS16: (uint16 id, value) pairs -- the id half-buffer holds 2-byte ids (10 B per pair over PCIe instead of 12)
enum LaneMode { LANE_NONE = 0, LANE_SINGLE = 1, LANE_PAIRS = 2, LANE_COUNTS = 3, LANE_PAIRS16 = 4 };

struct Lane {
    std::mutex mu;
    hipStream_t stream = nullptr;
    double *h_vals[2] = {nullptr, nullptr};
    uint32_t *h_ids[2] = {nullptr, nullptr};
    double *d_vals[2] = {nullptr, nullptr};
    uint32_t *d_ids[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    bool inflight[2] = {false, false};
    int cur = 0;
    size_t fill = 0;
    LaneMode mode = LANE_NONE;
    uint32_t single_id = 0;
    // lh_reserve_pairs .. lh_commit_pairs: the tail of the current half-buffer belongs to one producer, which writes
    // its samples in place; everybody else who wants the lane (another producer, a flush at the flip) waits on cv
    bool reserved = false;
    size_t granted = 0;
    std::condition_variable cv;
    // the half-buffers as the device sees them (hipHostMalloc memory is mapped): kernels read them over PCIe
    double *m_vals[2] = {nullptr, nullptr};
    uint32_t *m_ids[2] = {nullptr, nullptr};
};

// Lane mutex taken, and the lane free of a reservation (lh_reserve_pairs .. lh_commit_pairs)
struct LaneLock {
    std::unique_lock<std::mutex> g;
    explicit LaneLock(Lane &ln) : g(ln.mu) { ln.cv.wait(g, [&] { return !ln.reserved; }); }
};

// One D2H block per extract call.
struct ExtractLayout {
    size_t off_stats, off_pvals, off_pkeys, off_pvalid, off_err, total;
    size_t off_count = 0, off_sum = 0, off_nb = 0, off_vbits = 0; // compact form (off_pkeys is shared)
};

// lh_extract_rows_compact: count | sum | pkeys | nbuckets | valid bits | err
ExtractLayout extract_layout_compact(size_t nmetrics, size_t np)
{
    ExtractLayout L{};
    size_t o = 0;
    L.off_count = o; o += nmetrics * sizeof(uint64_t);
    L.off_sum = o; o += nmetrics * sizeof(double);
    L.off_pkeys = o; o += (nmetrics * np * sizeof(int16_t) + 7) & ~size_t(7);
    L.off_nb = o; o += (nmetrics * sizeof(uint32_t) + 7) & ~size_t(7);
    L.off_vbits = o; o += (nmetrics * sizeof(uint32_t) + 7) & ~size_t(7);
    L.off_err = o; o += 8;
    L.total = o;
    return L;
}

ExtractLayout extract_layout(size_t nmetrics, size_t np)
{
    ExtractLayout L;
    size_t o = 0;
    L.off_stats = o; o += nmetrics * sizeof(lh::ExtractOut);
    L.off_pvals = o; o += nmetrics * np * sizeof(double);
    L.off_pkeys = o; o += (nmetrics * np * sizeof(int16_t) + 7) & ~size_t(7);
    L.off_pvalid = o; o += (nmetrics * np + 7) & ~size_t(7);
    L.off_err = o; o += 8;
    L.total = o;
    return L;
}

} // namespace

constexpr uint32_t kLaneRstat = 8; // first word of the host-fed lanes' report block in h_rstat (lh_create)

struct lh_engine {
    lh_config cfg{};
    int device = 0;
    int num_cus = 256;

    double *d_Tx = nullptr;
    double *d_D = nullptr;
    uint32_t *d_err = nullptr;

    std::vector<EpochBuffer> bufs;
    int cur = 0;
    std::shared_mutex epoch_mu; // submitters shared, flip unique (histogramMu, metrics.go:121)
    // Narrow engines: an ingest step holds it shared from reading the current buffer's `counts` until its launches are
    // enqueued; widen_buffer holds it unique.  Lock order: epoch, lane, cells.
    std::shared_mutex cells_mu;
    bool narrow = false;                    // the epoch buffers start every interval on uint32 cells
    uint64_t widen_at = 0xffffffffull;      // LH_OPT_WIDEN_AT_SAMPLES (tests): a narrow buffer holds at most this many samples
    std::atomic<uint64_t> c_widenings{0}, c_store_bytes{0}; // (store bytes: kept where a store is allocated or freed)

    std::mutex streams_mu;
    std::vector<hipStream_t> epoch_streams; // streams that touched the current epoch buffer
    std::vector<hipEvent_t> flip_events;

    hipStream_t main_stream = nullptr; // device submits with stream == NULL
    hipStream_t xstream = nullptr;     // extract / clear


    std::vector<std::unique_ptr<Lane>> lanes;

    std::shared_mutex names_mu;
    std::unordered_map<std::string, uint32_t> name2id;
    std::vector<std::string> names;

    std::mutex xmu; // extract scratch
    unsigned char *d_xbuf = nullptr;
    unsigned char *h_xbuf = nullptr;
    unsigned char *d_hxbuf = nullptr; // device-side address of the pinned h_xbuf (zero-copy results), or null
    uint32_t xseq = 0;                // completion-flag sequence of the zero-copy extract (xmu)
    size_t xbuf_bytes = 0;

    // K6 (lh_serialize): names, lifetime stores and the text buffer in HBM; all guarded by xmu
    std::string name_blob;              // names back to back (names_mu)
    std::vector<uint32_t> name_off{0};  // [names.size() + 1] (names_mu)
    char *d_names = nullptr;
    uint32_t *d_name_off = nullptr;
    size_t d_names_cap = 0, d_name_off_cap = 0, names_uploaded = 0;
    uint64_t *d_life = nullptr;         // [max_metrics][2] lifetime (count, sum), metrics.go:127
    char *d_text = nullptr;
    size_t d_text_cap = 0;
    char *d_blob = nullptr;             // SER_BLOB_MAX bytes

    // K4 (lh_snapshot_merge): plan arrays + pack buffer (xmu)
    uint64_t *d_mplan = nullptr;
    uint64_t *d_mbuf = nullptr;
    size_t mbuf_bytes = 0;
    lh_merge_info merge_info{};
    hipEvent_t merge_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; // around the merge's steps (xmu)
    bool merge_events_pending = false;
    hipEvent_t xev[3] = {nullptr, nullptr, nullptr}; // large extracts: before the kernel, after it, after the copy (xmu)
    bool xev_valid = false;
    bool merge_narrow = true;           // LH_OPT_MERGE_NARROW_CELLS: rows travel at 8 / 16 bits per cell where their counts allow

    // counters (metrics.go:112-117): their own name table, the lifetime store and the "ever touched" flags in HBM
    std::shared_mutex cnames_mu;
    std::unordered_map<std::string, uint32_t> cname2id;
    std::vector<std::string> cnames;
    std::string cname_blob;
    std::vector<uint32_t> cname_off{0};
    char *d_cnames = nullptr;
    uint32_t *d_cname_off = nullptr;
    size_t d_cnames_cap = 0, d_cname_off_cap = 0, cnames_uploaded = 0;
    uint64_t *d_clife = nullptr;   // [max_counters] counterStore
    uint32_t *d_cknown = nullptr;  // [max_counters]
    std::atomic<uint64_t> c_counts{0};

    std::atomic<int> live_snapshots{0};

    // adaptive dispatch of mixed launches with few names: the single-pass kernel reports how many samples
    // missed its LDS windows; when that exceeds 2 % the engine uses the partitioned path (wider windows)
    std::atomic<uint64_t> small_samples{0};
    std::atomic<bool> small_disabled{false};
    // the region scatter reports records that found their LDS region full (a stream clustered by name); above 2 % of
    // an interval's samples later calls take the exact-layout scatter, and the regions get another try every 64 flips
    unsigned long long *h_rstat = nullptr, *d_rstat = nullptr; // pinned, device-visible: two blocks of 8 words (lh_create)
    std::atomic<bool> regions_disabled{false};
    std::atomic<uint64_t> region_samples{0}, c_region_ovf{0}, c_survey_reuse{0};
    uint64_t rstat_seen = 0, win_ovf = 0, win_samples = 0; // lh_flip only (under the epoch lock)
    int flips_since_regions_off = 0;

    // self-metrics (lh_get_counters)
    std::atomic<uint64_t> c_single{0}, c_small{0}, c_part{0}, c_direct{0}, c_launches{0}, c_flips{0}, c_busy{0},
        c_extracts{0}, c_waits{0}, c_misses{0};

    // Scratch of the partitioned mixed-ingest kernels: ONE shared block per engine for device-resident and large
    // launches (a stream that finds it last used by another stream waits on `scratch_done` first; such launches fill
    // the chip, so nothing is lost by running them one after another) -- the host-fed lanes' small launches have
    // blocks of their own, `aux` below.  Large launches are cut into sub-launches so that the block stays below
    // `scratch_cap` (lh_set_option(LH_OPT_SCRATCH_CAP_BYTES)).
    std::mutex scratch_mu;
    void *scratch_p = nullptr;
    size_t scratch_bytes = 0;
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr; // stream of the last launch that used the block
    bool scratch_used = false;
    int scratch_gen = 0;                  // generation of the last partitioned launch that used the block (scratch_mu)
    // Third generation: the survey's tables stay in the block between calls, and a stationary stream does not need a
    // new survey for every call (0.18 ms of a 1.2 ms call at config 4's slice size).  A call reuses them when they
    // were laid out for the window width it runs with, nothing else has used the block since, at most
    // `survey_every` - 1 calls have reused them and the self-metrics of the calls that have completed since the
    // survey stay healthy (region overflows + level-2 overflows + reduce-pass window misses < 2 % of the pairs).
    // The survey only decides WHERE a sample is counted: a stale one costs speed, never exactness.  (scratch_mu)
    lh::SurveyTables tables;              // (lh_dispatch.h; second generation, round 5: the same rules)
    uint32_t survey_every = 32;
    // Every lh_set_option that feeds a launch plan (LDS budgets, window cells, generation switches) is written under
    // scratch_mu and bumps tune_gen; a call takes ONE snapshot of `tune` (all its sub-launches share the survey's
    // tables, which are laid out for one plan) and tables are reused only by calls that saw the same tune_gen.
    uint64_t tune_gen = 0;
    uint64_t v3_seen_bad = 0, v3_seen_pairs = 0; // self-metrics / pairs at the last check
    uint64_t v2_seen_bad = 0, v2_seen_pairs = 0;
    // a scratch block that cannot be had sends the sub-launch through the scratch-free kernel (exact, slower)
    std::atomic<uint64_t> c_alloc_fail{0}, c_fallback{0};
    std::atomic<uint32_t> fail_allocs{0}; // LH_OPT_FAIL_SCRATCH_ALLOCS: the next N scratch allocations fail (tests)
    // A stream without skew among its names gives the third generation nothing to count in place: with more than 3/4
    // of the pairs forwarded to the reduce pass (65 536 uniform names: 88 %, 9.7 ms per 1e9 pairs) the first
    // generation's fixed two-level split is faster (7.9 ms).  Judged over completed calls; re-armed every 64 flips.
    uint64_t v3_seen_fwd = 0;
    uint32_t v3_last_call_log_w = 0;         // window width of the previous third-generation call (scratch_mu)
    std::atomic<bool> v3_disabled{false};
    uint32_t flips_since_v3_off = 0;
    std::atomic<uint64_t> c_scratch{0}, c_sublaunches{0}, c_part2{0}, c_part3{0};
    // Third generation (8 193 .. 65 536 names): the survey reports the window width that covers the stream's spans
    // (h_rstat[1], pinned); later calls use it.  A width that is too small only costs speed.
    bool v3_log_w_fixed = false;             // lh_set_option(LH_OPT_PART_V3_LOG_W) pinned it
    std::mutex probe_mu;                     // probe_width: once per engine
    bool probe_done = false;
    // Host-fed lane launches (one half-buffer of pairs each, read over PCIe in place) are link-bound in their first
    // pass and leave the link idle in the passes behind it.  On the ONE block above they run one after another -- the
    // link idles a third of the time (measured: 44 of the 52 GB/s lh_submit's single pass gets, and 2.3 G pairs/s at
    // 65 536 names where one launch's later passes take as long as its read).  Such launches therefore take the first
    // generation (no tables that outlive the launch) in one of `lane_blocks` small blocks of their own, so that lane
    // A's later passes run beside lane B's read.  (scratch_mu; lh_set_option(LH_OPT_LANE_SCRATCH_BLOCKS), 0 = the one
    // shared block as before)
    struct AuxScratch {
        void *p = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        hipStream_t stream = nullptr;
        bool used = false;
    };
    static constexpr uint32_t kAuxBlocks = 16;
    AuxScratch aux[kAuxBlocks];
    // (LH_OPT_LANE_SCRATCH_BLOCKS.  0 since round 6: a lane's half-buffer takes the direct path's cell table -- one pass, no
    // scratch -- which holds 0.89 - 0.90 of the link at every name count; in blocks of their own the lanes' partitioned launches
    // held 0.84 - 0.87 up to 8 192 names and 0.75 - 0.78 above: profiles/r06_hostfed_cells.jsonl)
    uint32_t lane_blocks = 0, aux_next = 0;
    // Above 8 192 names a lane's launch takes the third generation in its own block (lh_dispatch.h); the survey's tables
    // are shared by the lanes and only READ between surveys.  Two sets: a survey writes the set that is not in use and
    // becomes the active one; a set is rewritten only behind every lane launch in flight (the blocks' events).
    struct LaneTables {
        void *p[2] = {nullptr, nullptr};
        size_t bytes = 0;
        lh::SurveyTables t[2];
        hipEvent_t ready[2] = {nullptr, nullptr};   // recorded behind the launch that surveyed into the set
        hipStream_t ready_stream[2] = {nullptr, nullptr};
        int active = 0;
        uint64_t seen_bad = 0, seen_pairs = 0, seen_stale = 0;
    } lane_tables;
    bool lane_gen3 = true;                        // LH_OPT_LANE_GEN3
    uint32_t lane_g1_cap = lh::kLaneLevel1Workgroups;
    size_t scratch_cap = size_t(1536) << 20;     // 1.5 GiB
    bool scratch_cap_set = false, sublaunch_set = false; // lh_set_option was called: the caller's bound wins
    size_t sublaunch_pairs = size_t(1) << 29;

    bool lane_zero_copy = true;           // LH_OPT_LANE_ZERO_COPY: kernels read the pinned half-buffers in place
    lh::PartTuning tune;                  // lh_set_option; never the environment in the product build
    bool zero_copy_enabled = true;
    size_t zero_copy_max = 32768; // results up to this size are stored by the kernel straight into pinned memory
    std::mutex hD_mu;             // lh_expand_compact: the decompress table on the host, read back once
    std::vector<double> h_D;
    uint32_t flips_since_small_off = 0;   // adaptive dispatch re-arms the single-pass path every 64 flips (epoch_mu)
    bool small_forced_off = false;        // LH_OPT_SMALL_PATH = 0: never re-armed
};

struct lh_snapshot {
    lh_engine *e;
    int buf;
    bool life_applied = false;    // lh_snapshot_accumulate ran (xmu)
    bool counters_folded = false; // the interval's counter amounts are in the lifetime store (xmu)
};

namespace {

int use_device(lh_engine *e)
{
    HIPCHK(hipSetDevice(e->device));
    return LH_OK;
}

// Make `s` wait for the current epoch buffer's last clear the first time it is
// used in this epoch, and remember it so lh_flip can order the extract after it.
int epoch_touch_stream(lh_engine *e, hipStream_t s)
{
    std::lock_guard<std::mutex> g(e->streams_mu);
    for (hipStream_t t : e->epoch_streams)
        if (t == s) return LH_OK;
    HIPCHK(hipStreamWaitEvent(s, e->bufs[(size_t)e->cur].cleared, 0));
    e->epoch_streams.push_back(s);
    return LH_OK;
}

// an upper bound of any cell of the buffer: every sample adds 1 to one cell (saturates at "unknown")
void count_samples(EpochBuffer &b, size_t n)
{
    uint64_t old = __atomic_load_n(&b.nsamples, __ATOMIC_RELAXED), now;
    do {
        now = old > ~uint64_t(0) - n ? ~uint64_t(0) : old + n;
    } while (!__atomic_compare_exchange_n(&b.nsamples, &old, now, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}

// A narrow buffer moves to uint64 cells: every row's dirty span is copied into the wide store and zeroed behind the copy
// (k_widen_rows), and the kernels take the wide store from here on -- until the buffer is cleared, which returns it to
// its (clean) narrow store.  `quiesce`: the buffer is the CURRENT one and launches on other streams may still be adding
// to the narrow store -- wait for the device first (cells_mu is held unique: nothing new is enqueued meanwhile) and for
// the copy afterwards.  A snapshot's buffer is widened in stream order on the extract stream.
int widen_buffer(lh_engine *e, EpochBuffer &b, hipStream_t st, bool quiesce)
{
    if (!lh::cells_narrow(b.counts)) return LH_OK;
    const size_t M = e->cfg.max_metrics;
    if (quiesce) HIPCHK(hipDeviceSynchronize());
    if (!b.store64) {
        HIPCHK(hipMalloc((void **)&b.store64, M * LH_ROW_STRIDE * sizeof(uint64_t)));
        hipError_t me = hipMemsetAsync(b.store64, 0, M * LH_ROW_STRIDE * sizeof(uint64_t), st);
        if (me != hipSuccess) {
            (void)hipFree(b.store64);
            b.store64 = nullptr;
            HIPCHK(me);
        }
        e->c_store_bytes.fetch_add(M * LH_ROW_STRIDE * sizeof(uint64_t), std::memory_order_relaxed);
    }
    HIPCHK(lh::launch_widen_rows(b.store32, b.store64, b.ranges, (uint32_t)M, st));
    if (quiesce) HIPCHK(hipStreamSynchronize(st));
    if (quiesce) b.ingest_widened = true;
    b.counts = b.store64;
    e->c_widenings.fetch_add(1, std::memory_order_relaxed);
    return LH_OK;
}

// One ingest step's hold on the current buffer's cells (narrow engines only; a wide engine's pointer never changes).
// Taken BEFORE the step's launches are enqueued: the samples are reserved first, and a step that would take the narrow
// store past `widen_at` samples widens the buffer instead.
struct CellsHold {
    std::shared_lock<std::shared_mutex> g;
    int rc = LH_OK;
    CellsHold(lh_engine *e, EpochBuffer &b, size_t take, hipStream_t s)
    {
        if (!e->narrow) return;
        g = std::shared_lock<std::shared_mutex>(e->cells_mu);
        if (!lh::cells_narrow(b.counts)) return;
        const uint64_t r = __atomic_add_fetch(&b.reserved, (uint64_t)take, __ATOMIC_RELAXED);
        if (r <= e->widen_at && r >= take) return;
        g.unlock();
        {
            std::unique_lock<std::shared_mutex> u(e->cells_mu);
            rc = widen_buffer(e, b, s, true);
        }
        g.lock();
    }
};

int launch_single(lh_engine *e, uint32_t id, const double *d_v, size_t n, hipStream_t s)
{
    EpochBuffer &b = e->bufs[(size_t)e->cur];
    CellsHold hold(e, b, n, s);
    if (hold.rc) return hold.rc;
    count_samples(b, n);
    // a workgroup's uint32 LDS bins must not wrap even if every sample of the launch lands in one
    // bucket: keep one launch below 2^32 samples
    const size_t kMaxLaunch = size_t(1) << 31;
    while (n) {
        const size_t take = n < kMaxLaunch ? n : kMaxLaunch;
        HIPCHK(lh::launch_ingest_single(d_v, take, lh::cells_at(b.counts, (size_t)id * LH_ROW_STRIDE), b.ranges + 2 * (size_t)id,
                                        e->d_Tx, e->num_cus, s));
        e->c_single.fetch_add(take, std::memory_order_relaxed);
        e->c_launches.fetch_add(1, std::memory_order_relaxed);
        d_v += take;
        n -= take;
    }
    return LH_OK;
}

// ---- mixed (id, value) ingest: lh_dispatch.h chooses, these execute ------------------------------------------------

// A scratch block that cannot be had is not an error of the call (ingest never fails, metrics.go:251, 273): the caller
// of this function sends the sub-launch through the scratch-free direct kernel instead -- exact, slower -- and the
// engine counts it (lh_counters.scratch_alloc_failures / samples_fallback).  LH_OPT_FAIL_SCRATCH_ALLOCS makes the next N
// attempts fail (tests/test_gpu_faults.py).
bool scratch_alloc(lh_engine *e, void **p, size_t bytes)
{
    uint32_t left = e->fail_allocs.load(std::memory_order_relaxed);
    while (left && !e->fail_allocs.compare_exchange_weak(left, left - 1u, std::memory_order_relaxed)) {}
    hipError_t he = left ? hipErrorOutOfMemory : hipMalloc(p, bytes);
    if (he == hipSuccess) return true;
    if (!left) set_last_error("hipMalloc(scratch)", he); // (also clears the runtime's sticky error)
    *p = nullptr;
    e->c_alloc_fail.fetch_add(1, std::memory_order_relaxed);
    return false;
}

int run_direct(lh_engine *e, EpochBuffer &b, lh::Ids d_ids, const double *d_v, size_t take, hipStream_t s)
{
    // (round 6: whole tiles through a per-workgroup LDS table of cells, the rest one global atomic per sample)
    HIPCHK(lh::launch_ingest_pairs_cells(d_ids, d_v, take, b.counts, b.ranges, e->cfg.max_metrics, e->d_Tx, e->d_err,
                                         e->num_cus, s));
    e->c_direct.fetch_add(take, std::memory_order_relaxed);
    return LH_OK;
}

int run_small(lh_engine *e, EpochBuffer &b, lh::Ids d_ids, const double *d_v, size_t take, hipStream_t s)
{
    HIPCHK(lh::launch_ingest_pairs_small(d_ids, d_v, take, b.counts, b.ranges, e->cfg.max_metrics, e->d_Tx, e->d_err,
                                         e->num_cus, s));
    e->small_samples.fetch_add(take, std::memory_order_relaxed);
    e->c_small.fetch_add(take, std::memory_order_relaxed);
    return LH_OK;
}

int run_fallback(lh_engine *e, EpochBuffer &b, lh::Ids d_ids, const double *d_v, size_t take, hipStream_t s)
{
    e->c_fallback.fetch_add(take, std::memory_order_relaxed);
    return run_direct(e, b, d_ids, d_v, take, s);
}

// One call's view of the shared block: ONE survey per call, not one per sub-launch (second and third generation).  The
// survey's tables live in the shared scratch block, which every stream of the engine shares: the block's lock is held
// from the call's first partitioned sub-launch to its last, so that no other stream's launch can overwrite them in
// between (enqueueing is all that happens under the lock; the kernels run later, in stream order behind `scratch_done`).
struct PairsCall {
    lh::DispatchState st;
    uint64_t tune_gen = 0;
    bool surveyed = false; // the block holds the survey this call runs on (its own, or a reused one)
    bool judged = false;   // the reuse decision is made once per call, at its first surveyed-generation launch
    bool log_w_fixed = false; // lh_set_option pinned the third generation's window width
    std::unique_lock<std::mutex> lock;
};

// A lane's half-buffer through the first generation in one of the lanes' own blocks (lh_engine::aux).  The lock is held
// only while the block is picked and the launch enqueued (the lanes' small launches used to keep it for the
// rest of the call, direct and small kernels included).
int run_lane_block(lh_engine *e, EpochBuffer &b, PairsCall &c, const lh::Step &st, lh::Ids d_ids, const double *d_v,
                   hipStream_t s)
{
    // (a call whose earlier sub-launch went through the shared block already holds the lock, and keeps it)
    struct Hold {
        PairsCall &c;
        const bool had;
        Hold(PairsCall &c_, std::mutex &m) : c(c_), had(c_.lock.owns_lock()) { if (!had) c.lock = std::unique_lock<std::mutex>(m); }
        ~Hold() { if (!had) c.lock.unlock(); }
    } hold(c, e->scratch_mu);
    const uint32_t nb = e->lane_blocks;
    if (!nb) return run_fallback(e, b, d_ids, d_v, st.take, s); // (the option changed under the call: exact anyway)
    lh_engine::AuxScratch *a = nullptr;
    for (uint32_t i = 0; i < nb && !a; i++)
        if (e->aux[i].used && e->aux[i].stream == s) a = &e->aux[i]; // stream order is all it needs
    for (uint32_t i = 0; i < nb && !a; i++)
        if (!e->aux[i].used) a = &e->aux[i];
    for (uint32_t i = 0; i < nb && !a; i++)
        if (hipEventQuery(e->aux[i].done) == hipSuccess) a = &e->aux[i];
    if (!a) a = &e->aux[e->aux_next++ % nb];                          // all busy: behind one of them
    (void)hipGetLastError(); // hipEventQuery's "not ready" is not an error of this call
    if (a->bytes < st.scratch) {
        if (a->p) {
            if (a->used) HIPCHK(hipEventSynchronize(a->done));
            HIPCHK(hipFree(a->p));
            a->p = nullptr;
            a->bytes = 0;
            a->used = false;
        }
        if (!scratch_alloc(e, &a->p, st.scratch_alloc)) return run_fallback(e, b, d_ids, d_v, st.take, s);
        a->bytes = st.scratch_alloc;
    }
    if (a->used && a->stream != s) HIPCHK(hipStreamWaitEvent(s, a->done, 0));
    if (st.kind == lh::PATH_GEN3) {
        lh_engine::LaneTables &lt = e->lane_tables;
        if (!lt.p[0]) { // both sets at once, on first use
            const size_t tb = lh::part3_tables_bytes(e->cfg.max_metrics);
            void *p0 = nullptr, *p1 = nullptr;
            if (!tb || !scratch_alloc(e, &p0, tb)) return run_fallback(e, b, d_ids, d_v, st.take, s);
            if (!scratch_alloc(e, &p1, tb)) { (void)hipFree(p0); return run_fallback(e, b, d_ids, d_v, st.take, s); }
            lt.p[0] = p0;
            lt.p[1] = p1;
            lt.bytes = tb;
            for (hipEvent_t &ev : lt.ready)
                if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        // what the lanes' third-generation launches completed since the last look reported (their own block of pinned words)
        // (pairs first: a report that lands between the reads then shows as pairs without their overflows, never the reverse)
        const unsigned long long *lr = e->h_rstat + kLaneRstat;
        const uint64_t pairs = __atomic_load_n(&lr[6], __ATOMIC_ACQUIRE);
        const uint64_t bad = __atomic_load_n(&lr[0], __ATOMIC_RELAXED) + __atomic_load_n(&lr[4], __ATOMIC_RELAXED) +
                             __atomic_load_n(&lr[5], __ATOMIC_RELAXED);
        const uint64_t stale = __atomic_load_n(&lr[7], __ATOMIC_RELAXED);
        int set = lt.active;
        // A stale survey (h_rstat[7]: the hot windows take fewer pairs than when the tables were new) counts only once the
        // set has served kLaneStaleMinAge launches: lanes may carry different streams, each a few per cent off the launch
        // that surveyed, and a survey is a large part of a lane-sized launch -- they must not take turns re-surveying.
        const bool heed_stale = lt.t[set].valid && lt.t[set].age >= lh::kLaneStaleMinAge;
        const bool healthy = lh::healthy_share((bad - lt.seen_bad) + (heed_stale ? stale - lt.seen_stale : 0), pairs - lt.seen_pairs);
        lt.seen_bad = bad;
        lt.seen_stale = stale;
        lt.seen_pairs = pairs;
        size_t survey_n = 0;
        if (lh::survey_reusable(lt.t[set], 3, st.tune.v3_log_w, c.tune_gen, e->survey_every, healthy)) {
            lt.t[set].age++;
            e->c_survey_reuse.fetch_add(1, std::memory_order_relaxed);
            // (the tables are complete once the launch that surveyed them is: another lane's stream waits for that)
            if (lt.ready_stream[set] != s) HIPCHK(hipStreamWaitEvent(s, lt.ready[set], 0));
        } else {
            // This launch surveys its own pairs into the OTHER set, behind whatever may still read that set: every block's
            // last launch -- not only those that read this set themselves: a block's launches run one behind the other, and
            // an earlier launch that reads the set may still be running under a later one that does not (with 16 lanes on
            // 8 blocks and two surveys in quick succession that lost counts once in ten runs).  A survey is one launch in
            // `survey_every`: the bubble does not show.
            set ^= 1;
            for (uint32_t i = 0; i < lh_engine::kAuxBlocks; i++)
                if (&e->aux[i] != a && e->aux[i].used) HIPCHK(hipStreamWaitEvent(s, e->aux[i].done, 0));
            survey_n = st.take;
            lt.t[set].valid = true;
            lt.t[set].gen = 3;
            lt.t[set].log_w = st.tune.v3_log_w;
            lt.t[set].tune_gen = c.tune_gen;
            lt.t[set].age = 1;
            lt.active = set;
        }
        {
            //  a surveying launch that fails leaves the set it was to fill unwritten: the set must not stay
            // marked valid, or later launches would wait on an event that was never recorded and take their names from
            // whatever the fresh allocation holds
            hipError_t le = lh::launch_ingest_pairs_part3(d_ids, d_v, st.take, survey_n, b.counts, b.ranges, e->cfg.max_metrics,
                                                          e->d_Tx, e->d_err, a->p, a->bytes, lt.p[set], e->num_cus, st.tune,
                                                          e->d_rstat ? e->d_rstat + kLaneRstat : nullptr,
                                                          e->d_rstat ? reinterpret_cast<uint32_t *>(e->d_rstat + 1) : nullptr, s);
            if (le == hipSuccess && survey_n) {
                le = hipEventRecord(lt.ready[set], s);
                lt.ready_stream[set] = s;
            }
            if (le != hipSuccess) {
                if (survey_n) lt.t[set].valid = false;
                HIPCHK(le);
            }
        }
        e->region_samples.fetch_add(st.take, std::memory_order_relaxed);
        e->c_part3.fetch_add(st.take, std::memory_order_relaxed);
    } else {
        HIPCHK(lh::launch_ingest_pairs_part(d_ids, d_v, st.take, b.counts, b.ranges, e->cfg.max_metrics, e->d_Tx, e->d_err,
                                            a->p, a->bytes, e->num_cus, st.tune, s));
    }
    HIPCHK(hipEventRecord(a->done, s));
    a->stream = s;
    a->used = true;
    e->c_part.fetch_add(st.take, std::memory_order_relaxed);
    e->c_sublaunches.fetch_add(1, std::memory_order_relaxed);
    return LH_OK;
}

// scratch_mu held.  The shared block, at least `bytes` large; false: cannot be had.
bool shared_block(lh_engine *e, PairsCall &c, size_t bytes, int *rc)
{
    *rc = LH_OK;
    if (e->scratch_bytes >= bytes) return true;
    void *np = nullptr;
    bool ok = scratch_alloc(e, &np, bytes);
    if (!ok && e->scratch_p) { // the old block and the new one did not fit side by side: give the old one up first
        if (e->scratch_used && hipEventSynchronize(e->scratch_done) != hipSuccess) { *rc = LH_EDEVICE; return false; }
        (void)hipFree(e->scratch_p);
        e->scratch_p = nullptr;
        e->scratch_bytes = 0;
        e->scratch_used = false;
        e->tables.valid = false;
        c.surveyed = false;
        e->c_scratch.store(0, std::memory_order_relaxed);
        ok = scratch_alloc(e, &np, bytes);
    }
    if (!ok) return false;
    if (e->scratch_p) {
        if (e->scratch_used && hipEventSynchronize(e->scratch_done) != hipSuccess) { *rc = LH_EDEVICE; (void)hipFree(np); return false; }
        (void)hipFree(e->scratch_p); // earlier launches have finished reading it
    }
    e->scratch_p = np;
    e->scratch_bytes = bytes;
    e->scratch_used = false;
    e->c_scratch.store(bytes, std::memory_order_relaxed);
    c.surveyed = false; // a new block: the tables of this call's earlier sub-launches went with the old one
    e->tables.valid = false;
    return true;
}

// The reuse decision, once per call: what the launches completed since the last look reported (pinned words).
void judge_tables(lh_engine *e, PairsCall &c, int gen, uint32_t layout)
{
    c.judged = true;
    bool healthy = true;
    if (gen == 3) {
        // ([7]: pairs a STALE survey kept out of the hot windows -- the values moved under it; stale_judge, lh_kernels_part2.h)
        const uint64_t pairs = __atomic_load_n(&e->h_rstat[6], __ATOMIC_ACQUIRE); // of the shared block's launches that reported
        const uint64_t bad = __atomic_load_n(&e->h_rstat[0], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[4], __ATOMIC_RELAXED) +
                             __atomic_load_n(&e->h_rstat[5], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[7], __ATOMIC_RELAXED);
        const uint64_t fwd = __atomic_load_n(&e->h_rstat[3], __ATOMIC_RELAXED);
        healthy = lh::healthy_share(bad - e->v3_seen_bad, pairs - e->v3_seen_pairs);
        if (lh::names_without_skew(pairs - e->v3_seen_pairs, fwd - e->v3_seen_fwd, healthy, c.st.call_log_w == e->v3_last_call_log_w))
            e->v3_disabled.store(true); // (takes effect at the next call)
        e->v3_last_call_log_w = c.st.call_log_w;
        e->v3_seen_fwd = fwd;
        e->v3_seen_bad = bad;
        e->v3_seen_pairs = pairs;
    } else {
        // second generation: region overflows (k_scatter3 adds them to the pinned word as it retires) against the pairs
        // ENQUEUED through it since the last look -- an upper bound of what has completed, so the share is a lower
        // bound; lh_flip's per-interval judgement (regions_disabled) is the backstop
        // ... and the pairs a stale survey kept out of the hot windows (k_scatter3's last workgroup: stale_judge)
        const uint64_t bad = __atomic_load_n(&e->h_rstat[0], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[7], __ATOMIC_RELAXED),
                       pairs = e->c_part2.load(std::memory_order_relaxed);
        healthy = lh::healthy_share(bad - e->v2_seen_bad, pairs - e->v2_seen_pairs);
        e->v2_seen_bad = bad;
        e->v2_seen_pairs = pairs;
    }
    if (!c.surveyed && lh::survey_reusable(e->tables, gen, layout, c.tune_gen, e->survey_every, healthy)) {
        c.surveyed = true; // this call runs on an earlier call's survey
        e->tables.age++;
        e->c_survey_reuse.fetch_add(1, std::memory_order_relaxed);
    } else if (!c.surveyed) {
        e->tables.valid = true; // (the launch that follows surveys)
        e->tables.gen = gen;
        e->tables.log_w = layout;
        e->tables.tune_gen = c.tune_gen;
        e->tables.age = 1;
    }
}

// A partitioned sub-launch in the engine's shared block.  n_left: what is left of the call (the first sub-launch
// surveys all of it).
int run_shared(lh_engine *e, EpochBuffer &b, PairsCall &c, const lh::Step &st, lh::Ids d_ids, const double *d_v,
               size_t n_left, hipStream_t s)
{
    if (!c.lock.owns_lock()) c.lock = std::unique_lock<std::mutex>(e->scratch_mu);
    int rc = LH_OK;
    if (!shared_block(e, c, st.scratch, &rc)) return rc != LH_OK ? rc : run_fallback(e, b, d_ids, d_v, st.take, s);
    if (e->scratch_used && e->scratch_stream != s) HIPCHK(hipStreamWaitEvent(s, e->scratch_done, 0));
    const int gen = st.kind == lh::PATH_GEN2 ? 2 : st.kind == lh::PATH_GEN3 ? 3 : 1;
    if (gen != e->scratch_gen) c.surveyed = false; // the generations lay their tables out differently
    if (gen == 1) {
        HIPCHK(lh::launch_ingest_pairs_part(d_ids, d_v, st.take, b.counts, b.ranges, e->cfg.max_metrics, e->d_Tx,
                                            e->d_err, e->scratch_p, e->scratch_bytes, e->num_cus, st.tune, s));
        c.surveyed = false; // its records start at offset 0 of the block: the survey's tables are gone
        e->tables.valid = false;
    } else {
        // what the tables were laid out for: the window width (third generation) / the scatter shape (second: the exact
        // layout has no per-partition regions)
        if (!c.judged) judge_tables(e, c, gen, gen == 3 ? st.tune.v3_log_w : (st.tune.v2_shape & 7u));
        const size_t survey_n = c.surveyed ? 0 : n_left;
        //  judge_tables has marked the tables valid for the launch that is about to fill them: if that launch
        // cannot be enqueued they are not
        hipError_t le;
        if (gen == 2) {
            le = lh::launch_ingest_pairs_part2(d_ids, d_v, st.take, survey_n, b.counts, b.ranges, e->cfg.max_metrics,
                                               e->d_Tx, e->d_err, e->scratch_p, e->scratch_bytes, e->num_cus, st.tune,
                                               e->d_rstat, s);
        } else {
            le = lh::launch_ingest_pairs_part3(d_ids, d_v, st.take, survey_n, b.counts, b.ranges, e->cfg.max_metrics,
                                               e->d_Tx, e->d_err, e->scratch_p, e->scratch_bytes, nullptr, e->num_cus, st.tune,
                                               e->d_rstat, e->d_rstat ? reinterpret_cast<uint32_t *>(e->d_rstat + 1) : nullptr,
                                               s);
        }
        if (le != hipSuccess) {
            if (survey_n) { e->tables.valid = false; c.surveyed = false; }
            HIPCHK(le);
        }
        if (gen == 2) {
            if (st.tune.v2_shape & 2u) e->region_samples.fetch_add(st.take, std::memory_order_relaxed);
            e->c_part2.fetch_add(st.take, std::memory_order_relaxed);
        } else {
            e->region_samples.fetch_add(st.take, std::memory_order_relaxed);
            e->c_part3.fetch_add(st.take, std::memory_order_relaxed);
        }
        c.surveyed = true;
    }
    e->scratch_gen = gen;
    HIPCHK(hipEventRecord(e->scratch_done, s));
    e->scratch_stream = s;
    e->scratch_used = true;
    e->c_part.fetch_add(st.take, std::memory_order_relaxed);
    e->c_sublaunches.fetch_add(1, std::memory_order_relaxed);
    return LH_OK;
}

// What a call's path choice reads: one snapshot, taken under the lock every plan option is written under.
void snapshot_dispatch(lh_engine *e, PairsCall &c)
{
    bool log_w_fixed;
    {
        std::lock_guard<std::mutex> g(e->scratch_mu);
        c.st.tune = e->tune;
        c.tune_gen = e->tune_gen;
        c.st.lane_blocks = e->lane_blocks;
        c.st.lane_gen3 = e->lane_gen3;
        c.st.lane_g1_cap = e->lane_g1_cap;
        c.st.scratch_cap = e->scratch_cap;
        c.st.scratch_cap_set = e->scratch_cap_set;
        c.st.sublaunch_pairs = e->sublaunch_pairs;
        c.st.sublaunch_set = e->sublaunch_set;
        log_w_fixed = e->v3_log_w_fixed;
    }
    c.log_w_fixed = log_w_fixed;
    c.st.max_metrics = e->cfg.max_metrics;
    c.st.num_cus = e->num_cus;
    c.st.lane_samples = (size_t)e->cfg.lane_samples;
    c.st.small_disabled = e->small_disabled.load(std::memory_order_relaxed);
    c.st.regions_disabled = e->regions_disabled.load(std::memory_order_relaxed);
    c.st.v3_disabled = e->v3_disabled.load(std::memory_order_relaxed);
    // Third generation: the window width the last completed survey reported (0 until one has run).  Read ONCE per call:
    // the sub-launches of a call share the survey's per-partition tables, which are laid out for one width, and the
    // call's own survey stores its report while later sub-launches are still being enqueued.
    c.st.call_log_w = c.st.tune.v3_log_w;
    {
        const uint32_t raw = (uint32_t)__atomic_load_n(&e->h_rstat[1], __ATOMIC_RELAXED), lw = raw & 0xffu;
        if (!log_w_fixed && lw >= 10 && lw <= 14) c.st.call_log_w = lw;
        // bit 8: more than 1/8 of the sampled mass outside the second generation's cold windows at this name count
        c.st.call_yield = e->cfg.max_metrics <= 8192 && (raw & 0x100u) != 0;
        // second generation (<= 8 192 names): its survey reports 13 / 14 in the same word -- 14: spans wider than the
        // 8 192-bin reduce windows carry at least 5 % of the sampled mass
        c.st.call_wide = e->cfg.max_metrics <= 8192 && lw == 14;
    }
}

// The engine's first large third-generation call: no survey has reported a window width yet, and a wide stream on the
// default (narrowest) width counts most of its records through the overflow paths -- exact, but a 1e9-pair call of
// 21-decade values took 636 ms instead of 8.9 (profiles/r06_first_call.txt).  The survey's first three kernels run
// alone on the call's pairs and the host WAITS for the class they report (~0.3 ms behind whatever the stream still
// holds).  Once per engine: later changes of the stream are followed by the surveys' reports as before (one or two
// slow calls; the health judge re-surveys).  Device-resident calls only: a lane's half-buffer is too small to matter.
void probe_width(lh_engine *e, PairsCall &c, lh::Ids d_ids, const double *d_v, size_t n, hipStream_t s)
{
    constexpr size_t kProbeMinPairs = size_t(1) << 24;
    // (1 025 .. 8 192 names too: there the probe's report decides whether the call is the third generation's at all --
    // bit 8 of the word, lh_kernels_part2.h -- 1e9 pairs of normal(0, 1e3) over 8 192 names: first call 56 ms without it)
    if (c.st.max_metrics <= 1024 || c.log_w_fixed || n < kProbeMinPairs || !c.st.tune.v3 || c.st.v3_disabled) return;
    if (__atomic_load_n(&e->h_rstat[1], __ATOMIC_RELAXED) != 0 || !e->d_rstat) return;
    if (lh::peel_first((uintptr_t)d_ids.p, d_ids.width, (uintptr_t)d_v, n)) { d_ids = d_ids.plus(1); d_v += 1; n -= 1; }
    std::lock_guard<std::mutex> g(e->probe_mu);
    if (e->probe_done) return;
    e->probe_done = true; // (whatever happens below: the probe is an optimisation, tried once)
    const size_t tb = lh::part3_tables_bytes(e->cfg.max_metrics);
    void *p = nullptr;
    if (!tb || hipMalloc(&p, tb) != hipSuccess) { (void)hipGetLastError(); return; }
    n = std::min(n, size_t(1) << 30);
    lh::PartTuning pt = c.st.tune;
    if (c.st.max_metrics <= 8192) pt.v2_yield = true; // (the probe's plan is the third generation's whatever it will report)
    if (lh::launch_part3_probe(d_ids, d_v, n, e->cfg.max_metrics, e->d_Tx, p, e->num_cus, pt,
                               reinterpret_cast<uint32_t *>(e->d_rstat + 1), s) == hipSuccess &&
        hipStreamSynchronize(s) == hipSuccess) {
        const uint32_t raw = (uint32_t)__atomic_load_n(&e->h_rstat[1], __ATOMIC_ACQUIRE), lw = raw & 0xffu;
        if (lw >= 10 && lw <= 14) c.st.call_log_w = lw;
        if (c.st.max_metrics <= 8192) c.st.call_yield = (raw & 0x100u) != 0;
    } else {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s); // (the block below is freed: nothing may still write it)
    }
    (void)hipFree(p);
}

int launch_pairs(lh_engine *e, lh::Ids d_ids, const double *d_v, size_t n, hipStream_t s, bool host_fed = false)
{
    EpochBuffer &b = e->bufs[(size_t)e->cur];
    PairsCall c;
    snapshot_dispatch(e, c);
    if (!host_fed) probe_width(e, c, d_ids, d_v, n, s);
    bool peel = lh::peel_first((uintptr_t)d_ids.p, d_ids.width, (uintptr_t)d_v, n);
    while (n) {
        lh::Step st;
        if (peel) { // the odd-aligned first sample: the direct kernel, so that the rest of the call is vector-aligned
            st.take = 1;
            peel = false;
        } else {
            st = lh::choose_step(c.st, (uintptr_t)d_ids.p, d_ids.width, (uintptr_t)d_v, n, host_fed);
        }
        CellsHold hold(e, b, st.take, s);
        if (hold.rc) return hold.rc;
        int rc;
        if (st.kind == lh::PATH_SMALL) {
            rc = run_small(e, b, d_ids, d_v, st.take, s);
        } else if (st.kind == lh::PATH_DIRECT) {
            rc = run_direct(e, b, d_ids, d_v, st.take, s);
        } else if (st.lane_block) {
            rc = run_lane_block(e, b, c, st, d_ids, d_v, s);
        } else {
            rc = run_shared(e, b, c, st, d_ids, d_v, n, s);
        }
        if (rc) return rc; // a device error: what was enqueued before it is counted and stays in the interval
        count_samples(b, st.take); // (behind the enqueue: the merge's cell bound counts what is in the buffer)
        e->c_launches.fetch_add(1, std::memory_order_relaxed);
        d_ids = d_ids.plus(st.take);
        d_v += st.take;
        n -= st.take;
    }
    return LH_OK;
}

int launch_counts(lh_engine *e, const uint32_t *d_ids, const uint64_t *d_amounts, size_t n, hipStream_t s)
{
    EpochBuffer &b = e->bufs[(size_t)e->cur];
    HIPCHK(lh::launch_count_add(d_ids, d_amounts, n, b.ccur, b.cflag, e->cfg.max_counters, e->d_err, e->num_cus, s));
    e->c_counts.fetch_add(n, std::memory_order_relaxed);
    e->c_launches.fetch_add(1, std::memory_order_relaxed);
    return LH_OK;
}

// Lane mutex held, epoch lock held (shared or unique).
int lane_launch(lh_engine *e, Lane &ln)
{
    if (ln.fill == 0) return LH_OK;
    const int h = ln.cur;
    const size_t n = ln.fill;
    int rc = epoch_touch_stream(e, ln.stream);
    if (rc) return rc;
    // Every ingest kernel reads its input once, front to back: with the half-buffers mapped into the device's address
    // space the kernel fetches them over PCIe itself and no copy engine, no staging copy in HBM stands between the
    // producer's store and the bucket (one SDMA engine moves ~28 GB/s, which was the host-fed path's ceiling on some
    // boxes: VERDICT r2 weak #7).  LH_OPT_LANE_ZERO_COPY = 0 restores hipMemcpyAsync into HBM first.
    const bool zc = e->lane_zero_copy && ln.m_vals[h] && ln.m_ids[h] && ln.mode != LANE_COUNTS;
    const double *dv = zc ? ln.m_vals[h] : ln.d_vals[h];
    const uint32_t *di = zc ? ln.m_ids[h] : ln.d_ids[h];
    if (!zc) HIPCHK(hipMemcpyAsync(ln.d_vals[h], ln.h_vals[h], n * sizeof(double), hipMemcpyHostToDevice, ln.stream));
    if (ln.mode == LANE_PAIRS) {
        if (!zc) HIPCHK(hipMemcpyAsync(ln.d_ids[h], ln.h_ids[h], n * sizeof(uint32_t), hipMemcpyHostToDevice, ln.stream));
        rc = launch_pairs(e, di, dv, n, ln.stream, true);
    } else if (ln.mode == LANE_PAIRS16) {
        if (!zc) HIPCHK(hipMemcpyAsync(ln.d_ids[h], ln.h_ids[h], n * sizeof(uint16_t), hipMemcpyHostToDevice, ln.stream));
        rc = launch_pairs(e, reinterpret_cast<const uint16_t *>(di), dv, n, ln.stream, true);
    } else if (ln.mode == LANE_COUNTS) { // the value half-buffer carries uint64 amounts
        HIPCHK(hipMemcpyAsync(ln.d_ids[h], ln.h_ids[h], n * sizeof(uint32_t), hipMemcpyHostToDevice, ln.stream));
        rc = launch_counts(e, ln.d_ids[h], reinterpret_cast<const uint64_t *>(ln.d_vals[h]), n, ln.stream);
    } else {
        rc = launch_single(e, ln.single_id, dv, n, ln.stream);
    }
    if (rc) return rc;
    HIPCHK(hipEventRecord(ln.done[h], ln.stream));
    ln.inflight[h] = true;
    ln.cur ^= 1;
    ln.fill = 0;
    ln.mode = LANE_NONE;
    if (ln.inflight[ln.cur]) { // back-pressure: never drop (metrics.go:273-295 is synchronous)
        if (hipEventQuery(ln.done[ln.cur]) != hipSuccess) e->c_waits.fetch_add(1, std::memory_order_relaxed);
        HIPCHK(hipEventSynchronize(ln.done[ln.cur]));
        ln.inflight[ln.cur] = false;
    }
    return LH_OK;
}

Lane &pick_lane(lh_engine *e)
{
    const size_t h = std::hash<std::thread::id>()(std::this_thread::get_id());
    return *e->lanes[h % e->lanes.size()];
}

int flush_all_lanes(lh_engine *e)
{
    for (auto &lp : e->lanes) {
        LaneLock g(*lp); // a producer that holds a reservation commits first (lh_commit_pairs never blocks on the epoch)
        int rc = lane_launch(e, *lp);
        if (rc) return rc;
    }
    return LH_OK;
}

void free_engine(lh_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (auto &lp : e->lanes) {
        if (!lp) continue;
        for (int h = 0; h < 2; h++) {
            if (lp->h_vals[h]) (void)hipHostFree(lp->h_vals[h]);
            if (lp->h_ids[h]) (void)hipHostFree(lp->h_ids[h]);
            if (lp->d_vals[h]) (void)hipFree(lp->d_vals[h]);
            if (lp->d_ids[h]) (void)hipFree(lp->d_ids[h]);
            if (lp->done[h]) (void)hipEventDestroy(lp->done[h]);
        }
        if (lp->stream) (void)hipStreamDestroy(lp->stream);
    }
    for (auto &b : e->bufs) {
        if (b.store32) (void)hipFree(b.store32);
        if (b.store64) (void)hipFree(b.store64);
        if (b.ranges) (void)hipFree(b.ranges);
        if (b.ccur) (void)hipFree(b.ccur);
        if (b.cflag) (void)hipFree(b.cflag);
        if (b.cleared) (void)hipEventDestroy(b.cleared);
    }
    for (hipEvent_t ev : e->flip_events) (void)hipEventDestroy(ev);
    if (e->scratch_p) (void)hipFree(e->scratch_p);
    if (e->scratch_done) (void)hipEventDestroy(e->scratch_done);
    for (auto &a : e->aux) {
        if (a.p) (void)hipFree(a.p);
        if (a.done) (void)hipEventDestroy(a.done);
    }
    for (void *p : e->lane_tables.p)
        if (p) (void)hipFree(p);
    for (hipEvent_t ev : e->lane_tables.ready)
        if (ev) (void)hipEventDestroy(ev);
    if (e->d_Tx) (void)hipFree(e->d_Tx);
    if (e->d_D) (void)hipFree(e->d_D);
    if (e->d_err) (void)hipFree(e->d_err);
    if (e->h_rstat) (void)hipHostFree(e->h_rstat);
    if (e->d_xbuf) (void)hipFree(e->d_xbuf);
    if (e->h_xbuf) (void)hipHostFree(e->h_xbuf);
    if (e->d_names) (void)hipFree(e->d_names);
    if (e->d_name_off) (void)hipFree(e->d_name_off);
    if (e->d_life) (void)hipFree(e->d_life);
    if (e->d_text) (void)hipFree(e->d_text);
    if (e->d_blob) (void)hipFree(e->d_blob);
    if (e->d_mplan) (void)hipFree(e->d_mplan);
    for (hipEvent_t ev : e->xev)
        if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->merge_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->d_cnames) (void)hipFree(e->d_cnames);
    if (e->d_cname_off) (void)hipFree(e->d_cname_off);
    if (e->d_clife) (void)hipFree(e->d_clife);
    if (e->d_cknown) (void)hipFree(e->d_cknown);
    if (e->d_mbuf) (void)hipFree(e->d_mbuf);
    if (e->main_stream) (void)hipStreamDestroy(e->main_stream);
    if (e->xstream) (void)hipStreamDestroy(e->xstream);

    delete e;
}

int ensure_xbuf(lh_engine *e, size_t bytes)
{
    if (bytes <= e->xbuf_bytes) return LH_OK;
    if (e->d_xbuf) (void)hipFree(e->d_xbuf);
    if (e->h_xbuf) (void)hipHostFree(e->h_xbuf);
    e->d_xbuf = nullptr;
    e->h_xbuf = nullptr;
    e->d_hxbuf = nullptr;
    e->xbuf_bytes = 0;
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc((void **)&e->d_xbuf, cap));
    HIPCHK(hipHostMalloc((void **)&e->h_xbuf, cap, hipHostMallocDefault));
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, e->h_xbuf, 0) == hipSuccess) e->d_hxbuf = static_cast<unsigned char *>(dp);
    else (void)hipGetLastError(); // no mapping: results always travel by copy
    e->xbuf_bytes = cap;
    return LH_OK;
}

int create_impl(const lh_config *cfg_in, lh_engine *e)
{
    int ndev = 0;
    hipError_t dc = hipGetDeviceCount(&ndev);
    if (dc != hipSuccess || ndev <= 0) {
        std::snprintf(g_last_error, sizeof(g_last_error), "no HIP device visible (hipGetDeviceCount: %s, count %d)",
                      hipGetErrorString(dc), ndev);
        return LH_ENODEVICE;
    }
    if (cfg_in->device < 0 || cfg_in->device >= ndev) return LH_EINVAL;
    e->device = cfg_in->device;
    HIPCHK(hipSetDevice(e->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, e->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::snprintf(g_last_error, sizeof(g_last_error), "device %d is %s; this library carries gfx950 code only",
                      e->device, prop.gcnArchName);
        return LH_ENODEVICE;
    }
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;

    HIPCHK(hipStreamCreateWithFlags(&e->main_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&e->xstream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->scratch_done, hipEventDisableTiming));
    for (auto &a : e->aux) HIPCHK(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));

    HIPCHK(hipMalloc((void **)&e->d_Tx, sizeof(double) * LH_NTHRESH));
    // (padded with zeros to a row's stride: k_extract_wave reads whole 4-bin groups of the table beside the cells)
    HIPCHK(hipMalloc((void **)&e->d_D, sizeof(double) * LH_ROW_STRIDE));
    HIPCHK(hipMemset(e->d_D, 0, sizeof(double) * LH_ROW_STRIDE));
    // Two report blocks of 8 words: [0..7] the launches on the shared block, [8..15] the host-fed lanes' launches (ADVICE r5:
    // each judge reads only its own producers' reports).  Word [1] (the survey's window class) is shared by both.
    HIPCHK(hipHostMalloc((void **)&e->h_rstat, 128, hipHostMallocDefault));
    for (int i = 0; i < 16; i++) e->h_rstat[i] = 0; // [0] level-1 region overflows [1] window class [2..5] part3 self-metrics [6] pairs they cover [7] pairs a stale survey cost its hot windows
    {
        void *dp = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dp, e->h_rstat, 0));
        e->d_rstat = static_cast<unsigned long long *>(dp);
    }
    HIPCHK(hipMalloc((void **)&e->d_err, 16)); // [0] bad id flag, [1] window misses, [2] extract done counter
    HIPCHK(hipMemsetAsync(e->d_err, 0, 16, e->xstream));
    HIPCHK(lh::launch_gen_tables(e->d_Tx, e->d_D, e->xstream));

    const size_t M = e->cfg.max_metrics;
    HIPCHK(hipMalloc((void **)&e->d_life, M * 2 * sizeof(uint64_t)));
    HIPCHK(hipMemsetAsync(e->d_life, 0, M * 2 * sizeof(uint64_t), e->xstream));
    HIPCHK(hipMalloc((void **)&e->d_blob, lh::SER_BLOB_MAX));
    const size_t NC = e->cfg.max_counters;
    if (NC) {
        HIPCHK(hipMalloc((void **)&e->d_clife, NC * sizeof(uint64_t)));
        HIPCHK(hipMalloc((void **)&e->d_cknown, NC * sizeof(uint32_t)));
        HIPCHK(hipMemsetAsync(e->d_clife, 0, NC * sizeof(uint64_t), e->xstream));
        HIPCHK(hipMemsetAsync(e->d_cknown, 0, NC * sizeof(uint32_t), e->xstream));
    }
    // the cell width of the epoch buffers (lh_cells.h): 32-bit above 8 192 names unless the caller chose
    e->narrow = e->cfg.cell_bits == 32 || (e->cfg.cell_bits == 0 && e->cfg.max_metrics > 8192);
    e->bufs.resize(e->cfg.num_buffers);
    for (auto &b : e->bufs) {
        if (NC) {
            HIPCHK(hipMalloc((void **)&b.ccur, NC * sizeof(uint64_t)));
            HIPCHK(hipMalloc((void **)&b.cflag, NC * sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(b.ccur, 0, NC * sizeof(uint64_t), e->xstream));
            HIPCHK(hipMemsetAsync(b.cflag, 0, NC * sizeof(uint32_t), e->xstream));
        }
        if (e->narrow) {
            HIPCHK(hipMalloc((void **)&b.store32, M * LH_ROW_STRIDE * sizeof(uint32_t)));
            HIPCHK(hipMemsetAsync(b.store32, 0, M * LH_ROW_STRIDE * sizeof(uint32_t), e->xstream));
            e->c_store_bytes.fetch_add(M * LH_ROW_STRIDE * sizeof(uint32_t), std::memory_order_relaxed);
            b.counts = lh::cells_tagged(b.store32, 4);
        } else {
            HIPCHK(hipMalloc((void **)&b.store64, M * LH_ROW_STRIDE * sizeof(uint64_t)));
            HIPCHK(hipMemsetAsync(b.store64, 0, M * LH_ROW_STRIDE * sizeof(uint64_t), e->xstream));
            e->c_store_bytes.fetch_add(M * LH_ROW_STRIDE * sizeof(uint64_t), std::memory_order_relaxed);
            b.counts = b.store64;
        }
        HIPCHK(hipMalloc((void **)&b.ranges, M * 2 * sizeof(uint32_t)));
        HIPCHK(lh::launch_init_ranges(b.ranges, (uint32_t)M, e->xstream));
        HIPCHK(hipEventCreateWithFlags(&b.cleared, hipEventDisableTiming));
        HIPCHK(hipEventRecord(b.cleared, e->xstream));
        b.state = BUF_FREE;
    }
    e->cur = 0;
    e->bufs[0].state = BUF_CURRENT;

    for (uint32_t i = 0; i < e->cfg.num_lanes; i++) {
        std::unique_ptr<Lane> ln(new (std::nothrow) Lane());
        if (!ln) return LH_ENOMEM;
        HIPCHK(hipStreamCreateWithFlags(&ln->stream, hipStreamNonBlocking));
        for (int h = 0; h < 2; h++) {
            HIPCHK(hipHostMalloc((void **)&ln->h_vals[h], e->cfg.lane_samples * sizeof(double), hipHostMallocDefault));
            HIPCHK(hipHostMalloc((void **)&ln->h_ids[h], e->cfg.lane_samples * sizeof(uint32_t), hipHostMallocDefault));
            HIPCHK(hipMalloc((void **)&ln->d_vals[h], e->cfg.lane_samples * sizeof(double)));
            HIPCHK(hipMalloc((void **)&ln->d_ids[h], e->cfg.lane_samples * sizeof(uint32_t)));
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, ln->h_vals[h], 0) == hipSuccess) ln->m_vals[h] = static_cast<double *>(dp);
            if (hipHostGetDevicePointer(&dp, ln->h_ids[h], 0) == hipSuccess) ln->m_ids[h] = static_cast<uint32_t *>(dp);
            (void)hipGetLastError(); // (not mapped: the lane copies, as before)
            HIPCHK(hipEventCreateWithFlags(&ln->done[h], hipEventDisableTiming));
        }
        e->lanes.push_back(std::move(ln));
    }
    HIPCHK(hipStreamSynchronize(e->xstream));
    return LH_OK;
}

} // namespace

extern "C" {

int lh_abi_version(void) { return LH_ABI_VERSION; }

const char *lh_strerror(int code)
{
    switch (code) {
    case LH_OK: return "ok";
    case LH_EINVAL: return "invalid argument";
    case LH_ENOMEM: return "out of memory";
    case LH_EDEVICE: return "HIP runtime error";
    case LH_ENODEVICE: return "no usable gfx950 device";
    case LH_EBUSY: return "no free epoch buffer (release a snapshot first)";
    case LH_ERANGE: return "metric id out of range";
    case LH_ESTATE: return "invalid state for this call";
    default: return "unknown error";
    }
}

const char *lh_last_error(void) { return g_last_error; }

int lh_default_config(lh_config *cfg)
{
    if (!cfg) return LH_EINVAL;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(lh_config);
    cfg->device = 0;
    cfg->max_metrics = 1024;
    cfg->num_buffers = 2;
    cfg->num_lanes = 4;
    cfg->max_counters = 1024;
    cfg->lane_samples = 1u << 20; // 8 MiB of float64 per half-buffer
    return LH_OK;
}

int lh_create(const lh_config *cfg, lh_engine **out)
{
    if (!cfg || !out) return LH_EINVAL;
    *out = nullptr;
    // (ABI <= 6 callers pass the struct without cell_bits: the default width)
    constexpr uint32_t kConfigV6 = (uint32_t)offsetof(lh_config, cell_bits);
    if (cfg->struct_size != sizeof(lh_config) && cfg->struct_size != kConfigV6) return LH_EINVAL;
    const uint32_t cell_bits = cfg->struct_size == sizeof(lh_config) ? cfg->cell_bits : 0u;
    if (cell_bits != 0 && cell_bits != 32 && cell_bits != 64) return LH_EINVAL;
    if (cfg->max_metrics == 0 || cfg->num_buffers < 2 || cfg->num_buffers > 16 || cfg->num_lanes == 0 ||
        cfg->num_lanes > 256 || cfg->lane_samples < 2 || cfg->lane_samples > (1ull << 28) ||
        cfg->max_counters > (1u << 24))
        return LH_EINVAL;
    lh_engine *e = new (std::nothrow) lh_engine();
    if (!e) return LH_ENOMEM;
    std::memcpy(&e->cfg, cfg, cfg->struct_size);
    e->cfg.struct_size = (uint32_t)sizeof(lh_config);
    e->cfg.cell_bits = cell_bits;
    int rc = create_impl(cfg, e);
    if (rc != LH_OK) {
        free_engine(e);
        return rc;
    }
    *out = e;
    return LH_OK;
}

int lh_destroy(lh_engine *e)
{
    if (!e) return LH_EINVAL;
    if (e->live_snapshots.load() != 0) return LH_ESTATE;
    free_engine(e);
    return LH_OK;
}

int lh_intern(lh_engine *e, const char *name, size_t len, uint32_t *id)
{
    if (!e || (!name && len) || !id) return LH_EINVAL;
    std::string key(name ? name : "", len);
    {
        std::shared_lock<std::shared_mutex> g(e->names_mu);
        auto it = e->name2id.find(key);
        if (it != e->name2id.end()) { *id = it->second; return LH_OK; }
    }
    std::unique_lock<std::shared_mutex> g(e->names_mu);
    auto it = e->name2id.find(key);
    if (it != e->name2id.end()) { *id = it->second; return LH_OK; }
    if (e->names.size() >= e->cfg.max_metrics) return LH_ERANGE;
    if (e->name_blob.size() + key.size() > (size_t(1) << 31)) return LH_ERANGE; // uint32 offsets in HBM (K6)
    const uint32_t nid = (uint32_t)e->names.size();
    e->names.push_back(key);
    e->name_blob += key;
    e->name_off.push_back((uint32_t)e->name_blob.size());
    e->name2id.emplace(std::move(key), nid);
    *id = nid;
    return LH_OK;
}

int lh_lookup(lh_engine *e, const char *name, size_t len, uint32_t *id)
{
    if (!e || (!name && len) || !id) return LH_EINVAL;
    std::string key(name ? name : "", len);
    std::shared_lock<std::shared_mutex> g(e->names_mu);
    auto it = e->name2id.find(key);
    if (it == e->name2id.end()) return LH_ERANGE;
    *id = it->second;
    return LH_OK;
}

int lh_num_metrics(lh_engine *e, uint32_t *n)
{
    if (!e || !n) return LH_EINVAL;
    std::shared_lock<std::shared_mutex> g(e->names_mu);
    *n = (uint32_t)e->names.size();
    return LH_OK;
}

int lh_metric_name(lh_engine *e, uint32_t id, char *buf, size_t cap, size_t *len)
{
    if (!e || !len || (!buf && cap)) return LH_EINVAL;
    std::shared_lock<std::shared_mutex> g(e->names_mu);
    if (id >= e->names.size()) return LH_ERANGE;
    const std::string &s = e->names[id];
    *len = s.size();
    if (cap) std::memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
    return LH_OK;
}

int lh_submit(lh_engine *e, uint32_t id, const double *v, size_t n)
{
    if (!e || (!v && n)) return LH_EINVAL;
    if (id >= e->cfg.max_metrics) return LH_ERANGE;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    Lane &ln = pick_lane(e);
    LaneLock g(ln);
    if (ln.fill && (ln.mode != LANE_SINGLE || ln.single_id != id)) {
        rc = lane_launch(e, ln);
        if (rc) return rc;
    }
    const size_t cap = (size_t)e->cfg.lane_samples;
    while (n) {
        ln.mode = LANE_SINGLE;
        ln.single_id = id;
        const size_t take = (cap - ln.fill) < n ? (cap - ln.fill) : n;
        std::memcpy(ln.h_vals[ln.cur] + ln.fill, v, take * sizeof(double));
        ln.fill += take;
        v += take;
        n -= take;
        if (ln.fill == cap) {
            rc = lane_launch(e, ln);
            if (rc) return rc;
        }
    }
    return LH_OK;
}

// lh_submit_pairs / lh_submit_pairs16: the caller's batch is copied into the lane's pinned half-buffers
extern "C++" {
template <typename IDT>
static int submit_pairs_t(lh_engine *e, const IDT *ids, const double *v, size_t n)
{
    if (!e || ((!v || !ids) && n)) return LH_EINVAL;
    if (n == 0) return LH_OK;
    constexpr LaneMode MODE = sizeof(IDT) == 2 ? LANE_PAIRS16 : LANE_PAIRS;
    if (sizeof(IDT) == 4 || e->cfg.max_metrics < 65536u)
        for (size_t i = 0; i < n; i++)
            if (ids[i] >= e->cfg.max_metrics) return LH_ERANGE;
    int rc = use_device(e);
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    Lane &ln = pick_lane(e);
    LaneLock g(ln);
    if (ln.fill && ln.mode != MODE) {
        rc = lane_launch(e, ln);
        if (rc) return rc;
    }
    const size_t cap = (size_t)e->cfg.lane_samples;
    while (n) {
        ln.mode = MODE;
        const size_t take = (cap - ln.fill) < n ? (cap - ln.fill) : n;
        std::memcpy(ln.h_vals[ln.cur] + ln.fill, v, take * sizeof(double));
        std::memcpy(reinterpret_cast<IDT *>(ln.h_ids[ln.cur]) + ln.fill, ids, take * sizeof(IDT));
        ln.fill += take;
        v += take;
        ids += take;
        n -= take;
        if (ln.fill == cap) {
            rc = lane_launch(e, ln);
            if (rc) return rc;
        }
    }
    return LH_OK;
}

} // extern "C++"

int lh_submit_pairs(lh_engine *e, const uint32_t *ids, const double *v, size_t n) { return submit_pairs_t(e, ids, v, n); }

// (uint16 ids are valid for ANY engine with at least 65 536 names, not only for one with exactly that many: the four
// narrow entry points agree -- ADVICE r4; below 65 536 names submit_pairs_t checks every id)
int lh_submit_pairs16(lh_engine *e, const uint16_t *ids, const double *v, size_t n) { return submit_pairs_t(e, ids, v, n); }

// In-place staging (SURVEY.md 8b "Ownership": "or the ring is C-allocated (hipHostMalloc) and Go writes into it in
// place"): the producer gets the free tail of a pinned half-buffer, writes its (id, value) pairs there -- the one
// and only host-side store of a sample -- and commits how many it wrote.  Between the two calls the lane belongs to
// the caller: other producers that hash to it and the flush at the flip wait for the commit.
extern "C++" {
template <typename IDT>
static int reserve_pairs_t(lh_engine *e, size_t want, IDT **ids, double **vals, size_t *granted, uint32_t *token)
{
    if (!e || !ids || !vals || !granted || !token || want == 0) return LH_EINVAL;
    constexpr LaneMode MODE = sizeof(IDT) == 2 ? LANE_PAIRS16 : LANE_PAIRS;
    int rc = use_device(e);
    if (rc) return rc;
    // lock order as everywhere: epoch (shared) before lane.  lh_commit_pairs takes no epoch lock at all, so a flip that
    // holds the epoch exclusively and waits for this lane's reservation to end cannot deadlock with the commit.
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    // the caller's own lane if it is free, else the next free one, else wait for the caller's own
    const size_t nl = e->lanes.size();
    const size_t h0 = std::hash<std::thread::id>()(std::this_thread::get_id()) % nl;
    Lane *lane = nullptr;
    std::unique_lock<std::mutex> g;
    for (size_t k = 0; k < nl && !lane; k++) {
        Lane &c = *e->lanes[(h0 + k) % nl];
        std::unique_lock<std::mutex> t(c.mu, std::try_to_lock);
        if (t.owns_lock() && !c.reserved) { lane = &c; g = std::move(t); }
    }
    if (!lane) {
        lane = e->lanes[h0].get();
        g = std::unique_lock<std::mutex>(lane->mu);
        lane->cv.wait(g, [&] { return !lane->reserved; });
    }
    Lane &ln = *lane;
    const size_t cap = (size_t)e->cfg.lane_samples;
    if (ln.fill && (ln.mode != MODE || ln.fill == cap)) {
        rc = lane_launch(e, ln);
        if (rc) return rc;
    }
    ln.mode = MODE;
    ln.reserved = true;
    ln.granted = cap - ln.fill < want ? cap - ln.fill : want;
    *ids = reinterpret_cast<IDT *>(ln.h_ids[ln.cur]) + ln.fill;
    *vals = ln.h_vals[ln.cur] + ln.fill;
    *granted = ln.granted;
    *token = 0;
    for (size_t k = 0; k < nl; k++)
        if (e->lanes[k].get() == lane) *token = (uint32_t)k + 1u;
    return LH_OK;
}

} // extern "C++"

int lh_reserve_pairs(lh_engine *e, size_t want, uint32_t **ids, double **vals, size_t *granted, uint32_t *token)
{
    return reserve_pairs_t(e, want, ids, vals, granted, token);
}

int lh_reserve_pairs16(lh_engine *e, size_t want, uint16_t **ids, double **vals, size_t *granted, uint32_t *token)
{
    return reserve_pairs_t(e, want, ids, vals, granted, token);
}

int lh_commit_pairs(lh_engine *e, uint32_t token, size_t n)
{
    if (!e || token == 0 || token > e->lanes.size()) return LH_EINVAL;
    Lane &ln = *e->lanes[token - 1u];
    int rc = LH_OK;
    {
        std::lock_guard<std::mutex> g(ln.mu);
        if (!ln.reserved) return LH_ESTATE;
        // more than was granted: nothing is published, but the reservation ENDS (a lane left reserved would block
        // every later flip, sync and submit on it for good) -- the caller's pairs are not ingested
        const bool bad = n > ln.granted;
        if (!bad) ln.fill += n; // a half-buffer this fills is launched by the next call that takes the lane (reserve, submit, flip)
        ln.reserved = false;
        ln.granted = 0;
        if (ln.fill == 0) ln.mode = LANE_NONE;
        rc = bad ? LH_ESTATE : LH_OK;
    }
    ln.cv.notify_all();
    return rc;
}

int lh_commit_pairs16(lh_engine *e, uint32_t token, size_t n) { return lh_commit_pairs(e, token, n); }

int lh_submit_device(lh_engine *e, uint32_t id, const double *d_v, size_t n, void *stream)
{
    if (!e || (!d_v && n)) return LH_EINVAL;
    if (id >= e->cfg.max_metrics) return LH_ERANGE;
    if (((uintptr_t)d_v & 7) != 0) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : e->main_stream;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    rc = epoch_touch_stream(e, s);
    if (rc) return rc;
    return launch_single(e, id, d_v, n, s);
}

int lh_submit_pairs_device(lh_engine *e, const uint32_t *d_ids, const double *d_v, size_t n, void *stream)
{
    if (!e || ((!d_v || !d_ids) && n)) return LH_EINVAL;
    if (((uintptr_t)d_v & 7) != 0 || ((uintptr_t)d_ids & 3) != 0) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : e->main_stream;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    rc = epoch_touch_stream(e, s);
    if (rc) return rc;
    return launch_pairs(e, d_ids, d_v, n, s);
}

int lh_submit_pairs16_device(lh_engine *e, const uint16_t *d_ids, const double *d_v, size_t n, void *stream)
{
    if (!e || ((!d_v || !d_ids) && n)) return LH_EINVAL;
    if (((uintptr_t)d_v & 7) != 0 || ((uintptr_t)d_ids & 1) != 0) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : e->main_stream;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    rc = epoch_touch_stream(e, s);
    if (rc) return rc;
    return launch_pairs(e, d_ids, d_v, n, s);
}

int lh_intern_counter(lh_engine *e, const char *name, size_t len, uint32_t *id)
{
    if (!e || (!name && len) || !id) return LH_EINVAL;
    std::string key(name ? name : "", len);
    {
        std::shared_lock<std::shared_mutex> g(e->cnames_mu);
        auto it = e->cname2id.find(key);
        if (it != e->cname2id.end()) { *id = it->second; return LH_OK; }
    }
    std::unique_lock<std::shared_mutex> g(e->cnames_mu);
    auto it = e->cname2id.find(key);
    if (it != e->cname2id.end()) { *id = it->second; return LH_OK; }
    if (e->cnames.size() >= e->cfg.max_counters) return LH_ERANGE;
    if (e->cname_blob.size() + key.size() > (size_t(1) << 31)) return LH_ERANGE;
    const uint32_t nid = (uint32_t)e->cnames.size();
    e->cnames.push_back(key);
    e->cname_blob += key;
    e->cname_off.push_back((uint32_t)e->cname_blob.size());
    e->cname2id.emplace(std::move(key), nid);
    *id = nid;
    return LH_OK;
}

int lh_num_counters(lh_engine *e, uint32_t *n)
{
    if (!e || !n) return LH_EINVAL;
    std::shared_lock<std::shared_mutex> g(e->cnames_mu);
    *n = (uint32_t)e->cnames.size();
    return LH_OK;
}

int lh_counter_name(lh_engine *e, uint32_t id, char *buf, size_t cap, size_t *len)
{
    if (!e || !len || (!buf && cap)) return LH_EINVAL;
    std::shared_lock<std::shared_mutex> g(e->cnames_mu);
    if (id >= e->cnames.size()) return LH_ERANGE;
    const std::string &s = e->cnames[id];
    *len = s.size();
    if (cap) std::memcpy(buf, s.data(), s.size() < cap ? s.size() : cap);
    return LH_OK;
}

int lh_submit_counts(lh_engine *e, const uint32_t *ids, const uint64_t *amounts, size_t n)
{
    if (!e || ((!ids || !amounts) && n)) return LH_EINVAL;
    if (n == 0) return LH_OK;
    for (size_t i = 0; i < n; i++)
        if (ids[i] >= e->cfg.max_counters) return LH_ERANGE;
    int rc = use_device(e);
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    Lane &ln = pick_lane(e);
    LaneLock g(ln);
    if (ln.fill && ln.mode != LANE_COUNTS) {
        rc = lane_launch(e, ln);
        if (rc) return rc;
    }
    const size_t cap = (size_t)e->cfg.lane_samples;
    while (n) {
        ln.mode = LANE_COUNTS;
        const size_t take = (cap - ln.fill) < n ? (cap - ln.fill) : n;
        std::memcpy(ln.h_vals[ln.cur] + ln.fill, amounts, take * sizeof(uint64_t)); // same 8-byte slots as float64
        std::memcpy(ln.h_ids[ln.cur] + ln.fill, ids, take * sizeof(uint32_t));
        ln.fill += take;
        amounts += take;
        ids += take;
        n -= take;
        if (ln.fill == cap) {
            rc = lane_launch(e, ln);
            if (rc) return rc;
        }
    }
    return LH_OK;
}

int lh_submit_counts_device(lh_engine *e, const uint32_t *d_ids, const uint64_t *d_amounts, size_t n, void *stream)
{
    if (!e || ((!d_ids || !d_amounts) && n)) return LH_EINVAL;
    if (((uintptr_t)d_amounts & 7) != 0 || ((uintptr_t)d_ids & 3) != 0) return LH_EINVAL;
    if (e->cfg.max_counters == 0) return LH_ERANGE;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : e->main_stream;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    rc = epoch_touch_stream(e, s);
    if (rc) return rc;
    return launch_counts(e, d_ids, d_amounts, n, s);
}

int lh_flush(lh_engine *e)
{
    if (!e) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
    return flush_all_lanes(e);
}

int lh_sync(lh_engine *e)
{
    if (!e) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    std::vector<hipStream_t> ss;
    {
        std::shared_lock<std::shared_mutex> eg(e->epoch_mu);
        rc = flush_all_lanes(e);
        if (rc) return rc;
        std::lock_guard<std::mutex> g(e->streams_mu);
        ss = e->epoch_streams;
    }
    for (hipStream_t s : ss) HIPCHK(hipStreamSynchronize(s));
    uint32_t err = 0;
    HIPCHK(hipMemcpy(&err, e->d_err, sizeof(err), hipMemcpyDeviceToHost));
    if (err) {
        HIPCHK(hipMemset(e->d_err, 0, 8));
        return LH_ERANGE;
    }
    return LH_OK;
}

int lh_flip(lh_engine *e, lh_snapshot **out)
{
    if (!e || !out) return LH_EINVAL;
    *out = nullptr;
    int rc = use_device(e);
    if (rc) return rc;
    std::unique_lock<std::shared_mutex> eg(e->epoch_mu);
    int next = -1;
    for (size_t i = 0; i < e->bufs.size(); i++)
        if (e->bufs[i].state == BUF_FREE) { next = (int)i; break; }
    if (next < 0) {
        e->c_busy.fetch_add(1, std::memory_order_relaxed);
        return LH_EBUSY;
    }
    lh_snapshot *s = new (std::nothrow) lh_snapshot();
    if (!s) return LH_ENOMEM;
    rc = flush_all_lanes(e);
    if (rc) { delete s; return rc; }
    {
        std::lock_guard<std::mutex> g(e->streams_mu);
        while (e->flip_events.size() < e->epoch_streams.size()) {
            hipEvent_t ev;
            hipError_t he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (he != hipSuccess) { set_last_error("hipEventCreateWithFlags", he); delete s; return LH_EDEVICE; }
            e->flip_events.push_back(ev);
        }
        for (size_t i = 0; i < e->epoch_streams.size(); i++) {
            hipError_t he = hipEventRecord(e->flip_events[i], e->epoch_streams[i]);
            if (he == hipSuccess) he = hipStreamWaitEvent(e->xstream, e->flip_events[i], 0);
            if (he != hipSuccess) { set_last_error("flip event chain", he); delete s; return LH_EDEVICE; }
        }
        e->epoch_streams.clear();
    }
    s->e = e;
    s->buf = e->cur;
    e->bufs[(size_t)e->cur].state = BUF_SNAPSHOT;
    e->cur = next;
    e->bufs[(size_t)next].state = BUF_CURRENT;
    e->live_snapshots.fetch_add(1);
    e->c_flips.fetch_add(1, std::memory_order_relaxed);
    {
        // Region overflows of the launches that have completed so far (the kernels add to pinned memory).  A
        // launch's overflows can arrive one flip after its samples were counted, so both are accumulated and the ratio
        // is judged only over windows of at least 2^21 region-path samples: an idle interval that merely collects a
        // late count must not switch the path.
        const uint64_t now = __atomic_load_n(e->h_rstat, __ATOMIC_RELAXED) + __atomic_load_n(e->h_rstat + kLaneRstat, __ATOMIC_RELAXED);
        const uint64_t ov = now - e->rstat_seen;
        e->rstat_seen = now;
        if (ov) e->c_region_ovf.fetch_add(ov, std::memory_order_relaxed);
        e->win_ovf += ov;
        e->win_samples += e->region_samples.exchange(0);
        if (e->win_samples >= (uint64_t(1) << 21)) {
            if (e->win_ovf * 50 > e->win_samples) {
                e->regions_disabled.store(true);
                e->flips_since_regions_off = 0;
            }
            e->win_ovf = 0;
            e->win_samples = 0;
        }
        if (e->v3_disabled.load(std::memory_order_relaxed) && ++e->flips_since_v3_off > 64) {
            e->flips_since_v3_off = 0;
            e->v3_disabled.store(false);
        }
        if (e->regions_disabled.load(std::memory_order_relaxed) && ++e->flips_since_regions_off > 64) {
            e->flips_since_regions_off = 0;
            e->regions_disabled.store(false);
        }
    }
    // adaptive dispatch is not one-way: the few-name single-pass kernel gets another interval every 64 flips
    // (its window-miss counter turns it off again if the stream is still too wide for it)
    if (e->small_disabled.load(std::memory_order_relaxed) && !e->small_forced_off) {
        if (++e->flips_since_small_off >= 64) {
            e->flips_since_small_off = 0;
            e->small_samples.store(0);
            e->small_disabled.store(false);
        }
    }
    *out = s;
    return LH_OK;
}

static int after_extract(lh_engine *e, uint32_t err, uint32_t nfall);

int lh_extract(lh_snapshot *s, const double *p, size_t np, lh_stats *stats, double *pvals, int16_t *pkeys,
               uint8_t *pvalid, size_t nmetrics)
{
    return lh_extract_rows(s, 0, nmetrics, p, np, stats, pvals, pkeys, pvalid);
}

namespace {
// results on the host: copied into the caller's arrays, or (view) handed out in place
int finish_extract(lh_engine *e, const ExtractLayout &L, size_t nmetrics, size_t np, lh_stats *stats, double *pvals,
                   int16_t *pkeys, uint8_t *pvalid, lh_extract_view *view)
{
    if (view) {
        view->stats = reinterpret_cast<const lh_stats *>(e->h_xbuf + L.off_stats);
        view->pvals = reinterpret_cast<const double *>(e->h_xbuf + L.off_pvals);
        view->pkeys = reinterpret_cast<const int16_t *>(e->h_xbuf + L.off_pkeys);
        view->pvalid = e->h_xbuf + L.off_pvalid;
        view->nmetrics = nmetrics;
        view->np = np;
    } else {
        std::memcpy(stats, e->h_xbuf + L.off_stats, nmetrics * sizeof(lh_stats));
        if (np) {
            std::memcpy(pvals, e->h_xbuf + L.off_pvals, nmetrics * np * sizeof(double));
            if (pkeys) std::memcpy(pkeys, e->h_xbuf + L.off_pkeys, nmetrics * np * sizeof(int16_t));
            if (pvalid) std::memcpy(pvalid, e->h_xbuf + L.off_pvalid, nmetrics * np);
        }
    }
    uint32_t err, nfall;
    std::memcpy(&err, e->h_xbuf + L.off_err, 4);
    std::memcpy(&nfall, e->h_xbuf + L.off_err + 4, 4);
    return after_extract(e, err, nfall);
}

int extract_impl(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np, lh_stats *stats,
                 double *pvals, int16_t *pkeys, uint8_t *pvalid, lh_extract_view *view, lh_extract_compact *cview = nullptr);

// uint64(float64) as Go compiles it for amd64 (metrics.go:374): the host-side twin of the kernels'
// d_f64_to_u64_amd64, for lh_expand_compact
uint64_t f64_to_u64_amd64(double f)
{
    const double two63 = 9223372036854775808.0;
    if (f != f) return 0x8000000000000000ull;
    if (f < two63) {
        if (f <= -two63) return 0x8000000000000000ull;
        return (uint64_t)(long long)f;
    }
    const double g = f - two63;
    if (g >= two63) return 0;
    return (uint64_t)(long long)g ^ 0x8000000000000000ull;
}
} // namespace

int lh_extract_rows(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np, lh_stats *stats,
                    double *pvals, int16_t *pkeys, uint8_t *pvalid)
{
    if (!s || (nmetrics && !stats) || (np && nmetrics && (!p || !pvals))) return LH_EINVAL;
    return extract_impl(s, first, nmetrics, p, np, stats, pvals, pkeys, pvalid, nullptr);
}

int lh_extract_rows_view(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np,
                         lh_extract_view *view)
{
    if (!s || !view || (np && nmetrics && !p)) return LH_EINVAL;
    std::memset(view, 0, sizeof(*view));
    return extract_impl(s, first, nmetrics, p, np, nullptr, nullptr, nullptr, nullptr, view);
}

int lh_extract_rows_compact(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np,
                            lh_extract_compact *view)
{
    if (!s || !view || (np && nmetrics && !p)) return LH_EINVAL;
    std::memset(view, 0, sizeof(*view));
    return extract_impl(s, first, nmetrics, p, np, nullptr, nullptr, nullptr, nullptr, nullptr, view);
}

// The full form from the compact one, on the host: every field is a function of count, sum and the keys alone
// (processHistograms, metrics.go:349-356, 374, 378-385), evaluated with the operations the kernels use -- an IEEE
// divide, the amd64 conversion, a read of the device-generated table D[] -- so the results equal lh_extract_rows' bit
// for bit (tests/test_gpu_extract_compact.py).
int lh_expand_compact(lh_engine *e, const lh_extract_compact *c, lh_stats *stats, double *pvals, int16_t *pkeys,
                      uint8_t *pvalid)
{
    if (!e || !c || (c->nmetrics && (!c->count || !c->sum || !c->nbuckets))) return LH_EINVAL;
    if (c->np > LH_MAX_PERCENTILES || (c->np && c->nmetrics && (!c->pkeys || !c->pvalid_bits))) return LH_EINVAL;
    const double *D = nullptr;
    if (pvals && c->np) {
        std::lock_guard<std::mutex> g(e->hD_mu);
        if (e->h_D.empty()) {
            int rc = use_device(e);
            if (rc) return rc;
            e->h_D.resize(LH_NKEYS);
            hipError_t he = hipMemcpy(e->h_D.data(), e->d_D, sizeof(double) * LH_NKEYS, hipMemcpyDeviceToHost);
            if (he != hipSuccess) { e->h_D.clear(); set_last_error("hipMemcpy(D)", he); return LH_EDEVICE; }
        }
        D = e->h_D.data();
    }
    const size_t n = c->nmetrics, np = c->np;
    for (size_t m = 0; m < n; m++) {
        if (stats) {
            lh_stats &o = stats[m];
            o.count = c->count[m];
            o.sum = c->sum[m];
            o.avg = c->sum[m] / (double)c->count[m]; // metrics.go:356 (0/0 = NaN when empty)
            o.agg_sum_add = f64_to_u64_amd64(c->sum[m]);
            o.nbuckets = c->nbuckets[m];
            o.present = c->count[m] ? 1u : 0u;
        }
        const uint32_t bits = np ? c->pvalid_bits[m] : 0u;
        for (size_t i = 0; i < np; i++) {
            const bool ok = (bits >> i) & 1u;
            const int16_t key = c->pkeys[m * np + i];
            if (pvals) pvals[m * np + i] = ok ? D[(uint16_t)key ^ 0x8000u] : 0.0; // D is indexed by the dense bin
            if (pkeys) pkeys[m * np + i] = ok ? key : (int16_t)0;
            if (pvalid) pvalid[m * np + i] = ok ? 1 : 0;
        }
    }
    return LH_OK;
}

// Measurement helper (loghisto_gpu_tuning.h): device time of the engine's last LARGE extract (results through HBM and one
// copy: more than 32 KiB), the kernel and the device-to-host copy apart.
int lh_tool_last_extract_ms(lh_engine *e, float *kernel_ms, float *copy_ms)
{
    if (!e || !kernel_ms || !copy_ms) return LH_EINVAL;
    std::lock_guard<std::mutex> g(e->xmu);
    if (!e->xev_valid) return LH_ESTATE;
    int rc = use_device(e);
    if (rc) return rc;
    HIPCHK(hipEventElapsedTime(kernel_ms, e->xev[0], e->xev[1]));
    HIPCHK(hipEventElapsedTime(copy_ms, e->xev[1], e->xev[2]));
    return LH_OK;
}

namespace {
int extract_impl(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np, lh_stats *stats,
                 double *pvals, int16_t *pkeys, uint8_t *pvalid, lh_extract_view *view, lh_extract_compact *cview)
{
    lh_engine *e = s->e;
    if (np > LH_MAX_PERCENTILES || (uint64_t)first + nmetrics > e->cfg.max_metrics) return LH_EINVAL;
    if (nmetrics == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    static_assert(sizeof(lh::ExtractOut) == sizeof(lh_stats), "lh_stats layout");
    std::lock_guard<std::mutex> g(e->xmu);
    const ExtractLayout L = cview ? extract_layout_compact(nmetrics, np) : extract_layout(nmetrics, np);
    rc = ensure_xbuf(e, L.total + 16);
    if (rc) return rc;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    // Small results (the latency path: one or a few metrics) are written by the kernel straight into the pinned
    // host block through its device mapping: one launch + one sync, no separate copy.  Large ones go through
    // HBM and one DMA (fine-grained stores over PCIe would be slower than the copy engine); from 2 048 names on the
    // scan itself is the wave-per-metric kernel (65 536 names: ~290 -> ~60 us), so that the 9 MB transfer is what
    // remains -- measured: cutting it into chunks pipelined behind the kernels gained nothing over one copy (round 2, on
    // this stream; round 5, the copies on a stream of their own: 288 us for one block, 321 / 377 / 1 635 us for 2 / 4 / 8 --
    // the copies do not overlap the kernels and every extra hipMemcpyAsync costs ~10 us, profiles/r05_extract_wave.txt), and
    // lh_extract_rows_view hands the pinned block out in place instead of copying it once more on the host.
    const bool zero_copy = e->d_hxbuf != nullptr && L.total <= e->zero_copy_max && e->zero_copy_enabled;
    unsigned char *xb = zero_copy ? e->d_hxbuf : e->d_xbuf;
    lh::ExtractNotify nt;
    volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(e->h_xbuf + L.total);
    if (zero_copy) {
        if (++e->xseq == 0) e->xseq = 1;
        *flag = 0;
        nt.done_ctr = e->d_err + 2;
        nt.host_flag = reinterpret_cast<uint32_t *>(e->d_hxbuf + L.total);
        nt.seq = e->xseq;
    }
    if (!zero_copy) { // (the small results' latency path records nothing)
        for (hipEvent_t &ev : e->xev)
            if (!ev) HIPCHK(hipEventCreate(&ev));
        e->xev_valid = false;
        HIPCHK(hipEventRecord(e->xev[0], e->xstream));
    }
    lh::ExtractCompact cx;
    if (cview) {
        cx.count = reinterpret_cast<uint64_t *>(xb + L.off_count);
        cx.sum = reinterpret_cast<double *>(xb + L.off_sum);
        cx.nbuckets = reinterpret_cast<uint32_t *>(xb + L.off_nb);
        cx.vbits = reinterpret_cast<uint32_t *>(xb + L.off_vbits);
        cx.pkeys = reinterpret_cast<int16_t *>(xb + L.off_pkeys);
    }
    HIPCHK(lh::launch_extract(lh::cells_at(b.counts, (size_t)first * LH_ROW_STRIDE), b.ranges + 2 * (size_t)first,
                              (uint32_t)nmetrics, p, (uint32_t)np, e->d_D,
                              reinterpret_cast<lh::ExtractOut *>(xb + L.off_stats),
                              reinterpret_cast<double *>(xb + L.off_pvals),
                              reinterpret_cast<int16_t *>(xb + L.off_pkeys), xb + L.off_pvalid,
                              e->d_err, reinterpret_cast<uint32_t *>(xb + L.off_err), e->xstream, nt, cx));
    if (zero_copy) {
        // spin on the completion word the last workgroup stores after its system-scope release: no driver call
        // on the latency path; after 2 ms (a flip queued behind long ingest kernels) fall back to the stream
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (*flag != nt.seq) {
            __builtin_ia32_pause();
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIPCHK(hipStreamSynchronize(e->xstream));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        HIPCHK(hipEventRecord(e->xev[1], e->xstream));
        HIPCHK(hipMemcpyAsync(e->h_xbuf, e->d_xbuf, L.total, hipMemcpyDeviceToHost, e->xstream));
        HIPCHK(hipEventRecord(e->xev[2], e->xstream));
        HIPCHK(hipStreamSynchronize(e->xstream));
        e->xev_valid = true;
    }
    if (cview) {
        cview->count = reinterpret_cast<const uint64_t *>(e->h_xbuf + L.off_count);
        cview->sum = reinterpret_cast<const double *>(e->h_xbuf + L.off_sum);
        cview->nbuckets = reinterpret_cast<const uint32_t *>(e->h_xbuf + L.off_nb);
        cview->pvalid_bits = reinterpret_cast<const uint32_t *>(e->h_xbuf + L.off_vbits);
        cview->pkeys = reinterpret_cast<const int16_t *>(e->h_xbuf + L.off_pkeys);
        cview->nmetrics = nmetrics;
        cview->np = np;
        uint32_t err, nfall;
        std::memcpy(&err, e->h_xbuf + L.off_err, 4);
        std::memcpy(&nfall, e->h_xbuf + L.off_err + 4, 4);
        return after_extract(e, err, nfall);
    }
    return finish_extract(e, L, nmetrics, np, stats, pvals, pkeys, pvalid, view);
}
} // namespace

// The sticky bad-id flag and the single-pass kernel's window-miss counter ride along with every extract.
static int after_extract(lh_engine *e, uint32_t err, uint32_t nfall)
{
    e->c_extracts.fetch_add(1, std::memory_order_relaxed);
    if (nfall) {
        e->c_misses.fetch_add(nfall, std::memory_order_relaxed);
        const uint64_t seen = e->small_samples.exchange(0);
        // more than 2 % of the samples fell outside the LDS windows: the names' spans are wider than
        // 16384/names bins; the partitioned path gives every name 4 096 bins
        if ((uint64_t)nfall * 50 > seen && e->cfg.max_metrics > 4) e->small_disabled.store(true);
        HIPCHK(hipMemsetAsync(reinterpret_cast<uint32_t *>(e->d_err) + 1, 0, 4, e->xstream));
    }
    if (err) {
        HIPCHK(hipMemsetAsync(e->d_err, 0, 4, e->xstream));
        return LH_ERANGE;
    }
    return LH_OK;
}

int lh_buckets(lh_snapshot *s, uint32_t id, int16_t *keys, uint64_t *counts, size_t cap, size_t *n)
{
    if (!s || !n || (cap && (!keys || !counts))) return LH_EINVAL;
    lh_engine *e = s->e;
    if (id >= e->cfg.max_metrics) return LH_ERANGE;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    uint32_t r[2];
    rc = ensure_xbuf(e, 64);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->h_xbuf, b.ranges + 2 * (size_t)id, sizeof(r), hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    std::memcpy(r, e->h_xbuf, sizeof(r));
    *n = 0;
    if (r[0] > r[1]) return LH_OK;
    const size_t span = (size_t)r[1] - r[0] + 1;
    rc = ensure_xbuf(e, span * sizeof(uint64_t));
    if (rc) return rc;
    const size_t cb = lh::cell_bytes_of(b.counts);
    HIPCHK(hipMemcpyAsync(e->h_xbuf, lh::cells_base(lh::cells_at(b.counts, (size_t)id * LH_ROW_STRIDE + r[0])), span * cb,
                          hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    const uint64_t *row64 = reinterpret_cast<const uint64_t *>(e->h_xbuf);
    const uint32_t *row32 = reinterpret_cast<const uint32_t *>(e->h_xbuf);
    size_t k = 0;
    for (size_t i = 0; i < span; i++) {
        const uint64_t c = cb == 4 ? row32[i] : row64[i];
        if (!c) continue;
        if (k < cap) {
            keys[k] = (int16_t)(uint16_t)((r[0] + i) ^ 0x8000u);
            counts[k] = c;
        }
        k++;
    }
    *n = k;
    return LH_OK;
}

int lh_buckets_all(lh_snapshot *s, uint32_t first, size_t nmetrics, uint64_t *offsets, int16_t *keys,
                   uint64_t *counts, size_t cap, size_t *total)
{
    if (!s || !offsets || !total || (cap && (!keys || !counts))) return LH_EINVAL;
    lh_engine *e = s->e;
    if ((uint64_t)first + nmetrics > e->cfg.max_metrics) return LH_EINVAL;
    *total = 0;
    offsets[0] = 0;
    if (nmetrics == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    const uint64_t *rows = lh::cells_at(b.counts, (size_t)first * LH_ROW_STRIDE);
    const uint32_t *rng = b.ranges + 2 * (size_t)first;
    // pass 1: occupied cells per row
    const size_t off_bytes = (nmetrics + 1) * sizeof(uint64_t);
    rc = ensure_xbuf(e, off_bytes + nmetrics * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *d_nc = reinterpret_cast<uint32_t *>(e->d_xbuf + off_bytes);
    HIPCHK(lh::launch_count_cells(rows, rng, (uint32_t)nmetrics, d_nc, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + off_bytes, d_nc, nmetrics * sizeof(uint32_t), hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    const uint32_t *h_nc = reinterpret_cast<const uint32_t *>(e->h_xbuf + off_bytes);
    uint64_t run = 0;
    for (size_t i = 0; i < nmetrics; i++) {
        offsets[i] = run;
        run += h_nc[i];
    }
    offsets[nmetrics] = run;
    *total = (size_t)run;
    if (run == 0 || run > cap) return LH_OK; // caller sizes its arrays from *total and calls again
    // pass 2: compaction at the prefix offsets
    const size_t keys_off = (off_bytes + 15) & ~size_t(15);
    const size_t vals_off = (keys_off + run * sizeof(int16_t) + 15) & ~size_t(15);
    const size_t need = vals_off + run * sizeof(uint64_t);
    std::vector<uint64_t> keep(offsets, offsets + nmetrics + 1); // ensure_xbuf may reallocate the staging
    rc = ensure_xbuf(e, need);
    if (rc) return rc;
    std::memcpy(e->h_xbuf, keep.data(), off_bytes);
    HIPCHK(hipMemcpyAsync(e->d_xbuf, e->h_xbuf, off_bytes, hipMemcpyHostToDevice, e->xstream));
    HIPCHK(lh::launch_compact_cells(rows, rng, (uint32_t)nmetrics, reinterpret_cast<const uint64_t *>(e->d_xbuf),
                                    reinterpret_cast<int16_t *>(e->d_xbuf + keys_off),
                                    reinterpret_cast<uint64_t *>(e->d_xbuf + vals_off), e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + keys_off, e->d_xbuf + keys_off, need - keys_off, hipMemcpyDeviceToHost,
                          e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    std::memcpy(keys, e->h_xbuf + keys_off, run * sizeof(int16_t));
    std::memcpy(counts, e->h_xbuf + vals_off, run * sizeof(uint64_t));
    return LH_OK;
}

int lh_snapshot_rows(lh_snapshot *s, void **d_counts, uint32_t *nrows)
{
    if (!s || !d_counts) return LH_EINVAL;
    lh_engine *e = s->e;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    if (e->narrow) { // the caller indexes uint64 rows: the snapshot moves to its wide store first (lh_snapshot_cells does not)
        int rc = use_device(e);
        if (rc) return rc;
        std::lock_guard<std::mutex> g(e->xmu);
        const bool moves = lh::cells_narrow(b.counts);
        rc = widen_buffer(e, b, e->xstream, false);
        if (rc) return rc;
        // the view is the caller's from here on, on whatever stream: the move must have happened
        if (moves) HIPCHK(hipStreamSynchronize(e->xstream));
    }
    *d_counts = b.counts;
    if (nrows) *nrows = e->cfg.max_metrics;
    return LH_OK;
}

int lh_snapshot_cells(lh_snapshot *s, void **d_cells, uint32_t *nrows, uint32_t *cell_bytes)
{
    if (!s || !d_cells || !cell_bytes) return LH_EINVAL;
    lh_engine *e = s->e;
    std::lock_guard<std::mutex> g(e->xmu);
    const EpochBuffer &b = e->bufs[(size_t)s->buf];
    *d_cells = lh::cells_base(b.counts);
    *cell_bytes = lh::cell_bytes_of(b.counts);
    if (nrows) *nrows = e->cfg.max_metrics;
    return LH_OK;
}

int lh_cell_bytes(lh_engine *e) { return !e ? LH_EINVAL : (e->narrow ? 4 : 8); }

size_t lh_row_stride(void) { return LH_ROW_STRIDE; }

int lh_snapshot_ranges(lh_snapshot *s, void **d_ranges)
{
    if (!s || !d_ranges) return LH_EINVAL;
    *d_ranges = s->e->bufs[(size_t)s->buf].ranges;
    return LH_OK;
}

int lh_snapshot_mark_dirty(lh_snapshot *s, uint32_t first_row, uint32_t nrows, uint32_t lo_bin, uint32_t hi_bin)
{
    if (!s) return LH_EINVAL;
    lh_engine *e = s->e;
    if ((uint64_t)first_row + nrows > e->cfg.max_metrics || lo_bin > hi_bin || hi_bin >= LH_NKEYS) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    HIPCHK(lh::launch_mark_dirty(e->bufs[(size_t)s->buf].ranges, first_row, nrows, lo_bin, hi_bin, e->xstream));
    // the caller wrote cells of its own: nothing is known about their size any more (a later merge uses uint64 cells)
    __atomic_store_n(&e->bufs[(size_t)s->buf].nsamples, ~uint64_t(0), __ATOMIC_RELAXED);
    return LH_OK;
}

// ---- K4: RCCL merge ------------------------------------------------------------------------------
namespace {
// The few RCCL entry points K4 needs, resolved at run time from the caller's RCCL (no link-time
// dependency: single-GPU users never load it).  Enum values are NCCL's stable public ABI.
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_reducescatter_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
constexpr int kNcclUint32 = 3, kNcclUint64 = 5, kNcclSum = 0, kNcclMin = 3;
std::mutex g_rccl_mu;
std::string g_rccl_path = "librccl.so";
void *g_rccl_handle = nullptr;
nccl_allreduce_fn g_allreduce = nullptr;
nccl_reducescatter_fn g_reducescatter = nullptr;

int rccl_resolve()
{
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_allreduce && g_reducescatter) return LH_OK;
    g_rccl_handle = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl_handle) {
        std::snprintf(g_last_error, sizeof(g_last_error), "dlopen(%s): %s", g_rccl_path.c_str(), dlerror());
        return LH_EDEVICE;
    }
    g_allreduce = reinterpret_cast<nccl_allreduce_fn>(dlsym(g_rccl_handle, "ncclAllReduce"));
    g_reducescatter = reinterpret_cast<nccl_reducescatter_fn>(dlsym(g_rccl_handle, "ncclReduceScatter"));
    if (!g_allreduce || !g_reducescatter) {
        std::snprintf(g_last_error, sizeof(g_last_error), "%s lacks ncclAllReduce/ncclReduceScatter", g_rccl_path.c_str());
        return LH_EDEVICE;
    }
    return LH_OK;
}

#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        const int _r = (expr);                                                                 \
        if (_r != 0) {                                                                         \
            std::snprintf(g_last_error, sizeof(g_last_error), "%s: RCCL error %d", #expr, _r); \
            return LH_EDEVICE;                                                                 \
        }                                                                                      \
    } while (0)
} // namespace

int lh_set_rccl_library(const char *path)
{
    if (!path || !*path) return LH_EINVAL;
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (g_allreduce) return LH_ESTATE; // already resolved
    g_rccl_path = path;
    return LH_OK;
}

int lh_snapshot_merge(lh_snapshot *s, void *comm, int nranks, int rank, int plan, uint32_t nrows,
                      uint32_t *first_owned, uint32_t *last_owned)
{
    if (!s || !comm || nranks < 1 || nranks > 1024 || rank < 0 || rank >= nranks || nrows == 0) return LH_EINVAL;
    if (plan != LH_MERGE_ALLREDUCE && plan != LH_MERGE_REDUCE_SCATTER) return LH_EINVAL;
    lh_engine *e = s->e;
    if (nrows > e->cfg.max_metrics) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    rc = rccl_resolve();
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    hipStream_t st = e->xstream;
    const bool rs = plan == LH_MERGE_REDUCE_SCATTER;
    if (first_owned) *first_owned = 0;
    if (last_owned) *last_owned = rs ? 0 : nrows;
    e->merge_info = lh_merge_info{};
    e->merge_events_pending = false;
    for (hipEvent_t &ev : e->merge_ev)
        if (!ev) HIPCHK(hipEventCreate(&ev));
    HIPCHK(hipEventRecord(e->merge_ev[0], st));

    // plan arrays: P[max_metrics + 1], bstart[1025], info[16], brow[1025] (uint32), the plan kernels' partial sums, the
    // rows' wire classes (one byte each) -- allocated once, never moved
    const size_t M = e->cfg.max_metrics;
    if (!e->d_mplan) HIPCHK(hipMalloc((void **)&e->d_mplan, (M + 1 + 1025 + 16 + 520 + 5120 + (M + 7) / 8) * sizeof(uint64_t)));
    uint64_t *d_P = e->d_mplan, *d_bstart = d_P + M + 1, *d_info = d_bstart + 1025;
    uint32_t *d_brow = reinterpret_cast<uint32_t *>(d_info + 16);
    uint64_t *d_work = d_info + 16 + 520; // the plan kernels' per-row-block partial sums
    uint8_t *d_cls = reinterpret_cast<uint8_t *>(d_work + 5120);

    // 1. dirty ranges: min(lo), max(hi) over the ranks with ONE MIN all-reduce on (lo, ~hi), in place.  Two more things
    //    ride along, complemented the same way: this rank's sample count of the interval (clipped) -- every rank learns the
    //    largest one and all of them pick the same word type for the wire -- and, per row, the largest cell this rank
    //    holds in it (one pass over the rank's own dirty windows): nranks x the all-reduced maximum bounds every merged
    //    cell of the row, and a row whose bound fits 8 or 16 bits travels at that width (k_merge_widths).  A one-rank
    //    merge moves nothing between GPUs: no pass over the cells for it.
    const bool narrow = e->merge_narrow && nranks > 1;
    const size_t rwords = (size_t)nrows * 3 + 1;
    rc = ensure_xbuf(e, rwords * sizeof(uint32_t) + 256);
    if (rc) return rc;
    uint32_t *d_tmp = reinterpret_cast<uint32_t *>(e->d_xbuf);
    const uint64_t mine = __atomic_load_n(&b.nsamples, __ATOMIC_RELAXED);
    HIPCHK(lh::launch_merge_prep(d_tmp, b.ranges, b.counts, nrows, mine > 0xffffffffull ? 0xffffffffu : (uint32_t)mine,
                                 narrow, st));
    NCCLCHK(g_allreduce(d_tmp, d_tmp, rwords, kNcclUint32, kNcclMin, comm, st));
    HIPCHK(lh::launch_ranges_flip_hi(b.ranges, d_tmp, nrows, false, 0, st)); // (lo, ~~hi) back in place
    HIPCHK(hipEventRecord(e->merge_ev[1], st));
    const uint32_t *d_extra = d_tmp + (size_t)nrows * 2, *d_rowmaxc = d_extra + 1;

    // 2. the window plan, on the device: every row keeps its OWN merged window [lo_r, hi_r]; the windows are
    //    packed back to back (CSR, in wire words) and cut into nranks owner blocks of equal packed size.  The host only
    //    needs a few totals to size the collective; the plan kernel stores them straight into pinned memory and the host
    //    spins on a completion word (no copy, no stream synchronisation unless the stream is still busy with the
    //    interval's ingest after 2 ms).
    const uint32_t nblocks = rs ? (uint32_t)nranks : 1u;
    uint64_t info[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (e->d_hxbuf) {
        volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(e->h_xbuf + 128);
        if (++e->xseq == 0) e->xseq = 1;
        *flag = 0;
        HIPCHK(lh::launch_merge_plan(b.ranges, nrows, nblocks, (uint32_t)rank, (uint32_t)nranks, d_extra, d_rowmaxc, d_cls,
                                     d_P, d_bstart, d_brow, d_work, reinterpret_cast<uint64_t *>(e->d_hxbuf),
                                     reinterpret_cast<uint32_t *>(e->d_hxbuf + 128), e->xseq, st));
        HIPCHK(hipEventRecord(e->merge_ev[2], st));
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (*flag != e->xseq) {
            __builtin_ia32_pause();
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HIPCHK(hipStreamSynchronize(st));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        std::memcpy(info, e->h_xbuf, sizeof(info));
    } else {
        HIPCHK(lh::launch_merge_plan(b.ranges, nrows, nblocks, (uint32_t)rank, (uint32_t)nranks, d_extra, d_rowmaxc, d_cls,
                                     d_P, d_bstart, d_brow, d_work, d_info, nullptr, 0, st));
        HIPCHK(hipEventRecord(e->merge_ev[2], st));
        HIPCHK(hipMemcpyAsync(e->h_xbuf, d_info, sizeof(info), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::memcpy(info, e->h_xbuf, sizeof(info));
    }
    const uint64_t total = info[0], bmax = info[1]; // in wire words
    const uint32_t own_lo = rs ? (uint32_t)info[5] : 0u, own_hi = rs ? (uint32_t)info[6] : nrows;
    if (first_owned) *first_owned = own_lo;
    if (last_owned) *last_owned = own_hi;
    // no merged cell can reach 2^32 when nranks x (the largest per-rank sample count) stays below it (the same rule, on
    // the same all-reduced word, as k_merge_widths applied on the device)
    const bool words32 = info[4] < 0xffffffffull && info[4] * (uint64_t)nranks < (uint64_t(1) << 32);
    const size_t word = words32 ? sizeof(uint32_t) : sizeof(uint64_t);
    e->merge_info.packed_cells = info[7];
    e->merge_info.packed_words = total;
    e->merge_info.widest_row = (uint32_t)info[2];
    e->merge_info.occupied_rows = (uint32_t)info[3];
    e->merge_info.padded_words = rs ? (uint64_t)nranks * bmax : total;
    e->merge_info.cell_bytes = (uint32_t)word;
    e->merge_info.rows_8bit = (uint32_t)info[8];
    e->merge_info.rows_16bit = (uint32_t)info[9];
    if (total == 0) return LH_OK; // nothing anywhere (every rank computes the same plan: no hang)
    // a merged cell may not fit a narrow store's uint32: the snapshot moves to uint64 cells first (in stream order)
    if (!words32) {
        rc = widen_buffer(e, b, st, false);
        if (rc) return rc;
    }

    // 3. pack -> collective -> unpack.  The pack buffer is its own grow-only allocation.
    const uint64_t send_elems = rs ? (uint64_t)nranks * bmax : total, recv_elems = rs ? bmax : 0;
    const size_t need = (size_t)(send_elems + recv_elems) * word;
    if (need > e->mbuf_bytes) {
        if (e->d_mbuf) (void)hipFree(e->d_mbuf);
        e->d_mbuf = nullptr;
        e->mbuf_bytes = 0;
        const size_t cap = need + need / 8 + 4096;
        HIPCHK(hipMalloc((void **)&e->d_mbuf, cap));
        e->mbuf_bytes = cap;
    }
    unsigned char *send = reinterpret_cast<unsigned char *>(e->d_mbuf), *recv = send + (size_t)send_elems * word;
    e->merge_info.send_bytes = send_elems * word;
    e->merge_info.recv_bytes = (rs ? recv_elems : total) * word;
    HIPCHK(lh::launch_pack_rows(b.counts, b.ranges, d_cls, d_P, d_bstart, d_brow, nrows, nblocks, rs ? bmax : total, send,
                                words32, st));
    HIPCHK(hipEventRecord(e->merge_ev[3], st));
    const int dt = words32 ? kNcclUint32 : kNcclUint64;
    if (!rs) {
        NCCLCHK(g_allreduce(send, send, (size_t)total, dt, kNcclSum, comm, st));
        HIPCHK(hipEventRecord(e->merge_ev[4], st));
        HIPCHK(lh::launch_unpack_rows(b.counts, b.ranges, d_cls, d_P, d_bstart, 0, 0, nrows, send, words32, st));
    } else {
        // reduce-scatter by contiguous name blocks of equal packed size, every block padded to the largest one
        NCCLCHK(g_reducescatter(send, recv, (size_t)recv_elems, dt, kNcclSum, comm, st));
        HIPCHK(hipEventRecord(e->merge_ev[4], st));
        HIPCHK(lh::launch_unpack_rows(b.counts, b.ranges, d_cls, d_P, d_bstart, (uint32_t)rank, own_lo, own_hi - own_lo,
                                      recv, words32, st));
    }
    HIPCHK(hipEventRecord(e->merge_ev[5], st));
    e->merge_events_pending = true;
    // the rows now hold sums over the ranks: the per-rank bound no longer describes them
    __atomic_store_n(&b.nsamples, ~uint64_t(0), __ATOMIC_RELAXED);
    return LH_OK;
}

int lh_snapshot_merge_info(lh_snapshot *s, lh_merge_info *out)
{
    if (!s || !out) return LH_EINVAL;
    lh_engine *e = s->e;
    std::lock_guard<std::mutex> g(e->xmu);
    if (e->merge_events_pending) {
        int rc = use_device(e);
        if (rc) return rc;
        HIPCHK(hipEventSynchronize(e->merge_ev[5]));
        float *ms[5] = {&e->merge_info.ranges_ms, &e->merge_info.plan_ms, &e->merge_info.pack_ms,
                        &e->merge_info.collective_ms, &e->merge_info.unpack_ms};
        for (int i = 0; i < 5; i++) HIPCHK(hipEventElapsedTime(ms[i], e->merge_ev[i], e->merge_ev[i + 1]));
        HIPCHK(hipEventElapsedTime(&e->merge_info.span_ms, e->merge_ev[0], e->merge_ev[5]));
        e->merge_events_pending = false;
    }
    *out = e->merge_info;
    return LH_OK;
}

// ---------------------------------------------------------------------------
// K6: wire text on the device (lh_kernels_fmt.hip)
// ---------------------------------------------------------------------------
namespace {

// Names interned since the last call go to HBM (xmu held).
int upload_names(lh_engine *e, size_t *nnames)
{
    std::shared_lock<std::shared_mutex> g(e->names_mu);
    const size_t nn = e->names.size();
    *nnames = nn;
    if (e->names_uploaded == nn && e->d_name_off) return LH_OK;
    const size_t nbytes = e->name_blob.size();
    if (nbytes + 1 > e->d_names_cap) {
        if (e->d_names) (void)hipFree(e->d_names);
        e->d_names = nullptr;
        e->d_names_cap = 0;
        const size_t cap = 2 * nbytes + 4096;
        HIPCHK(hipMalloc((void **)&e->d_names, cap));
        e->d_names_cap = cap;
    }
    if (nn + 1 > e->d_name_off_cap) {
        if (e->d_name_off) (void)hipFree(e->d_name_off);
        e->d_name_off = nullptr;
        e->d_name_off_cap = 0;
        const size_t cap = 2 * (nn + 1) + 1024;
        HIPCHK(hipMalloc((void **)&e->d_name_off, cap * sizeof(uint32_t)));
        e->d_name_off_cap = cap;
    }
    // whole table each time it grew: a few MB at 65 536 names, and only in intervals that saw new names
    if (nbytes) HIPCHK(hipMemcpyAsync(e->d_names, e->name_blob.data(), nbytes, hipMemcpyHostToDevice, e->xstream));
    HIPCHK(hipMemcpyAsync(e->d_name_off, e->name_off.data(), (nn + 1) * sizeof(uint32_t), hipMemcpyHostToDevice,
                          e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream)); // the host vectors may grow once names_mu is dropped
    e->names_uploaded = nn;
    return LH_OK;
}

// label = pre "%s" post with "%%" -> '%'; exactly one %s
bool split_label(const char *label, std::string *pre, std::string *post)
{
    int seen = 0;
    std::string *cur = pre;
    for (const char *c = label; *c; c++) {
        if (c[0] == '%' && c[1] == 's') {
            if (seen++) return false;
            cur = post;
            c++;
        } else if (c[0] == '%' && c[1] == '%') {
            *cur += '%';
            c++;
        } else if (c[0] == '%') {
            return false;
        } else {
            *cur += *c;
        }
    }
    return seen == 1;
}

struct BlobBuilder {
    std::string bytes;
    bool dots;
    bool ok = true;
    void put(const std::string &s, bool is_key, uint32_t *off, uint32_t *len)
    {
        *off = (uint32_t)bytes.size();
        *len = (uint32_t)s.size();
        for (char c : s) bytes += (is_key && dots && c == '_') ? '.' : c;
        if (bytes.size() > lh::SER_BLOB_MAX) ok = false;
    }
    void key(const std::string &pre, const std::string &post, lh::SerKey *k)
    {
        uint32_t o, l;
        put(pre, true, &o, &l);
        k->pre_off = (uint16_t)o; k->pre_len = (uint16_t)l;
        put(post, true, &o, &l);
        k->post_off = (uint16_t)o; k->post_len = (uint16_t)l;
    }
};

} // namespace

int lh_serialize(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, const char *const *labels,
                 size_t np, const lh_line_format *fmt, uint32_t flags, char *out, size_t cap, size_t *len)
{
    if (!s || !fmt || !len || (np && (!p || !labels)) || (cap && !out)) return LH_EINVAL;
    if (!fmt->prefix || !fmt->sep || !fmt->suffix || np > LH_MAX_PERCENTILES) return LH_EINVAL;
    *len = 0;
    lh_engine *e = s->e;
    if (nmetrics == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;

    lh::SerArgs a{};
    BlobBuilder bb;
    bb.dots = (fmt->flags & LH_FMT_UNDERSCORE_TO_DOT) != 0;
    bb.put(fmt->prefix, false, &a.prefix_off, &a.prefix_len);
    bb.put(fmt->sep, false, &a.sep_off, &a.sep_len);
    bb.put(fmt->suffix, false, &a.suffix_off, &a.suffix_len);
    uint32_t nk = 0;
    bb.key("", "_count", &a.keys[nk++]); // metrics.go:349-351
    bb.key("", "_sum", &a.keys[nk++]);
    bb.key("", "_avg", &a.keys[nk++]);
    for (size_t i = 0; i < np; i++) {
        std::string pre, post;
        if (!labels[i] || !split_label(labels[i], &pre, &post)) return LH_EINVAL;
        bb.key(pre, post, &a.keys[nk++]);
    }
    if (flags & LH_SER_AGGREGATES) { // metrics.go:601-606
        bb.key("", "_agg_avg", &a.keys[nk++]);
        bb.key("", "_agg_count", &a.keys[nk++]);
        bb.key("", "_agg_sum", &a.keys[nk++]);
    }
    if (!bb.ok) return LH_EINVAL;

    std::lock_guard<std::mutex> g(e->xmu);
    size_t nnames = 0;
    rc = upload_names(e, &nnames);
    if (rc) return rc;
    if ((uint64_t)first + nmetrics > nnames) return LH_EINVAL;

    const uint64_t nlines = (uint64_t)nmetrics * nk;
    const uint32_t nb = lh::ser_blocks(nlines);
    const ExtractLayout L = extract_layout(nmetrics, np);
    const size_t off_lens = (L.total + 15) & ~size_t(15);
    const size_t off_bsum = off_lens + ((nlines * 4 + 15) & ~size_t(15));
    const size_t off_boff = off_bsum + (((size_t)nb * 4 + 15) & ~size_t(15));
    const size_t total_x = off_boff + ((size_t)nb + 1) * 8;
    rc = ensure_xbuf(e, total_x);
    if (rc) return rc;

    EpochBuffer &b = e->bufs[(size_t)s->buf];
    HIPCHK(hipMemcpyAsync(e->d_blob, bb.bytes.data(), bb.bytes.size(), hipMemcpyHostToDevice, e->xstream));
    HIPCHK(lh::launch_extract(lh::cells_at(b.counts, (size_t)first * LH_ROW_STRIDE), b.ranges + 2 * (size_t)first, (uint32_t)nmetrics,
                              p, (uint32_t)np, e->d_D, reinterpret_cast<lh::ExtractOut *>(e->d_xbuf + L.off_stats),
                              reinterpret_cast<double *>(e->d_xbuf + L.off_pvals),
                              reinterpret_cast<int16_t *>(e->d_xbuf + L.off_pkeys), e->d_xbuf + L.off_pvalid,
                              e->d_err, reinterpret_cast<uint32_t *>(e->d_xbuf + L.off_err), e->xstream));
    a.stats = reinterpret_cast<const lh::ExtractOut *>(e->d_xbuf + L.off_stats);
    a.pvals = reinterpret_cast<const double *>(e->d_xbuf + L.off_pvals);
    a.pvalid = e->d_xbuf + L.off_pvalid;
    a.life = e->d_life;
    a.names = e->d_names;
    a.name_off = e->d_name_off;
    a.blob = e->d_blob;
    a.blob_len = (uint32_t)bb.bytes.size();
    a.first = first;
    a.nmetrics = (uint32_t)nmetrics;
    a.np = (uint32_t)np;
    a.nkeys = nk;
    a.flags = bb.dots ? lh::SER_DOTS : 0u;
    uint32_t *d_lens = reinterpret_cast<uint32_t *>(e->d_xbuf + off_lens);
    uint32_t *d_bsum = reinterpret_cast<uint32_t *>(e->d_xbuf + off_bsum);
    uint64_t *d_boff = reinterpret_cast<uint64_t *>(e->d_xbuf + off_boff);
    HIPCHK(lh::launch_ser_len(a, d_lens, d_bsum, d_boff, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf, d_boff + nb, 8, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + 8, e->d_xbuf + L.off_err, 8, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream)); // also keeps bb.bytes alive until the blob copy is done
    uint64_t total = 0;
    uint32_t err, nfall;
    std::memcpy(&total, e->h_xbuf, 8);
    std::memcpy(&err, e->h_xbuf + 8, 4);
    std::memcpy(&nfall, e->h_xbuf + 12, 4);
    *len = (size_t)total;
    const int erc = after_extract(e, err, nfall); // LH_ERANGE: some sample carried a bad id (text is still produced)
    if (erc != LH_OK && erc != LH_ERANGE) return erc;
    if (total == 0 || total > cap) return erc;
    if (total + 16 > e->d_text_cap) {
        if (e->d_text) (void)hipFree(e->d_text);
        e->d_text = nullptr;
        e->d_text_cap = 0;
        const size_t want = (size_t)total + (size_t)total / 4 + 4096;
        HIPCHK(hipMalloc((void **)&e->d_text, want));
        e->d_text_cap = want;
    }
    HIPCHK(lh::launch_ser_write(a, d_lens, d_boff, e->d_text, e->xstream));
    HIPCHK(hipMemcpyAsync(out, e->d_text, (size_t)total, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    return erc;
}

int lh_snapshot_accumulate(lh_snapshot *s)
{
    if (!s) return LH_EINVAL;
    lh_engine *e = s->e;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    if (s->life_applied) return LH_OK;
    const size_t M = e->cfg.max_metrics;
    const ExtractLayout L = extract_layout(M, 0);
    rc = ensure_xbuf(e, L.total);
    if (rc) return rc;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    lh::ExtractOut *d_stats = reinterpret_cast<lh::ExtractOut *>(e->d_xbuf + L.off_stats);
    HIPCHK(lh::launch_extract(b.counts, b.ranges, (uint32_t)M, nullptr, 0, e->d_D, d_stats,
                              reinterpret_cast<double *>(e->d_xbuf + L.off_pvals),
                              reinterpret_cast<int16_t *>(e->d_xbuf + L.off_pkeys), e->d_xbuf + L.off_pvalid,
                              e->d_err, reinterpret_cast<uint32_t *>(e->d_xbuf + L.off_err), e->xstream));
    HIPCHK(lh::launch_life_add(d_stats, e->d_life, (uint32_t)M, e->xstream));
    s->life_applied = true;
    return LH_OK;
}

int lh_lifetime(lh_engine *e, uint32_t first, size_t n, uint64_t *count, uint64_t *sum)
{
    if (!e || (n && (!count || !sum))) return LH_EINVAL;
    if ((uint64_t)first + n > e->cfg.max_metrics) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    rc = ensure_xbuf(e, n * 16);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->h_xbuf, e->d_life + 2 * (size_t)first, n * 16, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    const uint64_t *h = reinterpret_cast<const uint64_t *>(e->h_xbuf);
    for (size_t i = 0; i < n; i++) { count[i] = h[2 * i]; sum[i] = h[2 * i + 1]; }
    return LH_OK;
}

int lh_format_f(lh_engine *e, const double *v, size_t n, char *out, size_t slot, uint32_t *lens)
{
    if (!e || (n && (!v || !out || !lens)) || slot < lh::SER_FMT_SLOT || n > (size_t(1) << 24)) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    const size_t off_out = (n * 8 + 15) & ~size_t(15);
    const size_t off_lens = off_out + n * lh::SER_FMT_SLOT;
    const size_t total = off_lens + n * 4;
    rc = ensure_xbuf(e, total);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->d_xbuf, v, n * 8, hipMemcpyHostToDevice, e->xstream));
    HIPCHK(lh::launch_format_f(reinterpret_cast<const double *>(e->d_xbuf), reinterpret_cast<char *>(e->d_xbuf + off_out),
                               reinterpret_cast<uint32_t *>(e->d_xbuf + off_lens), (uint32_t)n, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + off_out, e->d_xbuf + off_out, total - off_out, hipMemcpyDeviceToHost,
                          e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    std::memcpy(lens, e->h_xbuf + off_lens, n * 4);
    for (size_t i = 0; i < n; i++)
        std::memcpy(out + i * slot, e->h_xbuf + off_out + i * lh::SER_FMT_SLOT, lens[i]);
    return LH_OK;
}

namespace {
// the interval's counter amounts go into the lifetime store exactly once per snapshot (xmu held)
int fold_counters(lh_snapshot *s)
{
    lh_engine *e = s->e;
    if (s->counters_folded || !e->cfg.max_counters) return LH_OK;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    HIPCHK(lh::launch_count_fold(b.ccur, b.cflag, e->d_clife, e->d_cknown, e->cfg.max_counters, e->xstream));
    s->counters_folded = true;
    return LH_OK;
}
} // namespace

int lh_counters_collect(lh_snapshot *s, uint32_t first, size_t n, uint64_t *rate, uint8_t *present, uint64_t *total,
                        uint8_t *known)
{
    if (!s) return LH_EINVAL;
    lh_engine *e = s->e;
    if ((uint64_t)first + n > e->cfg.max_counters) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    rc = fold_counters(s);
    if (rc || n == 0) return rc;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    const size_t o_rate = 0, o_total = n * 8, o_flag = 2 * n * 8, o_known = o_flag + n * 4, bytes = o_known + n * 4;
    rc = ensure_xbuf(e, bytes);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->h_xbuf + o_rate, b.ccur + first, n * 8, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + o_total, e->d_clife + first, n * 8, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + o_flag, b.cflag + first, n * 4, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf + o_known, e->d_cknown + first, n * 4, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    if (rate) std::memcpy(rate, e->h_xbuf + o_rate, n * 8);
    if (total) std::memcpy(total, e->h_xbuf + o_total, n * 8);
    const uint32_t *hf = reinterpret_cast<const uint32_t *>(e->h_xbuf + o_flag);
    const uint32_t *hk = reinterpret_cast<const uint32_t *>(e->h_xbuf + o_known);
    for (size_t i = 0; i < n; i++) {
        if (present) present[i] = hf[i] ? 1 : 0;
        if (known) known[i] = hk[i] ? 1 : 0;
    }
    return LH_OK;
}

int lh_serialize_counters(lh_snapshot *s, uint32_t first, size_t n, const lh_line_format *fmt, char *out, size_t cap,
                          size_t *len)
{
    if (!s || !fmt || !len || (cap && !out)) return LH_EINVAL;
    if (!fmt->prefix || !fmt->sep || !fmt->suffix) return LH_EINVAL;
    *len = 0;
    lh_engine *e = s->e;
    if ((uint64_t)first + n > e->cfg.max_counters) return LH_EINVAL;
    if (n == 0) return LH_OK;
    int rc = use_device(e);
    if (rc) return rc;
    lh::SerArgs a{};
    BlobBuilder bb;
    bb.dots = (fmt->flags & LH_FMT_UNDERSCORE_TO_DOT) != 0;
    bb.put(fmt->prefix, false, &a.prefix_off, &a.prefix_len);
    bb.put(fmt->sep, false, &a.sep_off, &a.sep_len);
    bb.put(fmt->suffix, false, &a.suffix_off, &a.suffix_len);
    bb.key("", "", &a.keys[0]);      // metrics.go:487-489: the counter's name is the key
    bb.key("", "_rate", &a.keys[1]); // metrics.go:491-493
    if (!bb.ok) return LH_EINVAL;
    std::lock_guard<std::mutex> g(e->xmu);
    rc = fold_counters(s);
    if (rc) return rc;
    size_t nnames = 0;
    {   // counter names interned since the last call go to HBM
        std::shared_lock<std::shared_mutex> ng(e->cnames_mu);
        nnames = e->cnames.size();
        if (e->cnames_uploaded != nnames || !e->d_cname_off) {
            const size_t nbytes = e->cname_blob.size();
            if (nbytes + 1 > e->d_cnames_cap) {
                if (e->d_cnames) (void)hipFree(e->d_cnames);
                e->d_cnames = nullptr;
                e->d_cnames_cap = 0;
                HIPCHK(hipMalloc((void **)&e->d_cnames, 2 * nbytes + 4096));
                e->d_cnames_cap = 2 * nbytes + 4096;
            }
            if (nnames + 1 > e->d_cname_off_cap) {
                if (e->d_cname_off) (void)hipFree(e->d_cname_off);
                e->d_cname_off = nullptr;
                e->d_cname_off_cap = 0;
                HIPCHK(hipMalloc((void **)&e->d_cname_off, (2 * (nnames + 1) + 1024) * sizeof(uint32_t)));
                e->d_cname_off_cap = 2 * (nnames + 1) + 1024;
            }
            if (nbytes) HIPCHK(hipMemcpyAsync(e->d_cnames, e->cname_blob.data(), nbytes, hipMemcpyHostToDevice, e->xstream));
            HIPCHK(hipMemcpyAsync(e->d_cname_off, e->cname_off.data(), (nnames + 1) * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, e->xstream));
            HIPCHK(hipStreamSynchronize(e->xstream));
            e->cnames_uploaded = nnames;
        }
    }
    if ((uint64_t)first + n > nnames) return LH_EINVAL; // every counter in the range must have a name
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    const uint64_t nlines = (uint64_t)n * 2;
    const uint32_t nb = lh::ser_blocks(nlines);
    const size_t off_lens = 0;
    const size_t off_bsum = (nlines * 4 + 15) & ~size_t(15);
    const size_t off_boff = off_bsum + (((size_t)nb * 4 + 15) & ~size_t(15));
    rc = ensure_xbuf(e, off_boff + ((size_t)nb + 1) * 8);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(e->d_blob, bb.bytes.data(), bb.bytes.size(), hipMemcpyHostToDevice, e->xstream));
    a.c_total = e->d_clife;
    a.c_rate = b.ccur;
    a.c_known = e->d_cknown;
    a.c_present = b.cflag;
    a.names = e->d_cnames;
    a.name_off = e->d_cname_off;
    a.blob = e->d_blob;
    a.blob_len = (uint32_t)bb.bytes.size();
    a.first = first;
    a.nmetrics = (uint32_t)n;
    a.np = 0;
    a.nkeys = 2;
    a.flags = (bb.dots ? lh::SER_DOTS : 0u) | lh::SER_COUNTERS;
    uint32_t *d_lens = reinterpret_cast<uint32_t *>(e->d_xbuf + off_lens);
    uint32_t *d_bsum = reinterpret_cast<uint32_t *>(e->d_xbuf + off_bsum);
    uint64_t *d_boff = reinterpret_cast<uint64_t *>(e->d_xbuf + off_boff);
    HIPCHK(lh::launch_ser_len(a, d_lens, d_bsum, d_boff, e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf, d_boff + nb, 8, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream)); // also keeps bb.bytes alive until the blob copy is done
    uint64_t total = 0;
    std::memcpy(&total, e->h_xbuf, 8);
    *len = (size_t)total;
    if (total == 0 || total > cap) return LH_OK;
    if (total + 16 > e->d_text_cap) {
        if (e->d_text) (void)hipFree(e->d_text);
        e->d_text = nullptr;
        e->d_text_cap = 0;
        const size_t want = (size_t)total + (size_t)total / 4 + 4096;
        HIPCHK(hipMalloc((void **)&e->d_text, want));
        e->d_text_cap = want;
    }
    HIPCHK(lh::launch_ser_write(a, d_lens, d_boff, e->d_text, e->xstream));
    HIPCHK(hipMemcpyAsync(out, e->d_text, (size_t)total, hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    return LH_OK;
}

int lh_snapshot_stream(lh_snapshot *s, void **stream)
{
    if (!s || !stream) return LH_EINVAL;
    *stream = (void *)s->e->xstream;
    return LH_OK;
}

namespace {
// lh_release, narrow engines, behind the clear (xmu held): the store the buffer's next interval starts on.  Normally the
// narrow one again -- the widening left it clean, the wide store is clean behind the clear and kept for the next time.  A
// buffer whose ingest passed 2^32 samples in kStayWide consecutive intervals stays on the wide store and FREES the narrow
// one: at such rates the narrow store only ever holds the first milliseconds of an interval, and keeping both would cost
// 1.5 x what a 64-bit engine holds.  kBackToNarrow consecutive intervals below 2^31 samples bring it back (and free the
// wide store).  hipFree waits for the device: these are transitions, not per-interval work.
constexpr uint32_t kStayWide = 2, kBackToNarrow = 16;
int next_interval_width(lh_engine *e, EpochBuffer &b)
{
    const size_t M = e->cfg.max_metrics;
    if (b.store32) {
        b.wide_streak = b.ingest_widened ? b.wide_streak + 1 : 0;
        if (b.wide_streak >= kStayWide && !lh::cells_narrow(b.counts)) {
            HIPCHK(hipFree(b.store32));
            b.store32 = nullptr;
            e->c_store_bytes.fetch_sub(M * LH_ROW_STRIDE * sizeof(uint32_t), std::memory_order_relaxed);
            b.quiet_streak = 0;
            return LH_OK; // b.counts stays the wide store
        }
        b.counts = lh::cells_tagged(b.store32, 4);
        return LH_OK;
    }
    const uint64_t ns = __atomic_load_n(&b.nsamples, __ATOMIC_RELAXED);
    b.quiet_streak = ns <= e->widen_at / 2 ? b.quiet_streak + 1 : 0; // (half the bound: 2^31 samples)
    if (b.quiet_streak < kBackToNarrow) return LH_OK;
    b.quiet_streak = 0;
    uint32_t *p = nullptr;
    if (hipMalloc((void **)&p, M * LH_ROW_STRIDE * sizeof(uint32_t)) != hipSuccess) {
        (void)hipGetLastError();
        return LH_OK; // stays wide: nothing is lost
    }
    hipError_t me = hipMemsetAsync(p, 0, M * LH_ROW_STRIDE * sizeof(uint32_t), e->xstream);
    if (me != hipSuccess) {
        (void)hipFree(p);
        HIPCHK(me);
    }
    HIPCHK(hipFree(b.store64)); // (waits for the clear that was just enqueued)
    b.store64 = nullptr;
    b.store32 = p;
    e->c_store_bytes.fetch_add(M * LH_ROW_STRIDE * sizeof(uint32_t), std::memory_order_relaxed);
    e->c_store_bytes.fetch_sub(M * LH_ROW_STRIDE * sizeof(uint64_t), std::memory_order_relaxed);
    b.wide_streak = 0;
    b.counts = lh::cells_tagged(p, 4);
    return LH_OK;
}
} // namespace

int lh_release(lh_snapshot *s)
{
    if (!s) return LH_EINVAL;
    lh_engine *e = s->e;
    int rc = use_device(e);
    if (rc) return rc;
    EpochBuffer &b = e->bufs[(size_t)s->buf];
    // The first error is reported, but the buffer is recycled and the handle freed whatever happens on the way: a
    // failed step that returned early would leave the epoch buffer BUF_SNAPSHOT for good (every later lh_flip
    // LH_EBUSY) and leak the handle.
    int first_err = LH_OK;
    auto keep = [&](int r) { if (r != LH_OK && first_err == LH_OK) first_err = r; };
    auto hip = [&](hipError_t he) {
        if (he != hipSuccess) {
            set_last_error("lh_release", he);
            keep(LH_EDEVICE);
        }
    };
    {
        std::lock_guard<std::mutex> g(e->xmu);
        // the reference folds the interval's amounts into the lifetime totals at the epoch boundary whoever reads
        // them (metrics.go:435-458): a snapshot released before any counter call still folds
        keep(fold_counters(s));
        hip(lh::launch_clear(b.counts, b.ranges, e->cfg.max_metrics, e->xstream));
        if (e->narrow) keep(next_interval_width(e, b));
        if (e->cfg.max_counters) {
            hip(hipMemsetAsync(b.ccur, 0, (size_t)e->cfg.max_counters * sizeof(uint64_t), e->xstream));
            hip(hipMemsetAsync(b.cflag, 0, (size_t)e->cfg.max_counters * sizeof(uint32_t), e->xstream));
        }
        hip(hipEventRecord(b.cleared, e->xstream));
        __atomic_store_n(&b.nsamples, 0, __ATOMIC_RELAXED);
        __atomic_store_n(&b.reserved, 0, __ATOMIC_RELAXED);
        b.ingest_widened = false;
    }
    {
        std::unique_lock<std::shared_mutex> eg(e->epoch_mu);
        b.state = BUF_FREE;
    }
    e->live_snapshots.fetch_sub(1);
    delete s;
    return first_err;
}

int lh_get_counters(lh_engine *e, lh_counters *out)
{
    if (!e || !out) return LH_EINVAL;
    out->samples_single = e->c_single.load();
    out->samples_small = e->c_small.load();
    out->samples_partitioned = e->c_part.load();
    out->samples_direct = e->c_direct.load();
    out->launches = e->c_launches.load();
    out->flips = e->c_flips.load();
    out->flips_busy = e->c_busy.load();
    out->extracts = e->c_extracts.load();
    out->backpressure_waits = e->c_waits.load();
    out->window_misses = e->c_misses.load();
    out->small_path_disabled = e->small_disabled.load() ? 1u : 0u;
    out->regions_disabled = e->regions_disabled.load() ? 1u : 0u;
    out->scratch_bytes = e->c_scratch.load();
    {   //  the lanes' blocks and their shared survey tables were reported nowhere
        std::lock_guard<std::mutex> g(e->scratch_mu);
        uint64_t lane = 0;
        for (const lh_engine::AuxScratch &a : e->aux) lane += a.bytes;
        if (e->lane_tables.p[0]) lane += 2 * (uint64_t)e->lane_tables.bytes;
        out->lane_scratch_bytes = lane;
    }
    out->widenings = e->c_widenings.load();
    out->store_bytes = e->c_store_bytes.load();
    out->sublaunches = e->c_sublaunches.load();
    out->samples_partitioned_v2 = e->c_part2.load();
    out->counter_events = e->c_counts.load();
    out->region_overflows = e->c_region_ovf.load();
    out->samples_partitioned_v3 = e->c_part3.load();
    out->records_level1 = __atomic_load_n(&e->h_rstat[2], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[kLaneRstat + 2], __ATOMIC_RELAXED);
    out->records_level2 = __atomic_load_n(&e->h_rstat[3], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[kLaneRstat + 3], __ATOMIC_RELAXED);
    out->level2_overflows = __atomic_load_n(&e->h_rstat[4], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[kLaneRstat + 4], __ATOMIC_RELAXED);
    out->reduce_window_misses = __atomic_load_n(&e->h_rstat[5], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[kLaneRstat + 5], __ATOMIC_RELAXED);
    out->surveys_reused = e->c_survey_reuse.load();
    out->scratch_alloc_failures = e->c_alloc_fail.load();
    out->samples_fallback = e->c_fallback.load();
    out->survey_stale_pairs = __atomic_load_n(&e->h_rstat[7], __ATOMIC_RELAXED) + __atomic_load_n(&e->h_rstat[kLaneRstat + 7], __ATOMIC_RELAXED);
    {
        const uint64_t lw = __atomic_load_n(&e->h_rstat[1], __ATOMIC_RELAXED) & 0xffu;
        std::lock_guard<std::mutex> g(e->scratch_mu);
        out->window_log2 = e->v3_log_w_fixed ? e->tune.v3_log_w : (lw >= 10 && lw <= 14 ? lw : e->tune.v3_log_w);
    }
    return LH_OK;
}

// An option that feeds the launch plans: written under the scratch lock (launch_pairs snapshots `tune` under it), and
// the third generation's kept survey tables -- laid out for the old plan's LDS budget -- are not reused.
static int set_tune(lh_engine *e, const std::function<void(lh::PartTuning &)> &f)
{
    std::lock_guard<std::mutex> g(e->scratch_mu);
    f(e->tune);
    e->tune_gen++;
    e->tables.valid = false; // (the lanes' table sets carry the tune_gen they were planned under: not reused either)
    return LH_OK;
}

int lh_set_option(lh_engine *e, int option, uint64_t value)
{
    if (!e) return LH_EINVAL;
    switch (option) {
    case LH_OPT_TWO_LEVEL_ABOVE:
        if (value > 256) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.two_level_above = (uint32_t)value; });
    case LH_OPT_HOT_MIN_TILES:
        if (value < 1 || value > (1u << 20)) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.hot_min_tiles = (uint32_t)value; });
    case LH_OPT_HOT_WINDOWS:
        if (value > 1) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.hot = value != 0; });
    case LH_OPT_NAMES_PER_PARTITION:
        if (value < 1 || value > 64) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.names_per_part = (uint32_t)value; });
    case LH_OPT_EXTRACT_ZERO_COPY:
        if (value > 1 && (value < 4096 || value > (uint64_t(64) << 20))) return LH_EINVAL;
        e->zero_copy_enabled = value != 0;
        if (value > 1) e->zero_copy_max = (size_t)value;
        return LH_OK;
    case LH_OPT_SCRATCH_CAP_BYTES: {
        if (value < (uint64_t(64) << 20)) return LH_EINVAL;
        std::lock_guard<std::mutex> g(e->scratch_mu);
        e->scratch_cap = (size_t)value;
        e->scratch_cap_set = true;
        return LH_OK;
    }
    case LH_OPT_SUBLAUNCH_PAIRS: {
        if (value < (uint64_t(1) << 22) || value > (uint64_t(1) << 30)) return LH_EINVAL;
        size_t p2 = size_t(1) << 22;
        while (p2 * 2 <= value) p2 *= 2;
        std::lock_guard<std::mutex> g(e->scratch_mu);
        e->sublaunch_pairs = p2;
        e->sublaunch_set = true;
        return LH_OK;
    }
    case LH_OPT_PART_V2:
        if (value > 1) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.v2 = value != 0; });
    case LH_OPT_PART_V2_SHAPE:
        if (value > 3 && value != 6) return LH_EINVAL; // (6: the wide shape forced; the engine picks it itself from the survey's report)
        return set_tune(e, [&](lh::PartTuning &t) { t.v2_shape = (uint32_t)value; });
    case LH_OPT_PART_V2_MIN_PAIRS:
        if (value < (1u << 17) || value > (uint64_t(1) << 31)) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.v2_min_samples = (size_t)value; });
    case LH_OPT_PART_V3:
        if (value > 1) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.v3 = value != 0; });
    case LH_OPT_PART_V3_MIN_PAIRS:
        if (value < (1u << 17) || value > (uint64_t(1) << 31)) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.v3_min_samples = (size_t)value; });
    case LH_OPT_PART_V3_DIRECT_MAX_PAIRS:
        if (value > (uint64_t(1) << 30)) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.v3_direct_max = (size_t)value; });
    case LH_OPT_PART_V3_LOG_W:
        if (value != 0 && (value < 10 || value > 14)) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) {
            e->v3_log_w_fixed = value != 0;
            t.v3_log_w = value ? (uint32_t)value : 10u;
        });
    case LH_OPT_PART_MIN_PAIRS:
        if (value != 0 && (value < 65536 || value > (uint64_t(1) << 31))) return LH_EINVAL;
        return set_tune(e, [&](lh::PartTuning &t) { t.part_min_samples = (size_t)value; });
    case LH_OPT_LANE_SCRATCH_BLOCKS: {
        if (value > lh_engine::kAuxBlocks) return LH_EINVAL;
        std::lock_guard<std::mutex> g(e->scratch_mu);
        e->lane_blocks = (uint32_t)value;
        return LH_OK;
    }
    case LH_OPT_SURVEY_EVERY: {
        if (value < 1 || value > 1024) return LH_EINVAL;
        std::lock_guard<std::mutex> g(e->scratch_mu);
        e->survey_every = (uint32_t)value;
        e->tables.valid = false;
        e->lane_tables.t[0].valid = e->lane_tables.t[1].valid = false;
        return LH_OK;
    }
    case LH_OPT_LANE_GEN3: {
        if (value > 256) return LH_EINVAL; // 0 off, 1 on, 2 .. 256: on, with that many level-1 workgroups per launch (tuning runs)
        std::lock_guard<std::mutex> g(e->scratch_mu);
        e->lane_gen3 = value != 0;
        e->lane_g1_cap = value > 1 ? (uint32_t)value : lh::kLaneLevel1Workgroups;
        e->lane_tables.t[0].valid = e->lane_tables.t[1].valid = false;
        return LH_OK;
    }
    case LH_OPT_MERGE_NARROW_CELLS: {
        if (value > 1) return LH_EINVAL;
        std::lock_guard<std::mutex> g(e->xmu);
        e->merge_narrow = value != 0;
        return LH_OK;
    }
    case LH_OPT_FAIL_SCRATCH_ALLOCS:
        if (value > 0xffffffffull) return LH_EINVAL;
        e->fail_allocs.store((uint32_t)value);
        return LH_OK;
    case LH_OPT_WIDEN_AT_SAMPLES: {
        if (value < 1 || value > 0xffffffffull) return LH_EINVAL;
        std::unique_lock<std::shared_mutex> u(e->cells_mu);
        e->widen_at = value;
        return LH_OK;
    }
    case LH_OPT_LANE_ZERO_COPY:
        if (value > 1) return LH_EINVAL;
        e->lane_zero_copy = value != 0;
        return LH_OK;
    case LH_OPT_SMALL_PATH:
        if (value > 1) return LH_EINVAL;
        e->small_disabled.store(value == 0);
        e->small_forced_off = value == 0;
        return LH_OK;
    default:
        return LH_EINVAL;
    }
}

int lh_compress_device(lh_engine *e, const double *d_v, int16_t *d_keys, size_t n, void *stream)
{
    if (!e || ((!d_v || !d_keys) && n)) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    HIPCHK(lh::launch_compress(d_v, d_keys, n, e->d_Tx, false, stream ? (hipStream_t)stream : e->main_stream));
    return LH_OK;
}

int lh_compress_device_golog(lh_engine *e, const double *d_v, int16_t *d_keys, size_t n, void *stream)
{
    if (!e || ((!d_v || !d_keys) && n)) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    HIPCHK(lh::launch_compress(d_v, d_keys, n, e->d_Tx, true, stream ? (hipStream_t)stream : e->main_stream));
    return LH_OK;
}

int lh_codec_tables(lh_engine *e, double *Tx, double *D)
{
    if (!e) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    if (Tx) HIPCHK(hipMemcpy(Tx, e->d_Tx, sizeof(double) * LH_NTHRESH, hipMemcpyDeviceToHost));
    if (D) HIPCHK(hipMemcpy(D, e->d_D, sizeof(double) * LH_NKEYS, hipMemcpyDeviceToHost));
    return LH_OK;
}

// The path choice of launch_pairs without an engine or a device (loghisto_gpu_tuning.h): the same choose_step /
// peel_first calls, in the same order; a sub-launch whose scratch "cannot be had" (fail_allocs) takes the direct kernel
// exactly as run_lane_block / run_shared do.  (A lane's block, once allocated, serves the call's later lane launches; the
// shared block grows to the largest sub-launch.)
int lh_dispatch_probe(const lh_dispatch_query *q, lh_dispatch_step *steps, size_t cap, size_t *nsteps)
{
    if (!q || !nsteps || (cap && !steps) || q->struct_size != sizeof(lh_dispatch_query)) return LH_EINVAL;
    if (q->max_metrics == 0 || (q->id_width != 2 && q->id_width != 4) || q->n == 0) return LH_EINVAL;
    lh::DispatchState st;
    st.max_metrics = q->max_metrics;
    st.num_cus = q->num_cus ? (int)q->num_cus : 256;
    st.lane_samples = (size_t)q->lane_samples;
    st.lane_blocks = q->lane_blocks;
    st.small_disabled = q->small_disabled != 0;
    st.regions_disabled = q->regions_disabled != 0;
    st.v3_disabled = q->v3_disabled != 0;
    st.call_log_w = q->call_log_w ? q->call_log_w : 10u;
    if (q->scratch_cap) { st.scratch_cap = (size_t)q->scratch_cap; st.scratch_cap_set = true; }
    if (q->sublaunch_pairs) { st.sublaunch_pairs = (size_t)q->sublaunch_pairs; st.sublaunch_set = true; }
    st.tune.part_min_samples = (size_t)q->part_min_pairs;
    st.tune.v2_min_samples = (size_t)q->v2_min_pairs;
    st.tune.v3_min_samples = (size_t)q->v3_min_pairs;
    st.tune.v2 = !q->v2_off;
    st.tune.v3 = !q->v3_off;
    st.tune.hot = !q->hot_off;
    if (q->v2_shape_set) st.tune.v2_shape = q->v2_shape & 3u;
    st.lane_gen3 = !q->lane_gen3_off;
    uintptr_t ids = (uintptr_t)q->ids_addr, vals = (uintptr_t)q->vals_addr;
    size_t n = (size_t)q->n, k = 0, shared_bytes = 0, lane_bytes = 0;
    uint32_t fail = q->fail_allocs;
    bool peel = lh::peel_first(ids, q->id_width, vals, n);
    while (n) {
        lh::Step s;
        lh_dispatch_step o{};
        if (peel) {
            s.take = 1;
            o.peeled = 1;
            peel = false;
        } else {
            s = lh::choose_step(st, ids, q->id_width, vals, n, q->host_fed != 0);
        }
        if (s.take == 0 || s.take > n) return LH_ESTATE; // (cannot happen: every step makes progress inside the call)
        if (s.scratch) {
            size_t &have = s.lane_block ? lane_bytes : shared_bytes;
            if (have < s.scratch) {
                if (fail) { fail--; s.kind = lh::PATH_DIRECT; s.lane_block = false; s.scratch = 0; o.fell_back = 1; }
                else have = s.lane_block ? s.scratch_alloc : s.scratch;
            }
        }
        o.path = (uint32_t)s.kind;
        o.lane_block = s.lane_block ? 1u : 0u;
        o.take = s.take;
        o.scratch = s.scratch;
        if (k < cap) steps[k] = o;
        k++;
        ids += s.take * q->id_width;
        vals += s.take * sizeof(double);
        n -= s.take;
    }
    *nsteps = k;
    return LH_OK;
}

int lh_selftest_vlog(lh_engine *e, double *max_abs_err)
{
    if (!e || !max_abs_err) return LH_EINVAL;
    int rc = use_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(e->xmu);
    rc = ensure_xbuf(e, 64);
    if (rc) return rc;
    HIPCHK(lh::launch_vlog_selftest(reinterpret_cast<double *>(e->d_xbuf), e->xstream));
    HIPCHK(hipMemcpyAsync(e->h_xbuf, e->d_xbuf, sizeof(double), hipMemcpyDeviceToHost, e->xstream));
    HIPCHK(hipStreamSynchronize(e->xstream));
    std::memcpy(max_abs_err, e->h_xbuf, sizeof(double));
    return LH_OK;
}

} // extern "C"
