// lh_dispatch.h -- WHICH kernel path a mixed (id, value) launch takes, as pure functions of a snapshot of the
// engine's state: no HIP call, no engine pointer, no clock.  lh_engine.cc (launch_pairs) executes what they return;
// tests/test_dispatch.py enumerates them on a box without a GPU through lh_dispatch_probe
// (include/loghisto_gpu_tuning.h).  Every path is exact (Histogram(name, v) = histogramCache[name][compress(v)] += 1,
// metrics.go:273-295): the choice only decides how fast a launch runs.
//
// The paths, fastest applicable first (measured crossovers: DESIGN.md 5):
//   SMALL   <= 32 names, >= 65 536 pairs: one streaming pass, every name's window in LDS       (lh_kernels_small.hip)
//   GEN2    33 .. 8 192 names, >= 2^20 pairs: survey + region scatter of 2-byte records + reduce (lh_kernels_part2.h)
//   GEN3    8 193 .. 65 536 names, >= 3 * 2^20 pairs: hashed survey + two scatter levels + reduce (lh_kernels_part3.h)
//   GEN1    partition by name (4-byte records, one or two levels) + reduce (lh_kernels_part.hip): since round 6 no
//           device-resident call's default -- names without skew above 8 192, a generation switched off by option -- and a
//           host-fed lane launch (2^17 .. 2^22 pairs) over <= 8 192 names when the lanes have scratch blocks of their own
//           (above 8 192 names such a launch takes GEN3 there from 2^18 pairs, on survey tables the lanes share)
//   DIRECT  no scratch, no survey: whole tiles through a per-workgroup LDS table of (name, bin) cells, the rest one global
//           atomic per sample (launch_ingest_pairs_cells): small or misaligned launches, and any launch whose scratch
//           cannot be had.  The minimum sizes are where each path overtakes it: profiles/r06_small_calls.txt
#pragma once

#include "lh_kernels.h"

#include <stddef.h>
#include <stdint.h>

namespace lh {

enum PathKind : uint32_t { PATH_DIRECT = 0, PATH_SMALL = 1, PATH_GEN1 = 2, PATH_GEN2 = 3, PATH_GEN3 = 4 };

constexpr size_t kMaxLaunchPairs = size_t(1) << 30;   // one launch: LDS counters and record indices stay below 2^32
constexpr size_t kLaneBlockMaxPairs = size_t(1) << 22; // larger host-fed launches amortise their passes: the shared block
// A host-fed lane's half-buffer keeps the thresholds it was tuned with (profiles/r05_hostfed_native.jsonl): its launches are
// link-bound in their first pass whatever the kernel, and the lanes' blocks exist so that later passes run beside the next read.
constexpr size_t kLanePartMinPairs = size_t(1) << 17, kLaneV3MinPairs = size_t(1) << 18;
constexpr uint32_t kLaneLevel1Workgroups = 8;          // a lane's third-generation launch: level-1 workgroups (PartTuning::v3_g1_cap)

// What the choice reads.  lh_engine fills it once per call (under its scratch lock: every option that feeds a launch
// plan is written under that lock).
struct DispatchState {
    uint32_t max_metrics = 0;
    int num_cus = 256;
    size_t lane_samples = 0;       // lh_config.lane_samples: the lanes' blocks are sized for it
    PartTuning tune;               // lh_set_option
    uint32_t call_log_w = 10;      // third generation: the window width this call runs with (the last survey's report)
    bool call_wide = false;        // second generation: the last survey saw spans wider than the 8 192-bin reduce windows
    bool call_yield = false;       // 1 025 .. 8 192 names: ... and more than 1/8 of the mass outside the second generation's cold windows
    bool small_disabled = false;   // adaptive switches (lh_engine: window misses / region overflows / forwarded share)
    bool regions_disabled = false;
    bool v3_disabled = false;
    uint32_t lane_blocks = 0;      // scratch blocks of the host-fed lanes (0, the default: a half-buffer takes the direct path)
    bool lane_gen3 = true;         // 8 193 .. 65 536 names: a lane's launch takes the third generation (records in the lane's
                                   // block, the survey's tables shared by all lanes) instead of the first
    uint32_t lane_g1_cap = kLaneLevel1Workgroups; // ... with at most this many level-1 workgroups
    bool scratch_cap_set = false, sublaunch_set = false; // the caller bounded the block: LH_OPT_SCRATCH_CAP_BYTES / _SUBLAUNCH_PAIRS
    size_t scratch_cap = size_t(1536) << 20;
    size_t sublaunch_pairs = size_t(1) << 29;
};


// One sub-launch: the first `take` of the n pairs at (ids, vals).
struct Step {
    PathKind kind = PATH_DIRECT;
    size_t take = 0;
    size_t scratch = 0;       // bytes of scratch the launch wrapper needs (0: none)
    size_t scratch_alloc = 0; // what to allocate when the block at hand is smaller (a lane's block is sized once, for the
                              // lanes' largest launch: their launches are all of about one size)
    bool lane_block = false;  // in one of the lanes' own blocks: GEN1, or GEN3 with the lanes' shared survey tables
    PartTuning tune;          // the tuning to call the launch wrapper with (adaptive switches and width applied)
};

// ids / vals: the device addresses -- their alignment decides which kernels can take the launch (two ids and two
// values per vector load).  id_width: 2 or 4 bytes.  n >= 1.
Step choose_step(const DispatchState &st, uintptr_t ids, uint32_t id_width, uintptr_t vals, size_t n, bool host_fed);

// A slice taken at an odd sample index leaves BOTH arrays one element short of the fast kernels' vector alignment:
// that one sample goes through the direct kernel, the rest of the call is aligned again.
bool peel_first(uintptr_t ids, uint32_t id_width, uintptr_t vals, size_t n);

// ---- survey reuse (second and third generation) --------------------------------------------------------------------
// The survey's tables (hot names, region sizes, per-partition ranking) stay in the shared scratch block; metric streams
// are stationary from one interval to the next, so a call may run on an earlier call's survey.  It only decides WHERE a
// sample is counted: a stale one costs speed, never exactness.
struct SurveyTables {
    bool valid = false;        // the block holds the tables of a survey and nothing has overwritten them since
    int gen = 0;               // the generation that laid them out (2 or 3)
    uint32_t log_w = 0;        // what they were laid out for: the window width (third generation) / the scatter shape (second)
    uint32_t age = 0;          // calls that have used them
    uint64_t tune_gen = 0;     // the option generation they were planned under
};
// true: this call reuses the tables.  healthy: fewer than 2 % of the pairs of the calls completed since the survey took
// an overflow / window-miss path.
bool survey_reusable(const SurveyTables &t, int gen, uint32_t layout, uint64_t call_tune_gen, uint32_t survey_every,
                     bool healthy);

// Third generation -> first generation: names without skew give the second level nothing to count in place.  Judged over
// the calls that reported since the last look: at least 2^22 pairs, more than 3/4 of them forwarded to the reduce pass,
// by HEALTHY calls at an unchanged window width (a stale survey forwards most records whatever the names' skew).
bool names_without_skew(uint64_t pairs, uint64_t forwarded, bool healthy, bool same_width);

// launches a set of the lanes' shared survey tables serves before a stale-survey report may end its reuse
constexpr uint32_t kLaneStaleMinAge = 8;

// fewer than 2 % of `pairs` took an exact-but-slow path
inline bool healthy_share(uint64_t bad, uint64_t pairs) { return bad * 50 <= pairs; }

} // namespace lh
