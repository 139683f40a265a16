// lh_kernels.h -- launch interface between the host runtime (lh_engine.cc) and
// the gfx950 kernels (lh_kernels.hip).  Internal; the public ABI is
// include/loghisto_gpu.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "lh_cells.h"

// Cells from one row of an epoch buffer to the next.  A row is uint64[65536] (bin = key ^ 0x8000), but the rows are NOT
// packed back to back: names with similar value distributions keep their occupied windows at the same offset inside the
// row, and windows exactly 512 KiB apart share the low 19 address bits -- measured on 65 536 windows of 600 cells
// (tools/row_stride.hip, profiles/r05_row_stride.jsonl): the reduce passes' atomic flush 275 -> 238 us, the clear
// 94 -> 66 us, the wave-per-row read 79 -> 74 us with rows 256 bytes further apart; packing the windows tightly buys
// nothing beyond that.  Exposed to callers that index the device rows themselves as lh_row_stride().
#ifndef LH_ROW_SKEW_CELLS
#define LH_ROW_SKEW_CELLS 32 /* (>= 4: k_extract_wave reads whole 4-bin groups past bin 65 535 and relies on the zero padding
                                 behind the row and behind the D table -- lh_kernels.hip asserts it; an A/B build against
                                 the packed rows of ABI <= 4 is -DLH_ROW_SKEW_CELLS=4, not 0) */
#endif
#define LH_ROW_STRIDE ((size_t)65536 + LH_ROW_SKEW_CELLS)

namespace lh {

struct ExtractOut {          // device mirror of lh_stats, one per metric
    uint64_t count;
    double   sum;
    double   avg;
    uint64_t agg_sum_add;
    uint32_t nbuckets;
    uint32_t present;
};

// The ids of an (id, value) stream on the device: uint32 (width 4) or uint16 (width 2; at most 65 536 names).
struct Ids {
    const void *p = nullptr;
    uint32_t width = 4;
    Ids() = default;
    Ids(const uint32_t *q) : p(q), width(4) {}
    Ids(const uint16_t *q) : p(q), width(2) {}
    Ids plus(size_t n) const { Ids r = *this; r.p = static_cast<const char *>(p) + n * width; return r; }
    bool pair_aligned() const { return ((uintptr_t)p & (2u * width - 1u)) == 0; } // two ids per load
    const uint32_t *u32() const { return static_cast<const uint32_t *>(p); }
    const uint16_t *u16() const { return static_cast<const uint16_t *>(p); }
};

// Table generation (once per engine).
hipError_t launch_gen_tables(double *d_Tx, double *d_D, hipStream_t s);

// K1: ingest.  counts: [nmetrics][LH_ROW_STRIDE] cells -- uint64, or uint32 when bit 0 of the pointer is set (lh_cells.h:
// every `counts` / `row` below is such a tagged pointer); ranges: [nmetrics][2] u32 (lo,hi bin).
hipError_t launch_ingest_single(const double *d_v, size_t n, uint64_t *row, uint32_t *range,
                                const double *d_Tx, int num_cus, hipStream_t s);
hipError_t launch_ingest_pairs(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                               uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                               int num_cus, hipStream_t s);
// The same without scratch or survey, but through a per-workgroup LDS table of (name, bin) -> count (lh_kernels_part2.h:
// k_scatter_clustered on its own): what the engine's "direct" path launches.  Exact; robust against streams that fall into
// few cells, where one global atomic per sample serialises.
hipError_t launch_ingest_pairs_cells(Ids d_ids, const double *d_v, size_t n, uint64_t *counts, uint32_t *ranges,
                                     uint32_t nmetrics, const double *d_Tx, uint32_t *d_err, int num_cus, hipStream_t s);

// Dispatch settings of the partitioned mixed ingest (lh_set_option).  Every setting only chooses among
// exact kernel paths; none of them is read from the environment in the product build.
struct PartTuning {
    uint32_t names_per_part = 4;    // names per LDS-reduce partition: 4 gives every name a 4 096-bin window in P2
    uint32_t two_level_above = 32;  // second scatter level when a level-1 partition holds more names than this
    uint32_t hot_min_tiles = 32;    // hot-name windows in P1 when every workgroup gets at least this many tiles
    size_t part_min_samples = 0;    // smallest launch that is partitioned at all (0 = the defaults: 2^20, 3 * 2^20 above 8 192 names); below: the direct path
    bool hot = true;                // hot-name windows allowed at all
    bool v2 = true;                 // survey + 2-byte-record path (lh_kernels_part2.h) for <= 8 192 names
    size_t v2_min_samples = 0;      // 0 = default (2^20: where it overtakes the direct path's cell table)
    uint32_t v2_shape = 2;          // bit 0: two 512-thread workgroups per CU (128 partitions) instead of one 1 024-thread (256);
                                    // bit 1: fixed per-partition regions (k_scatter3) instead of the exact per-tile layout
    bool v3 = true;                 // hashed survey + region scatter + in-place second level (lh_kernels_part3.h) for 8 193 .. 65 536 names
    size_t v3_min_samples = 0;      // 0 = default (2^24)
    uint32_t v3_log_w = 10;         // log2 of the second level's window width, 10 .. 14: the engine follows the survey's report
    bool v2_yield = false;          // 1 025 .. 8 192 names: the last survey saw more than 1/8 of the sampled mass outside the second
                                    // generation's cold windows -- such launches are the third generation's (the engine follows the report)
    size_t v3_direct_max = 0;       // third generation: launches of at most this many pairs end in k_part_direct3 (one global
                                    // atomic per forwarded record) instead of the windowed reduce pass; 0 = default (2^22), 1 = never
    uint32_t v3_g1_cap = 0;         // third generation: at most this many level-1 workgroups (0 = one per CU).  A host-fed
                                    // launch reads its pairs over PCIe: 8 workgroups keep 0.6 MB in flight, several times
                                    // what the link's latency needs, and every level-1 workgroup owns its CU's whole LDS --
                                    // CUs the other lanes' later passes can use instead (measured 8 / 16 / 32 / 256:
                                    // profiles/r05_hostfed_native.jsonl)
};

// Partitioned mixed ingest (lh_kernels_part.hip).  part_scratch_bytes returns 0 when the launch
// should use the direct kernel instead (small n, one name, or too many names per partition).
size_t part_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune);
bool part_aligned(Ids d_ids, const double *d_v); // two ids per load (8 B / 4 B), 16-B values: vector loads

// Few names (<= 32): single streaming pass with every name's window in LDS (lh_kernels_small.hip).
bool small_supported(size_t n, uint32_t nmetrics, Ids d_ids, const double *d_v);
hipError_t launch_ingest_pairs_small(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                                     uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                                     int num_cus, hipStream_t s);
hipError_t launch_ingest_pairs_part(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                                    uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                                    void *scratch, size_t scratch_bytes, int num_cus, const PartTuning &tune,
                                    hipStream_t s);

// Counters (metrics.go:251-269): cur[id] += amount, flag[id] = touched; the fold adds the interval into the lifetime
// store (metrics.go:435-458).
hipError_t launch_count_add(const uint32_t *d_ids, const uint64_t *d_amounts, size_t n, uint64_t *cur, uint32_t *flag,
                            uint32_t ncounters, uint32_t *d_err, int num_cus, hipStream_t s);
hipError_t launch_count_fold(const uint64_t *cur, const uint32_t *flag, uint64_t *life, uint32_t *known,
                             uint32_t ncounters, hipStream_t s);

// Second generation (lh_kernels_part2.h): one survey per launch, 2-byte records, line-granular copy-out.
// part2_scratch_bytes returns 0 when the launch should take the first-generation path.
size_t part2_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune);
// survey_n: pairs of [d_ids, d_v) to survey before the scatter (the whole call), 0 = reuse the scratch block's tables
// region_stat: device-visible counter (pinned host memory) the region kernel adds its overflowed records to, or null
hipError_t launch_ingest_pairs_part2(Ids d_ids, const double *d_v, size_t n, size_t survey_n,
                                     uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                     uint32_t *d_err, void *scratch, size_t scratch_bytes, int num_cus,
                                     const PartTuning &tune, unsigned long long *region_stat, hipStream_t s);

// Third generation (lh_kernels_part3.h): 8 193 .. 65 536 names.  part3_scratch_bytes returns 0 when the launch should
// take another path.  span_stat: device-visible word (pinned host memory) that receives the survey's window class.
size_t part3_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune);
// The block in two parts (the host-fed lanes keep the records of their launches in blocks of their own and share one set
// of survey tables): tables = a function of the name count only; records = the rest.  tables == nullptr below: one block.
size_t part3_tables_bytes(uint32_t nmetrics);
size_t part3_records_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune);
hipError_t launch_ingest_pairs_part3(Ids d_ids, const double *d_v, size_t n, size_t survey_n,
                                     uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                     uint32_t *d_err, void *scratch, size_t scratch_bytes, void *tables, int num_cus,
                                     const PartTuning &tune, unsigned long long *region_stat, uint32_t *span_stat,
                                     hipStream_t s);

// The survey's window-width class of n pairs, alone (-> *span_stat): for a call that must choose a width before any survey
// has reported one.  `tables`: part3_tables_bytes(nmetrics) of device memory, scratch of the probe.
hipError_t launch_part3_probe(Ids d_ids, const double *d_v, size_t n, uint32_t nmetrics, const double *d_Tx, void *tables,
                              int num_cus, const PartTuning &tune, uint32_t *span_stat, hipStream_t s);

// K2: extract.  One workgroup per metric.
// ExtractNotify (optional): when the outputs live in host-mapped memory the last workgroup to finish stores
// `seq` into *host_flag (system-scope release after every workgroup's results), so the host can spin on a word
// of pinned memory instead of going through a stream synchronisation.  done_ctr: device uint32, zero between calls.
struct ExtractNotify { uint32_t *done_ctr = nullptr; uint32_t *host_flag = nullptr; uint32_t seq = 0; };
// The COMPACT form of the results (lh_extract_rows_compact): what cannot be derived on the host, as separate arrays --
// count, sum, occupied buckets, the selected keys and one word of valid bits per metric (bit i: percentile i has a
// bucket).  avg, uint64(sum), present and the percentile VALUES (D[key]) follow from these bit for bit.  With
// count == nullptr the kernels write the full form (out / pvals / pkeys / pvalid).
struct ExtractCompact {
    uint64_t *count = nullptr;
    double *sum = nullptr;
    uint32_t *nbuckets = nullptr;
    uint32_t *vbits = nullptr;
    int16_t *pkeys = nullptr;
};
hipError_t launch_extract(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics,
                          const double *h_p /* host */, uint32_t np, const double *d_D, ExtractOut *out,
                          double *pvals, int16_t *pkeys, uint8_t *pvalid, const uint32_t *err_in,
                          uint32_t *err_out, hipStream_t s, ExtractNotify notify = ExtractNotify(),
                          ExtractCompact compact = ExtractCompact());

// K5: occupied cells of every row as CSR arrays (ascending key within a row).
hipError_t launch_count_cells(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics, uint32_t *ncells,
                              hipStream_t s);
hipError_t launch_compact_cells(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics,
                                const uint64_t *offsets, int16_t *keys, uint64_t *vals, hipStream_t s);

// K4 helpers (multi-GPU merge): the collective itself is RCCL, called from lh_engine.cc.
// (lo, hi) -> (lo, ~hi) for the MIN all-reduce; with_extra: dst[2 * nrows] = ~extra rides along (its all-reduced
// complement is the largest `extra` of any rank).
hipError_t launch_ranges_flip_hi(uint32_t *dst, const uint32_t *src, uint32_t nrows, bool with_extra, uint32_t extra,
                                 hipStream_t s);
// The merge's first step: dst = (lo, ~hi) per row | ~extra | ~(row r's largest cell on this rank, clipped to 2^32 - 1) per
// row -- 3 * nrows + 1 words for ONE MIN all-reduce.  narrow == false: the third part is written as 0 (no cell is read).
hipError_t launch_merge_prep(uint32_t *dst, const uint32_t *ranges, const uint64_t *counts, uint32_t nrows, uint32_t extra,
                             bool narrow, hipStream_t s);
// Per-row windows packed CSR, in WIRE WORDS: uint64 (one cell each), or uint32 when nranks x ~extra_src[0] < 2^32 -- then
// row r travels at cls[r] = 8, 16 or 32 bits per cell, the narrowest that holds nranks x ~rowmaxc[r] (the all-reduced
// largest per-rank cell of the row; rowmaxc null: 32).  P[nrows+1] = exclusive prefix of the rows' words; the nblocks
// owner blocks are contiguous row ranges of equal PACKED size: brow[nblocks+1] their first rows, bstart[nblocks+1] the
// prefix there.  info[16] = {total words, largest block (words), widest row (cells), occupied rows, ~extra_src[0], first
// row and end row of block `rank`, total cells, rows at 8 bits, rows at 16 bits}; host_flag (device mapping of pinned
// memory, or null) receives `seq` after info is stored.  work: scratch of 5 120 uint64 (per-row-block partial sums).
hipError_t launch_merge_plan(const uint32_t *ranges, uint32_t nrows, uint32_t nblocks, uint32_t rank, uint32_t nranks,
                             const uint32_t *extra_src, const uint32_t *rowmaxc, uint8_t *cls, uint64_t *P,
                             uint64_t *bstart, uint32_t *brow, uint64_t *work, uint64_t *info, uint32_t *host_flag,
                             uint32_t seq, hipStream_t s);
// buf[k * bstride + ...]: block k's rows back to back, the rest of each block zeroed; words32: uint32 words on the wire.
hipError_t launch_pack_rows(const uint64_t *counts, const uint32_t *ranges, const uint8_t *cls, const uint64_t *P,
                            const uint64_t *bstart, const uint32_t *brow, uint32_t nrows, uint32_t nblocks,
                            uint64_t bstride, void *buf, bool words32, hipStream_t s);
hipError_t launch_unpack_rows(uint64_t *counts, const uint32_t *ranges, const uint8_t *cls, const uint64_t *P,
                              const uint64_t *bstart, uint32_t kblock, uint32_t first_row, uint32_t nrows_out,
                              const void *buf, bool words32, hipStream_t s);

// K6 (lh_kernels_fmt.hip): ProcessedMetricSet keys + Go "%f" + wire lines, one thread per (metric, key).
struct SerKey { uint16_t pre_off, pre_len, post_off, post_len; }; // key = pre + name + post, strings in the blob
constexpr uint32_t SER_DOTS = 1u;                                 // '_' -> '.' in the name (graphite.go:42)
constexpr uint32_t SER_COUNTERS = 2u;                             // counter mode: keys "<name>" and "<name>_rate"
constexpr uint32_t SER_MAX_KEYS = 3 + 32 + 3;                     // _count _sum _avg, percentiles, _agg_*
constexpr uint32_t SER_BLOB_MAX = 2048;
constexpr uint32_t SER_FMT_SLOT = 336;                            // longest "%f" of a float64 is 317 bytes
struct SerArgs {
    const ExtractOut *stats;  // [nmetrics] rows first .. first+nmetrics
    const double *pvals;      // [nmetrics][np]
    const uint8_t *pvalid;    // [nmetrics][np]
    const uint64_t *life;     // [max_metrics][2] lifetime (count, sum); read only when nkeys includes _agg_*
    const char *names;        // name bytes of every interned metric, back to back
    const uint32_t *name_off; // [num_names + 1]
    const char *blob;         // prefix | sep | suffix | key strings
    // counter mode (SER_COUNTERS): key 0 = lifetime total of a known counter (metrics.go:487-489), key 1 = the
    // interval's amount of a counter touched this interval (metrics.go:491-493)
    const uint64_t *c_total;
    const uint64_t *c_rate;
    const uint32_t *c_known;
    const uint32_t *c_present;
    uint32_t blob_len, first, nmetrics, np, nkeys, flags;
    uint32_t prefix_off, prefix_len, sep_off, sep_len, suffix_off, suffix_len;
    SerKey keys[SER_MAX_KEYS];
};
uint32_t ser_blocks(uint64_t nlines);
// lens[nlines], bsum[ser_blocks], boff[ser_blocks + 1]; boff[ser_blocks] receives the total byte count
hipError_t launch_ser_len(const SerArgs &a, uint32_t *lens, uint32_t *bsum, uint64_t *boff, hipStream_t s);
hipError_t launch_ser_write(const SerArgs &a, const uint32_t *lens, const uint64_t *boff, char *out, hipStream_t s);
hipError_t launch_life_add(const ExtractOut *stats, uint64_t *life, uint32_t n, hipStream_t s);
hipError_t launch_format_f(const double *d_v, char *d_out, uint32_t *d_lens, uint32_t n, hipStream_t s);

// K3: clear the dirty span of every row and reset the ranges.
hipError_t launch_clear(uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, hipStream_t s);
// A narrow store's dirty spans into the (zeroed) wide store, and zeroed behind the copy (lh_cells.h).
hipError_t launch_widen_rows(uint32_t *narrow, uint64_t *wide, const uint32_t *ranges, uint32_t nmetrics, hipStream_t s);
hipError_t launch_init_ranges(uint32_t *ranges, uint32_t nmetrics, hipStream_t s);
hipError_t launch_mark_dirty(uint32_t *ranges, uint32_t first, uint32_t nrows, uint32_t lo, uint32_t hi, hipStream_t s);

// Codec-only kernels (parity tests).
hipError_t launch_compress(const double *d_v, int16_t *d_keys, size_t n, const double *d_Tx, bool golog, hipStream_t s);
hipError_t launch_vlog_selftest(double *d_maxerr, hipStream_t s);

} // namespace lh
