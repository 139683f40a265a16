/*
 * loghisto_gpu.h -- C ABI of liblhgpu.so, the MI355X (gfx950) engine behind
 * loghisto's histogram hot path.
 *
 * The reference (spacejam/loghisto, Go) has no FFI on this path: the boundary is
 * the bodies of three Go functions, which a cgo binding replaces with calls into
 * this library (the binding is shown in INTEGRATION.md):
 *
 *   (*MetricSystem).Histogram        /root/reference/metrics.go:273-295
 *       -> lh_intern (once per name) + lh_submit / lh_submit_pairs (batched)
 *   collectRawMetrics, histogram part /root/reference/metrics.go:460-463
 *       -> lh_flip            (the epoch boundary: steal the interval's cells)
 *   processHistograms + percentile   /root/reference/metrics.go:336-418
 *       -> lh_extract         (count, sum, avg, uint64(sum), percentiles)
 *   RawMetricSet.Histograms          /root/reference/metrics.go:54-60
 *       -> lh_buckets         (occupied (key,count) cells of one metric)
 *   processMetrics/addAggregates key loops + the serializers' per-key line
 *                                    /root/reference/metrics.go:483-506, 590-608,
 *                                    graphite.go:37-48, opentsdb.go:45-58
 *       -> lh_snapshot_accumulate + lh_serialize (keys, Go's %f and the wire
 *          lines assembled on the device; optional, for large name spaces)
 *   compress / decompress            /root/reference/metrics.go:316-332
 *       -> computed on device inside lh_submit*; lh_compress_device and
 *          lh_codec_tables expose the codec for parity tests.
 *
 * Conventions:
 *   - every entry point returns an int status (LH_OK == 0); nothing aborts.
 *     The Go layer logs non-zero codes through glog and carries on, exactly as
 *     it does for percentile()'s error (metrics.go:379-384).
 *   - ingest is lossless and may block (back-pressure) but never drops
 *     (metrics.go:273-295 is synchronous); only emission may be dropped, by the
 *     caller.
 *   - lh_submit* are thread-safe and copy the caller's buffer before returning
 *     (cgo rule: C may not retain Go memory).
 *   - a sample belongs to exactly one snapshot: everything submitted before
 *     lh_flip returns is in that snapshot, everything after is in the next
 *     (metrics.go:460-463).
 *   - plain pointers and sizes only; `stream` arguments are hipStream_t passed
 *     as void* (NULL = the engine's own stream).
 *
 * Bucket keys are the reference's int16 keys.  Dense rows are indexed by
 * bin = (uint16)key ^ 0x8000, so ascending bin == ascending key == ascending
 * decompressed value.
 */
#ifndef LOGHISTO_GPU_H
#define LOGHISTO_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LH_ABI_VERSION 7           /* 7: uint32 cells above 8 192 names (lh_config.cell_bits, lh_snapshot_cells, lh_cell_bytes;
                                      lh_counters.widenings / store_bytes): half the lines under every flush, clear, extract
                                      read and merge pack, lh_create(65 536 names) 32 GiB instead of 64; lh_snapshot_rows still
                                      hands out uint64 rows; 6: lh_extract_rows_compact / lh_expand_compact (the results of many names at 42 B instead of
                                      139 B per name); 5: lh_row_stride() (rows of lh_snapshot_rows are no longer 65 536 cells apart), the
                                      tuning / test options moved to loghisto_gpu_tuning.h, ingest falls back to the
                                      scratch-free kernel when scratch cannot be had; 4: uint16-id pairs (lh_*pairs16*) */
#define LH_NKEYS 65536            /* int16 key space (metrics.go:316)        */
#define LH_NTHRESH 70980          /* extended-key thresholds incl. sentinel  */
#define LH_MAX_PERCENTILES 32

enum {
    LH_OK = 0,
    LH_EINVAL = 1,      /* bad argument                                      */
    LH_ENOMEM = 2,      /* host or device allocation failed                  */
    LH_EDEVICE = 3,     /* HIP runtime error (see lh_last_error)             */
    LH_ENODEVICE = 4,   /* no usable gfx950 device                           */
    LH_EBUSY = 5,       /* lh_flip: every epoch buffer still has a live snapshot */
    LH_ERANGE = 6,      /* metric id >= max_metrics / table full             */
    LH_ESTATE = 7       /* call not valid in this state                      */
};

typedef struct lh_engine lh_engine;
typedef struct lh_snapshot lh_snapshot;

typedef struct lh_config {
    uint32_t struct_size;   /* sizeof(lh_config), for ABI growth              */
    int32_t  device;        /* HIP device ordinal                             */
    uint32_t max_metrics;   /* dense rows per epoch buffer (512 KiB each)     */
    uint32_t num_buffers;   /* epoch buffers, >= 2                            */
    uint32_t num_lanes;     /* host staging lanes (one HIP stream each)       */
    uint32_t max_counters;  /* counter names (metrics.go:115), 8 B each per epoch buffer; 0 = no counters */
    uint64_t lane_samples;  /* samples per pinned half-buffer of a lane       */
    uint32_t cell_bits;     /* (ABI 7) width of a bucket cell in HBM: 64 = the reference's uint64 (metrics.go:278); 32 = an
                               epoch buffer counts in uint32 cells while its interval holds fewer than 2^32 samples -- no cell
                               can wrap -- and moves to a uint64 store of its own (allocated then, kept) before a submit could
                               pass that: exact for any stream, half the HBM and half the lines per window until then;
                               0 = default: 64 up to 8 192 names, 32 above.  A struct_size without this field means 0. */
    uint32_t reserved0;     /* 0 */
} lh_config;

/* Per-metric result of processHistograms (metrics.go:336-376). */
typedef struct lh_stats {
    uint64_t count;        /* totalCount                                      */
    double   sum;          /* totalSum (fixed parallel order)  */
    double   avg;          /* sum / float64(count); NaN when count == 0       */
    uint64_t agg_sum_add;  /* uint64(totalSum), amd64 conversion (metrics.go:374) */
    uint32_t nbuckets;     /* occupied buckets                                */
    uint32_t present;      /* 1 iff the metric received a sample this epoch   */
} lh_stats;

int lh_abi_version(void);
const char *lh_strerror(int code);
/* Thread-local text of the last HIP failure seen by this thread ("" if none). */
const char *lh_last_error(void);

int lh_default_config(lh_config *cfg);
/* NewMetricSystem (metrics.go:143) calls this; Stop (metrics.go:651) -> lh_destroy. */
int lh_create(const lh_config *cfg, lh_engine **out);
int lh_destroy(lh_engine *e);

/* name -> dense metric id (idempotent, thread-safe).  Replaces the string-keyed
 * outer map of histogramCache (metrics.go:119). */
int lh_intern(lh_engine *e, const char *name, size_t len, uint32_t *id);
int lh_lookup(lh_engine *e, const char *name, size_t len, uint32_t *id); /* LH_ERANGE if unknown */
int lh_num_metrics(lh_engine *e, uint32_t *n);
/* Copies up to cap bytes of the name (no terminator); *len receives the full length. */
int lh_metric_name(lh_engine *e, uint32_t id, char *buf, size_t cap, size_t *len);

/* Host-memory ingest: Histogram(name, v[i]) for i < n (metrics.go:273). */
int lh_submit(lh_engine *e, uint32_t id, const double *v, size_t n);
/* Mixed batch: Histogram(name(ids[i]), v[i]). */
int lh_submit_pairs(lh_engine *e, const uint32_t *ids, const double *v, size_t n);
/* In-place staging of a mixed batch (the other half of SURVEY.md 8b "Ownership": "the ring is C-allocated
 * (hipHostMalloc) and Go writes into it in place"; call shape of Histogram, metrics.go:273).  lh_reserve_pairs hands
 * out the free tail of one of the engine's pinned staging buffers: *ids / *vals point at room for *granted <= want
 * pairs, which the caller fills front to back -- the only host-side store a sample ever sees; the ingest kernel
 * later reads that memory over PCIe in place -- and lh_commit_pairs(token, n) publishes the first n <= *granted of
 * them (n = 0 gives the reservation back).  Between the two calls the buffer belongs to the caller: a flip, and
 * other producers that are sent to the same buffer, wait for the commit, so a producer commits before it blocks on
 * anything else.  Every reservation is committed exactly once; commit from any thread.  ids are not validated on
 * the host: an id >= max_metrics is skipped by the kernel and reported as LH_ERANGE by the next lh_sync / lh_flip
 * / lh_extract, as for lh_submit_pairs_device. */
int lh_reserve_pairs(lh_engine *e, size_t want, uint32_t **ids, double **vals, size_t *granted, uint32_t *token);
int lh_commit_pairs(lh_engine *e, uint32_t token, size_t n);
/* Device-memory ingest for GPU-resident producers; asynchronous on `stream`.
 * Ordering contract: the kernel is enqueued on `stream` (NULL = the engine's own non-blocking
 * stream, which does NOT synchronise with the legacy default stream).  The producer of the buffers
 * must therefore either run on the same stream or be complete before the call, and the buffers must
 * stay valid until the stream reaches this point (lh_sync / lh_flip + lh_extract imply it).  A
 * caller-owned stream must outlive the next lh_flip, which records an event on it. */
int lh_submit_device(lh_engine *e, uint32_t id, const double *d_v, size_t n, void *stream);
int lh_submit_pairs_device(lh_engine *e, const uint32_t *d_ids, const double *d_v, size_t n, void *stream);
/* The same three ways in with uint16 ids: 10 bytes per pair instead of 12 (SURVEY.md 8d "mixed stream: 12 B/sample ...
 * 10 B if ids are uint16, legal for <= 65 536 names").  Every mixed-ingest kernel is a template on the id width and
 * reads the narrow ids directly (two per 4-byte load), so nothing is widened on the way: the host-fed path moves 17 %
 * fewer bytes over PCIe, the device-resident path reads 17 % fewer from HBM.  Replaces the same Go lines as the
 * uint32 forms (body of Histogram, /root/reference/metrics.go:273-295); a binding uses them whenever the engine was
 * created with max_metrics <= 65 536.  A staging buffer holds one id width at a time: a producer that switches width
 * makes the library launch what the buffer holds first.  lh_commit_pairs16 == lh_commit_pairs (the token knows). */
int lh_submit_pairs16(lh_engine *e, const uint16_t *ids, const double *v, size_t n);
int lh_reserve_pairs16(lh_engine *e, size_t want, uint16_t **ids, double **vals, size_t *granted, uint32_t *token);
int lh_commit_pairs16(lh_engine *e, uint32_t token, size_t n);
int lh_submit_pairs16_device(lh_engine *e, const uint16_t *d_ids, const double *d_v, size_t n, void *stream);
/* Counters on the device.  Counter names have their own dense id space.
 *   (*MetricSystem).Counter           /root/reference/metrics.go:251-269  -> lh_intern_counter + lh_submit_counts
 *   collectRawMetrics, counter part   /root/reference/metrics.go:425-458  -> lh_flip (steals the interval's amounts
 *                                     together with the histogram cells) + lh_counters_collect (Rates = the
 *                                     interval's amounts of the names touched, Counters = lifetime totals of every
 *                                     name ever touched; the fold into the lifetime store happens once per snapshot)
 *   processMetrics / serializers      /root/reference/metrics.go:487-493  -> lh_serialize_counters ("<name>" and
 *                                     "<name>_rate" lines, Go's %f of float64(count))
 * Same conventions as the histogram ingest: lossless, thread-safe, the caller's buffers are copied. */
int lh_intern_counter(lh_engine *e, const char *name, size_t len, uint32_t *id);
int lh_num_counters(lh_engine *e, uint32_t *n);
int lh_counter_name(lh_engine *e, uint32_t id, char *buf, size_t cap, size_t *len);
/* Counter(name(ids[i]), amounts[i]) for i < n. */
int lh_submit_counts(lh_engine *e, const uint32_t *ids, const uint64_t *amounts, size_t n);
int lh_submit_counts_device(lh_engine *e, const uint32_t *d_ids, const uint64_t *d_amounts, size_t n, void *stream);
/* Counters [first, first+n) of the snapshot: rate[i] = amount added this interval, present[i] = 1 iff the name was
 * touched this interval (metrics.go:430-433), total[i] = lifetime total after this interval, known[i] = 1 iff the
 * name has ever been touched (metrics.go:435-458).  Any output may be NULL. */
int lh_counters_collect(lh_snapshot *s, uint32_t first, size_t n, uint64_t *rate, uint8_t *present, uint64_t *total,
                        uint8_t *known);
/* lh_serialize_counters (declared after lh_line_format below) writes them as wire lines. */

/* Push partially filled staging buffers to the device (asynchronous). */
int lh_flush(lh_engine *e);
/* Wait until every sample submitted so far is in the bucket arrays. */
int lh_sync(lh_engine *e);

/* Epoch boundary (metrics.go:460-463).  Returns LH_EBUSY if no epoch buffer is
 * free; the current epoch simply keeps accumulating in that case. */
int lh_flip(lh_engine *e, lh_snapshot **out);
/* processHistograms for metrics [0, nmetrics) of the snapshot.
 *   p[np]                     percentiles in [0,1] (metrics.go:145-155)
 *   stats[nmetrics]
 *   pvals[nmetrics*np]        always exactly some decompress(key)
 *   pkeys[nmetrics*np]        (may be NULL) the selected int16 keys
 *   pvalid[nmetrics*np]       (may be NULL) 0 where the reference returns
 *                             "Invalid percentile" (metrics.go:417) or count==0 */
int lh_extract(lh_snapshot *s, const double *p, size_t np, lh_stats *stats,
               double *pvals, int16_t *pkeys, uint8_t *pvalid, size_t nmetrics);
/* Same for metrics [first, first+nmetrics): the rows a rank owns after a
 * reduce-scatter merge. */
int lh_extract_rows(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np,
                    lh_stats *stats, double *pvals, int16_t *pkeys, uint8_t *pvalid);
/* The same without the last copy: the results are handed out IN PLACE, in the engine's pinned result buffer (a
 * cgo caller wraps them as slices without copying; at 65 536 names the copy into caller arrays would cost more
 * than the scan).  The pointers point into the engine's one pinned result block, which every call that returns
 * results through it rewrites (and may re-allocate): they stay valid until the next lh_extract*, lh_buckets*,
 * lh_serialize*, lh_counters_collect, lh_lifetime, lh_format_f or lh_snapshot_merge call on this engine, or
 * lh_release of the snapshot, whichever comes first.  A host layer that needs lifetime totals or counters for the
 * same interval reads the view first (or copies what it keeps) and then makes those calls. */
typedef struct lh_extract_view {
    const lh_stats *stats;   /* [nmetrics]      */
    const double *pvals;     /* [nmetrics * np] */
    const int16_t *pkeys;    /* [nmetrics * np] */
    const uint8_t *pvalid;   /* [nmetrics * np] */
    size_t nmetrics, np;
} lh_extract_view;
int lh_extract_rows_view(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np,
                         lh_extract_view *view);
/* The COMPACT form of the same results, for large name spaces (round 6).  processHistograms emits 3 + P floats per
 * name (metrics.go:349-356, 378-385), but only count, sum and the P selected KEYS are information: avg is
 * sum / float64(count), uint64(sum) a conversion, and every percentile value is exactly decompress(key) -- the table
 * D[] that lh_codec_tables exports.  This call brings 8 + 8 + 4 + 4 + 2 P bytes per name to the host (42 B at the nine
 * default percentiles against 139 B: at 65 536 names 2.75 MB instead of 9.1 MB over PCIe, which was more than half of
 * the flip -> results latency), in place in the engine's pinned block under lh_extract_rows_view's lifetime rule.
 *   count[nmetrics], sum[nmetrics], nbuckets[nmetrics]
 *   pkeys[nmetrics * np]     the selected int16 keys (0 where the percentile has no bucket)
 *   pvalid_bits[nmetrics]    bit i set iff percentile i has a bucket (np <= LH_MAX_PERCENTILES = 32)
 * lh_expand_compact derives the full form from it ON THE HOST, bit for bit what lh_extract_rows returns for the same
 * snapshot (present = count != 0, avg = sum / float64(count), agg_sum_add = uint64(sum) with the amd64 conversion,
 * pvals = D[key]); a binding that formats keys itself reads D[] once (lh_codec_tables) and never expands.  Any of
 * stats / pvals / pkeys / pvalid may be NULL. */
typedef struct lh_extract_compact {
    const uint64_t *count;
    const double *sum;
    const uint32_t *nbuckets;
    const uint32_t *pvalid_bits;
    const int16_t *pkeys;
    size_t nmetrics, np;
} lh_extract_compact;
int lh_extract_rows_compact(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, size_t np,
                            lh_extract_compact *view);
int lh_expand_compact(lh_engine *e, const lh_extract_compact *c, lh_stats *stats, double *pvals, int16_t *pkeys,
                      uint8_t *pvalid);
/* Occupied cells of one metric, ascending key. *n receives the number of
 * occupied cells even when it exceeds cap. */
int lh_buckets(lh_snapshot *s, uint32_t id, int16_t *keys, uint64_t *counts, size_t cap, size_t *n);
/* The same for metrics [first, first+nmetrics) at once, compacted on the device: CSR arrays with
 * offsets[nmetrics+1]; metric first+i owns keys/counts[offsets[i] .. offsets[i+1]).  *total receives
 * the number of occupied cells; if it exceeds cap only offsets/total are filled (size and call again).
 * This is RawMetricSet.Histograms (metrics.go:54-60) for every name in one crossing. */
int lh_buckets_all(lh_snapshot *s, uint32_t first, size_t nmetrics, uint64_t *offsets, int16_t *keys,
                   uint64_t *counts, size_t cap, size_t *total);
/* Dense device view of the snapshot for the multi-GPU merge: row r of metric r
 * is (uint64 *)d_counts + r * lh_row_stride(), LH_NKEYS cells long (bin = key ^ 0x8000).
 * The rows are NOT back to back: lh_row_stride() > LH_NKEYS (ABI 5; rows exactly 512 KiB
 * apart put every name's occupied window on the same low address bits -- DESIGN.md 4).
 * After an in-place reduction the caller must call lh_snapshot_mark_dirty so that
 * extract/clear cover the merged cells. */
int lh_snapshot_rows(lh_snapshot *s, void **d_counts, uint32_t *nrows);
size_t lh_row_stride(void); /* in cells */
/* (ABI 7) On an engine of 32-bit cells (lh_config.cell_bits) lh_snapshot_rows first moves the snapshot to its uint64 store
 * (one pass over the occupied windows; the store is allocated the first time: LH_ENOMEM if it cannot be had), so that the
 * view above stays what it was.  lh_snapshot_cells hands out the cells as they are: row r is at
 * (char *)d_cells + r * lh_row_stride() * cell_bytes, cell_bytes = 4 or 8.  lh_cell_bytes: the width an engine's epoch
 * buffers START every interval with (4 or 8). */
int lh_snapshot_cells(lh_snapshot *s, void **d_cells, uint32_t *nrows, uint32_t *cell_bytes);
int lh_cell_bytes(lh_engine *e);
int lh_snapshot_ranges(lh_snapshot *s, void **d_ranges /* uint32[nrows][2] lo,hi bins */);
int lh_snapshot_mark_dirty(lh_snapshot *s, uint32_t first_row, uint32_t nrows, uint32_t lo_bin, uint32_t hi_bin);
/* K4 -- multi-GPU merge of a snapshot across the ranks of an RCCL communicator (one process per GPU).
 * Ingest is data-parallel: every rank buckets its own slice of the stream for ALL names; the only
 * exchange is this integer SUM of the occupied window of the uint64 bucket matrix at the flip
 * (cells are a commutative sum, metrics.go:278, 292).  The reference is single-process: no counterpart.
 *   comm   ncclComm_t (as void*) created by the caller with the RCCL named by lh_set_rccl_library
 *   plan   LH_MERGE_ALLREDUCE: every rank ends with every merged row;
 *          LH_MERGE_REDUCE_SCATTER: rank r ends with the merged rows of a contiguous block of names and
 *          extracts those with lh_extract_rows.  The blocks tile [0, nrows) in rank order and hold equal numbers of
 *          PACKED CELLS, not of names (names ranked by frequency would otherwise put every wide window into block 0
 *          and pad the other blocks up to it): use the returned [first, last)
 *   first_owned / last_owned  receive the [first, last) rows holding merged data on this rank
 * Runs on the snapshot's stream; dirty ranges are merged first (one MIN all-reduce on (lo, ~hi)), then every
 * row's own merged window travels, packed back to back; the window plan (prefix sums, block sizes) is computed
 * on the device and only two totals come back to size the collective.  Extract/clear stay exact. */
enum { LH_MERGE_ALLREDUCE = 0, LH_MERGE_REDUCE_SCATTER = 1 };
int lh_snapshot_merge(lh_snapshot *s, void *comm, int nranks, int rank, int plan, uint32_t nrows,
                      uint32_t *first_owned, uint32_t *last_owned);
/* What the last lh_snapshot_merge on this engine moved.  Every row travels with its OWN merged window
 * [lo_r, hi_r] (packed back to back), so an outlier sample widens one row, never the matrix -- and, when the wire word
 * is uint32, at its own cell width: 8, 16 or 32 bits, the narrowest that holds nranks x (the largest cell any rank
 * has in that row; all-reduced with the dirty ranges), so that the uint32 SUM of the collective never carries from one
 * cell of a word into the next.  Most names of a skewed stream hold small counts per interval: config 4's windows
 * (65 536 Zipf names, 8 ranks) travel at ~1.1 bytes per cell instead of 4. */
typedef struct lh_merge_info {
    uint64_t packed_cells;   /* sum over rows of the merged window widths, in cells                          */
    uint64_t send_bytes;     /* bytes handed to the collective (reduce-scatter: nranks x largest block)      */
    uint64_t recv_bytes;     /* bytes this rank ends up with                                                  */
    uint32_t widest_row;     /* widest merged window, in cells                                                */
    uint32_t occupied_rows;  /* rows with at least one cell on some rank                                      */
    uint64_t padded_words;   /* reduce-scatter: nranks x largest owner block, in wire words (>= packed_words; the
                              * ratio is what the equal-block collective costs over the packed matrix);
                              * all-reduce: packed_words */
    uint32_t cell_bytes;     /* the wire word: 8 (one cell per word), or 4 when no merged cell of the interval can
                              * reach 2^32 (nranks x the largest per-rank sample count of the interval < 2^32)  */
    uint32_t rows_8bit;      /* occupied rows that travelled at 8 bits per cell (4 cells per word) ...          */
    /* Device time of the merge's steps, HIP events on the snapshot stream (lh_snapshot_merge_info waits for the
     * last one): dirty-range all-reduce (with the pass over the rows' largest cells), window plan, pack, the
     * collective, unpack; span = first event to last, host round trip for the plan totals included. */
    float ranges_ms, plan_ms, pack_ms, collective_ms, unpack_ms, span_ms;
    uint64_t packed_words;   /* sum over rows of ceil(window cells x bits / 32) (uint32 words), or packed_cells */
    uint32_t rows_16bit;     /* ... and at 16 bits per cell (2 per word); the other occupied rows: one per word  */
    uint32_t reserved;
} lh_merge_info;
int lh_snapshot_merge_info(lh_snapshot *s, lh_merge_info *out);
/* Path (or soname) of the RCCL shared object the communicator comes from; default "librccl.so".
 * Process-wide; call before the first lh_snapshot_merge. */
int lh_set_rccl_library(const char *path);
/* K6 -- the interval's histogram keys as wire text, assembled and formatted on the device.
 * Replaces, for the histogram part of a ProcessedMetricSet, one fmt.Sprintf + map insert per key:
 *   processMetrics    /root/reference/metrics.go:495-499   <name>_count, _sum, _avg, percentile labels
 *   addAggregates     /root/reference/metrics.go:590-608   <name>_agg_avg, _agg_count, _agg_sum
 *   GraphiteProtocol  /root/reference/graphite.go:37-48    "cockroach.<host>.<key, _ -> .> %f %d\n"
 *   OpenTSDBProtocol  /root/reference/opentsdb.go:45-58    "put <key> %d %f host=<host>\n"
 * One line per key:  prefix  key  sep  value  suffix  with value printed as Go's %f prints a float64
 * (exact decimal expansion, 6 fractional digits, round-half-even, "NaN", "+Inf", "-Inf") and
 * key = fmt.Sprintf(label, name).  Lines are metric-major for metrics [first, first+nmetrics) that have
 * samples in this interval, keys in the order _count, _sum, _avg, labels[0..np), then (with
 * LH_SER_AGGREGATES, for names whose lifetime count is > 0) _agg_avg, _agg_count, _agg_sum; keys whose
 * percentile is invalid (metrics.go:379-384) are omitted.  Go's map iteration order is random, so this
 * is one of the orders the reference can produce.
 *   labels[np]  Go format strings holding exactly one %s ("%s_99.9"); "%%" is a literal %
 *   fmt         NUL-terminated pieces; LH_FMT_UNDERSCORE_TO_DOT applies graphite.go:42 to the key
 *   out/cap     host buffer; *len receives the byte count even when it exceeds cap, in which case
 *               nothing is written (size and call again).  No trailing NUL.
 * Every name in the range must have been interned.  prefix+sep+suffix+labels are limited to 2 KiB. */
typedef struct lh_line_format {
    const char *prefix;     /* "cockroach.<host>."        | "put "              */
    const char *sep;        /* " "                        | " <unix time> "     */
    const char *suffix;     /* " <unix time>\n"           | " host=<host>\n"    */
    uint32_t    flags;      /* LH_FMT_UNDERSCORE_TO_DOT   | 0                   */
    uint32_t    reserved;
} lh_line_format;
enum { LH_FMT_UNDERSCORE_TO_DOT = 1 };
enum { LH_SER_AGGREGATES = 1 };
int lh_serialize(lh_snapshot *s, uint32_t first, size_t nmetrics, const double *p, const char *const *labels,
                 size_t np, const lh_line_format *fmt, uint32_t flags, char *out, size_t cap, size_t *len);
/* One line per exported key, counter-major: "<prefix><name><sep>%f<suffix>" for every known counter and
 * "<prefix><name>_rate<sep>%f<suffix>" for those touched this interval (same lh_line_format and size-then-call
 * protocol as lh_serialize). */
int lh_serialize_counters(lh_snapshot *s, uint32_t first, size_t n, const lh_line_format *fmt, char *out, size_t cap,
                          size_t *len);
/* processHistograms' lifetime side effect (metrics.go:359-376) for every row of the snapshot, kept in
 * HBM: life_sum[m] += uint64(totalSum_m) (amd64 conversion, wrapping add), life_count[m] += count_m.
 * Applied at most once per snapshot; later calls return LH_OK without effect. */
int lh_snapshot_accumulate(lh_snapshot *s);
/* The lifetime stores of metrics [first, first+n) (histogramCountStore, metrics.go:127). */
int lh_lifetime(lh_engine *e, uint32_t first, size_t n, uint64_t *count, uint64_t *sum);
/* Go's %f of n float64 values, formatted on the device (parity tests of the formatter):
 * value i occupies out[i*slot .. i*slot + lens[i]); slot must be >= 336. */
int lh_format_f(lh_engine *e, const double *v, size_t n, char *out, size_t slot, uint32_t *lens);
/* Stream on which the snapshot's extract/clear work is ordered (hipStream_t). */
int lh_snapshot_stream(lh_snapshot *s, void **stream);
/* Returns the snapshot's buffer to the pool (cleared asynchronously). */
int lh_release(lh_snapshot *s);

/* Self-metrics of the engine (exposed by the host layer as gauges through
 * RegisterGaugeFunc, metrics.go:299).  Monotonic since lh_create. */
typedef struct lh_counters {
    uint64_t samples_single;       /* through k_ingest_single                           */
    uint64_t samples_small;        /* mixed, single-pass LDS kernel (<= 32 names)       */
    uint64_t samples_partitioned;  /* mixed, partition + LDS reduce                     */
    uint64_t samples_direct;       /* mixed, one global atomic per sample (small launches) */
    uint64_t launches;             /* ingest launches of any kind                       */
    uint64_t flips;                /* successful lh_flip calls                          */
    uint64_t flips_busy;           /* lh_flip calls that returned LH_EBUSY              */
    uint64_t extracts;             /* lh_extract / lh_extract_rows calls                */
    uint64_t backpressure_waits;   /* submitters that had to wait for a staging half-buffer */
    uint64_t window_misses;        /* samples the single-pass kernel sent to global atomics */
    uint32_t small_path_disabled;  /* 1 while adaptive dispatch routes few-name streams through the partitioned path */
    uint32_t regions_disabled;     /* 1 while a name-clustered stream keeps the mixed ingest on the exact-layout scatter */
    uint64_t scratch_bytes;        /* HBM scratch of the partitioned mixed ingest: the engine's shared block (the host-fed
                                      lanes' own small blocks, LH_OPT_LANE_SCRATCH_BLOCKS, are not counted) */
    uint64_t sublaunches;          /* partitioned sub-launches (a large launch is cut so the scratch stays bounded) */
    uint64_t samples_partitioned_v2; /* of samples_partitioned: through the survey + 2-byte-record path          */
    uint64_t counter_events;         /* (id, amount) events through lh_submit_counts*                               */
    uint64_t region_overflows;       /* records the region scatter counted through the exact out-of-window path     */
    uint64_t samples_partitioned_v3; /* of samples_partitioned: through the 8 193 .. 65 536-name path               */
    uint64_t window_log2;            /* that path's second-level window width (log2 bins) for the next call         */
    uint64_t records_level1;         /* that path: 4-byte records its first level wrote (samples no hot window took)  */
    uint64_t records_level2;         /* ... records its second level forwarded to the reduce pass                     */
    uint64_t level2_overflows;       /* ... records that found a second-level region full (exact path)                */
    uint64_t reduce_window_misses;   /* ... records outside their window in the reduce pass (exact path)              */
    uint64_t surveys_reused;         /* calls that ran on an earlier call's survey (LH_OPT_SURVEY_EVERY)              */
    uint64_t scratch_alloc_failures; /* scratch blocks of the mixed ingest that could not be allocated (ABI 5)         */
    uint64_t samples_fallback;       /* samples that therefore went through the scratch-free kernel: exact, slower     */
    uint64_t survey_stale_pairs;     /* pairs a kept survey's hot windows no longer took (the values moved under it: the
                                        launch reports them, the next call surveys again)                              */
    uint64_t lane_scratch_bytes;     /* (ABI 6) HBM the host-fed lanes' launches hold beside scratch_bytes: their scratch
                                        blocks (up to LH_OPT_LANE_SCRATCH_BLOCKS of them, ~0.2 GiB each at 65 536 names;
                                        LH_OPT_SCRATCH_CAP_BYTES bounds the shared block only) and the two sets of survey
                                        tables the lanes share                                                         */
    uint64_t widenings;              /* (ABI 7) epoch buffers of 32-bit cells that moved to uint64 cells (an interval about to
                                        hold 2^32 samples, a merge whose sums may pass it, lh_snapshot_rows)                */
    uint64_t store_bytes;            /* (ABI 7) HBM of the epoch buffers' cell stores, wide stores of narrow engines included */
} lh_counters;
int lh_get_counters(lh_engine *e, lh_counters *out);

/* Settings.  Every option only chooses among EXACT kernel paths or sizes a buffer: no option (and no environment
 * variable -- the library never calls getenv) can change a result.  Takes effect for later calls; not synchronised
 * with concurrent submits (set options before the producers start).  These are the operational ones; the keys that
 * only exist to steer the mixed ingest's path choice in tests and tuning runs (generation switches, size thresholds,
 * window widths, the allocation-failure hook) live in loghisto_gpu_tuning.h and are not part of the drop-in contract.
 *   LH_OPT_EXTRACT_ZERO_COPY  0 / 1: small extract results (<= 32 KiB) are stored by the kernel straight into pinned
 *                             host memory (default 1); a value >= 4096 also sets that size limit (measured: beyond
 *                             32 KiB the copy engine wins -- 1 024 names, 158 KB: 48 us by copy, 95 us by stores)
 *   LH_OPT_SCRATCH_CAP_BYTES  upper bound of the mixed ingest's SHARED scratch block (default 1.5 GiB, >= 64 MiB); the
 *                             host-fed lanes' own blocks are bounded by LH_OPT_LANE_SCRATCH_BLOCKS x a lane-sized launch's
 *                             need and reported as lh_counters.lane_scratch_bytes
 *   LH_OPT_SUBLAUNCH_PAIRS    largest partitioned sub-launch, 2^22 .. 2^30 pairs, rounded down to a power of two
 *                             (default 2^29; a sub-launch is halved until its scratch fits the cap).
 *                             Engines with more than 8 192 names (two scatter levels: ~1.2 GB of chunk pools and
 *                             ~0.35 ms of fixed work per launch) are NOT cut by default -- a 1e9-pair launch over
 *                             65 536 names takes a 9 GB block -- unless one of these two options was set, and then
 *                             not below 2^28 pairs.  A block that cannot be had never fails a call: the sub-launch goes
 *                             through the scratch-free kernel (lh_counters.scratch_alloc_failures / samples_fallback)
 *   LH_OPT_SURVEY_EVERY       33 .. 65 536 names: a call may run on the survey of an earlier call (hot names, region
 *                             sizes, per-partition ranking stay in the scratch block) until this many calls have used
 *                             it (default 32; 1 = every call surveys).  Only while the stream looks the same: a survey
 *                             is also repeated when the window width or scatter shape changed, when anything else used
 *                             the block, or when more than 2 % of the pairs of the calls completed since took an
 *                             overflow / window-miss path or stayed out of the hot windows that took them when the
 *                             survey was new (the stream's values moved: lh_counters.survey_stale_pairs).  A stale
 *                             survey costs speed -- one call's worth -- never exactness
 *   LH_OPT_LANE_SCRATCH_BLOCKS  0 .. 16 (default 0 since round 6; 16 before): 0 = a host-fed mixed launch (lh_submit_pairs*,
 *                             lh_commit_pairs*: one staging half-buffer, at most 2^22 pairs) takes the direct path -- one
 *                             link-bound pass through a per-workgroup LDS table of cells, no scratch: 0.89 - 0.90 of the
 *                             link at every name count.  n > 0 = such launches run partitioned in one of n scratch blocks
 *                             of their own (first generation up to 8 192 names; above, the third generation on survey
 *                             tables the lanes share), one lane's later passes beside another lane's read (0.84 - 0.87 of
 *                             the link up to 8 192 names, 0.75 - 0.78 above; lh_counters.lane_scratch_bytes)
 *   LH_OPT_LANE_ZERO_COPY     0 / 1: the ingest kernels read the pinned staging buffers of lh_submit* / lh_reserve_pairs
 *                             in place over PCIe (default 1) instead of after a hipMemcpyAsync into HBM (0)
 *   LH_OPT_MERGE_NARROW_CELLS 0 / 1 (default 1): lh_snapshot_merge over more than one rank sends a row at 8 or 16 bits per
 *                             cell when nranks x its largest per-rank cell fits (lh_merge_info); 0 = every cell a whole
 *                             word.  A rank that sets 0 makes every rank of that merge send whole words (the bound
 *                             is all-reduced): ranks need not agree */
enum {
    LH_OPT_EXTRACT_ZERO_COPY = 5,
    LH_OPT_SCRATCH_CAP_BYTES = 6,
    LH_OPT_SUBLAUNCH_PAIRS = 7,
    LH_OPT_LANE_ZERO_COPY = 15,
    LH_OPT_SURVEY_EVERY = 16,
    LH_OPT_LANE_SCRATCH_BLOCKS = 18,
    LH_OPT_MERGE_NARROW_CELLS = 22
};
int lh_set_option(lh_engine *e, int option, uint64_t value);

/* Codec access for parity tests. */
/* key[i] = compress(d_v[i]) on device (metrics.go:316-322). */
int lh_compress_device(lh_engine *e, const double *d_v, int16_t *d_keys, size_t n, void *stream);
/* Same arithmetic but through the device restatement of Go's math.Log instead of
 * the fast path + threshold table (cross-check of the two device routes). */
int lh_compress_device_golog(lh_engine *e, const double *d_v, int16_t *d_keys, size_t n, void *stream);
/* Copies the device-generated tables to host: Tx[LH_NTHRESH] (thresholds in
 * x = 1+|v| space) and D[LH_NKEYS] (decompress by bin).  Either may be NULL. */
int lh_codec_tables(lh_engine *e, double *Tx, double *D);
/* max |v_log_f32(m) - log2(m)| over all 2^23 fp32 mantissas m in [1,2): the
 * measured bound the fast path's guard band rests on. */
int lh_selftest_vlog(lh_engine *e, double *max_abs_err);

#ifdef __cplusplus
}
#endif
#endif /* LOGHISTO_GPU_H */
