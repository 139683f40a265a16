/* loghisto_gpu_tuning.h -- test and tuning hooks of liblhgpu.so.  NOT part of the drop-in contract
 * (include/loghisto_gpu.h): a binding of the reference never needs anything in here.
 *
 * The mixed (id, value) ingest picks one of five exact kernel paths by name count, launch size, alignment and what
 * the stream has looked like so far (loghisto_amd/csrc/lh_dispatch.h).  The parity tests must be able to send SMALL
 * inputs down every path of the very library that ships, and measurement runs must be able to hold one path fixed:
 * these lh_set_option keys do that.  Like every option they only choose among exact paths or size a buffer -- none
 * can change a result -- and the product library never reads the environment.  The numbers are stable across ABI
 * versions (they were declared in loghisto_gpu.h up to ABI 4). */
#ifndef LOGHISTO_GPU_TUNING_H
#define LOGHISTO_GPU_TUNING_H

#include "loghisto_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/*   LH_OPT_TWO_LEVEL_ABOVE    first generation: second scatter level when a level-1 partition holds more names than
 *                             this (default 32, i.e. above 8 192 names; 0 forces it above 4 names per partition)
 *   LH_OPT_HOT_MIN_TILES      first generation: hot-name windows in the scatter pass when every workgroup gets >= this
 *                             many 4 096-sample tiles (default 32); 1 exercises the path on small inputs
 *   LH_OPT_HOT_WINDOWS        0 / 1: hot-name windows allowed in any generation's scatter pass (default 1)
 *   LH_OPT_NAMES_PER_PARTITION first generation: names per LDS-reduce partition, 1..64 (default 4 = 4 096-bin windows)
 *   LH_OPT_SMALL_PATH         0 / 1: the single-pass kernel for <= 32 names (1 also re-arms it after adaptive
 *                             dispatch turned it off)
 *   LH_OPT_PART_V2            0 / 1: the second generation (survey + 2-byte records; 33 .. 8 192 names; default 1)
 *   LH_OPT_PART_V2_MIN_PAIRS  smallest launch that takes it (default 2^20, where it overtakes the direct path's cell table;
 *                             2^25 until round 6, the crossover with the first generation while every call surveyed
 *                             itself; >= 2^17)
 *   LH_OPT_PART_V2_SHAPE      bit 0: two 512-thread scatter workgroups per CU over <= 128 partitions instead of one
 *                             1 024-thread workgroup over <= 256; bit 1: fixed per-partition LDS regions (records
 *                             placed by the classifying phase) instead of the exact per-tile layout.  Default 2.
 *                             6 (bits 1 + 2, <= 1 024 names): the WIDE shape, 512 partitions of two names with 16 384-bin
 *                             reduce windows -- the engine takes it by itself while the survey reports value spans wider
 *                             than 8 192 bins (two-signed streams over 40 decades); setting 6 forces it.
 *                             With bit 1 set the engine falls back to the exact layout while more than 2 % of an
 *                             interval's samples overflow their regions (a stream clustered by name), see
 *                             lh_counters.regions_disabled
 *   LH_OPT_PART_V3            0 / 1: the third generation (hashed survey, region scatter of 4-byte records, a second
 *                             level that counts each partition's frequent names in place; 8 193 .. 65 536 names --
 *                             BASELINE config 4's name count; default 1)
 *   LH_OPT_PART_V3_MIN_PAIRS  smallest launch that takes it (default 3 * 2^20 for device-resident calls -- below, the direct
 *                             path's cell table is faster -- and 2^18 for a host-fed lane's half-buffer; >= 2^17)
 *   LH_OPT_PART_V3_LOG_W      log2 of its second-level window width, 10 .. 14 (32 .. 4 names per fine partition; 14: 4
 *                             names in twice the LDS); 0 (default) = follow the survey: every call's survey reports the
 *                             smallest width within half of which, around their name's mean, 99 % of the sampled values
 *                             lie, and the following calls use it (lh_counters.window_log2)
 *   LH_OPT_PART_V3_DIRECT_MAX_PAIRS  third generation: a launch of at most this many pairs ends in a reduce pass without
 *                             LDS windows, one global atomic per forwarded record (a host-fed lane's half-buffer leaves a
 *                             fine partition some hundred records: the windowed pass's fixed cost per slot bounded the lanes);
 *                             0 = the default, 2^22 (the largest lane launch); 1 = never; <= 2^30
 *   LH_OPT_PART_MIN_PAIRS     smallest mixed launch that takes a partitioned path at all, whatever the generation (below it:
 *                             the direct path -- no scratch, no survey: whole tiles through a per-workgroup LDS table of
 *                             cells, the rest one global atomic per sample); 0 = the defaults: 2^20 pairs up to 8 192 names,
 *                             3 * 2^20 above, 2^17 for a host-fed lane's half-buffer (131 072 for all until round 6:
 *                             profiles/r06_small_calls.txt); >= 65 536 otherwise
 *   LH_OPT_LANE_GEN3          0 / 1 (default 1): above 8 192 names a host-fed lane launch takes the third generation in the
 *                             lane's own scratch block, on survey tables the lanes share (read-only between surveys, two
 *                             sets); 0 = the first generation's two scatter levels, as up to ABI 4
 *   LH_OPT_FAIL_SCRATCH_ALLOCS  the next `value` scratch allocations of the mixed ingest fail as if the device were out
 *                             of memory (tests/test_gpu_faults.py: a call still counts every pair exactly once)
 *   LH_OPT_WIDEN_AT_SAMPLES   1 .. 2^32 - 1 (the default): an epoch buffer of 32-bit cells moves to uint64 cells before the
 *                             interval's samples pass this (tests/test_gpu_cells32.py: the widening without 2^32 samples) */
enum {
    LH_OPT_TWO_LEVEL_ABOVE = 1,
    LH_OPT_HOT_MIN_TILES = 2,
    LH_OPT_HOT_WINDOWS = 3,
    LH_OPT_NAMES_PER_PARTITION = 4,
    LH_OPT_SMALL_PATH = 8,
    LH_OPT_PART_V2 = 9,
    LH_OPT_PART_V2_MIN_PAIRS = 10,
    LH_OPT_PART_V2_SHAPE = 11,
    LH_OPT_PART_V3 = 12,
    LH_OPT_PART_V3_MIN_PAIRS = 13,
    LH_OPT_PART_V3_LOG_W = 14,
    LH_OPT_PART_MIN_PAIRS = 17,
    LH_OPT_FAIL_SCRATCH_ALLOCS = 19,
    LH_OPT_LANE_GEN3 = 20,
    LH_OPT_PART_V3_DIRECT_MAX_PAIRS = 21,
    LH_OPT_WIDEN_AT_SAMPLES = 23
};

/* The path choice as a function: what an engine in the described state would do with a call of n pairs.  No device is
 * touched (and none is needed: tests/test_dispatch.py enumerates a few thousand states on a box without a GPU).
 * Paths: 0 direct (one global atomic per sample), 1 single pass (<= 32 names), 2 first generation, 3 second, 4 third. */
typedef struct lh_dispatch_query {
    uint32_t struct_size;     /* sizeof(lh_dispatch_query) */
    uint32_t max_metrics;
    uint64_t n;               /* pairs of the call */
    uint64_t ids_addr;        /* the arrays' addresses: only their alignment matters */
    uint64_t vals_addr;
    uint32_t id_width;        /* 2 or 4 */
    uint32_t host_fed;        /* a lane's half-buffer (lh_submit_pairs*, lh_commit_pairs*) */
    uint32_t num_cus;         /* 0 = 256 */
    uint32_t lane_blocks;     /* LH_OPT_LANE_SCRATCH_BLOCKS */
    uint64_t lane_samples;    /* lh_config.lane_samples */
    /* adaptive switches (lh_counters.small_path_disabled / regions_disabled; the third generation's skew switch) */
    uint32_t small_disabled, regions_disabled, v3_disabled;
    uint32_t call_log_w;      /* the third generation's window width of this call, 10 .. 14 (0 = 10) */
    /* options, 0 = the default unless stated */
    uint64_t scratch_cap;     /* LH_OPT_SCRATCH_CAP_BYTES */
    uint64_t sublaunch_pairs; /* LH_OPT_SUBLAUNCH_PAIRS */
    uint64_t part_min_pairs, v2_min_pairs, v3_min_pairs;
    uint32_t v2_off, v3_off, hot_off; /* 1 = LH_OPT_PART_V2 / _V3 / LH_OPT_HOT_WINDOWS set to 0 */
    uint32_t v2_shape_set, v2_shape;  /* v2_shape_set = 1: LH_OPT_PART_V2_SHAPE = v2_shape */
    uint32_t fail_allocs;     /* the first this-many scratch allocations fail */
    uint32_t lane_gen3_off;   /* 1 = LH_OPT_LANE_GEN3 set to 0 */
} lh_dispatch_query;

typedef struct lh_dispatch_step {
    uint32_t path;            /* 0 .. 4 */
    uint32_t lane_block;      /* in one of the lanes' own scratch blocks */
    uint64_t take;            /* pairs of this sub-launch */
    uint64_t scratch;         /* bytes of scratch it runs in (0: none) */
    uint32_t fell_back;       /* its scratch could not be had: through the direct kernel instead (path == 0) */
    uint32_t peeled;          /* the odd-aligned first sample of a call */
} lh_dispatch_step;

/* steps[0 .. *nsteps) in order; at most cap are written, *nsteps receives how many there are. */
int lh_dispatch_probe(const lh_dispatch_query *q, lh_dispatch_step *steps, size_t cap, size_t *nsteps);

/* Measurement helpers (bench.py; nothing on a product path calls them).
 * lh_tool_device_alloc / _free: plain hipMalloc'ed memory for a bench's input stream (the allocator the engine's own
 * buffers come from, not a framework's caching allocator).
 * lh_tool_read_ceiling: average / minimum time of `reps` launches of a kernel that ONLY reads [d_ptr, d_ptr + bytes) with
 * k_ingest_single's access pattern (three untimed launches first) -- what this box gives a pure read, measured in the
 * process that measures the headline.  d_ptr 16-byte aligned, bytes >= 64 KiB. */
int lh_tool_device_alloc(size_t bytes, void **d_ptr);
int lh_tool_device_free(void *d_ptr);
int lh_tool_read_ceiling(const void *d_ptr, size_t bytes, int reps, void *stream, float *avg_ms, float *min_ms);
/* Device time of the engine's last extract whose results went through HBM and one copy (more than 32 KiB of results:
 * from ~240 names on), HIP events on the snapshot stream: the K2 kernel(s) alone, and the device-to-host copy of the
 * results behind them.  LH_ESTATE when the engine has not run such an extract. */
int lh_tool_last_extract_ms(lh_engine *e, float *kernel_ms, float *copy_ms);

#ifdef __cplusplus
}
#endif
#endif
