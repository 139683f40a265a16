// lh_kernels_part3.h -- third generation of the partitioned mixed (id, value) ingest: 8 193 .. 65 536 names, gfx950.
// Included at the end of lh_kernels_part.hip after lh_kernels_part2.h (same translation unit: it reuses the plan
// kernels, the chunk descriptor format, the survey's g_stat layout, hidden_store_* and lh_bin_fast).
// Reference semantics are unchanged: Histogram(name, v) = histogramCache[name][compress(v)] += 1
// (metrics.go:273-295, 316-322).
//
// Config 4 of BASELINE.json runs 65 536 names per GPU.  Round 2 left that name count on the first-generation
// two-level path (k_scatter_samples -> k_scatter_records -> k_part_hist): 7.3 ms per 1e9 pairs, 20 % of the HBM
// roofline, 28 B of traffic per sample.  What the survey path (lh_kernels_part2.h) changed for <= 8 192 names does
// not carry over as it is -- its per-name LDS table is 8 B per name (512 KiB at 65 536 names) and its 2-byte record
// needs name-in-partition and bin offset in 15 bits -- so this file rebuilds the three passes around what does:
//
//   survey   k_survey_count_h     the same ~1 M sampled pairs, counted per workgroup in an LDS HASH table (a
//                                 workgroup sees <= 2 048 distinct names whatever the name space) and merged into the
//                                 per-name tables with one global atomic per (workgroup, name)
//            k_survey_pick/plan_h hot names claim one slot of a 1 024-entry hash table (the more frequent name wins a
//                                 collision); the plan gives the winners LDS windows exactly as k_survey_plan does
//            k_survey_remap       per level-1 partition (id & 255): its 256 names ranked by sampled count.  Rank r is
//                                 the name's index in the second level: ranks < Kp are counted in place there, rank
//                                 r >= Kp goes on to fine partition r / mpp2
//   level 1  k_scatter4           k_scatter3's region scatter with a HASHED hot-name lookup (8 KiB of LDS instead of
//                                 8 B per name) and 4-byte records (partition | local name | bin: no per-name cold
//                                 window, so no per-name table at all).  One LDS gather + one returning LDS atomic +
//                                 one LDS store per sample, two barriers per 8 192-sample tile
//   level 2  k_split_records      one workgroup per level-1 work slot.  Names are skewed inside a partition too: the Kp
//                                 most frequent of its 256 names get LDS windows here (96 KiB) and their records end
//                                 in this pass (under Zipf(1) they are ~70 % of the partition's records); the rest is
//                                 split into ns fine partitions through LDS regions, exactly like level 1
//   reduce   k_part_hist3         one workgroup per fine-partition work slot: mpp2 names x W bins of windows (16-bit cells, two slots per CU;
//                                 W >= 2^13: twice the cells, one slot per CU)
//                                 (128 KiB), windows placed from the slot's own first chunk and the survey
//
// Window width W (1 024 .. 16 384 bins, names per fine partition 32, 16, 8, 8, 4) follows the stream: the survey reports
// the smallest width within half of which, around their name's sampled mean, 99 % of the samples lie (k_survey_mass,
// k_survey_plan_h -> pinned host word) and the engine uses it for the following calls.  A wrong W only costs speed: every record outside a window is counted through the small LDS
// overflow tables and global atomics; a record that finds an LDS region full likewise.  Everything stays exact.
//
// HBM traffic per sample at 65 536 Zipf names: 12 B read + (cold share ~0.65) x (4 B written + 4 B read) +
// (forwarded share ~0.2) x (4 B + 4 B) = ~19 B against 28 B before.

constexpr uint32_t V3_MAX_NAMES = 65536;
constexpr uint32_t V3_LOG_NP = 8, V3_NP = 256;          // level-1 partitions: id & 255
constexpr uint32_t V3_HN = 1024;                        // hot-name hash entries (8 B each)
constexpr uint32_t V3_TILE = 8192;                      // samples per level-1 tile
constexpr uint32_t LINE4 = 16;                          // 4-byte records per 64-byte line
#ifndef LH_V3_PIECE
#define LH_V3_PIECE 2
#endif
// Level 1 copies whole PIECES of V3_PIECE consecutive lines out of a partition's region (128 bytes at 2): what the
// scattered record writes cost the memory system next to the 12 GB streaming read falls with their size
// (profiles/r03_level1_experiments.txt; round 4, profiles/r04_level1_experiments.txt: 64 -> 128 bytes takes the 1e9-pair
// call from 5.18 to 4.74 ms, 256 bytes give it back -- 4.91 -- because the 32 KiB of extra region come out of the hot
// windows); up to V3_PIECE * 16 - 1 records stay behind in the region, so every capacity grows by 16 records per line.
constexpr uint32_t V3_PIECE = LH_V3_PIECE, PIECE4 = V3_PIECE * LINE4;
constexpr uint32_t V3_MISSQ = 512;                      // records a tile can queue for the exact path (per parity)
// Smallest launch that takes this path.  2^24 while every call surveyed (0.18 ms); with one survey per 32 calls the third
// generation beats the first at every size measured (round 4, profiles/r04_level1_experiments.txt: 65 536 names, 0.26 M /
// 1 M / 2 M / 4 M / 8 M pairs 0.17 / 0.30 / 0.36 / 0.41 / 0.49 ms against 0.49 / 0.53 / 0.58 / 0.63 / 0.69) -- lane-sized
// host-fed launches included.
constexpr size_t V3_MIN_SAMPLES = size_t(3) << 20;     // device-resident calls: below, the cell-table kernel is faster (r06_small_calls.txt)
constexpr size_t V3_TABLES_SAMPLES = size_t(1) << 18;  // (the launch size the survey tables' offsets are computed at: any valid one)
constexpr size_t V3_DIRECT_MAX = size_t(1) << 22; // launches up to this many pairs: reduce pass without windows (k_part_direct3)
constexpr uint32_t SVH_GRID = 256, SVH_SLOTS = 4096;    // hashed survey: 256 workgroups x 2 048 samples
constexpr uint32_t V3_EXTRA1 = 768;                     // level-1 work slots beyond one per partition
constexpr uint32_t V3_EXTRA2 = 1024;                    // fine work slots beyond one per fine partition
constexpr uint32_t V3_MAX_NS = 64;                      // fine partitions per level-1 partition (at most)
constexpr uint32_t PEEL_WORDS = 24576;                  // level 2: 96 KiB of windows for the partition's top names
constexpr uint32_t P3_WINWORDS = 32768;                 // reduce: 32 768 window cells per slot
// The reduce pass keeps its window cells as 16-bit fields, two to an LDS word (as level 1's hot windows do): 64 KiB per slot
// instead of 128, so that TWO workgroups share a CU and one slot's chain of dependent loads (slot -> chunk list -> records)
// and its flush run beside the other's -- config 4's slice has 2 167 slots, 8.5 per CU one after another, ~23 us each of
// which most is waiting.  An add that takes a field across a multiple of 2^14 hands 2^14 counts on to the row in HBM
// (add_one below); the chunk loop runs in rounds that end at a barrier, which bounds what a field can hold.
#ifndef LH_P3_PACKED
#define LH_P3_PACKED 1
#endif
constexpr uint32_t P3_PACK = LH_P3_PACKED ? 1u : 0u;    // log2 cells per LDS word
constexpr uint32_t V3_MAX_LOG_W = LH_P3_PACKED ? 14u : 13u; // windows of 1 024 .. 16 384 bins (the widest: 4 names x 2^14 packed cells = 128 KiB)
#ifndef LH_HOT_CAP_BIG
#define LH_HOT_CAP_BIG 4096u /* widest hot window of a name with >= 1/64 of the sampled mass (k_survey_plan_h) */
#endif
#ifndef LH_P3_WIDE_FROM
#define LH_P3_WIDE_FROM 13u /* windows from this width on: twice the cells per slot (128 KiB packed, 8 names of 2^13 bins or 4 of 2^14), one slot per CU.
                               Measured (tools/build_tuning.py -DLH_P3_WIDE_FROM=..): from 13, loguniform[1e-3, 1e18] at 65 536 names 7.65 -> 7.2 ms
                               per 1e9 pairs (32 fine partitions per partition instead of 64); from 12, lognormal sigma 5 6.1 -> 6.75 */
#endif
#ifndef LH_P3_SPILL_LOG
#define LH_P3_SPILL_LOG 14u /* an add that takes its field across a multiple of 2^14 hands 2^14 counts on to the row.  Any value
                               from 6 (an add carries at most 64) to 15 is exact; tools/round.sh p3spill runs the third
                               generation's suites on a build with 6, where every busy cell of every slot takes the hand-off
                               path many times over */
#endif
static_assert(LH_P3_SPILL_LOG >= 6u && LH_P3_SPILL_LOG <= 15u, "c <= 64 per add; (2^log - 1) + 2^15 counts of a round must fit 16 bits");
constexpr uint32_t P3_SPILL = 1u << LH_P3_SPILL_LOG;
#ifndef LH_P3_DEPTH
#define LH_P3_DEPTH (LH_P3_PACKED ? 1 : 2)              /* chunks in flight per wave */
#endif
static_assert(!LH_P3_PACKED || LH_P3_DEPTH == 1, "a round is ONE chunk per wave");
constexpr uint32_t SPLIT_TILE = 8192;                   // records per level-2 tile
constexpr uint32_t SPLIT_REG_WORDS = 12800;             // level 2: LDS regions (>= 5/4 tile + 28 per fine partition)

__device__ __forceinline__ uint32_t v3_hash(uint32_t id) { return (id * 0x9E3779B1u) >> 22; } // 10 bits

// Per-name survey statistics (zero-initialised): cs[M] u64 = sampled count (24 bits) | sum of the sampled bins << 24,
// then mninv[M] u32 (max of 65535 - bin) and mx[M] u32.  Count and sum share a word so that a workgroup merges a name
// with ONE global atomic: atomics to scattered addresses sustain only ~9 G/s on this part (measured: four per name
// took 250 us for 2 M of them).
struct SurveyStat { const unsigned long long *cs; const uint32_t *mninv; const uint32_t *mx; };
__device__ __forceinline__ uint32_t sv_count(const SurveyStat &S, uint32_t m) { return (uint32_t)(S.cs[m] & 0xffffffull); }
__device__ __forceinline__ uint32_t sv_mean(const SurveyStat &S, uint32_t m)
{
    const unsigned long long w = S.cs[m];
    const uint32_t c = (uint32_t)(w & 0xffffffull);
    return c ? (uint32_t)((w >> 24) / c) : 32768u;
}

// ---------------------------------------------------------------------------
// Survey for large name spaces
// ---------------------------------------------------------------------------
// LDS hash table: key[SVH_SLOTS] (id + 1, 0 = free) | cnt | sum | mninv | mx.  A workgroup inserts <= 2 048 samples
// into 4 096 slots, so linear probing always terminates.
template <typename IDT>
__global__ __launch_bounds__(1024) void k_survey_count_h(const IDT *__restrict__ ids, const double *__restrict__ v,
                                                         size_t n, uint32_t nmetrics, const double *__restrict__ Tx,
                                                         unsigned long long *__restrict__ g_cs,
                                                         uint32_t *__restrict__ g_mninv, uint32_t *__restrict__ g_mx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char svh_smem[];
    uint32_t *s_key = reinterpret_cast<uint32_t *>(svh_smem);
    uint32_t *s_cnt = s_key + SVH_SLOTS, *s_sum = s_cnt + SVH_SLOTS, *s_mninv = s_sum + SVH_SLOTS,
             *s_mx = s_mninv + SVH_SLOTS;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 5 * SVH_SLOTS; i += 1024) s_key[i] = 0;
    __syncthreads();
    const size_t npairs = n / 2; // an odd last sample is not surveyed
    const size_t stride = npairs / gridDim.x;
    const size_t i = (size_t)blockIdx.x * stride + tid;
    if (i < npairs && tid < (stride ? stride : npairs)) {
        typedef IdStream<IDT> IS;
        const typename IS::raw_t id2 = IS(ids).ld(i);
        const pd2_t x2 = reinterpret_cast<const pd2_t *>(v)[i];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t id = h ? IS::second(id2) : IS::first(id2);
            if (id < nmetrics) {
                const uint32_t bin = lh_bin_of(h ? x2.y : x2.x, Tx);
                uint32_t s = (id * 2654435761u) >> 20;
                for (;;) {
                    const uint32_t prev = atomicCAS(&s_key[s], 0u, id + 1u);
                    if (prev == 0u || prev == id + 1u) break;
                    s = (s + 1u) & (SVH_SLOTS - 1u);
                }
                atomicAdd(&s_cnt[s], 1u);
                atomicAdd(&s_sum[s], bin);
                atomicMax(&s_mninv[s], 65535u - bin);
                atomicMax(&s_mx[s], bin);
            }
        }
    }
    __syncthreads();
    for (uint32_t s = tid; s < SVH_SLOTS; s += 1024) {
        const uint32_t k = s_key[s];
        if (k) {
            const uint32_t m = k - 1u;
            atomicAdd(&g_cs[m], (unsigned long long)s_cnt[s] | ((unsigned long long)s_sum[s] << 24));
            // (a stale read only costs an atomic that changes nothing)
            if (s_mninv[s] > g_mninv[m]) atomicMax(&g_mninv[m], s_mninv[s]);
            if (s_mx[s] > g_mx[m]) atomicMax(&g_mx[m], s_mx[s]);
        }
    }
}

// g_aux (zero-initialised with the statistics): claim[V3_HN] | pc[V3_NP] | in[8] (k_survey_mass)
constexpr uint32_t AUX_CLAIM = 0, AUX_PC = V3_HN, AUX_IN = V3_HN + V3_NP, AUX_WORDS = V3_HN + V3_NP + 32;

// One thread per name: a name with >= 16 sampled values claims its hash slot (the larger count wins); per-partition
// sample counts.
__global__ __launch_bounds__(1024) void k_survey_pick(const SurveyStat S, uint32_t nmetrics,
                                                      uint32_t *__restrict__ g_aux)
{
    __shared__ uint32_t s_pc[V3_NP];
    const uint32_t tid = threadIdx.x;
    if (tid < V3_NP) s_pc[tid] = 0;
    __syncthreads();
    const uint32_t m = blockIdx.x * 1024u + tid;
    const uint32_t c = m < nmetrics ? sv_count(S, m) : 0u;
    if (c) {
        atomicAdd(&s_pc[m & (V3_NP - 1u)], c);
        if (c >= 16u) atomicMax(&g_aux[AUX_CLAIM + v3_hash(m)], (min(c, 65535u) << 16) | m);
    }
    __syncthreads();
    if (tid < V3_NP && s_pc[tid]) atomicAdd(&g_aux[AUX_PC + tid], s_pc[tid]);
}

// The window width of levels 2 - 3 from the sampled MASS (round 6): the same samples once more, each against its name's
// sampled mean bin -- how many lie within half a window of 2^10 .. 2^14 bins of it (names with >= 32 samples).  Until then
// the sampled mass was classed by the names' sampled min .. max SPANS: ONE far outlier among a hot name's thousand
// samples puts its whole mass into the widest class, and a stream with a 0.1 % tail of far outliers (every hot name has
// some) got 8 192-bin windows -- level 2 in lock step with 3 names counted in place, 17 000 reduce slots, row spans and
// extract to match: 7.5 ms per 1e9 pairs instead of 4.4, 1.89 instead of 0.82 at config 4's slice.
//   g_aux[AUX_IN + k - 10] samples within 2^(k-1) bins of their name's mean, k = 10 .. 14; g_aux[AUX_IN + 5] all of them;
//   g_aux[AUX_IN + 6] sampled pairs of neighbours in the stream, [AUX_IN + 7] those of one name (k_survey_mass2)
static_assert(AUX_IN + 8 <= AUX_WORDS, "room behind the partition counts");
template <typename IDT>
__global__ __launch_bounds__(1024) void k_survey_mass(const IDT *__restrict__ ids, const double *__restrict__ v, size_t n,
                                                      uint32_t nmetrics, const double *__restrict__ Tx, const SurveyStat S,
                                                      uint32_t *__restrict__ g_aux)
{
    __shared__ uint32_t s_in[8];
    const uint32_t tid = threadIdx.x;
    if (tid < 8) s_in[tid] = 0;
    __syncthreads();
    const size_t npairs = n / 2; // (the samples k_survey_count_h took)
    const size_t stride = npairs / gridDim.x;
    const size_t i = (size_t)blockIdx.x * stride + tid;
    uint32_t in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < npairs && tid < (stride ? stride : npairs)) {
        typedef IdStream<IDT> IS;
        const typename IS::raw_t id2 = IS(ids).ld(i);
        const pd2_t x2 = reinterpret_cast<const pd2_t *>(v)[i];
        in[6]++;
        in[7] += IS::first(id2) == IS::second(id2) ? 1u : 0u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t id = h ? IS::second(id2) : IS::first(id2);
            if (id < nmetrics && sv_count(S, id) >= 32u) {
                const uint32_t bin = lh_bin_of(h ? x2.y : x2.x, Tx), mean = sv_mean(S, id);
                const uint32_t d = bin > mean ? bin - mean : mean - bin;
                in[5]++;
#pragma unroll
                for (uint32_t k = 0; k < 5; k++) in[k] += d < (512u << k) ? 1u : 0u;
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) in[k] += __shfl_xor(in[k], d, 64);
        if ((tid & 63u) == 0 && in[k]) atomicAdd(&s_in[k], in[k]);
    }
    __syncthreads();
    if (tid < 8 && s_in[tid]) atomicAdd(&g_aux[AUX_IN + tid], s_in[tid]);
}

// One workgroup, thread t owns hash slot t.  Output:
//   g_hk[t]   hot entry {name | width << 16, LDS base (relative to the window area) | origin << 16}; a free slot is
//             {0, 0}: width 0 matches no sample (names are < 65 536: both halves are what one SDWA operand selects)
//   g_hs[s]   the same windows as a list for the flush {name, origin | width << 16, base, 0}
//   g_pt[p]   level-1 region of partition p {first record relative to the region area, capacity}
//   hdr       [0] hot names [1] cells used [2] surveyed samples [3] surveyed samples of hot names [4] log2 of the
//             window width that covers 95 % of the sampled mass (10 .. 13), also stored to *span_out (pinned)
//             [HDR_REGION] words of LDS of the level-1 regions, [HDR_CELLS] 16-bit cells of the window area (the plan
//             splits `avail_bytes` of LDS between the two: RegionFit in lh_kernels_part2.h)
__global__ __launch_bounds__(V2_BLOCK) void k_survey_plan_h(const SurveyStat S,
                                                            const uint32_t *__restrict__ g_aux, uint32_t cells_in,
                                                            uint32_t tile, uint32_t avail_bytes, uint32_t max_cells,
                                                            pu2_t *__restrict__ g_hk,
                                                            pu4_t *__restrict__ g_hs, pu2_t *__restrict__ g_pt,
                                                            uint32_t *__restrict__ hdr, uint32_t *span_out, uint32_t yield_log_w)
{
    __shared__ uint32_t s_a[V2_BLOCK / 64], s_b[V2_BLOCK / 64];
    __shared__ uint32_t s_cap[V3_NP], s_pcw[V3_NP];
    constexpr uint32_t GAP = 2; // unused 16-bit cells between two windows: equal bins of different names on different banks
    static_assert(V3_HN == V2_BLOCK, "one thread per hash slot");
    const uint32_t tid = threadIdx.x;
    const uint32_t claim = g_aux[AUX_CLAIM + tid];
    uint32_t name = 0, cnt = 0, want = 0, mean = 32768u, mn = 0, mx = 0;
    bool whole = false, lobe = false; // as k_survey_plan: a narrow span kept whole; two lobes either side of key 0
    if (claim) {
        name = claim & 0xffffu;
        cnt = sv_count(S, name);
        mn = 65535u - S.mninv[name];
        mx = S.mx[name];
        mean = sv_mean(S, name);
        const uint32_t w = (((mx - mn + 1u) * 3u / 4u) + 63u) & ~63u; // as k_survey_plan: 3/4 of the sampled span
        want = w < 64u ? 64u : w;
        if (mx - mn + 1u <= HOT_WHOLE_SPAN) {
            want = ((mx - mn + 1u) + 63u) & ~63u;
            mean = (mn + mx + 1u) >> 1;
            whole = true;
        }
        lobe = mn + 64u < 32768u && mx > 32768u + 64u && mx - mn > 1024u;
    }
    const uint32_t pc = tid < V3_NP ? g_aux[AUX_PC + tid] : 0u;
    uint32_t total_cnt, dummy;
    block_sum2(pc, 0, s_a, s_b, total_cnt, dummy);
    const uint32_t big = total_cnt / 64u;
    // the smallest window (as log2, 10 .. 14) within half of which, around their name's mean, 99 % of the samples lie
    // (k_survey_mass; until round 6: the class of the names' sampled min .. max spans that covers 95 % of the mass).
    // 2^14 bins: streams on both sides of key 0 over many decades (+-10^U(-3, 20) spans 9 211 bins: 8 192-bin windows
    // sent 9 % of the reduce pass's records to global atomics, 5.1 ms of a 12.4 ms call)
    const uint32_t mass = g_aux[AUX_IN + 5];
    uint32_t lw = V3_MAX_LOG_W;
    for (uint32_t k = 10; k < V3_MAX_LOG_W; k++)
        if ((unsigned long long)g_aux[AUX_IN + k - 10u] * 100u >= (unsigned long long)mass * 99u) { lw = k; break; }
    if (!mass) lw = 10;
    if (want) {
        // A name with >= 1/64 of the mass may have a window as wide as half of that (512 bins on a lognormal stream, up to
        // LH_HOT_CAP_BIG on a wide one): where values spread evenly over their span a cell earns cnt / span whatever the
        // window's width, so the cells belong to the most frequent names' whole spans, not to 512 bins each of many names
        // (21 decades at 65 536 names: 7.24 -> 6.65 ms per 1e9 pairs, sigma = 5: 6.08 -> 5.78).  By the stream's mass, not
        // by the name's sampled span: one far outlier among a top name's samples stretches that (a 0.1 % tail to 1e60 with
        // 4 096-bin windows by span alone: 5.09 -> 5.33 ms).
        uint32_t cap_big = 1u << (lw - 1u);
        cap_big = cap_big < 512u ? 512u : cap_big > LH_HOT_CAP_BIG ? LH_HOT_CAP_BIG : cap_big;
        const uint32_t cap = cnt >= big ? cap_big : 256u; // (the other names at half or a quarter of cap_big instead of 256: within 2 % either way)
        if (want > cap && !whole) want = cap;
    }
    auto pick = [&](uint32_t cells) {
        uint32_t flo = 15, fhi = (1u << 21) + 1;
        if (cells < 64) flo = fhi - 1;
        while (fhi - flo > 1) {
            const uint32_t mid = flo + (fhi - flo) / 2;
            const bool in = want && cnt >= mid;
            uint32_t tw, tn;
            block_sum2(in ? want + GAP : 0u, in ? 1u : 0u, s_a, s_b, tw, tn);
            if (tw <= cells && tn <= V2_MAX_SLOTS) fhi = mid; else flo = mid;
        }
        return fhi;
    };
    uint32_t tau = pick(cells_in);
    // level-1 regions: 1.5 x the partition's expected records per tile + 8 + one piece (the leftover), from the names'
    // counts with the hot names' reduced to what their windows leave (cold_share, lh_kernels_part2.h) -- round 6; until
    // then every name counted in full and the regions took their upper bound of the LDS.  What that frees goes to the windows: they are
    // chosen once more with the larger budget (a superset of the first choice, so the regions stay large enough).
    if (tid < V3_NP) s_pcw[tid] = pc;
    __syncthreads();
    if (want && cnt >= tau) atomicSub(&s_pcw[name & (V3_NP - 1u)], cnt - cold_share(cnt, want, mx - mn + 1u));
    __syncthreads();
    uint32_t cap = 0;
    if (tid < V3_NP) {
        const uint32_t est = total_cnt ? (uint32_t)(((unsigned long long)s_pcw[tid] * tile) / total_cnt) : tile / V3_NP;
        cap = (est * 6u / 4u + 8u + PIECE4 + 3u) & ~3u; // (the leftover is < PIECE4)
        if (cap > tile + PIECE4) cap = tile + PIECE4;
        s_cap[tid] = cap;
    }
    __syncthreads();
    uint32_t region_words = 0;
    for (uint32_t i = 0; i < V3_NP; i++) {
        if (i == tid) g_pt[tid] = (pu2_t){region_words, cap};
        region_words += s_cap[i];
    }
    uint32_t cells = avail_bytes > region_words * 4u ? ((avail_bytes - region_words * 4u) / 2u) & ~63u : 0u;
    if (cells > max_cells) cells = max_cells;
    tau = pick(cells);
    const bool hot = want && cnt >= tau;
    const uint32_t sw = hot ? want + GAP : 0u, sn = hot ? 1u : 0u;
    uint32_t incw = sw, incn = sn;
    const uint32_t lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t yw = __shfl_up(incw, d, 64), yn = __shfl_up(incn, d, 64);
        if ((int)lane >= d) { incw += yw; incn += yn; }
    }
    __syncthreads();
    if (lane == 63) { s_a[wave] = incw; s_b[wave] = incn; }
    __syncthreads();
    uint32_t basew = 0, basen = 0, totw = 0, totn = 0;
#pragma unroll
    for (int w = 0; w < V2_BLOCK / 64; w++) {
        if (w < (int)wave) { basew += s_a[w]; basen += s_b[w]; }
        totw += s_a[w];
        totn += s_b[w];
    }
    const uint32_t cellpos = basew + incw - sw, slot = basen + incn - sn;
    uint32_t hot_cnt_total, dummy2;
    block_sum2(hot ? cnt : 0u, 0, s_a, s_b, hot_cnt_total, dummy2);
    pu2_t e = (pu2_t){0u, 0u};
    if (hot) {
        uint32_t o = mean > want / 2 ? mean - want / 2 : 0u;
        if (lobe) o = mean >= 32768u ? (mx + 1u > want ? mx + 1u - want : 0u) : mn; // the outer end of the heavier lobe
        if (o > 65536u - want) o = 65536u - want;
        e = (pu2_t){name | (want << 16), cellpos | (o << 16)};
        g_hs[slot] = (pu4_t){name, o | (want << 16), cellpos, 0u};
    }
    g_hk[tid] = e;
    if (tid == 0) {
        hdr[HDR_REGION] = region_words;
        hdr[HDR_CELLS] = cells;
        hdr[0] = totn;
        hdr[1] = totw;
        hdr[2] = total_cnt;
        hdr[3] = hot_cnt_total;
        hdr[4] = lw;
        hdr[HDR_BASE] = 0; // no launch has run on these tables yet (stale_judge, lh_kernels_part2.h)
        // (<= 8 192 names, yield_log_w = the second generation's cold window there: bit 8 keeps such a stream here)
        const uint32_t keep = yield_log_w >= 10u && yield_log_w <= 14u &&
                              (unsigned long long)g_aux[AUX_IN + yield_log_w - 10u] * 8u < (unsigned long long)mass * 7u &&
                              g_aux[AUX_IN + 7] * 2u < g_aux[AUX_IN + 6] ? 0x100u : 0u; // (not clustered by name)
        if (span_out && mass) __hip_atomic_store(span_out, lw | keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup per level-1 partition p, thread l = old local name (id = l << 8 | p).  Ranks the partition's names by
// sampled count (ties by l, so that names beyond nmetrics rank last) and lays out the level-2 regions of the partition:
//   g_remap[p * 256 + l] = rank, g_inv[p * 256 + rank] = l
//   g_pt2[p * V3_MAX_NS + s] = {first record of fine partition s's region (relative to the region area), capacity}
__global__ __launch_bounds__(256) void k_survey_remap(const SurveyStat S, const pu2_t *__restrict__ g_hk,
                                                      uint32_t nmetrics, uint32_t kp, uint32_t log_mpp2, uint32_t ns,
                                                      uint8_t *__restrict__ g_remap, uint8_t *__restrict__ g_inv,
                                                      pu2_t *__restrict__ g_pt2)
{
    __shared__ uint32_t s_c[256], s_w[V3_MAX_NS], s_cap[V3_MAX_NS];
    __shared__ uint32_t s_tot;
    const uint32_t p = blockIdx.x, l = threadIdx.x;
    const uint32_t m = (l << V3_LOG_NP) | p;
    uint32_t c = m < nmetrics ? sv_count(S, m) : 0u;
    // what reaches the second level: a name with a hot window in level 1 leaves it about a quarter of its samples
    if (c) {
        const uint32_t hx = g_hk[v3_hash(m)].x;
        if ((hx & 0xffffu) == m && (hx >> 16)) c = (c + 3u) / 4u;
    }
    s_c[l] = c;
    if (l < V3_MAX_NS) s_w[l] = 0;
    if (l == 0) s_tot = 0;
    __syncthreads();
    uint32_t rank = 0;
    for (uint32_t k = 0; k < 256; k++) {
        const uint32_t ck = s_c[k];
        rank += (ck > c || (ck == c && k < l)) ? 1u : 0u;
    }
    g_remap[p * 256u + l] = (uint8_t)rank;
    g_inv[p * 256u + rank] = (uint8_t)l;
    if (c) {
        // expected share of the records this partition forwards: a name counted in place leaves half of its records
        // at worst (its window is placed on the same survey)
        const uint32_t s = rank >> log_mpp2;
        atomicAdd(&s_w[s < ns ? s : ns - 1u], rank < kp ? (c + 1u) / 2u : c);
        atomicAdd(&s_tot, c);
    }
    __syncthreads();
    if (l < ns) {
        const uint32_t tot = s_tot;
        const uint32_t est = tot ? (uint32_t)(((unsigned long long)s_w[l] * SPLIT_TILE) / tot) : SPLIT_TILE / ns;
        s_cap[l] = (est * 5u / 4u + 24u + 3u) & ~3u;
    }
    __syncthreads();
    if (l < ns) {
        // The shares come from a few hundred sampled values per partition: a fine partition of cold names is
        // regularly off by 30 % (measured with exact capacities: 3 % of all samples found their region full).  The
        // region area is sized for a tile that forwards everything; what the estimates leave of it is shared out.
        uint32_t base = 0, sum = 0;
        for (uint32_t i = 0; i < ns; i++) {
            if (i < l) base += s_cap[i];
            sum += s_cap[i];
        }
        const uint32_t spare = (sum < SPLIT_REG_WORDS ? (SPLIT_REG_WORDS - sum) / ns : 0u) & ~3u;
        g_pt2[p * V3_MAX_NS + l] = (pu2_t){base + l * spare, s_cap[l] + spare};
    }
}

// The exact path of a record that found no room in LDS, for use INSIDE the tile loops: the same three atomics as
// v2_global_add, issued from inline asm and without the range pre-check.  Atomics that return nothing have no result
// register, so hiding them from the compiler's s_waitcnt bookkeeping is as safe as hiding stores (k_scatter2); a
// visible global operation in a rarely taken branch makes the compiler wait for vmcnt(0) at the join, and the
// pre-check's load stalls the wave -- and, through the tile barrier, the workgroup -- for a memory round trip.
__device__ __forceinline__ void v3_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges, uint32_t m,
                                              uint32_t bin, uint32_t c)
{
    lh::cell_add_hidden(counts, (size_t)m * LH_ROW_STRIDE + bin, c);
    // The range: LOOK, then widen (an unconditional min / max pair per record serialises on the name's two words when a
    // whole stream takes this path: profiles/r06_first_call.txt).  The look is a load the compiler does not see either,
    // with its own wait -- which also waits for the tile loop's prefetch; this path is off the loop's fast path.
    uint32_t *r = ranges + 2 * (size_t)m;
    pu2_t rg;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(rg) : "v"(r) : "memory");
    if (bin < rg.x) asm volatile("global_atomic_umin %0, %1, off" : : "v"(r), "v"(bin) : "memory");
    if (bin > rg.y) asm volatile("global_atomic_umax %0, %1, off offset:4" : : "v"(r), "v"(bin) : "memory");
}

// compress of a sample inside the guard band of a threshold (or not finite), without touching memory: Go's log
// restated operation by operation (lh_codec.h route 1; tests/test_gpu_parity.py holds it equal to the table compare
// of lh_bin_of at +-3 ulp of every threshold).  ~60 float64 instructions for 1 sample in ~4 000, against a table
// load whose wait drains the tile loop's prefetch.
__device__ __forceinline__ uint32_t v3_bin_exact(double v)
{
    const double x = 1.0 + fabs(v);
    const uint32_t eb = ((uint32_t)__double2hiint(x)) >> 20;
    return bin_from_kext(eb < 0x7ffu ? d_kext_golog(x) : 0, v); // NaN / +Inf -> int16(...) == 0
}

// ---------------------------------------------------------------------------
// Level 1: compress + hashed hot windows + region scatter of 4-byte records
// Record: partition << 24 | local name (id >> 8) << 16 | bin  (the first generation's format).
// The kernel handles whole tiles only; the launcher gives the last n % V3_TILE pairs to k_ingest_pairs.
// ---------------------------------------------------------------------------
struct Scatter4Lds {
    uint32_t cnt[V3_NP];                 // records in the partition's region (leftover + this tile's, may exceed cap)
    uint32_t cfill[V3_NP], cbase[V3_NP]; // open chunk of the partition: records in it, its index
    pu2_t pt[V3_NP];                     // {region base (LDS word index), capacity in records}
    uint32_t ov_key[OV_SLOTS], ov_cnt[OV_SLOTS];
    uint32_t missq[2][V3_MISSQ];
    uint32_t missn[2];
    uint32_t dummy[64];
    uint32_t pool_next, ovn, nrec, spills, pad[2]; // spills: hot cells that handed 2^15 counts on to the row (k_scatter3)
    pu2_t hk[V3_HN];                     // hot-name hash table
};
static_assert(sizeof(Scatter4Lds) % 16 == 0, "the regions follow the struct in LDS");
// upper bound of the sum of the level-1 region capacities (k_survey_plan_h)
constexpr uint32_t v3_region_words(uint32_t tile) { return 6u * tile / 4u + (12u + PIECE4) * V3_NP; }

template <int BATCH, typename IDT>
__global__ __launch_bounds__(1024, 4) void k_scatter4(const IDT *__restrict__ ids, const double *__restrict__ v,
                                                      size_t ntiles, uint32_t nmetrics, const double *__restrict__ Tx,
                                                      const pu2_t *__restrict__ g_hk, const pu4_t *__restrict__ g_hs,
                                                      const uint32_t *__restrict__ g_hdr,
                                                      const pu2_t *__restrict__ g_pt, uint32_t *__restrict__ g_hot,
                                                      uint32_t *__restrict__ records,
                                                      uint32_t *__restrict__ cdesc, uint32_t chunks_per_wg,
                                                      uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                      uint32_t *__restrict__ err, uint32_t *__restrict__ g_stats,
                                                      uint32_t *__restrict__ g_resume)
{
    constexpr int BLOCK = 1024, NPT = V3_NP;
    static_assert(BLOCK == 4 * NPT, "flush: four threads per partition");
    static_assert(BLOCK * V2_SPT == (int)V3_TILE, "eight samples per thread per tile");
    // ONE LDS allocation: [Scatter4Lds][regions][hot windows]
    extern __shared__ __attribute__((aligned(16))) unsigned char v3_smem[];
    Scatter4Lds &L = *reinterpret_cast<Scatter4Lds *>(v3_smem);
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(v3_smem);
    constexpr uint32_t REG_W = sizeof(Scatter4Lds) / 4;                 // word offset of the regions
    // the plan's split of the LDS behind the struct (k_survey_plan_h): regions, then the hot windows -- 16-BIT cells, two
    // to a word, handled exactly as k_scatter3's (a cell that reaches 2^15 hands them on to the row in HBM)
    const uint32_t region_words = g_hdr[HDR_REGION], cells = g_hdr[HDR_CELLS];
    const uint32_t win_h = 2u * (REG_W + region_words);                 // halfword offset of the hot windows
    uint32_t *win = lds32 + win_h / 2;                                  // [cells / 2] words
    constexpr uint32_t CNT_W = offsetof(Scatter4Lds, cnt) / 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t pool_base = blockIdx.x * chunks_per_wg;
    // the LDS address of the block (0 here; not a constant the compiler can fold)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)v3_smem;

    for (uint32_t i = tid; i < V3_HN; i += BLOCK) L.hk[i] = g_hk[i]; // (hot base: a CELL index inside the window area)
    for (uint32_t i = tid; i < cells / 2; i += BLOCK) win[i] = 0;
    if (tid < NPT) {
        pu2_t e = g_pt[tid];
        e.x = 4u * (e.x + REG_W) + lds_base; // LDS address: a record's address is one shift-add
        L.pt[tid] = e;
        L.cnt[tid] = 0;
        L.cfill[tid] = CHUNK;
        L.cbase[tid] = INVALID;
    }
    ov_init(L.ov_key, L.ov_cnt, tid, BLOCK);
    if (tid == 0) { L.pool_next = 0; L.ovn = 0; L.nrec = 0; L.spills = 0; L.missn[0] = 0; L.missn[1] = 0; }
    __syncthreads();
    pu2_t my_pt = L.pt[tid >> 2];       // the flush phase's partition (constant over the launch)
    my_pt.x = (my_pt.x - lds_base) >> 2; // (word index)
    uint32_t nrec = 0;                  // records this thread's partition emitted (q == 0 counts)

    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
    constexpr int NPAIR = V2_SPT / 2;
    // two register sets used by alternate tiles, loads issued a whole tile period ahead, the loop's global stores
    // hidden from the compiler's s_waitcnt bookkeeping: see k_scatter2 in lh_kernels_part2.h
    typename IS::raw_t ida[NPAIR], idb[NPAIR];
    pd2_t vaa[NPAIR], vab[NPAIR];
    auto load_tile = [&](size_t tile, typename IS::raw_t (&di)[NPAIR], pd2_t (&dv)[NPAIR]) {
        if (tile >= ntiles) tile = ntiles - 1; // the two tiles past the end that the pipeline touches (uniform)
        if (ABL & 32u) tile = blockIdx.x;
        const size_t it = tile * (V3_TILE / 2) + tid;
        const pd2_t *vt = vp + tile * (V3_TILE / 2) + tid;
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            di[j] = (ABL & 32u) ? ip.ld(it + (size_t)j * BLOCK) : ip.ld_nt(it + (size_t)j * BLOCK);
            dv[j] = (ABL & 32u) ? vt[j * BLOCK] : __builtin_nontemporal_load(vt + j * BLOCK);
        }
    };
    load_tile(blockIdx.x, ida, vaa);
    asm volatile("" : "+v"(ida[0]), "+v"(ida[1]), "+v"(ida[2]), "+v"(ida[3]), "+v"(vaa[0]), "+v"(vaa[1]), "+v"(vaa[2]),
                      "+v"(vaa[3]));
    static_assert(NPAIR == 4, "the asm above names four register pairs");
    load_tile((size_t)blockIdx.x + gridDim.x, idb, vab);
    // The first two tiles arrive before the loop starts (once per workgroup).  What it buys: the compiler schedules these
    // sixteen loads in another order than the loop's, and its s_waitcnt pass must cover both orders -- it then waits for a
    // whole register set (vmcnt(8)) where the set's first load would do (vmcnt(15)), and with the copy-out's stores in the
    // counter but not in its books (hidden_store_*) that stricter wait reaches into the loads issued a moment ago.
    asm volatile("" : "+v"(idb[0]), "+v"(idb[1]), "+v"(idb[2]), "+v"(idb[3]), "+v"(vab[0]), "+v"(vab[1]), "+v"(vab[2]),
                      "+v"(vab[3])); // (an empty asm that reads the registers: the compiler waits for their loads here)

    // ---- phase 1: classify and place.  Straight-line code, BATCH samples at a time: their table reads, then their LDS
    // atomics, then their record stores are in flight together.  An id >= nmetrics is not tested per sample: one max3 +
    // compare per batch finds the lanes that hold one, and only those lanes mark the sample (no table entry matches such
    // an id, so it is not hot; the mark keeps it from becoming a record).
    auto classify = [&](typename IS::raw_t (&idv)[NPAIR], pd2_t (&val)[NPAIR], const uint32_t par) {
#pragma unroll
        for (int h = 0; h < V2_SPT; h += BATCH) {
            uint32_t raw[BATCH], bin[BATCH], rank[BATCH];
            pu2_t he[BATCH], pe[BATCH];
            bool unc[BATCH], hot[BATCH], full[BATCH], spill[BATCH], inval[BATCH];
            uint32_t waddr[BATCH], sh[BATCH];
            uint32_t idmax = 0;
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const int j = h + k;
                raw[k] = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
                idmax = max(idmax, raw[k]);
                he[k] = L.hk[v3_hash(raw[k])];
                pe[k] = L.pt[raw[k] & (NPT - 1u)];
            }
            bool anyunc = false;
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const int j = h + k;
                bin[k] = lh_bin_fast((j & 1) ? val[j >> 1].y : val[j >> 1].x, unc[k]);
                if (ABL & 16u) { // the same index once more, on a value the compiler cannot tell from the first
                    double x2 = (j & 1) ? val[j >> 1].y : val[j >> 1].x;
                    asm volatile("" : "+v"(x2));
                    bool u2;
                    bin[k] = (bin[k] + lh_bin_fast(x2, u2)) >> 1;
                }
                anyunc |= unc[k];
            }
            if (anyunc) { // inside the guard band of a bucket threshold (1 sample in ~4 000): Go's log, exactly
#pragma unroll
                for (int k = 0; k < BATCH; k++) {
                    const int j = h + k;
                    if (unc[k]) bin[k] = v3_bin_exact((j & 1) ? val[j >> 1].y : val[j >> 1].x);
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; k++) inval[k] = false;
            if (idmax >= nmetrics) { // this lane holds an id >= nmetrics: reported by lh_sync / lh_extract, the sample skipped
                atomicOr(err, 1u);
#pragma unroll
                for (int k = 0; k < BATCH; k++) inval[k] = raw[k] >= nmetrics;
            }
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const uint32_t hrel = bin[k] - (he[k].y >> 16);
                hot[k] = (he[k].x & 0xffffu) == raw[k] && hrel < (he[k].x >> 16);
                // the hot cell (a halfword of the window area: word cell / 2, field cell & 1) or the partition's counter
                const uint32_t cell = (he[k].y & 0xffffu) + hrel + win_h;
                static_assert(CNT_W == 0, "the partition counters open the LDS block");
                waddr[k] = hot[k] ? cell >> 1 : raw[k] & (NPT - 1u);
                sh[k] = hot[k] ? (cell & 1u) << 4 : 0u;
                asm volatile("" : "=v"(rank[k])); // (no value for the lanes that skip the atomic)
                if (ABL & 8u) { asm volatile("" : "+v"(waddr[k]), "+v"(sh[k])); rank[k] = lane & 15u; }
                else if (!inval[k]) rank[k] = atomicAdd(lds32 + waddr[k], 1u << sh[k]);
                if (ABL & 2u) rank[k] &= 15u;
                spill[k] = hot[k] && ((rank[k] >> sh[k]) & 0xffffu) == 0x7fffu; // this add made the cell 2^15
            }
            if ((spill[0] || spill[1] || spill[2] || spill[3]) && !(ABL & 8u)) { // 2^15 counts of the cell move to the row in HBM
                static_assert(BATCH == 4, "the test above names four samples");
#pragma unroll
                for (int k = 0; k < BATCH; k++)
                    if (spill[k]) {
                        atomicSub(lds32 + waddr[k], 0x8000u << sh[k]);
                        hidden_global_add(counts, ranges, raw[k], bin[k], 0x8000u);
                        atomicAdd(&L.spills, 1u);
                    }
            }
            bool anyfull = false;
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const bool fits = !hot[k] && !inval[k] && rank[k] < pe[k].y;
                full[k] = !hot[k] && !inval[k] && !fits; // the region is full: counted exactly below
                anyfull |= full[k];
                // record: partition << 24 | local name << 16 | bin = the id's two bytes swapped above the bin
                if (fits)
                    *(__attribute__((address_space(3))) uint32_t *)(uintptr_t)(pe[k].x + 4u * rank[k]) =
                        __builtin_amdgcn_perm(raw[k], bin[k], 0x04050100u);
            }
            if (anyfull) { // no room: queued, counted exactly by the flush phase
#pragma unroll
                for (int k = 0; k < BATCH; k++)
                    if (full[k]) {
                        atomicAdd(&L.ovn, 1u);
                        const uint32_t key = (raw[k] << 16) | bin[k];
                        const uint32_t at = atomicAdd(&L.missn[par], 1u);
                        if (at < V3_MISSQ) L.missq[par][at] = key;
                        else if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v3_global_add(counts, ranges, raw[k], bin[k], 1);
                    }
            }
        }
    };
    uint32_t seq_lines = tid >> 2; // (ABL 64: this thread group's position in the workgroup's sequential stream)
    auto flush = [&](const uint32_t par) {
        if (ABL & 2u) return;
        __syncthreads();                                   // barrier A: the tile's records are in the regions
        if (ABL & 1u) {
            if (tid < NPT) L.cnt[tid] = 0;
            if (tid == BLOCK - 1) { L.missn[0] = 0; L.missn[1] = 0; }
            __syncthreads();
            return;
        }
        // ---- phase 2: four threads per partition (p = tid / 4; thread q copies 16-byte piece q of every line)
        if (tid == BLOCK - 1) L.missn[par ^ 1u] = 0; // the other parity's queue was drained in the previous tile
        {
            uint32_t t2 = tid;
            asm volatile("" : "+v"(t2)); // keeps this phase's address arithmetic inside the loop (see k_scatter2)
            const uint32_t p = t2 >> 2, q = t2 & 3u;
            const pu2_t e = my_pt;
            // whole pieces only: `full` lines leave (a multiple of V3_PIECE), up to PIECE4 - 1 records stay behind
            const uint32_t c = min(L.cnt[p], e.y), full = c / PIECE4 * V3_PIECE, left = c - full * LINE4;
            if (full) {
                const uint32_t cf = L.cfill[p], cb = L.cbase[p];
                const uint32_t room = (CHUNK - cf) / LINE4; // lines left in the open chunk (0: none open; whole pieces)
                uint32_t first = 0;
                if (full > room && q == 0) {
                    const uint32_t tag = p << CD_SHIFT;
                    const uint32_t over = full - room;
                    const uint32_t k = (over + CHUNK / LINE4 - 1) / (CHUNK / LINE4);
                    first = pool_base + atomicAdd(&L.pool_next, k);   // k consecutive chunks
                    if (cb != INVALID) hidden_store_u32(cdesc + cb, tag | CHUNK);   // the old chunk is now full
#pragma nounroll
                    for (uint32_t i = 0; i + 1 < k; i++) hidden_store_u32(cdesc + first + i, tag | CHUNK);
                    L.cbase[p] = first + k - 1;
                    L.cfill[p] = (over - (k - 1) * (CHUNK / LINE4)) * LINE4;
                } else if (q == 0) {
                    L.cfill[p] = cf + full * LINE4;
                }
                first = __builtin_amdgcn_mov_dpp(first, 0x00, 0xf, 0xf, false); // quad_perm [0,0,0,0]: q == 0's value
                const uint32_t dA = cb * CHUNK + cf, dB = first * CHUNK - room * LINE4;
                const uint32_t *src = lds32 + e.x;
#pragma nounroll
                for (uint32_t l = 0; l < full; l++) {
                    const uint32_t dst = (l < room ? dA : dB) + l * LINE4;
                    const pu4_t r4 = *reinterpret_cast<const pu4_t *>(src + l * LINE4 + q * 4);
                    if (ABL & 4u) asm volatile("" : : "v"(r4), "v"(dst));
                    else if (ABL & 64u) // the same bytes to ONE sequential stream per workgroup (its pool, front to back)
                        hidden_store_u4(records + (size_t)pool_base * CHUNK + ((seq_lines += BLOCK / 4) % (chunks_per_wg * (CHUNK / LINE4))) * LINE4 + q * 4, r4);
                    else hidden_store_u4(records + dst + q * 4, r4);
                }
                // the leftover (less than a piece) moves to the front of the region: thread q moves 16-byte pieces q,
                // q + 4, .. -- source and destination are at least one piece apart, every slot is read by its writer
#pragma unroll
                for (uint32_t j = 0; j < V3_PIECE; j++)
                    if (j * LINE4 + q * 4 < left)
                        *reinterpret_cast<pu4_t *>(lds32 + e.x + j * LINE4 + q * 4) =
                            *reinterpret_cast<const pu4_t *>(src + (full + j) * LINE4 + q * 4);
            }
            if (q == 0) { L.cnt[p] = left; nrec += full * LINE4; }
        }
        // the tile's overflowed records, one per thread: aggregated in the small LDS table, else a global atomic
        {
            const uint32_t nq = min(L.missn[par], V3_MISSQ);
            for (uint32_t i = tid; i < nq; i += BLOCK) {
                const uint32_t key = L.missq[par][i];
                if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v3_global_add(counts, ranges, key >> 16, key & 0xffffu, 1);
            }
            if (L.missn[par] > V3_MISSQ) { // (uniform) the tile overflowed past its queue, straight into the table: emptied
                __syncthreads();           // here, the table takes the next tile's cells too instead of staying full
                for (uint32_t i = tid; i < OV_SLOTS; i += BLOCK)
                    if (L.ov_key[i] != OV_EMPTY) {
                        v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
                        L.ov_key[i] = OV_EMPTY;
                        L.ov_cnt[i] = 0;
                    }
            }
        }
        __syncthreads();                                   // barrier B: counters and regions are ready for the next tile
    };
    // Tiles in PAIRS, both halves of the body unconditional, a last single tile peeled off: see k_scatter3 (the
    // compiler's count of the loads in flight must be exact, or every classification opens with a wait for the OTHER
    // register set's loads)
    // A stream CLUSTERED BY NAME (sorted by name, or whole batches of one producer) fills one partition's region with
    // every tile and sends the rest -- most of the tile -- through the exact overflow path: a 1e9-pair call took 1.9 s
    // (profiles/r06_first_call.txt).  A workgroup whose last two tiles overflowed by more than an eighth stops here; the tiles
    // it leaves are counted by k_scatter_clustered (its turn in g_resume), whose whole LDS is one (name, bin) table.
    uint32_t par = 0, ovn_seen = 0;
    bool gave_up = false;
    size_t tile = blockIdx.x;
    const size_t G = gridDim.x;
    for (; tile + G < ntiles; tile += 2 * G) {
        classify(ida, vaa, par);
        load_tile(tile + 2 * G, ida, vaa);
        flush(par);
        par ^= 1u;
        classify(idb, vab, par);
        load_tile(tile + 3 * G, idb, vab);
        flush(par);
        par ^= 1u;
        // (uniform: L.ovn is at rest between flush's last barrier and the next tile)
        const uint32_t ovn_now = (uint32_t)__builtin_amdgcn_readfirstlane(L.ovn);
        if (ovn_now - ovn_seen > V3_TILE / 4u) { // more than an eighth of the two tiles just done
            tile += 2 * G;
            gave_up = true;
            break;
        }
        ovn_seen = ovn_now;
    }
    if (!gave_up && tile < ntiles) { // (workgroup-uniform) the workgroup's last tile when it has an odd number of them
        classify(ida, vaa, par);
        flush(par);
        par ^= 1u;
        tile += G;
    }
    if (tid == 0) g_resume[blockIdx.x] = (uint32_t)min(tile, ntiles); // the first tile of this workgroup's turn left undone

    // ---- drain: the regions' leftovers (< one line each) and the open chunks' descriptors
    {
        const uint32_t p = tid >> 2, q = tid & 3u;
        const uint32_t left = L.cnt[p]; // < PIECE4 after a flush
        uint32_t d = INVALID;
        if (left && q == 0) {
            uint32_t cf = L.cfill[p], cb = L.cbase[p]; // (cf is a multiple of PIECE4: the leftover fits the open chunk)
            if (cf == CHUNK) { // no open chunk, or it is exactly full
                if (cb != INVALID) cdesc[cb] = (p << CD_SHIFT) | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[p] = cb;
            }
            d = cb * CHUNK + cf;
            L.cfill[p] = cf + left;
            nrec += left;
        }
        d = __builtin_amdgcn_mov_dpp(d, 0x00, 0xf, 0xf, false);
#pragma unroll
        for (uint32_t j = 0; j < V3_PIECE; j++)
            if (j * LINE4 + q * 4 < left)
                *reinterpret_cast<pu4_t *>(records + d + j * LINE4 + q * 4) =
                    *reinterpret_cast<const pu4_t *>(lds32 + ((L.pt[p].x - lds_base) >> 2) + j * LINE4 + q * 4);
    }
    if (nrec) atomicAdd(&L.nrec, nrec);
    __syncthreads();
    if (tid < NPT && L.cbase[tid] != INVALID) cdesc[L.cbase[tid]] = (tid << CD_SHIFT) | L.cfill[tid];
    // self-metrics of the launch (k_v3_report adds them to the engine's pinned words): records that found their region
    // full, records this level emitted
    if (tid == 0) {
        if (L.ovn) atomicAdd(&g_stats[0], L.ovn);
        atomicAdd(&g_stats[1], L.nrec);
    }

    // ---- the hot windows leave as one coalesced copy into the workgroup's slice of g_hot; k_hot_reduce adds the
    // workgroups' copies up and touches every row cell once (see k_scatter3); the overflow table
    {
        pu4_t *dst = reinterpret_cast<pu4_t *>(g_hot + (size_t)blockIdx.x * (cells / 2));
        const pu4_t *src = reinterpret_cast<const pu4_t *>(win);
        for (uint32_t i = tid; i < cells / 8; i += BLOCK) dst[i] = src[i];
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += BLOCK)
        if (L.ov_key[i] != OV_EMPTY) v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
}

// ---------------------------------------------------------------------------
// Window placement shared by level 2 and the reduce pass: the union of what the survey saw of the name and of what
// the slot's first chunk holds; centred on the span when it fits, on the survey's mean bin otherwise.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t v3_place(uint32_t mn, uint32_t mx, uint32_t sv_cnt, uint32_t sv_mean, uint32_t W)
{
    uint32_t centre = 32768u; // nothing known about the name: key 0
    if (mn != INVALID) centre = (mx - mn < W || !sv_cnt) ? (mn + mx + 1u) >> 1 : sv_mean;
    uint32_t org = centre > W / 2 ? centre - W / 2 : 0u;
    if (org > 65536u - W) org = 65536u - W;
    return org;
}

// ---------------------------------------------------------------------------
// Level 2: one workgroup per level-1 work slot (all records of one partition p1).
//   tbl[old local] = rank | window origin << 8   (rank: k_survey_remap)
//   rank < kp            counted here: LDS window rank x W (the record ends in this pass)
//   otherwise / outside  4-byte record  fine << 24 | rank % mpp2 << 16 | bin  into fine partition rank / mpp2
// Output chunk tag q = fine << 8 | p1.
// ---------------------------------------------------------------------------
struct SplitLds {
    uint32_t cnt[V3_MAX_NS], cfill[V3_MAX_NS], cbase[V3_MAX_NS];
    pu2_t pt[V3_MAX_NS];
    uint32_t ov_key[OV_SLOTS], ov_cnt[OV_SLOTS];
    uint32_t missq[2][V3_MISSQ];
    uint32_t missn[2];
    uint32_t dummy[64];
    uint32_t pool_next, ovn, nfwd, pad0[3];
    uint32_t tbl[256];
    uint32_t name[32], mn[32], mx[32], svc[32], svm[32], org[32]; // the names counted here
};
static_assert(sizeof(SplitLds) % 16 == 0, "the regions follow the struct in LDS");
constexpr size_t SPLIT_LDS_BYTES = sizeof(SplitLds) + (size_t)(SPLIT_REG_WORDS + PEEL_WORDS) * 4;
static_assert(SPLIT_LDS_BYTES <= 160 * 1024, "level 2 must fit one CU's LDS");
static_assert(5 * SPLIT_TILE / 4 + 28 * V3_MAX_NS <= SPLIT_REG_WORDS, "k_survey_remap's capacities fit the region area");

__global__ __launch_bounds__(1024, 4) void k_split_records(const uint32_t *__restrict__ in_records,
                                                           const uint32_t *__restrict__ in_cdesc,
                                                           const uint32_t *__restrict__ in_sorted,
                                                           const uint32_t *__restrict__ in_part_start,
                                                           const uint32_t *__restrict__ in_slots,
                                                           const uint32_t *__restrict__ in_nslots,
                                                           const uint32_t *__restrict__ pool_start, uint32_t nmetrics,
                                                           uint32_t kp, uint32_t log_mpp2, uint32_t log_w, uint32_t ns,
                                                           const uint8_t *__restrict__ g_remap,
                                                           const uint8_t *__restrict__ g_inv,
                                                           const pu2_t *__restrict__ g_pt2, const SurveyStat S,
                                                           uint32_t *__restrict__ records, uint32_t *__restrict__ cdesc,
                                                           uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                           uint32_t *__restrict__ g_stats)
{
    constexpr int BLOCK = 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char v3_smem[];
    SplitLds &L = *reinterpret_cast<SplitLds *>(v3_smem);
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(v3_smem);
    constexpr uint32_t REG_W = sizeof(SplitLds) / 4, WIN_W = REG_W + SPLIT_REG_WORDS;
    constexpr uint32_t CNT_W = offsetof(SplitLds, cnt) / 4, DUMMY_W = offsetof(SplitLds, dummy) / 4;
    uint32_t *win = lds32 + WIN_W;
    const uint32_t slot = blockIdx.x;
    if (slot >= *in_nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p1 = in_slots[3 * slot], first = in_slots[3 * slot + 1], cnt = in_slots[3 * slot + 2];
    const uint32_t *list = in_sorted + in_part_start[p1] + first;
    const uint32_t pool_base = pool_start[slot];
    const uint32_t W = 1u << log_w, mmask = (1u << log_mpp2) - 1u;

    // ---- setup: the partition's rank table, the survey's view of the names counted here, the slot's first chunk
    const uint32_t c0 = list[0];
    const uint32_t n0 = in_cdesc[c0] & CD_MASK;
    const uint32_t srec = tid < n0 ? in_records[(size_t)c0 * CHUNK + tid] : 0u;
    if (tid < 256) L.tbl[tid] = g_remap[p1 * 256u + tid];
    if (tid < 32) {
        uint32_t name = INVALID, mn = INVALID, mx = 0, svc = 0, svm = 0;
        if (tid < kp) {
            const uint32_t m = ((uint32_t)g_inv[p1 * 256u + tid] << V3_LOG_NP) | p1;
            if (m < nmetrics) {
                name = m;
                svc = sv_count(S, m);
                if (svc) {
                    mn = 65535u - S.mninv[m];
                    mx = S.mx[m];
                    svm = sv_mean(S, m);
                }
            }
        }
        L.name[tid] = name;
        L.mn[tid] = mn;
        L.mx[tid] = mx;
        L.svc[tid] = svc;
        L.svm[tid] = svm;
    }
    for (uint32_t i = tid; i < kp << log_w; i += BLOCK) win[i] = 0;
    if (tid < V3_MAX_NS) {
        pu2_t e = tid < ns ? g_pt2[p1 * V3_MAX_NS + tid] : (pu2_t){0u, 0u};
        e.x += REG_W;
        L.pt[tid] = e;
        L.cnt[tid] = 0;
        L.cfill[tid] = CHUNK;
        L.cbase[tid] = INVALID;
    }
    ov_init(L.ov_key, L.ov_cnt, tid, BLOCK);
    if (tid == 0) { L.pool_next = 0; L.ovn = 0; L.nfwd = 0; L.missn[0] = 0; L.missn[1] = 0; }
    __syncthreads();
    if (tid < n0) {
        const uint32_t r = L.tbl[(srec >> 16) & 0xffu], b = srec & 0xffffu;
        if (r < kp) {
            if (b < L.mn[r]) atomicMin(&L.mn[r], b);
            if (b > L.mx[r]) atomicMax(&L.mx[r], b);
        }
    }
    __syncthreads();
    if (tid < 32) L.org[tid] = v3_place(L.mn[tid], L.mx[tid], L.svc[tid], L.svm[tid], W);
    __syncthreads();
    if (tid < 256) {
        const uint32_t r = L.tbl[tid];
        L.tbl[tid] = r | ((r < kp ? L.org[r] : 0u) << 8);
    }
    __syncthreads();

    // ---- main loop: tiles of 8 chunks (8 192 records), two 16-byte loads per thread, two register sets
    constexpr uint32_t CPT = SPLIT_TILE / CHUNK;
    const uint32_t ntiles = (cnt + CPT - 1) / CPT;
    pu4_t ra[2], rb[2];
    uint32_t cna[2], cnb[2]; // records of the chunk each load came from
    // The chunk a load reads comes out of two dependent loads (slot list -> descriptor).  Issued per tile they would
    // stall every wave of the workgroup at the same point; instead lane l of a wave holds them for tile tb + l / 2, load
    // l % 2, fetched 32 tiles at a time (tiles are requested in increasing order).
    uint32_t my_cid = 0, my_cn = 0, batch = 0xffffffffu;
    auto load_tile = [&](uint32_t tile, pu4_t (&r)[2], uint32_t (&cn)[2]) {
        if ((tile >> 5) != batch) { // workgroup-uniform
            batch = tile >> 5;
            const uint32_t t = (batch << 5) + (lane >> 1);
            uint32_t ci = t * CPT + (lane & 1u) * 4u + (wave >> 2); // 256 threads (4 waves) per chunk
            const bool in = t < ntiles && ci < cnt;
            ci = in ? ci : 0u;
            my_cid = list[ci];
            my_cn = in ? (in_cdesc[my_cid] & CD_MASK) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t idx = (tile & 31u) * 2u + (uint32_t)j;
            const uint32_t cid = __builtin_amdgcn_readlane(my_cid, idx);
            cn[j] = __builtin_amdgcn_readlane(my_cn, idx);
            r[j] = __builtin_nontemporal_load(reinterpret_cast<const pu4_t *>(in_records + (size_t)cid * CHUNK) + (tid & 255u));
        }
    };
    auto classify = [&](const pu4_t (&r)[2], const uint32_t (&cn)[2], const uint32_t par) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t rr[4] = {r[j].x, r[j].y, r[j].z, r[j].w};
            uint32_t t[4], where[4], rank[4], rec[4];
            pu2_t pe[4];
            uint32_t fwd = 0, full = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = L.tbl[(rr[k] >> 16) & 0xffu];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = (tid & 255u) * 4u + (uint32_t)k < cn[j];
                const uint32_t nw = t[k] & 0xffu, bin = rr[k] & 0xffffu, rel = bin - (t[k] >> 8);
                const bool here = valid && nw < kp && rel < W;
                const uint32_t fine = nw >> log_mpp2;
                where[k] = here ? WIN_W + (nw << log_w) + rel : valid ? CNT_W + fine : DUMMY_W + lane;
                rec[k] = (fine << 24) | ((nw & mmask) << 16) | bin;
                pe[k] = L.pt[fine];
                if (valid && !here) fwd |= 1u << k;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) rank[k] = atomicAdd(lds32 + where[k], 1u);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool fits = (fwd & (1u << k)) && rank[k] < pe[k].y;
                if ((fwd & (1u << k)) && !fits) full |= 1u << k;
                lds32[fits ? pe[k].x + rank[k] : DUMMY_W + lane] = rec[k];
            }
            if (full) { // the fine partition's region is full: the record is counted exactly by the flush phase
                atomicAdd(&L.ovn, (uint32_t)__popc(full));
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (full & (1u << k)) {
                        const uint32_t old = (rr[k] >> 16) & 0xffu, bin = rr[k] & 0xffffu;
                        const uint32_t key = (((old << V3_LOG_NP) | p1) << 16) | bin;
                        const uint32_t at = atomicAdd(&L.missn[par], 1u);
                        if (at < V3_MISSQ) L.missq[par][at] = key;
                        else if (!ov_add(L.ov_key, L.ov_cnt, key, 1u))
                            v2_global_add(counts, ranges, key >> 16, bin, 1); // (see k_split_waves)
                    }
            }
        }
    };
    // one wave per fine partition: 16 lines (64 lanes x 16 bytes) per step
    auto flush = [&](const uint32_t par) {
        __syncthreads();
        if (tid == BLOCK - 1) L.missn[par ^ 1u] = 0;
        for (uint32_t s = wave; s < ns; s += BLOCK / 64) {
            const pu2_t e = L.pt[s];
            const uint32_t c = min(L.cnt[s], e.y), full = c / LINE4, left = c % LINE4;
            if (full) { // wave-uniform
                const uint32_t cf = L.cfill[s], cb = L.cbase[s];
                const uint32_t room = (CHUNK - cf) / LINE4;
                uint32_t firstc = 0;
                if (full > room) {
                    const uint32_t tag = ((s << V3_LOG_NP) | p1) << CD_SHIFT;
                    const uint32_t over = full - room;
                    const uint32_t k = (over + CHUNK / LINE4 - 1) / (CHUNK / LINE4);
                    if (lane == 0) {
                        firstc = pool_base + atomicAdd(&L.pool_next, k);
                        if (cb != INVALID) hidden_store_u32(cdesc + cb, tag | CHUNK);
#pragma nounroll
                        for (uint32_t i = 0; i + 1 < k; i++) hidden_store_u32(cdesc + firstc + i, tag | CHUNK);
                        L.cbase[s] = firstc + k - 1;
                        L.cfill[s] = (over - (k - 1) * (CHUNK / LINE4)) * LINE4;
                    }
                    firstc = __builtin_amdgcn_readfirstlane(firstc);
                } else if (lane == 0) {
                    L.cfill[s] = cf + full * LINE4;
                }
                const uint32_t dA = cb * CHUNK + cf, dB = firstc * CHUNK - room * LINE4;
                const uint32_t *src = lds32 + e.x;
                const uint32_t q = lane & 3u;
#pragma nounroll
                for (uint32_t l = lane >> 2; l < full; l += 16) {
                    const uint32_t dst = (l < room ? dA : dB) + l * LINE4;
                    const pu4_t r4 = *reinterpret_cast<const pu4_t *>(src + l * LINE4 + q * 4);
                    hidden_store_u4(records + dst + q * 4, r4);
                }
                // the last partial line moves to the front (LDS operations of one wave execute in order: the reads of
                // line 0 above are done)
                if (lane < 4 && lane * 4 < left)
                    *reinterpret_cast<pu4_t *>(lds32 + e.x + lane * 4) =
                        *reinterpret_cast<const pu4_t *>(src + full * LINE4 + lane * 4);
            }
            if (lane == 0) { L.cnt[s] = left; if (full) atomicAdd(&L.nfwd, full * LINE4); }
        }
        {
            const uint32_t nq = min(L.missn[par], V3_MISSQ);
            for (uint32_t i = tid; i < nq; i += BLOCK) {
                const uint32_t key = L.missq[par][i];
                if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v3_global_add(counts, ranges, key >> 16, key & 0xffffu, 1);
            }
        }
        __syncthreads();
    };
    load_tile(0, ra, cna);
    asm volatile("" : "+v"(ra[0]), "+v"(ra[1]));
    load_tile(1, rb, cnb);
    uint32_t par = 0;
    for (uint32_t tile = 0; tile < ntiles; tile += 2) {
        classify(ra, cna, par);
        load_tile(tile + 2, ra, cna);
        flush(par);
        par ^= 1u;
        if (tile + 1 < ntiles) {
            classify(rb, cnb, par);
            load_tile(tile + 3, rb, cnb);
            flush(par);
            par ^= 1u;
        }
    }

    // ---- drain: leftovers (< one line per fine partition) and the open chunks' descriptors
    if (tid < ns) {
        const uint32_t s = tid, left = L.cnt[s];
        if (left) {
            uint32_t cf = L.cfill[s], cb = L.cbase[s];
            const uint32_t tag = ((s << V3_LOG_NP) | p1) << CD_SHIFT;
            if (cf == CHUNK) {
                if (cb != INVALID) cdesc[cb] = tag | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[s] = cb;
            }
            const uint32_t *src = lds32 + L.pt[s].x;
            for (uint32_t i = 0; i < left; i++) records[(size_t)cb * CHUNK + cf + i] = src[i];
            L.cfill[s] = cf + left;
            atomicAdd(&L.nfwd, left);
        }
    }
    __syncthreads();
    if (tid == 0) { // self-metrics: records forwarded to the reduce pass, records that found a region full
        atomicAdd(&g_stats[2], L.nfwd);
        if (L.ovn) atomicAdd(&g_stats[3], L.ovn);
    }
    if (tid < ns && L.cbase[tid] != INVALID) cdesc[L.cbase[tid]] = (((tid << V3_LOG_NP) | p1) << CD_SHIFT) | L.cfill[tid];

    // ---- flush the windows of the names counted here and the overflow table
    for (uint32_t r = wave; r < kp; r += BLOCK / 64) {
        const uint32_t name = L.name[r];
        if (name == INVALID) continue;
        const uint32_t org = L.org[r];
        uint32_t mn = INVALID, mx = 0;
        for (uint32_t i = lane; i < W; i += 64) {
            const uint32_t c = win[(r << log_w) + i];
            if (c) {
                const uint32_t b = org + i;
                lh::cell_add(counts, (size_t)name * LH_ROW_STRIDE + b, c);
                mn = min(mn, b);
                mx = max(mx, b);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor(mn, d, 64));
            mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
        }
        if (lane == 0 && mn != INVALID) {
            uint32_t *rg = ranges + 2 * (size_t)name;
            if (mn < rg[0]) atomicMin(&rg[0], mn);
            if (mx > rg[1]) atomicMax(&rg[1], mx);
        }
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += BLOCK)
        if (L.ov_key[i] != OV_EMPTY) v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
}

// ---------------------------------------------------------------------------
// Level 2 without workgroup barriers (at most 16 fine partitions per partition, i.e. windows of 1 024 or 2 048 bins).
// k_split_records runs in lock step: classify a tile, barrier, copy lines out, barrier -- measured 4.2 us per
// 8 192-record tile with the waves waiting at barriers for two thirds of it.  Here every WAVE owns its regions (one per
// fine partition, 800 words of LDS per wave in all) and its open chunks, walks its own chunks of the slot (wave,
// wave + 16, ...) and flushes after every 512 records; the 16 waves of the workgroup only share the windows of the
// names counted here (LDS atomics), the chunk pool (one LDS counter) and the small overflow table.  No barrier
// between setup and drain, so each wave's LDS round trips and loads hide behind the other fifteen.
// ---------------------------------------------------------------------------
constexpr uint32_t SW_WAVES = 16, SW_MAX_NS = 16, SW_AREA = SPLIT_REG_WORDS / SW_WAVES; // 800 words per wave
struct SplitWLds {
    uint32_t cnt[SW_WAVES * SW_MAX_NS], cfill[SW_WAVES * SW_MAX_NS], cbase[SW_WAVES * SW_MAX_NS];
    pu2_t pt[SW_MAX_NS]; // fine partition -> {first word of its region inside a wave's area, capacity in records}
    uint32_t ov_key[OV_SLOTS], ov_cnt[OV_SLOTS];
    uint32_t dummy[64];
    uint32_t pool_next, ovn, nfwd, next_tail; // next_tail: the next chunk of the slot's last quarter, taken by whoever is free
    uint32_t tbl[256];
    uint32_t name[32], mn[32], mx[32], svc[32], svm[32], org[32];
};
static_assert(sizeof(SplitWLds) % 16 == 0, "the regions follow the struct in LDS");
constexpr size_t SPLITW_LDS_BYTES = sizeof(SplitWLds) + (size_t)(SPLIT_REG_WORDS + PEEL_WORDS) * 4;
static_assert(SPLITW_LDS_BYTES <= 160 * 1024, "level 2 must fit one CU's LDS");

__global__ __launch_bounds__(1024, 4) void k_split_waves(const uint32_t *__restrict__ in_records,
                                                         const uint32_t *__restrict__ in_cdesc,
                                                         const uint32_t *__restrict__ in_sorted,
                                                         const uint32_t *__restrict__ in_part_start,
                                                         const uint32_t *__restrict__ in_slots,
                                                         const uint32_t *__restrict__ in_nslots,
                                                         const uint32_t *__restrict__ pool_start, uint32_t nmetrics,
                                                         uint32_t kp, uint32_t log_mpp2, uint32_t log_w, uint32_t log_ns,
                                                         const uint8_t *__restrict__ g_remap,
                                                         const uint8_t *__restrict__ g_inv,
                                                         const pu2_t *__restrict__ g_pt2, const SurveyStat S,
                                                         uint32_t *__restrict__ records, uint32_t *__restrict__ cdesc,
                                                         uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                         uint32_t *__restrict__ g_stats)
{
    constexpr int BLOCK = 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char v3_smem[];
    SplitWLds &L = *reinterpret_cast<SplitWLds *>(v3_smem);
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(v3_smem);
    constexpr uint32_t REG_W = sizeof(SplitWLds) / 4, WIN_W = REG_W + SPLIT_REG_WORDS;
    constexpr uint32_t CNT_W = offsetof(SplitWLds, cnt) / 4, DUMMY_W = offsetof(SplitWLds, dummy) / 4;
    uint32_t *win = lds32 + WIN_W;
    const uint32_t slot = blockIdx.x;
    if (slot >= *in_nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p1 = in_slots[3 * slot], first = in_slots[3 * slot + 1], cnt = in_slots[3 * slot + 2];
    const uint32_t *list = in_sorted + in_part_start[p1] + first;
    const uint32_t pool_base = pool_start[slot];
    const uint32_t W = 1u << log_w, mmask = (1u << log_mpp2) - 1u, ns = 1u << log_ns;

    // Requested before the setup and its four barriers (a slot's loads are a dependent chain: slot -> chunk list ->
    // records): this wave's first 64 dealt chunk indices, their descriptors, and -- not waiting for the descriptors --
    // its first two chunks.  Three quarters of the slot's chunks are dealt out (wave, wave + 16, ...); see below.
    auto load_chunk = [&](uint32_t cidx, u4_t (&dst)[CHUNK / 256]) {
        const u4_t *src = reinterpret_cast<const u4_t *>(in_records + (size_t)cidx * CHUNK) + lane;
#pragma unroll
        for (uint32_t k = 0; k < CHUNK / 256; k++) dst[k] = __builtin_nontemporal_load(src + k * 64);
    };
    constexpr uint32_t WSTEP = BLOCK / 64, DEPTH = 2;
    u4_t buf[DEPTH][CHUNK / 256];
    const uint32_t cnt_dealt = (cnt - cnt / 4u) & ~(WSTEP - 1u);
    const uint32_t mine = cnt_dealt / WSTEP; // dealt chunks of this wave
    uint32_t nb = min(mine, 64u), my_cid = 0, my_cn = 0;
    const uint32_t c0 = list[0];
    if (lane < nb) my_cid = list[wave + lane * WSTEP];
    const uint32_t srec0 = in_records[(size_t)c0 * CHUNK + tid]; // (masked by n0 below: a chunk is 1 024 slots whatever it holds)
    const uint32_t n0 = in_cdesc[c0] & CD_MASK;
    if (lane < nb) my_cn = in_cdesc[my_cid] & CD_MASK;
#pragma unroll
    for (uint32_t d = 0; d < DEPTH; d++) // (unconditional: a wave without dealt chunks reads chunk 0 and ignores it)
        load_chunk(__builtin_amdgcn_readlane(my_cid, min(d, max(nb, 1u) - 1u)), buf[d]);

    // ---- setup (as k_split_records): rank table, the names counted here, window origins
    const uint32_t srec = tid < n0 ? srec0 : 0u;
    if (tid < 256) L.tbl[tid] = g_remap[p1 * 256u + tid];
    if (tid < 32) {
        uint32_t name = INVALID, mn = INVALID, mx = 0, svc = 0, svm = 0;
        if (tid < kp) {
            const uint32_t m = ((uint32_t)g_inv[p1 * 256u + tid] << V3_LOG_NP) | p1;
            if (m < nmetrics) {
                name = m;
                svc = sv_count(S, m);
                if (svc) {
                    mn = 65535u - S.mninv[m];
                    mx = S.mx[m];
                    svm = sv_mean(S, m);
                }
            }
        }
        L.name[tid] = name;
        L.mn[tid] = mn;
        L.mx[tid] = mx;
        L.svc[tid] = svc;
        L.svm[tid] = svm;
    }
    for (uint32_t i = tid; i < kp << log_w; i += BLOCK) win[i] = 0;
    if (tid < SW_WAVES * SW_MAX_NS) { L.cnt[tid] = 0; L.cfill[tid] = CHUNK; L.cbase[tid] = INVALID; }
    if (tid < SW_MAX_NS) {
        // a wave's share of the partition's region layout (k_survey_remap): 1/16 of every capacity
        uint32_t off = 0, cap = 0;
        if (tid < ns) {
            for (uint32_t i = 0; i <= tid; i++) {
                const uint32_t c = (g_pt2[p1 * V3_MAX_NS + i].y / SW_WAVES) & ~3u;
                if (i < tid) off += c; else cap = c;
            }
        }
        L.pt[tid] = (pu2_t){off, cap};
    }
    ov_init(L.ov_key, L.ov_cnt, tid, BLOCK);
    if (tid == 0) { L.pool_next = 0; L.ovn = 0; L.nfwd = 0; L.next_tail = 0; }
    __syncthreads();
    if (tid < n0) {
        const uint32_t r = L.tbl[(srec >> 16) & 0xffu], b = srec & 0xffffu;
        if (r < kp) {
            if (b < L.mn[r]) atomicMin(&L.mn[r], b);
            if (b > L.mx[r]) atomicMax(&L.mx[r], b);
        }
    }
    __syncthreads();
    if (tid < 32) L.org[tid] = v3_place(L.mn[tid], L.mx[tid], L.svc[tid], L.svm[tid], W);
    __syncthreads();
    if (tid < 256) {
        const uint32_t r = L.tbl[tid];
        L.tbl[tid] = r | ((r < kp ? L.org[r] : 0u) << 8);
    }
    __syncthreads();

    // ---- this wave's chunks
    const uint32_t wreg = REG_W + wave * SW_AREA;      // first word of this wave's regions
    const uint32_t wcnt = CNT_W + wave * SW_MAX_NS;    // word offset of this wave's region counters
    uint32_t novf = 0, nfwd = 0;                       // self-metrics (per lane, summed at the end)
    // flush: lane group g = lane >> gsh owns fine partition g (8 lanes each at ns = 8, 4 at ns = 16); lane q of the
    // group copies 16-byte piece q & 3 of lines q >> 2, q >> 2 + G / 4, ..
    const uint32_t gsh = 6u - log_ns, G = 1u << gsh, fs = lane >> gsh, fq = lane & (G - 1u);
    const pu2_t fpe = L.pt[fs];
    const uint32_t ftag = ((fs << V3_LOG_NP) | p1) << CD_SHIFT;
    auto classify8 = [&](const u4_t &a, const u4_t &b, uint32_t base_idx, uint32_t cn) {
        // records a.x .. a.w sit at base_idx + lane * 4 + 0 .. 3, b's 256 records later
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u4_t &r4 = j ? b : a;
            const uint32_t rr[4] = {r4.x, r4.y, r4.z, r4.w};
            uint32_t t[4], where[4], rank[4], rec[4];
            pu2_t pe[4];
            uint32_t fwd = 0, full = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) t[k] = L.tbl[(rr[k] >> 16) & 0xffu];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = base_idx + (uint32_t)j * 256u + lane * 4u + (uint32_t)k < cn;
                const uint32_t nw = t[k] & 0xffu, bin = rr[k] & 0xffffu, rel = bin - (t[k] >> 8);
                const bool here = valid && nw < kp && rel < W;
                const uint32_t fine = (nw >> log_mpp2) & (SW_MAX_NS - 1u);
                where[k] = here ? WIN_W + (nw << log_w) + rel : valid ? wcnt + fine : DUMMY_W + lane;
                rec[k] = (fine << 24) | ((nw & mmask) << 16) | bin;
                pe[k] = L.pt[fine];
                if (valid && !here) fwd |= 1u << k;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) rank[k] = atomicAdd(lds32 + where[k], 1u);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool fits = (fwd & (1u << k)) && rank[k] < pe[k].y;
                if ((fwd & (1u << k)) && !fits) full |= 1u << k;
                lds32[fits ? wreg + pe[k].x + rank[k] : DUMMY_W + lane] = rec[k];
            }
            if (full) { // this wave's region of the fine partition is full: the record is counted exactly right away
                novf += (uint32_t)__popc(full);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (full & (1u << k)) {
                        const uint32_t old = (rr[k] >> 16) & 0xffu, bin = rr[k] & 0xffffu;
                        const uint32_t id = (old << V3_LOG_NP) | p1;
                        // (v2_global_add: its range update LOOKS before it adds.  v3_global_add's unconditional min / max
                        // pair on the name's two range words serialises when a whole stream takes this path -- a wide
                        // stream's first call, on the default width: 593 ms of a 1e9-pair call, 25 ms with the look)
                        if (!ov_add(L.ov_key, L.ov_cnt, (id << 16) | bin, 1u)) v2_global_add(counts, ranges, id, bin, 1);
                    }
            }
        }
    };
    auto flush = [&]() {
        // the stores above and the reads below belong to the same wave: LDS executes a wave's operations in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (fs < ns) {
            const uint32_t c = min(lds32[wcnt + fs], fpe.y), full = c / LINE4, left = c % LINE4;
            if (full) {
                const uint32_t ci = wave * SW_MAX_NS + fs;
                const uint32_t cf = L.cfill[ci], cb = L.cbase[ci];
                const uint32_t room = (CHUNK - cf) / LINE4;
                uint32_t firstc = 0;
                if (fq == 0) {
                    if (full > room) {
                        const uint32_t over = full - room;
                        const uint32_t k = (over + CHUNK / LINE4 - 1) / (CHUNK / LINE4);
                        firstc = pool_base + atomicAdd(&L.pool_next, k);
                        if (cb != INVALID) hidden_store_u32(cdesc + cb, ftag | CHUNK);
#pragma nounroll
                        for (uint32_t i = 0; i + 1 < k; i++) hidden_store_u32(cdesc + firstc + i, ftag | CHUNK);
                        L.cbase[ci] = firstc + k - 1;
                        L.cfill[ci] = (over - (k - 1) * (CHUNK / LINE4)) * LINE4;
                    } else {
                        L.cfill[ci] = cf + full * LINE4;
                    }
                    nfwd += full * LINE4;
                }
                firstc = __shfl(firstc, lane & ~(G - 1u), 64); // the group leader's value
                const uint32_t dA = cb * CHUNK + cf, dB = firstc * CHUNK - room * LINE4;
                const uint32_t *src = lds32 + wreg + fpe.x;
#pragma nounroll
                for (uint32_t l = fq >> 2; l < full; l += G / 4) {
                    const uint32_t dst = (l < room ? dA : dB) + l * LINE4;
                    const pu4_t r4 = *reinterpret_cast<const pu4_t *>(src + l * LINE4 + (fq & 3u) * 4);
                    hidden_store_u4(records + dst + (fq & 3u) * 4, r4);
                }
                // the last partial line moves to the front (after every read of line 0: same wave, in order)
                if (fq < 4 && fq * 4 < left)
                    *reinterpret_cast<pu4_t *>(lds32 + wreg + fpe.x + fq * 4) =
                        *reinterpret_cast<const pu4_t *>(src + full * LINE4 + fq * 4);
                if (fq == 0) lds32[wcnt + fs] = left;
            } else if (fq == 0 && c != lds32[wcnt + fs]) {
                lds32[wcnt + fs] = c; // (overflowed without filling a line: cannot happen with capacities >= 16)
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // Three quarters of the slot's chunks are dealt out (wave, wave + 16, ...); the last quarter is taken chunk by chunk
    // by whichever wave is free.  (Time stamps of single slots: wave 0 was done after 90 - 105 us of a 140 - 160 us
    // slot and waited for the slowest wave: partially filled chunks and the names' skew make the waves' loads uneven.)
    for (uint32_t b0 = 0; b0 < mine; b0 += 64) {
        auto fetch = [&](uint32_t k, uint32_t at) { // k: position in the batch (wave-uniform)
            load_chunk(__builtin_amdgcn_readlane(my_cid, min(k, nb - 1u)), buf[at]);
        };
        if (b0) { // (more than 1 024 dealt chunks in the slot; the first batch was requested before the setup)
            nb = min(mine - b0, 64u);
            if (lane < nb) {
                my_cid = list[wave + (b0 + lane) * WSTEP];
                my_cn = in_cdesc[my_cid] & CD_MASK;
            }
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) fetch(d, d);
        }
        for (uint32_t k = 0; k < nb; k += DEPTH) {
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) {
                if (k + d < nb) { // wave-uniform
                    const uint32_t cnd = __builtin_amdgcn_readlane(my_cn, k + d);
                    classify8(buf[d][0], buf[d][1], 0u, cnd);
                    flush();
                    if (cnd > 512u) {
                        classify8(buf[d][2], buf[d][3], 512u, cnd);
                        flush();
                    }
                }
                fetch(k + d + DEPTH, d);
            }
        }
    }
    for (;;) { // the last quarter
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&L.next_tail, 1u);
        pos = cnt_dealt + __builtin_amdgcn_readfirstlane(pos);
        if (pos >= cnt) break;
        const uint32_t cid = __builtin_amdgcn_readfirstlane(list[pos]);
        const uint32_t cn1 = __builtin_amdgcn_readfirstlane(in_cdesc[cid] & CD_MASK);
        load_chunk(cid, buf[0]);
        classify8(buf[0][0], buf[0][1], 0u, cn1);
        flush();
        if (cn1 > 512u) {
            classify8(buf[0][2], buf[0][3], 512u, cn1);
            flush();
        }
    }

    // ---- drain: leftovers (< one line per wave and fine partition) and the open chunks' descriptors
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            novf += __shfl_xor(novf, d, 64);
            nfwd += __shfl_xor(nfwd, d, 64);
        }
    }
    __syncthreads();
    if (tid < SW_WAVES * SW_MAX_NS) {
        const uint32_t w = tid / SW_MAX_NS, fsub = tid % SW_MAX_NS, left = L.cnt[tid];
        if (fsub < ns && left) {
            uint32_t cf = L.cfill[tid], cb = L.cbase[tid];
            const uint32_t tag = ((fsub << V3_LOG_NP) | p1) << CD_SHIFT;
            if (cf == CHUNK) {
                if (cb != INVALID) cdesc[cb] = tag | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[tid] = cb;
            }
            const uint32_t *src = lds32 + REG_W + w * SW_AREA + L.pt[fsub].x;
            for (uint32_t i = 0; i < left; i++) records[(size_t)cb * CHUNK + cf + i] = src[i];
            L.cfill[tid] = cf + left;
            atomicAdd(&L.nfwd, left);
        }
    }
    if (lane == 0) {
        if (nfwd) atomicAdd(&L.nfwd, nfwd);
        if (novf) atomicAdd(&L.ovn, novf);
    }
    __syncthreads();
    if (tid < SW_WAVES * SW_MAX_NS && (tid % SW_MAX_NS) < ns && L.cbase[tid] != INVALID)
        cdesc[L.cbase[tid]] = ((((tid % SW_MAX_NS) << V3_LOG_NP) | p1) << CD_SHIFT) | L.cfill[tid];
    if (tid == 0) { // self-metrics: records forwarded to the reduce pass, records that found a region full
        atomicAdd(&g_stats[2], L.nfwd);
        if (L.ovn) atomicAdd(&g_stats[3], L.ovn);
    }

    // ---- flush the windows of the names counted here and the overflow table
    for (uint32_t r = wave; r < kp; r += BLOCK / 64) {
        const uint32_t name = L.name[r];
        if (name == INVALID) continue;
        const uint32_t org = L.org[r];
        uint32_t mn = INVALID, mx = 0;
        for (uint32_t i = lane; i < W; i += 64) {
            const uint32_t c = win[(r << log_w) + i];
            if (c) {
                const uint32_t b = org + i;
                lh::cell_add(counts, (size_t)name * LH_ROW_STRIDE + b, c);
                mn = min(mn, b);
                mx = max(mx, b);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor(mn, d, 64));
            mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
        }
        if (lane == 0 && mn != INVALID) {
            uint32_t *rg = ranges + 2 * (size_t)name;
            if (mn < rg[0]) atomicMin(&rg[0], mn);
            if (mx > rg[1]) atomicMax(&rg[1], mx);
        }
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += BLOCK)
        if (L.ov_key[i] != OV_EMPTY) v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
}

// ---------------------------------------------------------------------------
// Reduce: one workgroup per fine-partition work slot.  Fine partition q = fine << 8 | p1 holds the names of ranks
// fine * mpp2 .. + mpp2 - 1 of level-1 partition p1; record = fine << 24 | rank % mpp2 << 16 | bin.
// ---------------------------------------------------------------------------
// Windows of 2^13 and 2^14 bins (round 6): 8 / 4 names in twice the cells -- 128 KiB packed, ONE slot per CU.
constexpr uint32_t P3_LDSWORDS = P3_WINWORDS >> P3_PACK;
constexpr uint32_t P3_WIDE_FROM = LH_P3_PACKED ? LH_P3_WIDE_FROM : 99u; // (32-bit window cells: 65 536 of them do not fit a CU)
constexpr size_t p3_lds_bytes(uint32_t ldswords) { return (ldswords + 6 * 32 + 2 * OV_SLOTS + 2) * sizeof(uint32_t) + 16; }
constexpr size_t P3_LDS_BYTES = p3_lds_bytes(P3_LDSWORDS), P3_LDS_BYTES_WIDE = p3_lds_bytes(2 * P3_LDSWORDS);
static_assert(!LH_P3_PACKED || 2 * (P3_LDS_BYTES + 1024) <= 160 * 1024, "two slots per CU");
static_assert(!LH_P3_PACKED || P3_LDS_BYTES_WIDE <= 160 * 1024, "a slot of 2^14-bin windows fits one CU's LDS");

__global__ __launch_bounds__(P2_BLOCK, LH_P3_PACKED ? 8 : 4) void k_part_hist3(const uint32_t *__restrict__ records,
                                                         const uint32_t *__restrict__ cdesc,
                                                         const uint32_t *__restrict__ sorted,
                                                         const uint32_t *__restrict__ part_start,
                                                         const uint32_t *__restrict__ slots,
                                                         const uint32_t *__restrict__ nslots,
                                                         const uint32_t *__restrict__ part_chunks, uint32_t nmetrics,
                                                         uint32_t log_mpp2, uint32_t log_w,
                                                         const uint8_t *__restrict__ g_inv, const SurveyStat S,
                                                         uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                         uint32_t *__restrict__ g_stats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_org = h + ((1u << (log_mpp2 + log_w)) > P3_WINWORDS ? 2 * P3_LDSWORDS : P3_LDSWORDS), *s_mn = s_org + 32, *s_mx = s_mn + 32, *s_name = s_mx + 32, *s_svc = s_name + 32,
             *s_svm = s_svc + 32;
    uint32_t *ov_key = s_svm + 32, *ov_cnt = ov_key + OV_SLOTS;
    uint32_t *s_all = ov_cnt + OV_SLOTS; // [2]: lowest and highest bin of the slot's first chunk, whatever the name
    const uint32_t slot = blockIdx.x;
    if (slot >= *nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = slots[3 * slot], first = slots[3 * slot + 1], cnt = slots[3 * slot + 2];
    const uint32_t p1 = q & (V3_NP - 1u), fine = q >> V3_LOG_NP;
    const uint32_t *list = sorted + part_start[q] + first;
    const uint32_t mpp2 = 1u << log_mpp2, W = 1u << log_w, words = mpp2 << log_w;

    // A slot is a chain of dependent round trips with little work between them (slot -> chunk list -> descriptor ->
    // records -> LDS -> flush; config 4's slice: 2 167 slots, 8.5 per CU one after another).  Everything the chunk list
    // decides is therefore requested at once, BEFORE the window setup and its barriers: the slot's first chunk (window
    // placement), this wave's first 64 chunk indices, their descriptors and -- without waiting for the descriptors: a
    // chunk is 4 KiB whatever it holds -- the first two chunks themselves.
    auto load_chunk = [&](uint32_t cidx, u4_t (&dst)[CHUNK / 256]) {
        const u4_t *src = reinterpret_cast<const u4_t *>(records + (size_t)cidx * CHUNK) + lane;
#pragma unroll
        for (uint32_t k = 0; k < CHUNK / 256; k++) dst[k] = __builtin_nontemporal_load(src + k * 64);
    };
    // (two workgroups per CU: 32 waves with ONE chunk each in flight are the 128 KiB per CU that 16 waves with two were, in
    // half the registers -- the kernel has 64 per thread)
    constexpr uint32_t WSTEP = P2_BLOCK / 64, DEPTH = LH_P3_DEPTH;
    u4_t buf[DEPTH][CHUNK / 256];
    const uint32_t mine = cnt > wave ? (cnt - wave + WSTEP - 1) / WSTEP : 0u; // chunks of this wave: wave, wave + 16, ..
    uint32_t nb = min(mine, 64u), my_cid = 0, my_cn = 0;
    // (the small loads first, the 8 KiB of chunks last: a wave's loads return in order, and whoever needs a small one
    // must not wait for the chunks behind it)
    const uint32_t c0 = list[0];
    if (lane < nb) my_cid = list[wave + lane * WSTEP];
    uint32_t nm = INVALID; // thread l < mpp2: the l-th name of the fine partition and what the survey saw of it
    if (tid < mpp2) nm = ((uint32_t)g_inv[p1 * 256u + (fine << log_mpp2) + tid] << V3_LOG_NP) | p1;
    const uint32_t srec0 = records[(size_t)c0 * CHUNK + tid]; // (all 1 024 slots of a chunk exist: masked by n0 below)
    const uint32_t n0 = cdesc[c0] & CD_MASK;
    if (lane < nb) my_cn = cdesc[my_cid] & CD_MASK;
    uint32_t svc = 0, svm = 0;
    if (nm < nmetrics) {
        svc = sv_count(S, nm);
        svm = sv_mean(S, nm);
    }
#pragma unroll
    for (uint32_t d = 0; d < DEPTH; d++) // (unconditional: a wave without chunks reads chunk 0 and ignores it)
        load_chunk(__builtin_amdgcn_readlane(my_cid, min(d, max(nb, 1u) - 1u)), buf[d]);
    const uint32_t srec = tid < n0 ? srec0 : 0u;
    for (uint32_t i = tid; i < (words >> P3_PACK); i += P2_BLOCK) h[i] = 0;
    ov_init(ov_key, ov_cnt, tid, P2_BLOCK);
    if (tid < mpp2) {
        s_name[tid] = nm < nmetrics ? nm : INVALID;
        s_svc[tid] = svc;
        s_svm[tid] = svm;
        s_mn[tid] = INVALID;
        s_mx[tid] = 0;
    }
    if (tid == 0) { s_all[0] = INVALID; s_all[1] = 0; }
    __syncthreads();
    // windows from the slot's own records (what reaches this pass may be the tail that a level-2 window left over);
    // a name the first chunk does not hold falls back to what the survey saw of it, and a name neither of them knows
    // -- the rule on a small launch: a lane-sized one over 65 536 names gives a tail name half a dozen records, the
    // survey's sample none -- to where the slot's OTHER names have their values (names of one stream mostly live in the
    // same decades; a window around key 0 sent 20 % of such a launch's forwarded records to the miss path)
    {
        uint32_t lo = INVALID, hi = 0;
        if (tid < n0) {
            const uint32_t l = (srec >> 16) & 0xffu, b = srec & 0xffffu;
            if (b < s_mn[l]) atomicMin(&s_mn[l], b);
            if (b > s_mx[l]) atomicMax(&s_mx[l], b);
            lo = hi = b;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, (uint32_t)__shfl_xor(lo, d, 64));
            hi = max(hi, (uint32_t)__shfl_xor(hi, d, 64));
        }
        if (lane == 0 && lo != INVALID) { atomicMin(&s_all[0], lo); atomicMax(&s_all[1], hi); }
    }
    __syncthreads();
    if (tid < mpp2) {
        uint32_t mn = s_mn[tid], mx = s_mx[tid];
        const uint32_t m = s_name[tid];
        if (mn == INVALID && m != INVALID) {
            if (s_svc[tid]) {
                mn = 65535u - S.mninv[m];
                mx = S.mx[m];
            } else if (s_all[0] != INVALID) {
                mn = s_all[0];
                mx = s_all[1];
            }
        }
        s_org[tid] = v3_place(mn, mx, s_svc[tid], s_svm[tid], W);
        s_mn[tid] = INVALID; // reused as the flush ranges
        s_mx[tid] = 0;
    }
    __syncthreads();

    uint32_t nmiss = 0; // records outside their window (self-metric)
    // add_one returns true when the add took a packed field across a multiple of P3_SPILL (spill_one must follow, in the same
    // round).  The add returns the word as it was; the lane whose add crosses takes P3_SPILL off the field again and adds
    // them to the row in HBM.  With P pending hand-offs a field holds floor(field / P3_SPILL) == P at every instant (an add
    // that changes the quotient raises P, a hand-off lowers both), so once a round's hand-offs are done -- at the round's
    // barrier -- the field is below P3_SPILL; within a round it grows by at most 2^15.  It never carries into its neighbour,
    // and every count is in exactly one place: exact for any slot, any stream.
    auto add_one = [&](uint32_t rec, uint32_t c) -> bool {
        const uint32_t l = (rec >> 16) & 0xffu, b = rec & 0xffffu;
        const uint32_t rel = b - s_org[l];
        if (rel < W) {
            const uint32_t i = (l << log_w) + rel;
            if (P3_PACK) {
                const uint32_t sh = (i & 1u) << 4;
                const uint32_t f = (atomicAdd(&h[i >> 1], c << sh) >> sh) & 0xffffu;
                return ((f + c) >> LH_P3_SPILL_LOG) != (f >> LH_P3_SPILL_LOG);
            }
            atomicAdd(&h[i], c);
        } else {
            nmiss += c;
            if (!ov_add(ov_key, ov_cnt, (l << 16) | b, c)) p2_global_add(counts, ranges, s_name[l], b, c);
        }
        return false;
    };
    auto spill_one = [&](uint32_t rec) {
        const uint32_t l = (rec >> 16) & 0xffu, b = rec & 0xffffu;
        const uint32_t i = (l << log_w) + (b - s_org[l]);
        atomicSub(&h[i >> 1], P3_SPILL << ((i & 1u) << 4));
        p2_global_add(counts, ranges, s_name[l], b, P3_SPILL);
    };
    auto reduce_chunk = [&](const u4_t (&r4)[CHUNK / 256], uint32_t cn) {
        const bool full = cn == CHUNK; // wave-uniform
#pragma unroll
        for (uint32_t k = 0; k < CHUNK / 256; k++) {
            const uint32_t rr[4] = {r4[k].x, r4[k].y, r4[k].z, r4[k].w};
            uint32_t crossed = 0; // bit t: record t's add took its field across 2^15
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (full) {
                    // constant streams: the whole wave carries one record value -> one lane adds 64
                    const uint32_t f0 = __builtin_amdgcn_readfirstlane(rr[t]);
                    if (__builtin_amdgcn_ballot_w64(rr[t] != f0) == 0ull) {
                        if (lane == 0 && add_one(rr[t], 64u)) crossed |= 1u << t;
                    } else if (add_one(rr[t], 1u)) {
                        crossed |= 1u << t;
                    }
                } else if (k * 256 + lane * 4 + t < cn) {
                    if (add_one(rr[t], 1u)) crossed |= 1u << t;
                }
            }
            if (P3_PACK && __builtin_amdgcn_ballot_w64(crossed != 0) != 0ull) { // (rare: once per 16 384 counts of a cell)
#pragma unroll 1
                for (uint32_t t = 0; t < 4; t++)
                    if (crossed & (1u << t)) spill_one(t == 0 ? rr[0] : t == 1 ? rr[1] : t == 2 ? rr[2] : rr[3]);
            }
        }
    };
    // Each wave walks chunks wave, wave + 16, ...; their indices and descriptors are fetched 64 at a time (lane l holds
    // the wave's l-th chunk of the batch; the first batch above) and two chunks (8 KiB per wave) are in flight while the
    // older is reduced: see k_part_hist2.
    // Packed cells: the loop's trip count is the WORKGROUP's (wave 0 has the most chunks; a wave that has run out keeps
    // fetching its last chunk and ignores it) and every round ends at a barrier.  That is what makes the 16-bit fields safe
    // whatever the slot holds: a lane settles its hand-offs inside its round, so every field is below P3_SPILL at a
    // barrier, and one round adds at most 32 waves x 1 024 records = 2^15 counts to it.
    constexpr bool ROUNDS = P3_PACK != 0; // (A/B on one box, three runs each: the rounds' barriers cost nothing -- slice 0.81 ms with or without)
    const uint32_t rounds = ROUNDS ? (cnt + WSTEP - 1) / WSTEP : mine;
    for (uint32_t b0 = 0; b0 < rounds; b0 += 64) {
        auto fetch = [&](uint32_t k, uint32_t at) { // k: position in the batch (wave-uniform)
            load_chunk(__builtin_amdgcn_readlane(my_cid, min(k, nb - 1u)), buf[at]);
        };
        if (b0) { // (more than 1 024 chunks in the slot)
            nb = mine > b0 ? min(mine - b0, 64u) : 0u;
            if (lane < nb) {
                my_cid = list[wave + (b0 + lane) * WSTEP];
                my_cn = cdesc[my_cid] & CD_MASK;
            }
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) fetch(d, d);
        }
        const uint32_t nb_all = ROUNDS ? min(rounds - b0, 64u) : nb;
        for (uint32_t k = 0; k < nb_all; k += DEPTH) {
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) {
                if (k + d < nb) reduce_chunk(buf[d], __builtin_amdgcn_readlane(my_cn, k + d)); // wave-uniform
                fetch(k + d + DEPTH, d);
            }
            if (ROUNDS) __syncthreads();
        }
    }
    if (__builtin_amdgcn_ballot_w64(nmiss != 0) != 0ull) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) nmiss += __shfl_xor(nmiss, d, 64);
        if (lane == 0) atomicAdd(&g_stats[4], nmiss);
    }
    __syncthreads();

    // Flush: one uint64 atomic per occupied cell.  Always atomics: a slot that holds all the chunks of its fine
    // partition is the only writer of its names' cells in THIS launch, but not in the epoch buffer -- another stream
    // (a lane's lh_submit, lh_submit_device on a caller's stream, the small and direct kernels) may add to the same
    // cell while this launch runs, and a plain load + store would lose that increment (round 3 flushed such slots
    // with read-modify-writes: 9 % of this pass on config 4's slice, and wrong under concurrent ingest;
    // tests/test_gpu_part3.py::test_concurrent_streams_into_the_same_names).  The atomics return nothing, so a
    // thread issues all 32 of its cells back to back and never waits for memory.  (One LDS range update per wave,
    // not per cell: the wave's 64 cells are consecutive bins of one name, its lowest and highest occupied bins are
    // those of the first and the last lane that found a count.)
    static_assert(P3_WINWORDS / P2_BLOCK == 32 && P3_WINWORDS % P2_BLOCK == 0, "a thread flushes 32 cells (64 of 2^14-bin windows)");
    constexpr uint32_t FL = P3_WINWORDS / P2_BLOCK;
    for (uint32_t k0 = 0; k0 < words / P2_BLOCK; k0 += FL)
#pragma unroll 8
    for (uint32_t k = k0; k < k0 + FL; k++) {
        const uint32_t i = tid + k * P2_BLOCK, l = i >> log_w, b = s_org[l] + (i & (W - 1));
        const uint32_t c = P3_PACK ? (h[i >> 1] >> ((i & 1u) << 4)) & 0xffffu : h[i];
        const unsigned long long occ = __builtin_amdgcn_ballot_w64(c != 0);
        if (occ != 0ull && lane == 0) {
            atomicMin(&s_mn[l], b + (uint32_t)__builtin_ctzll(occ));
            atomicMax(&s_mx[l], b + 63u - (uint32_t)__builtin_clzll(occ));
        }
        if (c)
            lh::cell_add(counts, (size_t)s_name[l] * LH_ROW_STRIDE + b, c);
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += P2_BLOCK)
        if (ov_key[i] != OV_EMPTY) p2_global_add(counts, ranges, s_name[ov_key[i] >> 16], ov_key[i] & 0xffffu, ov_cnt[i]);
    __syncthreads();
    if (tid < mpp2 && s_mn[tid] != INVALID) { // (no look at the old range first: that load would be the slot's last round trip)
        uint32_t *r = ranges + 2 * (size_t)s_name[tid];
        atomicMin(&r[0], s_mn[tid]);
        atomicMax(&r[1], s_mx[tid]);
    }
}

// Reduce pass of a SMALL launch (a host-fed lane's half-buffer: 1 - 2 M pairs over 65 536 names leave a fine partition
// some hundred records): no windows at all.  k_part_hist3 gives every one of its >= 2 112 slots 128 KiB of LDS to
// zero, place and flush and a chain of five dependent loads, whatever the slot holds -- 226 us per lane launch with every
// CU's LDS taken (profiles/r05_hostfed_kernel_trace.txt), which held 16 lanes at 4.2 - 4.3 G pairs/s.  Here every wave
// walks the level-2 chunk descriptors themselves (no plan, no slots: the tag of a descriptor is the fine partition) and
// adds each record with three atomics that return nothing.  Alone, a call of 2^21 pairs takes 0.23 ms so against 0.29 ms
// windowed, one of 2^22 the same either way (0.35 ms), and 16 lanes reach 4.5 - 4.6 G pairs/s (profiles/
// r05_hostfed_direct_reduce.jsonl, r05_direct_reduce_sweep.txt): the launcher takes this pass up to
// PartTuning::v3_direct_max pairs (2^22, the largest lane launch).
__global__ __launch_bounds__(256) void k_part_direct3(const uint32_t *__restrict__ records,
                                                      const uint32_t *__restrict__ cdesc, uint32_t nchunks,
                                                      uint32_t nmetrics, uint32_t log_mpp2,
                                                      const uint8_t *__restrict__ g_inv, uint64_t *__restrict__ counts,
                                                      uint32_t *__restrict__ ranges)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    for (uint32_t c0 = wave * 64u; c0 < nchunks; c0 += nwaves * 64u) {
        const uint32_t d = c0 + lane < nchunks ? cdesc[c0 + lane] : INVALID;
        unsigned long long live = __builtin_amdgcn_ballot_w64(d != INVALID && (d & CD_MASK) != 0u);
        while (live) {
            const uint32_t k = (uint32_t)__builtin_ctzll(live);
            live &= live - 1ull;
            const uint32_t dk = __builtin_amdgcn_readlane(d, k), cn = dk & CD_MASK, q = dk >> CD_SHIFT;
            const uint32_t p1 = q & (V3_NP - 1u), fine = q >> V3_LOG_NP;
            const uint32_t *src = records + (size_t)(c0 + k) * CHUNK;
            const uint8_t *inv = g_inv + p1 * 256u + (fine << log_mpp2);
            for (uint32_t i = lane; i < cn; i += 64u) {
                const uint32_t rec = src[i];
                const uint32_t name = ((uint32_t)inv[(rec >> 16) & 0xffu] << V3_LOG_NP) | p1;
                if (name < nmetrics) v3_global_add(counts, ranges, name, rec & 0xffffu, 1u); // three atomics, nothing returns
            }
        }
    }
}

// Last kernel of a launch: the launch's self-metrics (g_stats, device memory, zero again afterwards) are added to the
// engine's pinned words.  One thread: a system-scope atomic from every workgroup of the passes above cost 0.18 ms
// per launch (256 of them on one host address).
//   rstat[0] level-1 region overflows (the engine's clustered-stream switch)  [2] level-1 records  [3] records
//   forwarded by level 2  [4] level-2 region overflows  [5] reduce-pass window misses
__global__ void k_v3_report(uint32_t *__restrict__ g_stats, unsigned long long *__restrict__ rstat,
                            unsigned long long pairs, uint32_t *__restrict__ hdr)
{
    // rstat[6]: pairs of the launches that have reported (the engine judges the other words against THIS count, not
    // against what it has enqueued: the reports arrive when a launch completes)
    if (threadIdx.x == 5 && rstat) __hip_atomic_fetch_add(rstat + 6, pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // is the survey stale?  The pairs level 1 did NOT turn into records (its hot windows took them) against the same
    // share of the first launch on these tables: stale_judge, lh_kernels_part2.h.  The lanes' launches share a set of
    // tables read-only; only the first launch on a set writes its header word, behind the survey that filled the set.
    // A launch on a stale survey keeps its level-1 region overflows to itself: the regions are sized for what the hot
    // windows leave (k_survey_plan_h), so a survey whose windows take nothing any more overflows them -- and the engine
    // reads rstat[0] as a stream clustered by name (it leaves the region kernels for 64 flips).  The stale word alone
    // makes the next call survey again.
    bool stale = false;
    if (threadIdx.x == 0) {
        const unsigned long long rec = g_stats[1];
        stale = stale_judge(hdr, rstat, pairs > rec ? pairs - rec : 0ull, pairs);
    }
    if (threadIdx.x < 5) {
        const uint32_t v = g_stats[threadIdx.x];
        g_stats[threadIdx.x] = 0;
        const uint32_t at = threadIdx.x == 0 ? 0u : threadIdx.x + 1u;
        if (v && rstat && !stale) __hip_atomic_fetch_add(rstat + at, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One launch instead of five memsets (each a fill kernel of its own with a gap before it: ~25 us of a 1 ms call):
// both levels' chunk descriptors to INVALID, their plan words and the launch's self-metrics to zero.
__global__ __launch_bounds__(256) void k_v3_prepare(uint32_t *__restrict__ cd1, uint32_t n1, uint32_t *__restrict__ pc1,
                                                    uint32_t w1, uint32_t *__restrict__ cd2, uint32_t n2,
                                                    uint32_t *__restrict__ pc2, uint32_t w2,
                                                    uint32_t *__restrict__ g_stats)
{
    const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
    for (uint32_t i = i0; i < n1; i += step) cd1[i] = INVALID;
    for (uint32_t i = i0; i < n2; i += step) cd2[i] = INVALID;
    for (uint32_t i = i0; i < w1; i += step) pc1[i] = 0;
    for (uint32_t i = i0; i < w2; i += step) pc2[i] = 0;
    if (i0 < 8) g_stats[i0] = 0;
}

// ---------------------------------------------------------------------------
// plan + launcher
// ---------------------------------------------------------------------------
struct Part3Plan {
    uint32_t log_w, log_mpp2, mpp2, kp, mpp, ns, log_ns, nq, extra1, extra2, pool_extra;
    bool waves; // level 2 by k_split_waves (<= 16 fine partitions per partition)
    uint32_t region_words, cells, max_cells, avail_bytes, g1, chunks_per_wg, nchunks1, nchunks2;
    size_t lds_dyn;
    size_t off_stat, off_aux, off_hk, off_hs, off_hdr, off_pt, off_remap, off_inv, off_pt2, off_hot, off_resume;
    size_t off_rec1, off_cd1, off_sorted1, off_small1, off_rec2, off_cd2, off_sorted2, off_small2, off_gstats, total;
};

// 1 025 .. 8 192 names: this generation takes the launches of streams that leave more than 1/8 of their mass outside the
// second generation's cold windows (lh_kernels_part2.h; both generations' surveys report that in bit 8 of the width word,
// this one's against part2_cold_log_w of the name count) -- 1e9 pairs of normal(0, 1e3) over 8 192 names 57 -> 7.7 ms, of
// 21 decades over 4 096 names 49 -> 8.9.
static bool part2_yields_to_part3(size_t n, uint32_t nmetrics, const PartTuning &tune)
{
    if (!tune.v3 || !(tune.v2_shape & 2u) || nmetrics <= 1024u || nmetrics > V2_MAX_NAMES) return false;
    if (n < (tune.v3_min_samples ? tune.v3_min_samples : V3_MIN_SAMPLES) || n > (size_t(1) << 31)) return false;
    return tune.v2_yield;
}

static bool make_plan3(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune, Part3Plan &P)
{
    if (!tune.v3 || !(tune.v2_shape & 2u)) return false; // shape bit 1 clear: the engine asked for the exact layout
    if (n < (tune.v3_min_samples ? tune.v3_min_samples : V3_MIN_SAMPLES) || n > (size_t(1) << 31)) return false;
    if (nmetrics > V3_MAX_NAMES) return false;
    if (nmetrics <= V2_MAX_NAMES && !part2_yields_to_part3(n, nmetrics, tune)) return false;
    P.log_w = std::min(V3_MAX_LOG_W, std::max(10u, tune.v3_log_w));
    P.log_mpp2 = (P.log_w >= P3_WIDE_FROM ? 16u : 15u) - P.log_w; // mpp2 x W = 32 768 window cells in the reduce pass, 65 536 from W = 2^13 on
    P.mpp2 = 1u << P.log_mpp2;
    P.kp = PEEL_WORDS >> P.log_w;                // names counted in place by level 2: 24, 12, 6, 3, 1
    P.mpp = (nmetrics + V3_NP - 1) >> V3_LOG_NP; // names per level-1 partition: 33 .. 256
    P.ns = 1;
    P.log_ns = 0;
    while (P.ns * P.mpp2 < P.mpp) { P.ns *= 2; P.log_ns++; } // fine partitions per level-1 partition: <= 64
    P.nq = V3_NP * P.ns;
    P.waves = P.ns <= SW_MAX_NS;
    // Work slots beyond one per partition.  Every level-2 slot zeroes and flushes 96 KiB of windows and every reduce
    // slot 128 KiB, whatever it holds: a slot should see some 10^5 records (config 4's 1.25e8-pair slice: 494 and
    // 2 167 slots instead of 1 024 and 3 072 saved 0.1 ms of 1.3).
    P.extra1 = (uint32_t)std::min<size_t>(V3_EXTRA1, std::max<size_t>(64, n >> 19));
    P.extra2 = (uint32_t)std::min<size_t>(V3_EXTRA2, std::max<size_t>(64, n >> 20));
    // chunks a level-2 slot may strand partially filled: one per fine partition, and per wave in k_split_waves
    P.pool_extra = (P.waves ? SW_WAVES : 1u) * P.ns + 1u;
    // the LDS behind the struct is shared by the level-1 regions and the hot windows (16-bit cells): the plan sizes the
    // regions from the survey and gives the windows the rest (k_survey_plan_h); P.cells is the windows' share if the
    // regions needed their upper bound
    P.region_words = v3_region_words(V3_TILE);
    const size_t fixed = sizeof(Scatter4Lds) + (size_t)P.region_words * 4;
    P.max_cells = tune.hot ? 65472u : 0u; // (cell offsets are 16-bit fields)
    P.cells = std::min<uint32_t>((uint32_t)((V2_LDS_TOTAL - fixed) / 2) & ~63u, P.max_cells);
    P.avail_bytes = (uint32_t)(V2_LDS_TOTAL - sizeof(Scatter4Lds));
    P.lds_dyn = V2_LDS_TOTAL;
    const size_t ntiles = n / V3_TILE; // whole tiles; the rest goes through the direct kernel
    if (ntiles == 0) return false;
    size_t g1 = (size_t)num_cus;
    if (tune.v3_g1_cap && g1 > tune.v3_g1_cap) g1 = tune.v3_g1_cap;
    if (g1 > (ntiles + 3) / 4) g1 = (ntiles + 3) / 4;
    if (g1 < 1) g1 = 1;
    P.g1 = (uint32_t)g1;
    const size_t tiles_per_wg = (ntiles + g1 - 1) / g1;
    P.chunks_per_wg = (uint32_t)(tiles_per_wg * (V3_TILE / CHUNK) + V3_NP + 1);
    P.nchunks1 = P.g1 * P.chunks_per_wg;
    // level 2: every level-1 slot re-scatters its chunks into <= cnt + ns + 1 chunks of its own pool
    P.nchunks2 = P.nchunks1 + (V3_NP + P.extra1) * P.pool_extra;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~size_t(255); return at; };
    // the survey's tables first: their offsets depend on the name count only, so every sub-launch of a call finds them
    P.off_stat = take((size_t)nmetrics * 16);
    P.off_aux = take((size_t)AUX_WORDS * 4);
    P.off_hk = take((size_t)V3_HN * sizeof(pu2_t));
    P.off_hs = take((size_t)V2_MAX_SLOTS * sizeof(pu4_t));
    P.off_hdr = take(64);
    P.off_pt = take((size_t)V3_NP * sizeof(pu2_t));
    P.off_remap = take(65536);
    P.off_inv = take(65536);
    P.off_pt2 = take((size_t)V3_NP * V3_MAX_NS * sizeof(pu2_t));
    P.off_rec1 = take((size_t)P.nchunks1 * CHUNK * sizeof(uint32_t));
    P.off_cd1 = take((size_t)P.nchunks1 * sizeof(uint32_t));
    P.off_sorted1 = take((size_t)P.nchunks1 * sizeof(uint32_t));
    P.off_small1 = take(small_words(V3_NP, P.extra1) * sizeof(uint32_t));
    P.off_rec2 = take((size_t)P.nchunks2 * CHUNK * sizeof(uint32_t));
    P.off_cd2 = take((size_t)P.nchunks2 * sizeof(uint32_t));
    P.off_sorted2 = take((size_t)P.nchunks2 * sizeof(uint32_t));
    P.off_small2 = take(small_words(P.nq, P.extra2) * sizeof(uint32_t));
    P.off_gstats = take(64); // the launch's self-metrics: with the records, not with the tables (launches may share tables)
    P.off_hot = take((size_t)P.max_cells * 2 * P.g1); // level 1's hot windows, one copy per workgroup (k_hot_reduce)
    P.off_resume = take((size_t)P.g1 * 4);            // level 1: the first tile each workgroup left to k_scatter_clustered
    P.total = o;
    return true;
}

size_t part3_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune)
{
    Part3Plan P;
    return make_plan3(n, nmetrics, num_cus, tune, P) ? P.total : 0;
}

// The block in two parts: the survey's tables (everything before the level-1 records: a function of the name count
// only) and the launch's own records, descriptors and plans.  Launches that keep their records in blocks of their own
// can share ONE set of tables read-only (the host-fed lanes: lh_engine's LaneTables).
size_t part3_tables_bytes(uint32_t nmetrics)
{
    Part3Plan P;
    PartTuning t;
    t.v3_min_samples = V3_TABLES_SAMPLES;
    if (nmetrics <= V2_MAX_NAMES) t.v2_yield = true; // (the tables' size depends on the name count only)
    return make_plan3(V3_TABLES_SAMPLES, nmetrics, 256, t, P) ? P.off_rec1 : 0;
}

size_t part3_records_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune)
{
    Part3Plan P;
    return make_plan3(n, nmetrics, num_cus, tune, P) ? P.total - P.off_rec1 : 0;
}

// survey_n / region_stat: as launch_ingest_pairs_part2.  span_stat: device-visible word (pinned host memory) that
// receives the survey's window-width class, or null.  tables: null = the survey's tables are the first part3_tables_bytes
// of `scratch`; otherwise they live THERE (part3_tables_bytes large) and `scratch` holds only the launch's own
// part3_records_bytes.
template <typename IDT>
static hipError_t launch_part3_t(const IDT *d_ids, const double *d_v, size_t n, size_t survey_n,
                                     uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                     uint32_t *d_err, void *scratch, size_t scratch_bytes, void *tables, int num_cus,
                                     const PartTuning &tune, unsigned long long *region_stat, uint32_t *span_stat,
                                     hipStream_t s)
{
    Part3Plan P;
    if (!make_plan3(n, nmetrics, num_cus, tune, P) || !scratch) return hipErrorInvalidValue;
    if (scratch_bytes < (tables ? P.total - P.off_rec1 : P.total)) return hipErrorInvalidValue;
    if (!part_aligned(d_ids, d_v)) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attributes
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter4<4, IDT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)V2_LDS_TOTAL);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_split_records),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPLIT_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_split_waves),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)SPLITW_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_hist3),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LH_P3_PACKED ? P3_LDS_BYTES_WIDE : P3_LDS_BYTES));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_survey_count_h<IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(5 * SVH_SLOTS * 4));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_clustered<IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_count),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_scatter),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    // base: where the offsets of the tables apply; rec + (offset - r0): the records' part
    unsigned char *base = static_cast<unsigned char *>(tables ? tables : scratch);
    unsigned char *rec = static_cast<unsigned char *>(scratch);
    const size_t r0 = tables ? P.off_rec1 : 0;
    const LevelPtrs L1 = level_ptrs(rec, P.off_rec1 - r0, P.off_cd1 - r0, P.off_sorted1 - r0, P.off_small1 - r0, V3_NP, P.extra1);
    const LevelPtrs L2 = level_ptrs(rec, P.off_rec2 - r0, P.off_cd2 - r0, P.off_sorted2 - r0, P.off_small2 - r0, P.nq, P.extra2);
    unsigned long long *g_cs = reinterpret_cast<unsigned long long *>(base + P.off_stat);
    uint32_t *g_mninv = reinterpret_cast<uint32_t *>(g_cs + nmetrics), *g_mx = g_mninv + nmetrics;
    const SurveyStat S{g_cs, g_mninv, g_mx};
    uint32_t *g_aux = reinterpret_cast<uint32_t *>(base + P.off_aux);
    pu2_t *g_hk = reinterpret_cast<pu2_t *>(base + P.off_hk);
    pu4_t *g_hs = reinterpret_cast<pu4_t *>(base + P.off_hs);
    uint32_t *g_hdr = reinterpret_cast<uint32_t *>(base + P.off_hdr);
    uint32_t *g_stats = reinterpret_cast<uint32_t *>(rec + (P.off_gstats - r0)); // self-metrics of the launch (zeroed below, read and zeroed by k_v3_report)
    pu2_t *g_pt = reinterpret_cast<pu2_t *>(base + P.off_pt);
    uint8_t *g_remap = base + P.off_remap, *g_inv = base + P.off_inv;
    pu2_t *g_pt2 = reinterpret_cast<pu2_t *>(base + P.off_pt2);
    uint32_t *g_hot = reinterpret_cast<uint32_t *>(rec + (P.off_hot - r0));
    uint32_t *g_resume = reinterpret_cast<uint32_t *>(rec + (P.off_resume - r0));

    hipLaunchKernelGGL(k_v3_prepare, dim3(1024), dim3(256), 0, s, L1.cdesc, P.nchunks1, L1.pc,
                       (uint32_t)small_words(V3_NP, P.extra1), L2.cdesc, P.nchunks2, L2.pc,
                       (uint32_t)small_words(P.nq, P.extra2), g_stats);
    hipError_t e = hipSuccess;
    if (survey_n) {
        // stat and aux are adjacent (both multiples of 256 bytes apart): one memset
        e = hipMemsetAsync(base + P.off_stat, 0, P.off_hk - P.off_stat, s);
        if (e != hipSuccess) return e;
        const size_t sv_tiles = (survey_n / 2 + 1023) / 1024;
        const unsigned sv_grid = (unsigned)std::min<size_t>(SVH_GRID, std::max<size_t>(1, sv_tiles));
        hipLaunchKernelGGL(k_survey_count_h<IDT>, dim3(sv_grid), dim3(1024), 5 * SVH_SLOTS * 4, s, d_ids, d_v, survey_n,
                           nmetrics, d_Tx, g_cs, g_mninv, g_mx);
        hipLaunchKernelGGL(k_survey_pick, dim3((nmetrics + 1023) / 1024), dim3(1024), 0, s, S, nmetrics, g_aux);
        hipLaunchKernelGGL(k_survey_mass<IDT>, dim3(sv_grid), dim3(1024), 0, s, d_ids, d_v, survey_n, nmetrics, d_Tx, S, g_aux);
        hipLaunchKernelGGL(k_survey_plan_h, dim3(1), dim3(V2_BLOCK), 0, s, S, g_aux, P.cells, V3_TILE, P.avail_bytes, P.max_cells, g_hk, g_hs, g_pt,
                           g_hdr, span_stat, nmetrics <= V2_MAX_NAMES ? part2_cold_log_w(nmetrics, tune.v2_shape) : 0u);
        hipLaunchKernelGGL(k_survey_remap, dim3(V3_NP), dim3(256), 0, s, S, g_hk, nmetrics, P.kp, P.log_mpp2, P.ns,
                           g_remap, g_inv, g_pt2);
    }
    const size_t nt_full = n / V3_TILE, done = nt_full * V3_TILE;
    hipLaunchKernelGGL((k_scatter4<4, IDT>), dim3(P.g1), dim3(1024), P.lds_dyn, s, d_ids, d_v, nt_full, nmetrics, d_Tx, g_hk,
                       g_hs, g_hdr, g_pt, g_hot, L1.records, L1.cdesc, P.chunks_per_wg, counts, ranges,
                       d_err, g_stats, g_resume);
    hipLaunchKernelGGL(k_scatter_clustered<IDT>, dim3(P.g1), dim3(1024), CL_LDS_BYTES, s, d_ids, d_v, nt_full, nmetrics,
                       d_Tx, g_resume, counts, ranges, d_err, g_stats);
    hipLaunchKernelGGL(k_hot_reduce, dim3(V2_MAX_SLOTS), dim3(1024), 0, s, g_hot, P.g1, g_hs, g_hdr, counts, ranges, 0u, nullptr);
    if (done < n) {
        e = launch_ingest_pairs(d_ids + done, d_v + done, n - done, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s);
        if (e != hipSuccess) return e;
    }
    e = run_plan(L1, P.nchunks1, V3_NP, P.pool_extra, P.extra1, s);
    if (e != hipSuccess) return e;
    if (P.waves)
        hipLaunchKernelGGL(k_split_waves, dim3(V3_NP + P.extra1), dim3(1024), SPLITW_LDS_BYTES, s, L1.records, L1.cdesc,
                           L1.sorted, L1.part_start, L1.slots, L1.nslots, L1.pool_start, nmetrics, P.kp, P.log_mpp2,
                           P.log_w, P.log_ns, g_remap, g_inv, g_pt2, S, L2.records, L2.cdesc, counts, ranges, g_stats);
    else
        hipLaunchKernelGGL(k_split_records, dim3(V3_NP + P.extra1), dim3(1024), SPLIT_LDS_BYTES, s, L1.records,
                           L1.cdesc, L1.sorted, L1.part_start, L1.slots, L1.nslots, L1.pool_start, nmetrics, P.kp,
                           P.log_mpp2, P.log_w, P.ns, g_remap, g_inv, g_pt2, S, L2.records, L2.cdesc, counts, ranges,
                           g_stats);
    if (n <= (tune.v3_direct_max ? tune.v3_direct_max : V3_DIRECT_MAX)) {
        // small launch: one global atomic per forwarded record, straight from the level-2 chunks (k_part_direct3)
        const unsigned g = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (P.nchunks2 + 255) / 256)); // a wave per 64 descriptors
        hipLaunchKernelGGL(k_part_direct3, dim3(g), dim3(256), 0, s, L2.records, L2.cdesc, P.nchunks2, nmetrics,
                           P.log_mpp2, g_inv, counts, ranges);
    } else {
        e = run_plan(L2, P.nchunks2, P.nq, 0u, P.extra2, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_part_hist3, dim3(P.nq + P.extra2), dim3(P2_BLOCK), P.log_w >= P3_WIDE_FROM ? P3_LDS_BYTES_WIDE : P3_LDS_BYTES, s, L2.records, L2.cdesc,
                           L2.sorted, L2.part_start, L2.slots, L2.nslots, L2.pc, nmetrics, P.log_mpp2, P.log_w, g_inv, S,
                           counts, ranges, g_stats);
    }
    hipLaunchKernelGGL(k_v3_report, dim3(1), dim3(64), 0, s, g_stats, region_stat, (unsigned long long)n, g_hdr);
    return hipGetLastError();
}

// The survey's first three kernels alone: the window-width class of the n pairs lands in *span_stat (pinned).  For the
// engine's FIRST third-generation call, whose sub-launches must be laid out for a width before any survey has reported
// one.  `tables`: part3_tables_bytes(nmetrics) of device memory (scratch of the probe; nothing is kept).
template <typename IDT>
static hipError_t launch_part3_probe_t(const IDT *d_ids, const double *d_v, size_t n, uint32_t nmetrics, const double *d_Tx,
                                       void *tables, int num_cus, const PartTuning &tune, uint32_t *span_stat, hipStream_t s)
{
    Part3Plan P;
    if (!make_plan3(n, nmetrics, num_cus, tune, P) || !tables || !part_aligned(d_ids, d_v)) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_survey_count_h<IDT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(5 * SVH_SLOTS * 4));
    if (e != hipSuccess) return e;
    unsigned char *base = static_cast<unsigned char *>(tables);
    unsigned long long *g_cs = reinterpret_cast<unsigned long long *>(base + P.off_stat);
    uint32_t *g_mninv = reinterpret_cast<uint32_t *>(g_cs + nmetrics), *g_mx = g_mninv + nmetrics;
    const SurveyStat S{g_cs, g_mninv, g_mx};
    uint32_t *g_aux = reinterpret_cast<uint32_t *>(base + P.off_aux);
    e = hipMemsetAsync(base + P.off_stat, 0, P.off_hk - P.off_stat, s);
    if (e != hipSuccess) return e;
    const size_t sv_tiles = (n / 2 + 1023) / 1024;
    const unsigned sv_grid = (unsigned)std::min<size_t>(SVH_GRID, std::max<size_t>(1, sv_tiles));
    hipLaunchKernelGGL(k_survey_count_h<IDT>, dim3(sv_grid), dim3(1024), 5 * SVH_SLOTS * 4, s, d_ids, d_v, n, nmetrics, d_Tx,
                       g_cs, g_mninv, g_mx);
    hipLaunchKernelGGL(k_survey_pick, dim3((nmetrics + 1023) / 1024), dim3(1024), 0, s, S, nmetrics, g_aux);
    hipLaunchKernelGGL(k_survey_mass<IDT>, dim3(sv_grid), dim3(1024), 0, s, d_ids, d_v, n, nmetrics, d_Tx, S, g_aux);
    hipLaunchKernelGGL(k_survey_plan_h, dim3(1), dim3(V2_BLOCK), 0, s, S, g_aux, P.cells, V3_TILE, P.avail_bytes, P.max_cells,
                       reinterpret_cast<pu2_t *>(base + P.off_hk), reinterpret_cast<pu4_t *>(base + P.off_hs),
                       reinterpret_cast<pu2_t *>(base + P.off_pt), reinterpret_cast<uint32_t *>(base + P.off_hdr), span_stat,
                       nmetrics <= V2_MAX_NAMES ? part2_cold_log_w(nmetrics, tune.v2_shape) : 0u);
    return hipGetLastError();
}

hipError_t launch_part3_probe(Ids d_ids, const double *d_v, size_t n, uint32_t nmetrics, const double *d_Tx, void *tables,
                              int num_cus, const PartTuning &tune, uint32_t *span_stat, hipStream_t s)
{
    return d_ids.width == 2 ? launch_part3_probe_t(d_ids.u16(), d_v, n, nmetrics, d_Tx, tables, num_cus, tune, span_stat, s)
                            : launch_part3_probe_t(d_ids.u32(), d_v, n, nmetrics, d_Tx, tables, num_cus, tune, span_stat, s);
}

hipError_t launch_ingest_pairs_part3(Ids d_ids, const double *d_v, size_t n, size_t survey_n,
                                     uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                     uint32_t *d_err, void *scratch, size_t scratch_bytes, void *tables, int num_cus,
                                     const PartTuning &tune, unsigned long long *region_stat, uint32_t *span_stat,
                                     hipStream_t s)
{
    return d_ids.width == 2 ? launch_part3_t(d_ids.u16(), d_v, n, survey_n, counts, ranges, nmetrics, d_Tx, d_err, scratch,
                                             scratch_bytes, tables, num_cus, tune, region_stat, span_stat, s)
                            : launch_part3_t(d_ids.u32(), d_v, n, survey_n, counts, ranges, nmetrics, d_Tx, d_err, scratch,
                                             scratch_bytes, tables, num_cus, tune, region_stat, span_stat, s);
}
