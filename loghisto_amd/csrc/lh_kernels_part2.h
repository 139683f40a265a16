// lh_kernels_part2.h -- second generation of the partitioned mixed (id, value) ingest, gfx950.
// Included at the end of lh_kernels_part.hip (same translation unit: it reuses the plan kernels, the chunk
// descriptor format and lh_bin_of).  Reference semantics are unchanged:
// Histogram(name, v) = histogramCache[name][compress(v)] += 1 (metrics.go:273-295, 316-322).
//
// What round 1 measured on k_scatter_samples (profiles/r01n): 35 % of the HBM roofline,
// 1.44x the algorithmic traffic, ~72 VALU per wave-sample, 40 % LDS bank conflicts; the cold path (rank atomic,
// LDS sort, per-record copy-out with a 16-byte table read and a 4-byte store each) was the cost, and only 45 %
// of a Zipf(1.0) stream avoided it.  This version changes four things:
//
//   1. ONE survey per launch instead of one per workgroup (k_survey_count / k_survey_plan, ~1 M sampled pairs
//      spread over the whole launch): per-name count, mean, min and max bin.  It yields, for every name, a
//      4 096-bin "cold" window origin, and for the most frequent names LDS "hot" windows whose WIDTH follows the
//      name's measured spread -- as many names as fit the CU's LDS (one 1 024-thread workgroup per CU owns
//      ~90 KiB of windows: ~96 names at 1 024 names instead of 16).
//   2. 2-BYTE RECORDS.  A cold sample is stored as (name-in-partition << log_w | bin - window origin): 14 bits.
//      The record IS the LDS index P2 adds to, so P2 is a load and one ds_add per record -- no window search,
//      no overflow table, no bounds test.  Record traffic halves (cold samples cost 12 + 2 + 2 B instead of
//      12 + 4 + 4).  Samples outside their name's cold window (rare: the window spans 41 e-folds) are counted
//      exactly through a small LDS table and global atomics.
//   3. LINE-GRANULAR COPY-OUT.  The tile's records are laid out in LDS line by line (64-byte lines of 32
//      records, staged leftovers first), so the copy-out is one aligned 16-byte LDS read and one 16-byte global
//      store per 8 records instead of four LDS operations and a 4-byte store per record.
//   4. 8 192-sample tiles on 1 024 threads: half the barriers per sample.
//
// Everything stays exact whatever the survey estimates: windows only decide where a sample is counted.
//
// Two scatter kernels share the survey, the record format and P2:
//   k_scatter3  (default, shapes 2 / 3)  fixed per-partition LDS regions sized by the survey; a record is placed by the
//               phase that classifies it (the LDS atomic's return value is its slot): two barriers per tile
//   k_scatter2  (shapes 0 / 1)           exact per-tile layout (scan of the partitions' line counts, points 3 and 4
//               above): any distribution of a tile over the partitions at full speed; the engine falls back to it
//               while a stream is clustered by name (lh_engine.cc, regions_disabled)

constexpr int V2_BLOCK = 1024;                         // survey kernels
constexpr int V2_SPT = 8;                              // samples per thread per tile
constexpr uint32_t LINE2 = 32;                         // records per line (64 B)
constexpr uint32_t V2_MAX_NAMES = 8192;
constexpr uint32_t V2_MAX_SLOTS = 512;                 // hot names
constexpr uint32_t HOT_WHOLE_SPAN = 320;               // sampled spans up to this many bins get a hot window over all of it
// smallest launch that takes this path: where it overtakes the cell-table kernel (profiles/r06_small_calls.txt).  Until round 6
// 2^25, the crossover with the first generation while every call surveyed itself (profiles/r02_c3_sizes.txt); with one survey
// per 32 calls this path is the faster one at every size the first generation takes.
constexpr size_t V2_MIN_SAMPLES = size_t(1) << 20;
constexpr uint32_t V2_LDS_TOTAL = 160 * 1024;
constexpr uint32_t SV_GRID = 256;                      // survey workgroups (one 4 096-sample tile each)
constexpr uint32_t P2V2_WINWORDS = 32768;              // 128 KiB of uint32 windows per P2 workgroup
constexpr uint32_t V2_MISSQ = 1024;                    // out-of-window samples a tile can queue (per parity)
constexpr uint32_t P2V2_SLOT_EXTRA = 256;              // P2 work slots beyond one per partition

typedef uint16_t rec16_t;

#ifndef LH_HOT_CAP_BIG2
#define LH_HOT_CAP_BIG2 4096u /* widest hot window of a name with >= 1/64 of the sampled mass (k_survey_plan) */
#endif
#ifndef LH_SC3_CAP_NUM
#define LH_SC3_CAP_NUM 6
#endif
constexpr uint32_t SC3_CAP_NUM = LH_SC3_CAP_NUM; // region capacity = expected records * CAP_NUM / 4 + 8 + one piece (6: 1.5 x)
#ifndef LH_SC3_PIECE
#define LH_SC3_PIECE 1
#endif
// The region scatter copies whole PIECES of SC3_PIECE consecutive 64-byte lines out of a partition's region (see
// V3_PIECE in lh_kernels_part3.h); up to SC3_PIECE * 32 - 1 records stay behind.  Measured at 1 024 names
// (profiles/r04_level1_experiments.txt): 128-byte pieces are SLOWER here (2.89 -> 3.04 ms per 1e9 pairs) -- with 2-byte
// records and 45 % of the pairs cold a partition gathers a piece only every fifth tile, and the 16 KiB of LDS come out
// of the hot windows -- so this path stays at one line; the third generation (4-byte records, 67 % cold) gains 8 %.
constexpr uint32_t SC3_PIECE = LH_SC3_PIECE, PIECE2 = SC3_PIECE * LINE2;
// upper bound of the sum of the partitions' capacities; `tile` = samples between two flushes
constexpr uint32_t region_records(uint32_t tile, uint32_t np) { return SC3_CAP_NUM * tile / 4u + (40u + PIECE2) * np; }


// Two shapes of the scatter pass (both exact; lh_set_option(LH_OPT_PART_V2_SHAPE) picks one):
//   <1024, 256>  one 1 024-thread workgroup per CU, 8 192-sample tiles, up to 256 partitions (4 names each at
//                1 024 names -> 8 192-bin cold windows); the workgroup owns all of the CU's LDS
//   < 512, 128>  two 512-thread workgroups per CU, 4 096-sample tiles, up to 128 partitions (8 names each ->
//                4 096-bin cold windows); fewer hot cells each, but the two workgroups' phases overlap
template <int BLOCK, int NPT> struct Scatter2LdsT {
    static constexpr int TILE = BLOCK * V2_SPT;
    static constexpr uint32_t MAXLINES = (TILE + NPT * (LINE2 - 1)) / LINE2 + 1;
    static constexpr uint32_t OWNERS = (MAXLINES + 7) & ~7u;
    uint32_t cnt[NPT], sf[NPT], cfill[NPT], cbase[NPT]; // persistent per partition
    pu2_t tA[NPT];           // this tile: {staged before | emitted << 8, sorted base (records)}
    uint32_t lbase[NPT];     // first line of the partition in this tile's emission
    uint32_t newsf[NPT];
    uint32_t dA[NPT], dB[NPT], room[NPT]; // destination of emitted record u: u < room ? dA + u : dB + u
    uint16_t owner[OWNERS];  // line -> partition
    __attribute__((aligned(16))) rec16_t sorted[MAXLINES * LINE2];
    __attribute__((aligned(16))) rec16_t stage[NPT * LINE2];
    uint32_t ov_key[OV_SLOTS], ov_cnt[OV_SLOTS];
    uint32_t missq[2][V2_MISSQ]; // samples outside their cold window (name << 16 | bin), by tile parity
    uint32_t missn[2];
    uint32_t dummy[64];      // target of the LDS operations of samples that leave nothing (keeps phases 1/3 branch-free)
    uint32_t pool_next, nlines;
};

// Per-name entry of the survey's plan, 8 bytes: cold origin | hot origin << 16, hot LDS base | hot width << 16.
// A name without a hot window has width 0.
struct NameEntry { uint32_t org; uint32_t hot; };

// ---------------------------------------------------------------------------
// Is the survey stale?  A survey is shared by up to LH_OPT_SURVEY_EVERY calls, and nothing the scatter kernels report
// (region overflows, window misses) moves when the VALUES of a stream shift under it: the hot windows then sit where
// the samples no longer are, every sample becomes a record, and the call costs half as much again (1 024 names,
// lognormal survey, four-valued stream: 4.34 ms per 1e9 pairs instead of 3.0 -- profiles/r05_fewvalued.txt) for up
// to 31 more calls.  So the launches keep the one number that does move: the share of their samples the hot windows
// took.  The first launch on a survey's tables stores it (hdr[HDR_BASE]); a later launch whose share is more than
// STALE_DROP below that adds the difference, in pairs, to the engine's pinned word rstat[RSTAT_STALE], which
// lh_engine's judge_tables counts with the other signs of an unhealthy call: the next call surveys again.
// Exactness does not depend on any of it.
// ---------------------------------------------------------------------------
constexpr uint32_t HDR_HITS = 8, HDR_TILES = 9, HDR_TICKET = 10, HDR_BASE = 11; // words of the 64-byte survey header
constexpr uint32_t HDR_REGION = 5, HDR_CELLS = 6; // LDS the plan gave the level-1 regions (records) and the hot windows (cells)
constexpr uint32_t HDR_OVF = 7;                   // k_scatter3: records of the launch that found their region full
constexpr uint32_t RSTAT_STALE = 7;                // engine's pinned words: [7] pairs a stale survey kept out of the hot windows
constexpr uint32_t STALE_VALID = 0x80000000u;      // hdr[HDR_BASE] = STALE_VALID | share in 1/65 536ths
constexpr uint32_t STALE_DROP = 65536u / 20u;      // 5 % of the launch's pairs

// share: what the hot windows took of `pairs`, in 1/65 536ths (first level of the third generation: pairs that did NOT
// become records).  One thread, after every workgroup's account is in.
// Returns true when the launch ran on a stale survey (and said so).
__device__ __forceinline__ bool stale_judge(uint32_t *__restrict__ hdr, unsigned long long *__restrict__ rstat,
                                            unsigned long long taken, unsigned long long pairs)
{
    if (!pairs) return false;
    const uint32_t share = (uint32_t)((taken << 16) / pairs);
    const uint32_t base = hdr[HDR_BASE];
    if (!(base & STALE_VALID)) {
        hdr[HDR_BASE] = STALE_VALID | share;
    } else if (share + STALE_DROP < (base & 0x1ffffu) && rstat) {
        const unsigned long long lost = (((unsigned long long)((base & 0x1ffffu) - share)) * pairs) >> 16;
        __hip_atomic_fetch_add(rstat + RSTAT_STALE, lost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------
// Survey
// ---------------------------------------------------------------------------
// g_stat layout (zero-initialised): cnt[M] u32 | mninv[M] u32 (max of 65535 - bin) | mx[M] u32 | sum[M] u64
template <typename IDT>
__global__ __launch_bounds__(V2_BLOCK) void k_survey_count(const IDT *__restrict__ ids,
                                                           const double *__restrict__ v, size_t n,
                                                           uint32_t nmetrics, const double *__restrict__ Tx,
                                                           uint32_t *__restrict__ g_cnt,
                                                           uint32_t *__restrict__ g_mninv,
                                                           uint32_t *__restrict__ g_mx,
                                                           unsigned long long *__restrict__ g_sum)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sv_smem[];
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(sv_smem);
    uint32_t *s_sum = s_cnt + nmetrics, *s_mninv = s_sum + nmetrics, *s_mx = s_mninv + nmetrics;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 4 * nmetrics; i += V2_BLOCK) s_cnt[i] = 0;
    __syncthreads();
    // workgroup w reads the 4 096 samples (2 048 pairs) that start at pair w * stride: spread over the launch
    const size_t npairs = n / 2; // an odd last sample is not surveyed
    const size_t stride = npairs / gridDim.x;
    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const size_t i = (size_t)blockIdx.x * stride + (size_t)j * V2_BLOCK + tid;
        if (i < npairs && (size_t)j * V2_BLOCK + tid < (stride ? stride : npairs)) {
            const typename IS::raw_t id2 = ip.ld(i);
            const pd2_t x2 = vp[i];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t id = h ? IS::second(id2) : IS::first(id2);
                if (id < nmetrics) {
                    const uint32_t bin = lh_bin_of(h ? x2.y : x2.x, Tx);
                    atomicAdd(&s_cnt[id], 1u);
                    atomicAdd(&s_sum[id], bin);
                    atomicMax(&s_mninv[id], 65535u - bin);
                    atomicMax(&s_mx[id], bin);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t m = tid; m < nmetrics; m += V2_BLOCK) {
        const uint32_t c = s_cnt[m];
        if (c) {
            atomicAdd(&g_cnt[m], c);
            atomicAdd(&g_sum[m], (unsigned long long)s_sum[m]);
            atomicMax(&g_mninv[m], s_mninv[m]);
            atomicMax(&g_mx[m], s_mx[m]);
        }
    }
}

// The sampled MASS by width (round 6, as k_survey_mass of the third generation): the same samples once more, each against
// its name's sampled mean bin -- g_mass[k] = samples within 512 << k bins of it, k = 0 .. 4; g_mass[5] = all of them (names
// with >= 32 samples).  The plan caps the most frequent names' hot windows by it.  g_mass[6] = sampled PAIRS of neighbours
// in the stream, g_mass[7] = those of ONE name: under Zipf(1) over 8 192 names 2 %, in a stream sorted or run-clustered by
// name nearly all -- such a stream stays with this generation however wide it is (part2_yields_to_part3: the third
// generation's cell table took 104 ms per 1e9 pairs of 21 decades sorted by name over 8 192 names, this one's 41).
template <typename IDT>
__global__ __launch_bounds__(V2_BLOCK) void k_survey_mass2(const IDT *__restrict__ ids, const double *__restrict__ v, size_t n,
                                                           uint32_t nmetrics, const double *__restrict__ Tx,
                                                           const uint32_t *__restrict__ g_cnt,
                                                           const unsigned long long *__restrict__ g_sum,
                                                           uint32_t *__restrict__ g_mass)
{
    __shared__ uint32_t s_in[8];
    const uint32_t tid = threadIdx.x;
    if (tid < 8) s_in[tid] = 0;
    __syncthreads();
    const size_t npairs = n / 2;
    const size_t stride = npairs / gridDim.x;
    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
    uint32_t in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const size_t i = (size_t)blockIdx.x * stride + (size_t)j * V2_BLOCK + tid;
        if (i < npairs && (size_t)j * V2_BLOCK + tid < (stride ? stride : npairs)) {
            const typename IS::raw_t id2 = ip.ld(i);
            const pd2_t x2 = vp[i];
            in[6]++;
            in[7] += IS::first(id2) == IS::second(id2) ? 1u : 0u;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t id = h ? IS::second(id2) : IS::first(id2);
                const uint32_t c = id < nmetrics ? g_cnt[id] : 0u;
                if (c >= 32u) {
                    const uint32_t bin = lh_bin_of(h ? x2.y : x2.x, Tx), mean = (uint32_t)(g_sum[id] / c);
                    const uint32_t d = bin > mean ? bin - mean : mean - bin;
                    in[5]++;
#pragma unroll
                    for (uint32_t k = 0; k < 5; k++) in[k] += d < (512u << k) ? 1u : 0u;
                }
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
    if (tid < 8 && s_in[tid]) atomicAdd(&g_mass[tid], s_in[tid]);
}

// sums of a and b over the workgroup, returned to every thread
__device__ __forceinline__ void block_sum2(uint32_t a, uint32_t b, uint32_t *s_a, uint32_t *s_b, uint32_t &ta,
                                           uint32_t &tb)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
    }
    __syncthreads(); // s_a / s_b reuse
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    ta = 0;
    tb = 0;
#pragma unroll
    for (int w = 0; w < V2_BLOCK / 64; w++) { ta += s_a[w]; tb += s_b[w]; }
}

// What the plan needs to know of the region scatter (k_scatter3) to lay out ITS share of the LDS as well: round 6.
// Until then the host reserved the regions' upper bound (1.5 x a tile in which EVERY sample is a record + 72 records per
// partition: 60 KiB at 256 partitions) and the windows got the rest; the regions a Zipf stream needs are a third smaller,
// because the names with hot windows leave only their tails as records.  The plan now sizes the regions from the names'
// counts with the hot names' reduced (cold_share), and gives what that frees to the windows: tile = 0 -> no regions
// (k_scatter2's exact layout), cells as given.  A survey gone stale (the values moved, every sample is a record) then
// overflows the regions -- the launch says so itself (stale_judge) and its overflow count is NOT passed on to the engine,
// which would read it as a stream clustered by name.
// What a name with a hot window of `want` bins still sends to its partition's region, of `cnt` sampled values over a span
// of `span` bins: everything but 3/4 of the window's share of the span.  (A bell-shaped name keeps 80 - 99 % in a window
// of 3/4 of its span and is charged ~45 %; a name spread evenly over 21 decades keeps an eighth in its 512 bins and is
// charged 91 %: the regions must not overflow on a HEALTHY survey -- the engine reads overflows as a stream clustered by
// name and leaves the region kernel for 64 flips.)
__device__ __forceinline__ uint32_t cold_share(uint32_t cnt, uint32_t want, uint32_t span)
{
    const uint32_t w = want < span ? want : span;
    return cnt - (uint32_t)(((unsigned long long)cnt * 3u * w) / (4ull * span));
}

struct RegionFit {
    uint32_t tile;        // samples between two flushes (0: the kernel has no regions)
    uint32_t log_np;
    uint32_t avail_bytes; // LDS for regions + hot windows together
    uint32_t cell_bytes;  // 2 (k_scatter3: 16-bit cells) or 4
    uint32_t cell_gap;    // unused cells between two windows (equal bins of different names on different banks)
    uint32_t max_cells;
};

// One workgroup.  Thread t owns names [t * E, (t + 1) * E), E = ceil(M / 1024) <= 8.
// hdr: [0] hot names, [1] cells used, [2] surveyed samples, [3] surveyed samples of the hot names, [HDR_REGION] records of
// LDS of the regions, [HDR_CELLS] cells of the window area; g_pt[p] = {first record of partition p's region, capacity}
__global__ __launch_bounds__(V2_BLOCK) void k_survey_plan(const uint32_t *__restrict__ g_cnt,
                                                          const uint32_t *__restrict__ g_mninv,
                                                          const uint32_t *__restrict__ g_mx,
                                                          const unsigned long long *__restrict__ g_sum,
                                                          const uint32_t *__restrict__ g_mass,
                                                          uint32_t nmetrics, uint32_t log_w, uint32_t cells_in,
                                                          const RegionFit rf,
                                                          NameEntry *__restrict__ nt, pu4_t *__restrict__ hs,
                                                          uint32_t *__restrict__ hdr, pu2_t *__restrict__ g_pt,
                                                          uint32_t *span_out)
{
    __shared__ uint32_t s_a[V2_BLOCK / 64], s_b[V2_BLOCK / 64];
    __shared__ uint32_t s_pc[512], s_cap[512];
    const uint32_t gap = rf.cell_gap;
    constexpr uint32_t EMAX = V2_MAX_NAMES / V2_BLOCK;
    const uint32_t tid = threadIdx.x;
    const uint32_t E = (nmetrics + V2_BLOCK - 1) / V2_BLOCK;
    const uint32_t m0 = tid * E;
    const uint32_t W = 1u << log_w;
    uint32_t cnt[EMAX], want[EMAX], mean[EMAX], corg[EMAX];
    uint32_t mysum = 0, whole = 0; // whole: bit e = name e of this thread keeps its whole sampled span
    uint32_t lobe = 0, lo_end[EMAX], hi_end[EMAX]; // lobe: bit e = two lobes either side of key 0; the span's ends
    uint32_t span[EMAX];                           // the sampled span in bins
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        cnt[e] = 0;
        span[e] = 1;
        lo_end[e] = hi_end[e] = 0;
        want[e] = 0;
        mean[e] = 32768u;
        corg[e] = 32768u - W / 2;
        const uint32_t m = m0 + e;
        if (e < E && m < nmetrics) {
            const uint32_t c = g_cnt[m];
            cnt[e] = c;
            mysum += c;
            if (c) {
                const uint32_t mn = 65535u - g_mninv[m], mx = g_mx[m];
                span[e] = mx - mn + 1u;
                mean[e] = (uint32_t)(g_sum[m] / c);
                // cold window: centred on the sampled span when it fits, on the mean bin otherwise
                const uint32_t centre = (mx - mn < W) ? (mn + mx + 1) >> 1 : mean[e];
                uint32_t o = centre > W / 2 ? centre - W / 2 : 0u;
                if (o > 65536u - W) o = 65536u - W;
                corg[e] = o;
                // hot window the name would like: 3/4 of the sampled span (the span of a few thousand samples of
                // a bell-shaped bin distribution is ~ +-3.5 sigma; 3/4 of it keeps ~99 %), in steps of 64 bins
                if (c >= 16) {
                    uint32_t w = (((mx - mn + 1) * 3u / 4u) + 63u) & ~63u;
                    want[e] = w < 64u ? 64u : w;
                    // A sampled span of at most HOT_WHOLE_SPAN bins is kept WHOLE, centred on the span and exempt from the
                    // 256-bin cap below: the outer values of a few-valued stream carry as much as the inner ones (k = 8
                    // values 40.5 bins apart, a 285-bin span: 3.76 -> 2.98 ms per 1e9 pairs; lognormal, k = 4 and k = 16
                    // streams unchanged -- profiles/r05_fewvalued.txt, an A/B/A run on one box).  No name that gets a
                    // window on a bell-shaped stream has so narrow a span: at sigma = 1 the span of the ~1 000 samples such
                    // a name needs is ~660 bins.  (The whole span up to 512 bins cost lognormal streams 3 % in round 4.)
                    if (mx - mn + 1 <= HOT_WHOLE_SPAN) {
                        want[e] = ((mx - mn + 1) + 63u) & ~63u;
                        mean[e] = (mn + mx + 1) >> 1;
                        whole |= 1u << e;
                    }
                    // SIGNED values: two lobes of bins, one either side of key 0 (bin 32 768), and the mean bin lies in the
                    // gap between them -- a window centred there takes nothing (normal(0, 1e3) over 1 024 names: 3.76 ms
                    // per 1e9 pairs against 2.94 for one-signed values).  Such a name's window goes to the outer end of the
                    // lobe the mean leans to: that is where a lobe's mass is (bins are logarithmic in |v|).
                    if (mn + 64u < 32768u && mx > 32768u + 64u && mx - mn > 1024u) {
                        lobe |= 1u << e;
                        lo_end[e] = mn;
                        hi_end[e] = mx;
                    }
                }
            }
        }
    }
    uint32_t total_cnt, wide_cnt;
    {   // the sampled mass of the names whose span is wider than the 8 192-bin reduce window of four names per partition
        uint32_t mywide = 0;
#pragma unroll
        for (uint32_t e = 0; e < EMAX; e++)
            if (cnt[e] >= 16u && span[e] > 8192u) mywide += cnt[e];
        block_sum2(mysum, mywide, s_a, s_b, total_cnt, wide_cnt);
    }
    // ... reported to the engine (pinned word, as the third generation reports its window class): 14 = at least 5 % of the
    // sampled mass would miss 8 192-bin windows -- the following calls over <= 1 024 names take the WIDE shape (512
    // partitions of two names x 16 384 bins).  Otherwise the stream's width class by its mass (10 .. 13, k_survey_mass2):
    // above 1 024 names a stream wider than this generation's cold windows goes to the third (part2_yields_to_part3).
    uint32_t cls = 13;
    {
        const uint32_t mass = g_mass[5];
        for (uint32_t k = 10; k < 13; k++)
            if ((unsigned long long)g_mass[k - 10u] * 100u >= (unsigned long long)mass * 99u) { cls = k; break; }
        if (!mass) cls = 10;
        if ((unsigned long long)g_mass[3] * 100u < (unsigned long long)mass * 99u) cls = 14; // (not within +-4 096 bins)
        if ((unsigned long long)wide_cnt * 20ull >= total_cnt && nmetrics <= 1024u) cls = 14;
        // bit 8: more than 1/8 of the mass lies outside THIS generation's cold windows (2^log_w bins around the mean)
        // (... unless the stream is clustered by name: half of its neighbours are one name)
        if (nmetrics > 1024u && log_w >= 10u && log_w <= 14u && (unsigned long long)g_mass[log_w - 10u] * 8u < (unsigned long long)mass * 7u &&
            g_mass[7] * 2u < g_mass[6])
            cls |= 0x100u;
    }
    if (tid == 0 && span_out && total_cnt)
        __hip_atomic_store(span_out, cls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // names that carry at least 1/64 of the surveyed samples may have windows as wide as half the stream's width class
    // (the smallest of 2^10 .. 2^14 bins within half of which, around their name's mean, 99 % of the samples lie: 512 bins
    // on a lognormal stream, up to LH_HOT_CAP_BIG2 on a wide one -- where values spread evenly over their span a cell
    // earns cnt / span whatever the window's width, so the cells belong to the most frequent names' whole spans), the
    // others 256.  By the stream's mass, not the name's sampled span: one far outlier stretches that.
    const uint32_t big = total_cnt / 64u;
    uint32_t cap_big = 512u;
    {
        const uint32_t mass = g_mass[5];
        uint32_t lw = 14;
        for (uint32_t k = 10; k < 14; k++)
            if ((unsigned long long)g_mass[k - 10u] * 100u >= (unsigned long long)mass * 99u) { lw = k; break; }
        if (mass) cap_big = 1u << (lw - 1u);
        cap_big = cap_big < 512u ? 512u : cap_big > LH_HOT_CAP_BIG2 ? LH_HOT_CAP_BIG2 : cap_big;
    }
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        if (want[e]) {
            const uint32_t cap = cnt[e] >= big ? cap_big : 256u;
            if (want[e] > cap && !(whole & (1u << e))) want[e] = cap;
        }
    }
    // the smallest tau >= 16 such that the windows of every name with cnt >= tau fit `cells` and the slot table
    auto pick = [&](uint32_t cells) {
        uint32_t flo = 15, fhi = (1u << 21) + 1; // fhi selects nothing: feasible
        if (cells < 64) flo = fhi - 1;           // no LDS for hot windows at all
        while (fhi - flo > 1) {
            const uint32_t mid = flo + (fhi - flo) / 2;
            uint32_t sw = 0, sn = 0;
#pragma unroll
            for (uint32_t e = 0; e < EMAX; e++)
                if (want[e] && cnt[e] >= mid) { sw += want[e] + gap; sn++; }
            uint32_t tw, tn;
            block_sum2(sw, sn, s_a, s_b, tw, tn);
            if (tw <= cells && tn <= V2_MAX_SLOTS) fhi = mid; else flo = mid;
        }
        return fhi;
    };
    uint32_t cells = cells_in, region_recs = 0;
    uint32_t tau = pick(cells);
    if (rf.tile) {
        // the regions for THIS choice of hot names, then the windows once more with what the regions leave (a superset of
        // the first choice, so the regions stay large enough)
        const uint32_t np = 1u << rf.log_np;
        if (tid < 512) s_pc[tid] = 0;
        __syncthreads();
#pragma unroll
        for (uint32_t e = 0; e < EMAX; e++)
            if (cnt[e]) atomicAdd(&s_pc[(m0 + e) & (np - 1u)], (want[e] && cnt[e] >= tau) ? cold_share(cnt[e], want[e], span[e]) : cnt[e]);
        __syncthreads();
        uint32_t cap = 0;
        if (tid < np) {
            const uint32_t est = total_cnt ? (uint32_t)(((unsigned long long)s_pc[tid] * rf.tile) / total_cnt) : rf.tile / np;
            cap = (est * SC3_CAP_NUM / 4u + 8u + PIECE2 + 31u) & ~31u;
            if (cap > rf.tile + PIECE2) cap = rf.tile + PIECE2; // leftover (< one piece) + a whole tile
        }
        if (tid < 512) s_cap[tid] = cap;
        __syncthreads();
        uint32_t base = 0;
        for (uint32_t i = 0; i < np; i++) {
            if (i == tid && tid < np) g_pt[tid] = (pu2_t){base, cap};
            base += s_cap[i];
        }
        region_recs = base; // (a multiple of 32: every capacity is)
        const uint32_t left = rf.avail_bytes > region_recs * 2u ? rf.avail_bytes - region_recs * 2u : 0u;
        cells = (left / rf.cell_bytes) & ~63u;
        if (cells > rf.max_cells) cells = rf.max_cells; // (at least cells_in: the capacities add up to less than the bound)
        tau = pick(cells);
    }
    uint32_t sw = 0, sn = 0, sc = 0;
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++)
        if (want[e] && cnt[e] >= tau) { sw += want[e] + gap; sn++; sc += cnt[e]; }
    // exclusive scans of (cells, slots) in name order
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
    uint32_t cellpos = basew + incw - sw, slot = basen + incn - sn;
    uint32_t hot_cnt_total, dummy2;
    block_sum2(sc, 0, s_a, s_b, hot_cnt_total, dummy2);
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        const uint32_t m = m0 + e;
        if (e < E && m < nmetrics) {
            NameEntry ne;
            ne.org = corg[e];
            ne.hot = 0;
            if (want[e] && cnt[e] >= tau) {
                const uint32_t w = want[e];
                uint32_t o = mean[e] > w / 2 ? mean[e] - w / 2 : 0u;
                if (lobe & (1u << e)) o = mean[e] >= 32768u ? (hi_end[e] + 1u > w ? hi_end[e] + 1u - w : 0u) : lo_end[e];
                if (o > 65536u - w) o = 65536u - w;
                ne.org |= o << 16;
                ne.hot = cellpos | (w << 16);
                hs[slot] = (pu4_t){m, o | (w << 16), cellpos, 0u};
                cellpos += w + gap; // unused cells: windows of equal width do not start in the same LDS bank
                slot++;
            }
            nt[m] = ne;
        }
    }
    if (tid == 0) {
        hdr[0] = totn;
        hdr[1] = totw;
        hdr[2] = total_cnt;
        hdr[3] = hot_cnt_total;
        hdr[HDR_REGION] = region_recs;
        hdr[HDR_CELLS] = cells;
        hdr[HDR_OVF] = 0;
        hdr[HDR_HITS] = 0;  // k_scatter3's account of what the hot windows take (stale_report below)
        hdr[HDR_TILES] = 0;
        hdr[HDR_TICKET] = 0;
        hdr[HDR_BASE] = 0;  // no launch has run on these tables yet
    }
}

// ---------------------------------------------------------------------------
// P1 v2: compress + hot windows + 2-byte-record scatter
// ---------------------------------------------------------------------------
__device__ __forceinline__ void v2_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                              uint32_t m, uint32_t bin, uint64_t c)
{
    lh::cell_add(counts, (size_t)m * LH_ROW_STRIDE + bin, c);
    uint32_t *r = ranges + 2 * (size_t)m;
    if (bin < r[0]) atomicMin(&r[0], bin);
    if (bin > r[1]) atomicMax(&r[1], bin);
}

// Global stores the compiler's s_waitcnt bookkeeping does not see (k_scatter2 explains why).  The s_nop covers the
// "VMEM store of more than 8 bytes followed by a write of its data registers" hazard, which the compiler's hazard
// recogniser cannot handle for an instruction inside an asm block.
#ifndef LH_STORE_NT
#define LH_STORE_NT 0
#endif
__device__ __forceinline__ void hidden_store_u4(void *p, pu4_t v)
{
    if (LH_STORE_NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void hidden_store_u32(void *p, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
// counts[m][bin] += c and the row's range, for use INSIDE the tile loops: three atomics that return nothing, issued from
// inline asm and without v2_global_add's look at the range first (a visible global operation in a rarely taken branch
// makes the compiler wait for vmcnt(0) at the join, and the pre-check's load stalls the wave for a memory round trip)
__device__ __forceinline__ void hidden_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges, uint32_t m,
                                                  uint32_t bin, uint32_t c)
{
    lh::cell_add_hidden(counts, (size_t)m * LH_ROW_STRIDE + bin, c);
    uint32_t *r = ranges + 2 * (size_t)m;
    asm volatile("global_atomic_umin %0, %1, off\n\tglobal_atomic_umax %0, %1, off offset:4" : : "v"(r), "v"(bin) : "memory");
}

template <int BLOCK, int NPT, typename IDT>
__global__ __launch_bounds__(BLOCK, 4) void k_scatter2(const IDT *__restrict__ ids,
                                                       const double *__restrict__ v, size_t n, uint32_t nmetrics,
                                                       uint32_t log_np, uint32_t log_w,
                                                       const double *__restrict__ Tx,
                                                       const NameEntry *__restrict__ g_nt,
                                                       const pu4_t *__restrict__ g_hs,
                                                       const uint32_t *__restrict__ g_hdr, uint32_t cells,
                                                       rec16_t *__restrict__ records, uint32_t *__restrict__ cdesc,
                                                       uint32_t chunks_per_wg, uint64_t *__restrict__ counts,
                                                       uint32_t *__restrict__ ranges, uint32_t *__restrict__ err)
{
    // ONE LDS allocation: [Scatter2Lds][name table][hot windows].  Phases 1 and 3 address it through word / halfword
    // offsets from its base, so that a sample's single LDS operation has one form whatever the sample turns into.
    typedef Scatter2LdsT<BLOCK, NPT> LdsT;
    static_assert(sizeof(LdsT) % 16 == 0, "the name table follows the struct in LDS");
    static_assert(BLOCK >= 4 * NPT, "drain: one thread per 16-byte piece of a staged line");
    constexpr int V2_TILE = LdsT::TILE;
    extern __shared__ __attribute__((aligned(16))) unsigned char v2_smem[];
    LdsT &L = *reinterpret_cast<LdsT *>(v2_smem);
    NameEntry *nt = reinterpret_cast<NameEntry *>(v2_smem + sizeof(LdsT));         // [nmetrics]
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(v2_smem);
    rec16_t *lds16 = reinterpret_cast<rec16_t *>(v2_smem);
    const uint32_t win_w = (uint32_t)(sizeof(LdsT) / 4) + 2 * nmetrics;            // word offset of the hot windows
    uint32_t *win = lds32 + win_w;                                                 // [cells]
    constexpr uint32_t CNT_W = offsetof(LdsT, cnt) / 4, DUMMY_W = offsetof(LdsT, dummy) / 4;
    constexpr uint32_t SORTED_H = offsetof(LdsT, sorted) / 2, STAGE_H = offsetof(LdsT, stage) / 2;
    constexpr int V2_BLOCK = BLOCK;   // shadows the survey kernels' block size inside this kernel
    constexpr int NPMAX = NPT;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t np = 1u << log_np, pmask = np - 1, W = 1u << log_w;
    const uint32_t pool_base = blockIdx.x * chunks_per_wg;

    for (uint32_t i = tid; i < nmetrics; i += V2_BLOCK) {
        NameEntry ne = g_nt[i];
        ne.hot += win_w; // hot base as a word offset from the LDS base (base + win_w < 65 536: fits the low half)
        nt[i] = ne;
    }
    for (uint32_t i = tid; i < cells; i += V2_BLOCK) win[i] = 0;
    if (tid < NPMAX) { L.cnt[tid] = 0; L.sf[tid] = 0; L.cfill[tid] = CHUNK; L.cbase[tid] = INVALID; }
    ov_init(L.ov_key, L.ov_cnt, tid, V2_BLOCK);
    if (tid == 0) { L.pool_next = 0; L.missn[0] = 0; L.missn[1] = 0; }
    __syncthreads();

    const size_t ntiles = (n + V2_TILE - 1) / V2_TILE;
    const size_t npairs = (n + 1) / 2;
    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
    constexpr int NPAIR = V2_SPT / 2;
    // Two register sets, used by alternate tiles.  A set's loads are issued right after barrier A of the tile that
    // last used it, a whole tile period before they are needed, so HBM always has ~96 KiB per CU in flight while
    // the workgroup computes.
    //
    // For that to work the loop's STORES (copy-out, chunk descriptors) are issued from inline asm.  With ordinary
    // stores the compiler's wait for `nxt` became s_waitcnt vmcnt(0) at the top of the loop: the stores sit
    // between the loads and their use, their number is not a compile-time constant, so it drained everything --
    // the loads just issued included -- and the kernel ran load, then compute, then load (measured: 3.3 ms per 1e9
    // pairs whatever the workgroup shape).  A store has no result register, so hiding it from the compiler's
    // counter bookkeeping is safe: vmcnt retires in order and operations the compiler does not know about only
    // make its waits stricter.  (Hiding the LOADS instead is not safe: the register allocator may copy a
    // destination register before the data has landed.)
    typename IS::raw_t ida[NPAIR], idb[NPAIR];
    pd2_t vaa[NPAIR], vab[NPAIR];
    auto load_tile = [&](size_t tile, typename IS::raw_t (&di)[NPAIR], pd2_t (&dv)[NPAIR]) {
        const size_t pbase = tile * (V2_TILE / 2);
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            // pairs beyond the stream (last tile, and the two tiles past the end that the pipeline touches) re-read
            // the last pair; `lim` masks them.  The last pair of an odd-length stream reads one element past n inside
            // the same 16-byte granule; it is masked too.
            size_t i = pbase + (size_t)j * V2_BLOCK + tid;
            i = i < npairs ? i : npairs - 1;
            di[j] = ip.ld_nt(i);
            dv[j] = __builtin_nontemporal_load(vp + i);
        }
    };
    load_tile(blockIdx.x, ida, vaa);
    // The first tile must have landed before the second one is requested: otherwise the compiler interleaves the two
    // sets' loads, cannot tell them apart by age, and the merged wait at the loop header becomes vmcnt(0) for every
    // iteration.  (The empty asm "uses" the registers, which makes the compiler wait for them here.)
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

    // One tile.  `di` / `dv` hold the tile's samples; once they have been classified (after barrier A) the same
    // registers receive the loads of the tile two steps ahead.  The loop below alternates between the two register
    // sets, so no value ever has to be copied from one set to the other -- a copy would have to wait for loads that
    // were issued moments ago.
    auto process_tile = [&](size_t tile, typename IS::raw_t (&idv)[NPAIR], pd2_t (&val)[NPAIR], const uint32_t par) {
        // samples of this tile that exist (every tile but the last is full)
        const uint32_t lim = (tile + 1) * (size_t)V2_TILE <= n ? (uint32_t)V2_TILE : (uint32_t)(n - tile * (size_t)V2_TILE);
        uint32_t pr[V2_SPT];  // partition | rank << 8, INVALID when the sample left no record
        uint32_t rec[V2_SPT];
        uint32_t rare = 0;    // some sample of this thread carried an id >= nmetrics
        // ---- phase 1: classify.  Straight-line code, four samples at a time: their name-table reads, then their
        // LDS atomics, are in flight together (one LDS round trip per batch instead of one per sample).
#pragma unroll
        for (int h = 0; h < V2_SPT; h += 4) {
            uint32_t id[4], bin[4], where[4], rank[4];
            NameEntry ne[4];
            uint32_t unc = 0, miss = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = h + k;
                const uint32_t raw = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
                // padding beyond the stream is not an error; an id >= nmetrics is (reported, sample skipped)
                const bool live = 2u * ((uint32_t)(j >> 1) * V2_BLOCK + tid) + (uint32_t)(j & 1) < lim;
                const bool ok = live && raw < nmetrics;
                rare |= (live && !ok) ? 1u : 0u;
                id[k] = ok ? raw : INVALID;
                ne[k] = nt[ok ? raw : 0u];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = h + k;
                const double x = (j & 1) ? val[j >> 1].y : val[j >> 1].x;
                bool u;
                bin[k] = lh_bin_fast(x, u);
                if (u) unc |= 1u << k;
            }
            if (unc) { // inside the guard band of a bucket threshold (1 sample in ~4 000): exact table compare
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = h + k;
                    if (unc & (1u << k)) bin[k] = lh_bin_of((j & 1) ? val[j >> 1].y : val[j >> 1].x, Tx);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int j = h + k;
                const uint32_t hrel = bin[k] - (ne[k].org >> 16), crel = bin[k] - (ne[k].org & 0xffffu);
                const bool valid = id[k] != INVALID;
                const bool hot = valid && hrel < (ne[k].hot >> 16);
                const bool cold = valid && !hot && crel < W;
                const uint32_t p = id[k] & pmask;
                where[k] = hot ? (ne[k].hot & 0xffffu) + hrel : cold ? CNT_W + p : DUMMY_W + lane;
                rec[j] = ((id[k] >> log_np) << log_w) | crel;
                pr[j] = cold ? p : INVALID;
                if (valid && !hot && !cold) miss |= 1u << k;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) rank[k] = atomicAdd(lds32 + where[k], 1u);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (pr[h + k] != INVALID) pr[h + k] |= rank[k] << 8;
            if (miss) { // outside the name's cold window: queued, counted exactly after the copy-out (phase 4)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (miss & (1u << k)) {
                        const uint32_t key = (id[k] << 16) | bin[k];
                        const uint32_t at = atomicAdd(&L.missn[par], 1u);
                        if (at < V2_MISSQ) L.missq[par][at] = key;
                        else if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v2_global_add(counts, ranges, id[k], bin[k], 1);
                    }
            }
        }
        if (rare) atomicOr(err, 1u); // an id >= nmetrics: reported by lh_sync / lh_extract
        __syncthreads();                                   // barrier A: counts complete
        // this register set is free again: it receives the tile two steps ahead (a whole tile period in flight)
        load_tile(tile + 2 * (size_t)gridDim.x, idv, val);

        // ---- phase 2: per-partition bookkeeping, four threads per partition (p = tid / 4, q = tid % 4).
        // Every wave first scans the line counts of ALL partitions redundantly (K partitions per lane, one wave scan:
        // no barrier, no waiting for other waves), publishes the bases of the 16 partitions its own threads handle and
        // reads them back (same wave: LDS operations of one wave are ordered).  sf / cnt are only READ in this
        // phase; their new values are installed after barrier B.
        if (tid == V2_BLOCK - 1) L.missn[par ^ 1u] = 0; // the other parity's queue was drained in the previous tile
        {
            constexpr int K = NPT / 64; // partitions per lane of the scan
            uint32_t nf[K];
            uint32_t lane_lines = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                nf[k] = (L.sf[lane * K + k] + L.cnt[lane * K + k]) / LINE2;
                lane_lines += nf[k];
            }
            uint32_t inc = lane_lines;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d, 64);
                if ((int)lane >= d) inc += y;
            }
            uint32_t run = inc - lane_lines;
            // this wave's threads handle partitions [16 * wave, 16 * wave + 16): lanes 16 * wave / K .. own them
            const bool mine = lane >= 16u * wave / K && lane < (16u * wave + 16u) / K;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (mine) L.lbase[lane * K + k] = run;
                run += nf[k];
            }
            if (tid == V2_BLOCK - 1) L.nlines = inc; // lane 63 of the last wave: the total
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint32_t t2 = tid;
            asm volatile("" : "+v"(t2)); // keeps the address arithmetic below inside the loop (hoisted, it spills:
                                         // and a scratch reload here waits for the prefetch that was just issued)
            const uint32_t p = t2 >> 2, q = t2 & 3u;
            const uint32_t c = L.cnt[p], sf0 = L.sf[p], lb = L.lbase[p];
            const uint32_t total = sf0 + c, nfull = total / LINE2, out = nfull * LINE2;
            if (q == 0) {
                L.tA[p] = (pu2_t){sf0 | (out << 8), lb * LINE2};
                L.newsf[p] = total - out;
                if (nfull) {
                    const uint32_t tag = p << CD_SHIFT;
                    const uint32_t cf = L.cfill[p], cb = L.cbase[p];
                    const uint32_t room = CHUNK - cf; // multiple of LINE2 (0 when there is no open chunk)
                    uint32_t first = 0;
                    if (out > room) {
                        const uint32_t over = out - room;
                        const uint32_t k = (over + CHUNK - 1) / CHUNK;
                        first = pool_base + atomicAdd(&L.pool_next, k);   // k consecutive chunks
                        if (cb != INVALID) hidden_store_u32(cdesc + cb, tag | CHUNK);   // the old chunk is now full
#pragma nounroll
                        for (uint32_t i = 0; i + 1 < k; i++) hidden_store_u32(cdesc + first + i, tag | CHUNK);
                        L.cbase[p] = first + k - 1;
                        L.cfill[p] = over - (k - 1) * CHUNK;
                    } else {
                        L.cfill[p] = cf + out;
                    }
                    L.dA[p] = cb * CHUNK + cf;
                    L.dB[p] = first * CHUNK - room;
                    L.room[p] = room;
                }
            }
            if (nfull) {
#pragma nounroll
                for (uint32_t i = q; i < nfull; i += 4) L.owner[lb + i] = (uint16_t)p;
                // the staged leftovers open the partition's first line: one aligned 16-byte LDS copy per thread
                if (q * 8 < sf0)
                    *reinterpret_cast<pu4_t *>(&L.sorted[lb * LINE2 + q * 8]) =
                        *reinterpret_cast<const pu4_t *>(&L.stage[p * LINE2 + q * 8]);
            }
        }
        __syncthreads();                                   // barrier B: plan of the tile is visible

        // ---- phase 3: place the records (again one form for every sample: eight table reads, eight stores)
        if (tid < NPMAX) { L.sf[tid] = L.newsf[tid]; L.cnt[tid] = 0; }
        {
            pu2_t a[V2_SPT];
#pragma unroll
            for (int j = 0; j < V2_SPT; j++) a[j] = L.tA[pr[j] == INVALID ? 0u : (pr[j] & 0xffu)];
#pragma unroll
            for (int j = 0; j < V2_SPT; j++) {
                const uint32_t p = pr[j] & 0xffu;
                const uint32_t u = (a[j].x & 0xffu) + (pr[j] >> 8), out = a[j].x >> 8;
                const uint32_t at = pr[j] == INVALID ? 2 * DUMMY_W + lane
                                                     : (u < out ? SORTED_H + a[j].y + u : STAGE_H + p * LINE2 + (u - out));
                lds16[at] = (rec16_t)rec[j];
            }
        }
        __syncthreads();                                   // barrier C: lines complete

        // ---- phase 4: copy whole lines out, 16 bytes per lane
        const uint32_t npieces = L.nlines * 4;
        for (uint32_t i = tid; i < npieces; i += V2_BLOCK) {
            const uint32_t line = i >> 2, q = i & 3u;
            const uint32_t p = L.owner[line];
            const uint32_t u = (line - L.lbase[p]) * LINE2 + q * 8;
            const pu4_t r4 = *reinterpret_cast<const pu4_t *>(&L.sorted[line * LINE2 + q * 8]);
            const uint32_t dst = (u < L.room[p] ? L.dA[p] : L.dB[p]) + u;
            hidden_store_u4(records + dst, r4);
        }
        // the tile's out-of-window samples, one per thread: aggregated in the small LDS table, else a global atomic
        {
            const uint32_t nq = min(L.missn[par], V2_MISSQ);
            for (uint32_t i = tid; i < nq; i += V2_BLOCK) {
                const uint32_t key = L.missq[par][i];
                if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v2_global_add(counts, ranges, key >> 16, key & 0xffffu, 1);
            }
        }
        // (the next tile's barrier A separates this phase from the next bookkeeping)
    };
    for (size_t tile = blockIdx.x; tile < ntiles; tile += 2 * (size_t)gridDim.x) {
        process_tile(tile, ida, vaa, 0u);
        if (tile + gridDim.x < ntiles) process_tile(tile + gridDim.x, idb, vab, 1u); // workgroup-uniform
    }

    // ---- drain: staged remainders and the open chunks' descriptors
    __syncthreads();
    if (tid < NPMAX) {
        const uint32_t p = tid, sf = L.sf[p];
        L.dA[p] = INVALID;
        if (sf) {
            uint32_t cf = L.cfill[p], cb = L.cbase[p];
            if (cf == CHUNK) { // no open chunk, or it is exactly full
                if (cb != INVALID) cdesc[cb] = (p << CD_SHIFT) | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[p] = cb;
            }
            L.dA[p] = cb * CHUNK + cf;
            L.cfill[p] = cf + sf;
        }
    }
    __syncthreads();
    {
        const uint32_t p = tid >> 2, q = tid & 3u; // one thread per (partition, 16-byte piece of its staged line)
        const uint32_t d = p < (uint32_t)NPMAX ? L.dA[p] : INVALID;
        if (d != INVALID && q * 8 < L.sf[p])
            *reinterpret_cast<pu4_t *>(records + d + q * 8) = *reinterpret_cast<const pu4_t *>(&L.stage[p * LINE2 + q * 8]);
    }
    if (tid < np && L.cbase[tid] != INVALID) cdesc[L.cbase[tid]] = (tid << CD_SHIFT) | L.cfill[tid];

    // ---- flush the hot windows (one uint64 atomic per occupied bin) and the out-of-window table
    const uint32_t nhot = g_hdr[0];
    for (uint32_t s = wave; s < nhot; s += V2_BLOCK / 64) {
        const pu4_t h = g_hs[s];
        const uint32_t name = h.x, org = h.y & 0xffffu, width = h.y >> 16, base = h.z;
        uint32_t mn = INVALID, mx = 0;
        for (uint32_t i = lane; i < width; i += 64) {
            const uint32_t c = win[base + i];
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
            uint32_t *r = ranges + 2 * (size_t)name;
            if (mn < r[0]) atomicMin(&r[0], mn);
            if (mx > r[1]) atomicMax(&r[1], mx);
        }
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += V2_BLOCK)
        if (L.ov_key[i] != OV_EMPTY) v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
}

// ---------------------------------------------------------------------------
// P1 v3: fixed per-partition regions -- a cold record is placed by the phase that classifies it
// ---------------------------------------------------------------------------
// k_scatter2 lays a tile's records out exactly (scan of the partitions' line counts, owner table, staged
// leftovers), which takes two more phases and two more barriers per tile than the classification itself:
// measured (profiles/r02 ablations) 0.55 ms bookkeeping + 0.58 ms placement + 0.18 ms copy-out on top of 2.6 ms
// per 1e9 pairs, and 34 of the kernel's 67 VALU instructions per sample.
//
// Here every partition owns a fixed REGION of the workgroup's LDS, sized once per launch by the survey
// (1.5 x the partition's expected records per tile + 40, in whole lines; k_survey_parts).  The partition counter
// starts a tile at the number of records left over from the previous one, so the value the classification's LDS
// atomic returns IS the record's slot in the region: the record is stored at once, in phase 1.  After the tile's
// single counting barrier, four threads per partition copy the region's whole 64-byte lines to the partition's open
// chunk, move the last partial line to the front of the region and reset the counter to its length.  Two barriers
// per tile instead of three, no scan, no second pass over the samples.
//
// A record that finds its region full (a tile with several times the partition's expected share: a burst of one
// cold name) is counted exactly through the out-of-window queue instead.  The survey's per-partition shares use
// the names' TOTAL counts, hot samples included, so that a mispredicted hot window cannot cause that.
//
// The kernel handles whole tiles only; the launcher gives the last n % TILE pairs to k_ingest_pairs.
template <int NPT> struct Scatter3LdsT {
    uint32_t cnt[NPT];            // records in the partition's region (leftover + this tile's, may exceed cap)
    uint32_t cfill[NPT], cbase[NPT]; // open chunk of the partition: records in it, its index (persistent)
    pu2_t pt[NPT];                // {region base (LDS halfword index), capacity in records}
    uint32_t ov_key[OV_SLOTS], ov_cnt[OV_SLOTS];
    uint32_t missq[2][V2_MISSQ];
    uint32_t missn[2];
    uint32_t dummy[64];
    uint32_t pool_next, ovn; // ovn: records that found their region full (reported to the engine)
    uint32_t spills, pad0[3]; // spills: hot cells that handed 2^15 counts on to the row (16-bit cells, see k_scatter3)
};

// The classification's instruction count (round 6: 50.8 -> 45 VALU, 24 -> 19 SALU per wave-sample against round 5): ids >=
// nmetrics are CLAMPED to the table's extra entry nt[nmetrics] instead of tested per sample (one max3 + compare per batch
// finds the waves that hold one); the per-sample conditions stay lane masks in SGPRs; the LDS atomic and the record store
// are exec-masked instead of pointed at dummy words; the partition table holds the region's LDS ADDRESS.  Measured: the
// kernel's time does not move with its instruction count either way (+19 VALU per sample: +-0) -- see "what bounds the
// kernel" at k_scatter3.
// Ablation builds of the level-1 kernels (tools/build_tuning.py -DLH_ABL=bits; timing only, the counts are WRONG by
// construction; the product is built with 0): 1 = the copy-out body removed (both barriers stay, the counters are
// reset), 2 = no barriers and no copy-out at all (records wrap inside the first 16 slots of their region), 4 = the
// copy-out without its global stores, 8 = no LDS atomics (the slot is the lane), 16 = the bucket index computed twice
// (+19 VALU per sample: the slope of time over VALU work), 32 = every workgroup re-reads its first tile (the input
// comes from the L2 instead of HBM).
#ifndef LH_ABL
#define LH_ABL 0
#endif
constexpr uint32_t ABL = LH_ABL;
#ifndef LH_SC3_BATCH
#define LH_SC3_BATCH 4
#endif
constexpr int SC3_BATCH = LH_SC3_BATCH;              // samples classified together (4 or 8)
#ifndef LH_SC3_TILES_PER_FLUSH
#define LH_SC3_TILES_PER_FLUSH 1
#endif
constexpr uint32_t SC3_TILES_PER_FLUSH = LH_SC3_TILES_PER_FLUSH; // tiles classified between two flushes (1 or 2)
// entries of k_scatter3's name table in LDS: one per name, the entry of ids >= nmetrics, padded to 16 bytes
constexpr uint32_t sc3_nt_entries(uint32_t nmetrics) { return (nmetrics + 2u) & ~1u; }

template <int BLOCK, int NPT, int BATCH, typename IDT>
__global__ __launch_bounds__(BLOCK, 4) void k_scatter3(const IDT *__restrict__ ids,
                                                       const double *__restrict__ v, size_t ntiles, uint32_t nmetrics,
                                                       uint32_t log_np, uint32_t log_w,
                                                       const double *__restrict__ Tx,
                                                       const NameEntry *__restrict__ g_nt,
                                                       const pu4_t *__restrict__ g_hs,
                                                       uint32_t *__restrict__ g_hdr,
                                                       const pu2_t *__restrict__ g_pt, uint32_t *__restrict__ g_hot,
                                                       rec16_t *__restrict__ records,
                                                       uint32_t *__restrict__ cdesc, uint32_t chunks_per_wg,
                                                       uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                       uint32_t *__restrict__ err,
                                                       unsigned long long *__restrict__ rstat,
                                                       uint32_t *__restrict__ g_resume)
{
    // ONE LDS allocation: [Scatter3Lds][name table][regions][hot windows]
    typedef Scatter3LdsT<NPT> LdsT;
    static_assert(sizeof(LdsT) % 16 == 0, "the name table follows the struct in LDS");
    // copy-out threads per partition: four (a 16-byte piece of a line each) -- or two, in the WIDE shape <1024, 512>
    constexpr uint32_t TPP = BLOCK / NPT;
    static_assert(BLOCK == (int)(TPP * NPT) && (TPP == 2 || TPP == 4), "copy-out: two or four threads per partition");
    constexpr int V3_TILE = BLOCK * V2_SPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char v2_smem[];
    LdsT &L = *reinterpret_cast<LdsT *>(v2_smem);
    NameEntry *nt = reinterpret_cast<NameEntry *>(v2_smem + sizeof(LdsT));         // [nmetrics]
    uint32_t *lds32 = reinterpret_cast<uint32_t *>(v2_smem);
    rec16_t *lds16 = reinterpret_cast<rec16_t *>(v2_smem);
    // the plan's split of the LDS behind the name table (k_survey_plan: RegionFit): regions, then the hot windows
    const uint32_t region_recs = g_hdr[HDR_REGION], cells = g_hdr[HDR_CELLS];
    const uint32_t reg_h = (uint32_t)(sizeof(LdsT) / 2) + 4 * sc3_nt_entries(nmetrics); // halfword offset of the regions (16-byte aligned)
    // HOT WINDOWS OF 16-BIT CELLS (round 6), two to an LDS word: twice the cells in the same bytes -- under Zipf(1) over
    // 1 024 names 43 % -> 32 % of the pairs become records, and records are what the kernel pays for (see "what bounds the
    // kernel").  A sample adds 1 << 16 * (cell & 1) to the word; the atomic returns the word as it was, and the lane that
    // sees its own field at 2^15 - 1 (its add made it 2^15) takes 2^15 off the field and adds them to the row in HBM:
    // between that add and the subtraction at most one tile's other samples (8 191) can land on the field, so no field
    // ever carries into its neighbour, and every count is in exactly one place.  Exact for any stream; a spill is one in
    // 32 768 hits of ONE cell (a constant stream: 120 spills per workgroup and launch).
    const uint32_t win_h = reg_h + region_recs;                                    // halfword offset of the hot windows (even)
    uint32_t *win = lds32 + win_h / 2;                                             // [cells / 2] words
    constexpr uint32_t CNT_W = offsetof(LdsT, cnt) / 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t np = 1u << log_np, pmask = np - 1, W = 1u << log_w;
    const uint32_t pool_base = blockIdx.x * chunks_per_wg;
    // the LDS address of the block (0 here; not a constant the compiler can fold)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)v2_smem;

    for (uint32_t i = tid; i < nmetrics; i += BLOCK) nt[i] = g_nt[i]; // (hot base: a CELL index inside the window area)
    // the entry of an id >= nmetrics (such an id is clamped to it): no hot window, cold origin 65 535 -- with the
    // bin such a sample is given (0) it is neither hot nor cold and takes the rare path, which drops and reports it
    if (tid == 0) nt[nmetrics] = (NameEntry){0xffffu, 0u};
    for (uint32_t i = tid; i < cells / 2; i += BLOCK) win[i] = 0;
    if (tid < NPT) {
        pu2_t e = tid < np ? g_pt[tid] : (pu2_t){0u, 0u};
        e.x = 2u * (e.x + reg_h) + lds_base; // the region's LDS address: a record's address is one shift-add
        L.pt[tid] = e;
        L.cnt[tid] = 0;
        L.cfill[tid] = CHUNK;
        L.cbase[tid] = INVALID;
    }
    ov_init(L.ov_key, L.ov_cnt, tid, BLOCK);
    if (tid == 0) { L.pool_next = 0; L.ovn = 0; L.spills = 0; L.missn[0] = 0; L.missn[1] = 0; }
    __syncthreads();
    pu2_t my_pt = L.pt[tid / TPP]; // the flush phase's partition (constant over the launch)
    my_pt.x = (my_pt.x - lds_base) >> 1;  // (halfword index)

    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
    constexpr int NPAIR = V2_SPT / 2;
    // two register sets used by alternate tiles, loads issued a whole tile period ahead, stores hidden from the
    // compiler's s_waitcnt bookkeeping: see k_scatter2
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
    // atomics, then their record stores are in flight together.
    auto classify = [&](typename IS::raw_t (&idv)[NPAIR], pd2_t (&val)[NPAIR], const uint32_t par) {
#pragma unroll
        for (int h = 0; h < V2_SPT; h += BATCH) {
            uint32_t raw[BATCH], bin[BATCH], rank[BATCH], crel[BATCH];
            NameEntry ne[BATCH];
            pu2_t pe[BATCH];
            bool unc[BATCH], hot[BATCH], cold[BATCH], bad[BATCH];
            uint32_t waddr[BATCH], sh[BATCH];
            uint32_t idmax = 0;
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const int j = h + k;
                raw[k] = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
                idmax = max(idmax, raw[k]);
                ne[k] = nt[min(raw[k], nmetrics)]; // an id >= nmetrics reads the extra entry
                pe[k] = L.pt[raw[k] & pmask];
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
            if (anyunc) { // inside the guard band of a bucket threshold (1 sample in ~4 000): exact table compare
#pragma unroll
                for (int k = 0; k < BATCH; k++) {
                    const int j = h + k;
                    if (unc[k]) bin[k] = lh_bin_of((j & 1) ? val[j >> 1].y : val[j >> 1].x, Tx);
                }
            }
            if (idmax >= nmetrics) { // this lane holds an id >= nmetrics: bin 0 is outside the extra entry's windows
#pragma unroll
                for (int k = 0; k < BATCH; k++)
                    if (raw[k] >= nmetrics) bin[k] = 0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const uint32_t hrel = bin[k] - (ne[k].org >> 16);
                crel[k] = bin[k] - (ne[k].org & 0xffffu);
                hot[k] = hrel < (ne[k].hot >> 16);
                cold[k] = !hot[k] && crel[k] < W;
                // the hot cell (a halfword of the window area: word cell / 2, field cell & 1) or the partition's counter
                // (a word: cnt[] opens the LDS block)
                const uint32_t cell = (ne[k].hot & 0xffffu) + hrel + win_h;
                static_assert(CNT_W == 0, "the partition counters open the LDS block");
                waddr[k] = hot[k] ? cell >> 1 : raw[k] & pmask;
                sh[k] = hot[k] ? (cell & 1u) << 4 : 0u;
                asm volatile("" : "=v"(rank[k])); // (no value for the lanes that skip the atomic: `fits` below needs none)
                if (ABL & 8u) { rank[k] = lane & 15u; asm volatile("" : "+v"(waddr[k]), "+v"(sh[k])); }
                else if (hot[k] || cold[k]) rank[k] = atomicAdd(lds32 + waddr[k], 1u << sh[k]);
                if (ABL & 2u) rank[k] &= 15u;
            }
            bool anybad = false, anyspill = false;
            bool spill[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; k++) {
                const bool fits = cold[k] && rank[k] < pe[k].y;
                bad[k] = !hot[k] && !fits; // outside the name's cold window, or the region is full, or no such name
                anybad |= bad[k];
                spill[k] = hot[k] && ((rank[k] >> sh[k]) & 0xffffu) == 0x7fffu; // this add made the cell 2^15
                anyspill |= spill[k];
                if (fits) // pe.x is the region's LDS ADDRESS: one shift-add per record
                    *(__attribute__((address_space(3))) rec16_t *)(uintptr_t)(pe[k].x + 2u * rank[k]) =
                        (rec16_t)(((raw[k] >> log_np) << log_w) | crel[k]);
            }
            if (anyspill && !(ABL & 8u)) { // 2^15 counts of the cell move to the row in HBM
#pragma unroll
                for (int k = 0; k < BATCH; k++)
                    if (spill[k]) {
                        atomicSub(lds32 + waddr[k], 0x8000u << sh[k]);
                        hidden_global_add(counts, ranges, raw[k], bin[k], 0x8000u);
                        atomicAdd(&L.spills, 1u);
                    }
            }
            if (anybad) { // queued, counted exactly by the flush phase
#pragma unroll
                for (int k = 0; k < BATCH; k++)
                    if (bad[k]) {
                        if (raw[k] >= nmetrics) { atomicOr(err, 1u); continue; } // reported by lh_sync / lh_extract, the sample skipped
                        if (cold[k]) atomicAdd(&L.ovn, 1u); // the region is full: the engine watches this count
                        const uint32_t key = (raw[k] << 16) | bin[k];
                        const uint32_t at = atomicAdd(&L.missn[par], 1u);
                        if (at < V2_MISSQ) L.missq[par][at] = key;
                        else if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v2_global_add(counts, ranges, raw[k], bin[k], 1);
                    }
            }
        }
    };
    uint32_t seq_lines = tid / TPP; // (ABL 64: this thread group's position in the workgroup's sequential stream)
    auto flush = [&](const uint32_t par) {
        if (ABL & 2u) return;
        __syncthreads();                                   // barrier A: the records of the tile(s) are in the regions
        if (ABL & 1u) {
            if (tid < NPT) L.cnt[tid] = 0;
            if (tid == BLOCK - 1) { L.missn[0] = 0; L.missn[1] = 0; }
            __syncthreads();
            return;
        }
        // ---- phase 2: flush.  TPP threads per partition (p = tid / TPP; thread q copies the 16-byte pieces q, q + TPP,
        // .. of every line).  With TPP = 1 only the first NPT threads -- one wave per SIMD -- run this phase and the
        // other waves go straight to the barrier.
        if (tid == BLOCK - 1) L.missn[par ^ 1u] = 0; // the other parity's queue was drained in the previous tile
        if (tid < NPT * TPP) {
            uint32_t t2 = tid;
            asm volatile("" : "+v"(t2)); // keeps this phase's address arithmetic inside the loop (see k_scatter2)
            const uint32_t p = t2 / TPP, q = t2 % TPP;
            const pu2_t e = my_pt;
            // whole pieces only: `full` lines leave (a multiple of SC3_PIECE), fewer than PIECE2 records stay behind
            const uint32_t c = min(L.cnt[p], e.y), full = c / PIECE2 * SC3_PIECE, left = c - full * LINE2;
            if (full) {
                const uint32_t cf = L.cfill[p], cb = L.cbase[p];
                const uint32_t room = (CHUNK - cf) / LINE2; // lines left in the open chunk (0: none open; whole pieces)
                uint32_t first = 0;
                if (full > room && q == 0) {
                    const uint32_t tag = p << CD_SHIFT;
                    const uint32_t over = full - room;
                    const uint32_t k = (over + CHUNK / LINE2 - 1) / (CHUNK / LINE2);
                    first = pool_base + atomicAdd(&L.pool_next, k);   // k consecutive chunks
                    if (cb != INVALID) hidden_store_u32(cdesc + cb, tag | CHUNK);   // the old chunk is now full
#pragma nounroll
                    for (uint32_t i = 0; i + 1 < k; i++) hidden_store_u32(cdesc + first + i, tag | CHUNK);
                    L.cbase[p] = first + k - 1;
                    L.cfill[p] = (over - (k - 1) * (CHUNK / LINE2)) * LINE2;
                } else if (q == 0) {
                    L.cfill[p] = cf + full * LINE2;
                }
                if (TPP == 4) first = __builtin_amdgcn_mov_dpp(first, 0x00, 0xf, 0xf, false); // quad_perm [0,0,0,0]: q == 0's value
                if (TPP == 2) first = __builtin_amdgcn_mov_dpp(first, 0xa0, 0xf, 0xf, false); // quad_perm [0,0,2,2]
                const uint32_t dA = cb * CHUNK + cf, dB = first * CHUNK - room * LINE2;
                const rec16_t *src = lds16 + e.x;
#pragma nounroll
                for (uint32_t l = 0; l < full; l++) {
                    const uint32_t dst = (l < room ? dA : dB) + l * LINE2;
                    pu4_t r4[4 / TPP];
#pragma unroll
                    for (uint32_t i = 0; i < 4 / TPP; i++)
                        r4[i] = *reinterpret_cast<const pu4_t *>(src + l * LINE2 + (q + i * TPP) * 8);
#pragma unroll
                    for (uint32_t i = 0; i < 4 / TPP; i++) {
                        if (ABL & 4u) asm volatile("" : : "v"(r4[i]), "v"(dst));
                        else if (ABL & 64u) // the same bytes to ONE sequential stream per workgroup (its pool, front to back)
                            hidden_store_u4(records + (size_t)pool_base * CHUNK + ((seq_lines += BLOCK / 4) % (chunks_per_wg * (CHUNK / LINE2))) * LINE2 + (q + i * TPP) * 8, r4[i]);
                        else hidden_store_u4(records + dst + (q + i * TPP) * 8, r4[i]);
                    }
                }
                // the leftover (less than a piece) moves to the front of the region (its slots are this thread's own:
                // source and destination are at least one piece apart)
#pragma unroll
                for (uint32_t i = 0; i < SC3_PIECE * 4 / TPP; i++) {
                    const uint32_t piece = q + i * TPP; // 16-byte pieces 0 .. 4 * SC3_PIECE - 1 of the leftover
                    if (piece * 8 < left)
                        *reinterpret_cast<pu4_t *>(lds16 + e.x + piece * 8) =
                            *reinterpret_cast<const pu4_t *>(src + full * LINE2 + piece * 8);
                }
            }
            if (q == 0) L.cnt[p] = left;
        }
        // the tile's out-of-window samples, one per thread: aggregated in the small LDS table, else a global atomic
        {
            const uint32_t nq = min(L.missn[par], V2_MISSQ);
            for (uint32_t i = tid; i < nq; i += BLOCK) {
                const uint32_t key = L.missq[par][i];
                if (!ov_add(L.ov_key, L.ov_cnt, key, 1u)) v2_global_add(counts, ranges, key >> 16, key & 0xffffu, 1);
            }
            if (L.missn[par] > V2_MISSQ) { // (uniform) the tile overflowed past its queue, straight into the table: emptied
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
    // Each register set receives the tile two steps ahead as soon as its samples are classified.
    //
    // The loop takes the tiles in PAIRS and both halves of its body are unconditional (a last single tile is peeled off
    // below).  Round 6 found why: until then the second half sat under `if (tile + gridDim.x < ntiles)`, and the
    // compiler's s_waitcnt pass must assume the path on which that half did not run -- so it counted 8 loads in flight
    // where 16 are, and opened every classification with s_waitcnt vmcnt(7): a wait for the FIRST LOAD OF THE OTHER
    // REGISTER SET, issued one flush earlier, i.e. a whole memory round trip exposed per tile (with the input re-read
    // from the L2 the kernel was 23 % faster; profiles/r06_level1_ablations.txt).  With an unconditional body the count
    // is exact (vmcnt(15): the loads of the tile at hand only).
    static_assert(SC3_TILES_PER_FLUSH == 1, "two tiles per flush were measured (round 2: no gain) and removed");
    // A stream CLUSTERED BY NAME (sorted by name, whole batches of one producer) fills one partition's region with every
    // tile and sends the rest of the tile through the exact overflow path.  A workgroup whose last two tiles overflowed by
    // more than an eighth stops here; the tiles it leaves are counted by k_scatter_clustered (its turn in g_resume).  Only the
    // 1 024-thread shapes: that kernel's tile is theirs.
    uint32_t par = 0; // out-of-window queue of this flush group (the flush resets the other one)
    uint32_t ovn_seen = 0, my_tiles = 0;
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
        my_tiles += 2;
        if (BLOCK == 1024) {
            // (uniform: L.ovn is at rest between flush's last barrier and the next tile)
            const uint32_t ovn_now = (uint32_t)__builtin_amdgcn_readfirstlane(L.ovn);
            if (ovn_now - ovn_seen > (uint32_t)V3_TILE / 4u) { // more than an eighth of the two tiles just done
                tile += 2 * G;
                gave_up = true;
                break;
            }
            ovn_seen = ovn_now;
        }
    }
    if (!gave_up && tile < ntiles) { // (workgroup-uniform) the workgroup's last tile when it has an odd number of them
        classify(ida, vaa, par);
        flush(par);
        par ^= 1u;
        tile += G;
        my_tiles += 1;
    }
    if (tid == 0) {
        g_resume[blockIdx.x] = BLOCK == 1024 ? (uint32_t)min(tile, ntiles) : 0xffffffffu; // this workgroup's first undone tile
        // Up to 1 024 names the exact-layout scatter is the better STEADY state for such a stream (per 1e9 sorted pairs:
        // 4.9 ms / 8.6 ms at 21 decades, against 5.3 / 19.5 through the table), so what this workgroup leaves is reported
        // as overflow and the engine changes the path at the next flip, as it always did.  Above that it is not (8 192
        // names: 42 / 108 ms against 6.5 / 39) and only what really overflowed is reported.  profiles/r06_first_call.txt
        if (gave_up && nmetrics <= 1024u && tile < ntiles) L.ovn += (uint32_t)((ntiles - tile + G - 1) / G) * (uint32_t)V3_TILE;
    }

    // ---- drain: the regions' leftovers (< one line each) and the open chunks' descriptors
    {
        const uint32_t p = tid / TPP, q = tid % TPP;
        const uint32_t left = L.cnt[p]; // < PIECE2 after a flush
        uint32_t d = INVALID;
        if (left && q == 0) {
            uint32_t cf = L.cfill[p], cb = L.cbase[p]; // (cf is a multiple of PIECE2: the leftover fits the open chunk)
            if (cf == CHUNK) { // no open chunk, or it is exactly full
                if (cb != INVALID) cdesc[cb] = (p << CD_SHIFT) | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[p] = cb;
            }
            d = cb * CHUNK + cf;
            L.cfill[p] = cf + left;
        }
        d = TPP == 4 ? __builtin_amdgcn_mov_dpp(d, 0x00, 0xf, 0xf, false)   // quad_perm [0,0,0,0]: q == 0's value
                     : __builtin_amdgcn_mov_dpp(d, 0xa0, 0xf, 0xf, false);  // quad_perm [0,0,2,2]
#pragma unroll
        for (uint32_t j = 0; j < SC3_PIECE; j++)
#pragma unroll
            for (uint32_t i = 0; i < 4 / TPP; i++) {
                const uint32_t piece = q + i * TPP; // 16-byte pieces 0 .. 3 of the line
                if (left && j * LINE2 + piece * 8 < left)
                    *reinterpret_cast<pu4_t *>(records + d + j * LINE2 + piece * 8) =
                        *reinterpret_cast<const pu4_t *>(lds16 + my_pt.x + j * LINE2 + piece * 8);
            }
    }
    if (tid == 0) L.dummy[0] = 0; // (no sample targets the dummy words any more) the workgroup's hot-window hits
    __syncthreads();
    if (tid < np && L.cbase[tid] != INVALID) cdesc[L.cbase[tid]] = (tid << CD_SHIFT) | L.cfill[tid];
    // the engine watches this count (pinned host memory): a stream whose tiles overflow the regions is clustered by
    // name, and later calls take the exact-layout kernel instead
    if (tid == 0 && L.ovn) atomicAdd(&g_hdr[HDR_OVF], L.ovn); // (k_hot_reduce's judge passes it on)

    // ---- the hot windows leave as they are: one coalesced copy of the window area into the workgroup's slice of g_hot.
    // k_hot_reduce adds the workgroups' copies up and touches every row cell ONCE.  (Until round 6 every workgroup
    // flushed its windows with one uint64 atomic per occupied cell: 256 workgroups x 19 000 cells were ~100 us at the END
    // of every launch -- device-scope atomics of 256 workgroups on the same few thousand lines -- and the cost grew with
    // the very cells that save records.)
    {
        pu4_t *dst = reinterpret_cast<pu4_t *>(g_hot + (size_t)blockIdx.x * (cells / 2));
        const pu4_t *src = reinterpret_cast<const pu4_t *>(win);
        for (uint32_t i = tid; i < cells / 8; i += BLOCK) dst[i] = src[i]; // (cells is a multiple of 64, the area 16-byte aligned)
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += BLOCK)
        if (L.ov_key[i] != OV_EMPTY) v2_global_add(counts, ranges, L.ov_key[i] >> 16, L.ov_key[i] & 0xffffu, L.ov_cnt[i]);
    // the workgroup's part of the launch's account (stale_judge, by k_hot_reduce): its tiles, and the hits its cells
    // handed on to the rows already
    if (tid == 0) {
        const uint32_t mine = my_tiles; // (the tiles THIS kernel counted: a workgroup that gave up leaves the rest of its turn)
        if (L.spills) atomicAdd(&g_hdr[HDR_HITS], L.spills << 15);
        atomicAdd(&g_hdr[HDR_TILES], mine);
    }
}

// ---------------------------------------------------------------------------
// What the workgroups of k_scatter3 / k_scatter4 left undone when they found the stream clustered by name
// (g_resume[workgroup] = its first undone tile; >= ntiles: nothing, the workgroup returns at once -- every launch of an
// ordinary stream).  The same turn
// of tiles, counted in ONE open-addressed LDS table of (name << 16 | bin) -> count: a clustered tile holds a few names,
// i.e. a few hundred distinct cells, and the table (16 384 slots) is emptied into the rows whenever it is half full.
// Exact like every other path: a sample that finds its eight probe slots taken by other cells is one global atomic.
// ---------------------------------------------------------------------------
constexpr uint32_t CL_SLOTS = 16384, CL_PROBES = 8, CL_TILE = 8192; // (the tile of the 1 024-thread scatter kernels)
constexpr size_t CL_LDS_BYTES = (size_t)CL_SLOTS * 8 + 16; // + `used`, `gadds`
constexpr uint32_t CL_SMALL_SLOTS = 4096;                  // tiles of 1 024 pairs (small calls): two workgroups per CU
constexpr size_t CL_SMALL_LDS_BYTES = (size_t)CL_SMALL_SLOTS * 8 + 16;

template <typename IDT, int SPT = V2_SPT, uint32_t SLOTS = CL_SLOTS>
__global__ __launch_bounds__(1024) void k_scatter_clustered(const IDT *__restrict__ ids, const double *__restrict__ v,
                                                            size_t ntiles, uint32_t nmetrics,
                                                            const double *__restrict__ Tx,
                                                            const uint32_t *__restrict__ g_resume,
                                                            uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                            uint32_t *__restrict__ err, uint32_t *__restrict__ g_ovf)
{
    // (g_resume == null: the kernel on its own, every workgroup from its first tile -- launch_ingest_pairs_cells)
    size_t tile = g_resume ? g_resume[blockIdx.x] : blockIdx.x;
    if (tile >= ntiles) return;
    constexpr uint32_t BLOCK = 1024, TILE = BLOCK * SPT;
    static_assert(SPT == 1 || TILE == CL_TILE, "behind the scatter kernels the tile is theirs");
    extern __shared__ __attribute__((aligned(16))) unsigned char v3_smem[];
    uint32_t *key = reinterpret_cast<uint32_t *>(v3_smem), *cnt = key + SLOTS, *used = cnt + SLOTS;
    uint32_t *gadds = used + 1; // global adds this workgroup has made: emptied slots + samples that found no slot
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < SLOTS; i += BLOCK) { key[i] = OV_EMPTY; cnt[i] = 0; }
    if (tid == 0) { *used = 0; *gadds = 0; }
    __syncthreads();
    // Emptying the table: a thread's slots in three sweeps -- every count's atomic, then every range's look, then the
    // widenings -- so that the slots' memory round trips overlap instead of queueing behind one another (v2_global_add slot by
    // slot: a 1 024-pair call spent 9 of its 15 us here).
    auto drain = [&]() {
        constexpr uint32_t PER = SLOTS / BLOCK;
        uint32_t k[PER], mine = 0;
        pu2_t rg[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; j++) {
            const uint32_t i = tid + j * BLOCK;
            k[j] = key[i];
            if (k[j] != OV_EMPTY) {
                lh::cell_add(counts, (size_t)(k[j] >> 16) * LH_ROW_STRIDE + (k[j] & 0xffffu), cnt[i]);
                key[i] = OV_EMPTY;
                cnt[i] = 0;
                mine++;
            }
        }
        if (mine) { // (most threads of a small call hold nothing)
#pragma unroll
            for (uint32_t j = 0; j < PER; j++)
                if (k[j] != OV_EMPTY) rg[j] = *reinterpret_cast<const pu2_t *>(ranges + 2 * (size_t)(k[j] >> 16));
#pragma unroll
            for (uint32_t j = 0; j < PER; j++)
                if (k[j] != OV_EMPTY) {
                    uint32_t *r = ranges + 2 * (size_t)(k[j] >> 16);
                    const uint32_t bin = k[j] & 0xffffu;
                    if (bin < rg[j].x) atomicMin(&r[0], bin);
                    if (bin > rg[j].y) atomicMax(&r[1], bin);
                }
            atomicAdd(gadds, mine);
        }
        if (tid == 0) *used = 0;
    };
    const size_t first_tile = tile;
    for (; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * TILE + tid;
        uint32_t id[SPT];
        double x[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            id[j] = ids[base + (size_t)j * BLOCK];
            x[j] = __builtin_nontemporal_load(v + base + (size_t)j * BLOCK);
        }
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            bool unc;
            uint32_t bin = lh_bin_fast(x[j], unc);
            if (unc) bin = lh_bin_of(x[j], Tx); // (inside a threshold's guard band: the table compare)
            if (id[j] >= nmetrics) { atomicOr(err, 1u); continue; }
            const uint32_t k = (id[j] << 16) | bin;
            const uint32_t h0 = (k * 2654435761u) >> 16; // (the slot mask takes its low bits: 12 or 14 of the product's top 16)
            bool placed = false;
#pragma unroll 1
            for (uint32_t probe = 0; probe < CL_PROBES && !placed; probe++) {
                const uint32_t sl = (h0 + probe) & (SLOTS - 1u);
                const uint32_t prev = atomicCAS(&key[sl], OV_EMPTY, k);
                if (prev == OV_EMPTY) atomicAdd(used, 1u);
                if (prev == OV_EMPTY || prev == k) { atomicAdd(&cnt[sl], 1u); placed = true; }
            }
            if (!placed) { v2_global_add(counts, ranges, id[j], bin, 1); atomicAdd(gadds, 1u); }
        }
        __syncthreads();
        if (*used > SLOTS / 2u) { // (uniform: nothing adds between the barriers)
            drain();
        }
        __syncthreads();
    }
    drain();
    __syncthreads();
    // The table pays when a tile's samples meet in few cells -- long runs of one name (a sorted stream: one global add per
    // 20 samples).  Runs of 64 .. 256 pairs put 32 .. 128 names into every tile: nearly every sample is its own cell (0.7 ..
    // 0.9 adds per sample, 35 .. 43 ms per 1e9 pairs where the exact-layout path takes 10).  Such a launch reports what it
    // counted here as overflow, and the engine leaves the region scatter at the next flip as it did before this kernel
    // existed (profiles/r06_first_call.txt, section F).
    if (tid == 0 && g_ovf) {
        const size_t mine = ((tile - first_tile) / gridDim.x) * TILE;
        if ((size_t)*gadds * 2 > mine) atomicAdd(g_ovf, (uint32_t)mine);
    }
}

// The cell table ON ITS OWN: the path of calls too small for a partitioned path (and of calls whose scratch block cannot be
// had).  No scratch, no survey, exact.  Against one global atomic per sample (k_ingest_pairs: 8 - 11 G pairs/s on a Zipf /
// lognormal stream, but 0.6 G/s when the values are constant and 0.08 G/s -- 12 ns per sample -- when they all fall into ONE
// cell: same-address atomics serialise) the table adds what meets in a cell BEFORE it goes to memory.  Tiles of 8 192 pairs
// when there are enough of them to fill the device, of 1 024 otherwise; the pairs behind the last whole tile take
// k_ingest_pairs.  profiles/r06_small_calls.txt
template <typename IDT>
static hipError_t launch_cells_t(const IDT *d_ids, const double *d_v, size_t n, uint64_t *counts, uint32_t *ranges,
                                 uint32_t nmetrics, const double *d_Tx, uint32_t *d_err, int num_cus, hipStream_t s, size_t *done)
{
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attributes
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_clustered<IDT, V2_SPT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_clustered<IDT, 1, CL_SLOTS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_clustered<IDT, 1, CL_SMALL_SLOTS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_SMALL_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    *done = 0;
    // (measured per call size, Zipf names x lognormal values: the large table adds up hot cells across a workgroup's tiles and
    // wins from 2^17 pairs; below, its 128 KiB to initialise and scan are most of a call)
    if (n / CL_TILE >= 2 * (size_t)num_cus) {
        const size_t nt = n / CL_TILE;
        hipLaunchKernelGGL((k_scatter_clustered<IDT, V2_SPT>), dim3((unsigned)std::min<size_t>(nt, (size_t)num_cus)), dim3(1024),
                           CL_LDS_BYTES, s, d_ids, d_v, nt, nmetrics, d_Tx, nullptr, counts, ranges, d_err, nullptr);
        *done = nt * CL_TILE;
    } else if (n >= (size_t(1) << 17)) {
        const size_t nt = n / 1024;
        hipLaunchKernelGGL((k_scatter_clustered<IDT, 1, CL_SLOTS>), dim3((unsigned)std::min<size_t>(nt, (size_t)num_cus)), dim3(1024),
                           CL_LDS_BYTES, s, d_ids, d_v, nt, nmetrics, d_Tx, nullptr, counts, ranges, d_err, nullptr);
        *done = nt * 1024;
    } else if (n >= 1024) {
        const size_t nt = n / 1024;
        hipLaunchKernelGGL((k_scatter_clustered<IDT, 1, CL_SMALL_SLOTS>), dim3((unsigned)std::min<size_t>(nt, 2 * (size_t)num_cus)),
                           dim3(1024), CL_SMALL_LDS_BYTES, s, d_ids, d_v, nt, nmetrics, d_Tx, nullptr, counts, ranges, d_err, nullptr);
        *done = nt * 1024;
    }
    return hipGetLastError();
}

hipError_t launch_ingest_pairs_cells(Ids d_ids, const double *d_v, size_t n, uint64_t *counts, uint32_t *ranges,
                                     uint32_t nmetrics, const double *d_Tx, uint32_t *d_err, int num_cus, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t done = 0;
    hipError_t e = d_ids.width == 2 ? launch_cells_t(d_ids.u16(), d_v, n, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s, &done)
                                    : launch_cells_t(d_ids.u32(), d_v, n, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s, &done);
    if (e != hipSuccess) return e;
    if (done < n) e = launch_ingest_pairs(d_ids.plus(done), d_v + done, n - done, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s);
    return e;
}

// The hot windows of a launch's G workgroups, added up: one workgroup per hot name (g_hs), a thread per cell walks the
// G copies (consecutive threads read consecutive halfwords of one copy: coalesced), then ONE uint64 atomic per occupied
// cell and one range update per name.  With tile > 0 the kernel also closes the launch's account of what the hot windows
// took (stale_judge): the last workgroup to finish judges.
__global__ __launch_bounds__(1024) void k_hot_reduce(const uint32_t *__restrict__ g_hot, uint32_t G,
                                                     const pu4_t *__restrict__ g_hs, uint32_t *__restrict__ g_hdr,
                                                     uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                     uint32_t tile, unsigned long long *__restrict__ rstat)
{
    // 1 024 threads = 256 cells x 4 groups of copies: thread (c, q) adds the copies q, q + 4, .. of cell c (64 loads in
    // flight eight at a time instead of 256), the four partial sums meet in LDS
    __shared__ uint32_t s_part[4][256];
    __shared__ uint32_t s_mn, s_mx, s_hits;
    const uint32_t tid = threadIdx.x, lane = tid & 63, c = tid & 255u, q = tid >> 8;
    const uint32_t nhot = g_hdr[0], cells = g_hdr[HDR_CELLS];
    if (tid == 0) { s_mn = INVALID; s_mx = 0; s_hits = 0; }
    __syncthreads();
    if (blockIdx.x < nhot) {
        const pu4_t h = g_hs[blockIdx.x];
        const uint32_t name = h.x, org = h.y & 0xffffu, width = h.y >> 16, base = h.z;
        const uint16_t *hot16 = reinterpret_cast<const uint16_t *>(g_hot);
        uint32_t mn = INVALID, mx = 0, hits = 0;
        for (uint32_t i0 = 0; i0 < width; i0 += 256) { // (block-uniform trip count)
            const uint32_t i = i0 + c;
            uint32_t sum = 0;
            if (i < width) {
                const uint16_t *p = hot16 + base + i;
#pragma unroll 8
                for (uint32_t w = q; w < G; w += 4) sum += p[(size_t)w * cells];
            }
            s_part[q][c] = sum;
            __syncthreads();
            if (q == 0) {
                sum += s_part[1][c] + s_part[2][c] + s_part[3][c];
                if (sum) {
                    const uint32_t b = org + i;
                    lh::cell_add(counts, (size_t)name * LH_ROW_STRIDE + b, sum);
                    mn = min(mn, b);
                    mx = max(mx, b);
                    hits += sum;
                }
            }
            __syncthreads();
        }
        if (q == 0) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor(mn, d, 64));
                mx = max(mx, (uint32_t)__shfl_xor(mx, d, 64));
                hits += __shfl_xor(hits, d, 64);
            }
            if (lane == 0 && mn != INVALID) { atomicMin(&s_mn, mn); atomicMax(&s_mx, mx); }
            if (lane == 0 && hits) atomicAdd(&s_hits, hits);
        }
        __syncthreads();
        if (tid == 0 && s_mn != INVALID) {
            uint32_t *r = ranges + 2 * (size_t)name;
            if (s_mn < r[0]) atomicMin(&r[0], s_mn);
            if (s_mx > r[1]) atomicMax(&r[1], s_mx);
        }
    }
    if (tile && tid == 0) { // is the survey stale?  (stale_judge above.)
        if (s_hits) atomicAdd(&g_hdr[HDR_HITS], s_hits);
        __threadfence();
        if (atomicAdd(&g_hdr[HDR_TICKET], 1u) == gridDim.x - 1) {
            __threadfence();
            const unsigned long long taken = atomicAdd(&g_hdr[HDR_HITS], 0u);
            const unsigned long long pairs = (unsigned long long)atomicAdd(&g_hdr[HDR_TILES], 0u) * tile;
            const bool stale = stale_judge(g_hdr, rstat, taken, pairs);
            // the engine watches the region overflows (pinned host memory): a stream whose tiles overflow the regions is
            // clustered by name, and later calls take the exact-layout kernel instead -- unless the survey was stale
            const uint32_t ovf = atomicAdd(&g_hdr[HDR_OVF], 0u);
            if (ovf && !stale && rstat)
                __hip_atomic_fetch_add(rstat, (unsigned long long)ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            g_hdr[HDR_OVF] = 0;
            g_hdr[HDR_HITS] = 0;
            g_hdr[HDR_TILES] = 0;
            g_hdr[HDR_TICKET] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// P2 v2: the record is the LDS index
// ---------------------------------------------------------------------------
constexpr size_t P2V2_LDS_BYTES = (P2V2_WINWORDS + 3 * PART_MAX_MPP) * sizeof(uint32_t) + 16;

__global__ __launch_bounds__(P2_BLOCK) void k_part_hist2(const rec16_t *__restrict__ records,
                                                            const uint32_t *__restrict__ cdesc,
                                                            const uint32_t *__restrict__ sorted,
                                                            const uint32_t *__restrict__ part_start,
                                                            const uint32_t *__restrict__ slots,
                                                            const uint32_t *__restrict__ nslots, uint32_t log_np,
                                                            uint32_t mpp, uint32_t log_w, uint32_t nmetrics,
                                                            const NameEntry *__restrict__ g_nt,
                                                            uint64_t *__restrict__ counts,
                                                            uint32_t *__restrict__ ranges)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_org = h + P2V2_WINWORDS;
    uint32_t *s_mn = s_org + PART_MAX_MPP;
    uint32_t *s_mx = s_mn + PART_MAX_MPP;
    const uint32_t slot = blockIdx.x;
    if (slot >= *nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = slots[3 * slot], first = slots[3 * slot + 1], cnt = slots[3 * slot + 2];
    const uint32_t *list = sorted + part_start[p] + first;
    const uint32_t W = 1u << log_w, words = mpp << log_w;
    // one chunk (1 024 records = 2 KiB) per wave per iteration: two 16-byte loads per lane, double-buffered
    auto load_chunk = [&](uint32_t cidx, u4_t (&dst)[2]) {
        const u4_t *src = reinterpret_cast<const u4_t *>(records + (size_t)cidx * CHUNK) + lane;
        dst[0] = __builtin_nontemporal_load(src);
        dst[1] = __builtin_nontemporal_load(src + 64);
    };
    // Each wave walks chunks wave, wave + 16, ...  Their indices and descriptors are fetched 64 at a time (lane l
    // holds the wave's l-th chunk of the batch: two dependent loads per 64 chunks instead of two per chunk), and
    // DEPTH chunks are in flight per wave (8 KiB per wave, 128 KiB per CU) while the oldest is reduced: with one
    // workgroup per CU nothing else hides the load latency.  The first batch and its first DEPTH chunks are requested
    // HERE, before the windows are set up (a slot is a chain of dependent round trips: slot -> chunk list -> records ->
    // LDS -> flush), and a chunk's records are not held back for its descriptor: 2 KiB are read whatever it holds.
    constexpr uint32_t WSTEP = P2_BLOCK / 64, DEPTH = 4;
    u4_t buf[DEPTH][2];
    const uint32_t mine = cnt > wave ? (cnt - wave + WSTEP - 1) / WSTEP : 0u; // chunks of this wave
    uint32_t nb = min(mine, 64u), my_cid = 0, my_cn = 0;
    if (lane < nb) my_cid = list[wave + lane * WSTEP];
    uint32_t org0 = 0;
    if (tid < mpp) {
        const uint32_t m = (tid << log_np) | p;
        org0 = m < nmetrics ? (g_nt[m].org & 0xffffu) : 0u;
    }
    if (lane < nb) my_cn = cdesc[my_cid] & CD_MASK;
#pragma unroll
    for (uint32_t d = 0; d < DEPTH; d++) // (unconditional: a wave without chunks reads chunk 0 and ignores it)
        load_chunk(__builtin_amdgcn_readlane(my_cid, min(d, max(nb, 1u) - 1u)), buf[d]);
    for (uint32_t i = tid; i < words; i += P2_BLOCK) h[i] = 0;
    if (tid < mpp) {
        s_org[tid] = org0;
        s_mn[tid] = INVALID;
        s_mx[tid] = 0;
    }
    __syncthreads();

    auto reduce_chunk = [&](const u4_t (&r4)[2], uint32_t cn) {
        if (cn == CHUNK) { // full chunk (wave-uniform): sixteen unconditional LDS adds per lane
#pragma unroll
            for (uint32_t q = 0; q < 2; q++) {
                const uint32_t rr[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    atomicAdd(&h[rr[t] & 0x7fffu], 1u);
                    atomicAdd(&h[(rr[t] >> 16) & 0x7fffu], 1u);
                }
            }
            return;
        }
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            const uint32_t rr[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t at = q * 512 + lane * 8 + (uint32_t)t * 2; // record index of the low half
                if (at < cn) atomicAdd(&h[rr[t] & 0x7fffu], 1u);
                if (at + 1 < cn) atomicAdd(&h[(rr[t] >> 16) & 0x7fffu], 1u);
            }
        }
    };
    for (uint32_t b0 = 0; b0 < mine; b0 += 64) {
        // Loads are issued unconditionally (positions past the batch re-read its last chunk; the data is ignored) and
        // only the LDS work is conditional: with a load inside a branch the compiler cannot count the loads in
        // flight at the join and falls back to s_waitcnt vmcnt(0) before every chunk, which serialises the four
        // chunks this loop keeps in flight.
        auto fetch = [&](uint32_t k, uint32_t slot) { // k: position in the batch (wave-uniform)
            load_chunk(__builtin_amdgcn_readlane(my_cid, min(k, nb - 1u)), buf[slot]);
        };
        if (b0) { // (more than 1 024 chunks in the slot)
            nb = min(mine - b0, 64u);
            if (lane < nb) {
                my_cid = list[wave + (b0 + lane) * WSTEP];
                my_cn = cdesc[my_cid] & CD_MASK;
            }
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) fetch(d, d);
        }
        for (uint32_t k = 0; k < nb; k += DEPTH) {
#pragma unroll
            for (uint32_t d = 0; d < DEPTH; d++) { // fully unrolled: the slot index is a compile-time constant
                if (k + d < nb) reduce_chunk(buf[d], __builtin_amdgcn_readlane(my_cn, k + d)); // wave-uniform
                fetch(k + d + DEPTH, d);
            }
        }
    }
    __syncthreads();

    // flush: one uint64 atomic per occupied cell
    // (a wave's 64 cells are consecutive bins of ONE name when W >= 64: the name's range is updated once per wave, from
    // the first and the last lane that found a count, not once per occupied cell)
    for (uint32_t i = tid; i < words; i += P2_BLOCK) {
        const uint32_t c = h[i];
        const uint32_t l = i >> log_w, b = s_org[l] + (i & (W - 1));
        if (c)
            lh::cell_add(counts, (size_t)((l << log_np) | p) * LH_ROW_STRIDE + b, c);
        if (W >= 64u) {
            const unsigned long long occ = __builtin_amdgcn_ballot_w64(c != 0);
            if (occ != 0ull && (tid & 63u) == 0u) {
                atomicMin(&s_mn[l], b + (uint32_t)__builtin_ctzll(occ));
                atomicMax(&s_mx[l], b + 63u - (uint32_t)__builtin_clzll(occ));
            }
        } else if (c) {
            atomicMin(&s_mn[l], b);
            atomicMax(&s_mx[l], b);
        }
    }
    __syncthreads();
    if (tid < mpp && s_mn[tid] != INVALID) {
        uint32_t *r = ranges + 2 * (size_t)((tid << log_np) | p);
        if (s_mn[tid] < r[0]) atomicMin(&r[0], s_mn[tid]);
        if (s_mx[tid] > r[1]) atomicMax(&r[1], s_mx[tid]);
    }
}

// ---------------------------------------------------------------------------
// plan + launcher
// ---------------------------------------------------------------------------
struct Part2Plan {
    uint32_t shape;            // bit 0: 0 = <1024, 256>, 1 = <512, 128>; bit 1: direct record stores (k_scatter3);
                               // bit 2 (with bit 1, without bit 0, <= 1 024 names): WIDE, <1024, 512> -- two names per
                               // partition, 16 384-bin reduce windows
    uint32_t block, tile, lds_fixed;
    uint32_t log_np, np, mpp, log_w, cells, g1, chunks_per_wg, nchunks;
    uint32_t region_recs;      // shapes 2, 3: upper bound of the records of LDS the partitions' regions take
    RegionFit fit;             // shapes 2, 3: what k_survey_plan needs to split the LDS between regions and hot windows
    size_t lds_dyn;            // dynamic LDS of the scatter kernel
    size_t off_rec, off_cd, off_sorted, off_small, off_stat, off_nt, off_hs, off_hdr, off_pt, off_hot, off_resume, total;
};

// The cold window of this generation narrows with the name count (names per partition x window = 32 768 cells: 8 192 bins
// at 1 024 names, 1 024 at 8 192), and a sample outside it is a global atomic: normal(0, 1e3) over 8 192 names took 57 ms
// per 1e9 pairs, 21 decades over 4 096 names 49 (lognormal: 3.0).  Above 1 024 names (below, the WIDE shape answers) a
// launch of a stream that leaves more than 1/8 of its mass outside those windows (~45 ms per 1e9 pairs of misses: what the
// third generation costs on 64-bit cells) -- the last survey's report, PartTuning::v2_yield -- is left to the third
// generation, whose windows follow the stream (defined in lh_kernels_part3.h: it must take the launch).
static bool part2_yields_to_part3(size_t n, uint32_t nmetrics, const PartTuning &tune);
static uint32_t part2_cold_log_w(uint32_t nmetrics, uint32_t shape)
{
    const uint32_t names_per_part = (shape & 1u) ? 8u : 4u, npt = (shape & 1u) ? 128u : 256u;
    const uint32_t log_np = std::min(ilog2_ceil(npt), ilog2_ceil((nmetrics + names_per_part - 1) / names_per_part));
    const uint32_t mpp = (nmetrics + (1u << log_np) - 1) >> log_np;
    uint32_t lw = 0;
    while ((mpp << (lw + 1)) <= P2V2_WINWORDS && lw < 13u) lw++;
    return lw;
}

static bool make_plan2(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune, Part2Plan &P)
{
    if (!tune.v2 || n < (tune.v2_min_samples ? tune.v2_min_samples : V2_MIN_SAMPLES) || n > (size_t(1) << 31)) return false;
    if (nmetrics < 2 || nmetrics > V2_MAX_NAMES) return false;
    if (part2_yields_to_part3(n, nmetrics, tune)) return false;
    P.shape = tune.v2_shape & 7u;
    if ((P.shape & 4u) && ((P.shape & 3u) != 2u || nmetrics > 1024u)) P.shape &= 3u; // wide: the region kernel, one workgroup per CU, few names
    const bool half = P.shape & 1u, direct = P.shape & 2u, wide = P.shape & 4u;
    const uint32_t npt = wide ? 512u : half ? 128u : 256u, wgs_per_cu = half ? 2u : 1u;
    P.block = half ? 512u : 1024u;
    P.tile = P.block * V2_SPT;
    if (direct)
        P.lds_fixed = (uint32_t)(wide ? sizeof(Scatter3LdsT<512>) : half ? sizeof(Scatter3LdsT<128>) : sizeof(Scatter3LdsT<256>));
    else
        P.lds_fixed = (uint32_t)(half ? sizeof(Scatter2LdsT<512, 128>) : sizeof(Scatter2LdsT<1024, 256>));
    const uint32_t names_per_part = wide ? 2u : half ? 8u : 4u;
    const uint32_t want_np = (nmetrics + names_per_part - 1) / names_per_part;
    P.log_np = std::min(ilog2_ceil(npt), ilog2_ceil(want_np));
    P.np = 1u << P.log_np;
    P.mpp = (nmetrics + P.np - 1) >> P.log_np;
    if (P.mpp > PART_MAX_MPP) return false;
    uint32_t lw = 0;
    while ((P.mpp << (lw + 1)) <= P2V2_WINWORDS && lw < (wide ? 14u : 13u)) lw++;
    P.log_w = lw; // cold window = 2^log_w bins per name; the record (local << log_w | offset) is < 32 768
    // hot windows: whatever LDS is left beside the scatter structures and the per-name table
    const size_t budget = V2_LDS_TOTAL / wgs_per_cu;
    P.region_recs = direct ? region_records(P.tile * SC3_TILES_PER_FLUSH, P.np) : 0u;
    const size_t nt_bytes = (size_t)(direct ? sc3_nt_entries(nmetrics) : nmetrics) * sizeof(NameEntry);
    const size_t fixed = P.lds_fixed + nt_bytes + (size_t)P.region_recs * sizeof(rec16_t) + 256;
    // k_scatter3 (direct): 16-bit cells, and the plan sizes the regions itself (RegionFit) -- P.cells is what the windows
    // get if the regions need their upper bound; k_scatter2: 32-bit cells
    const uint32_t cell_bytes = direct ? 2u : 4u, max_cells = direct ? 65472u : 40000u & ~63u; // (cell offsets are 16-bit fields)
    P.cells = fixed + 4096 <= budget ? (uint32_t)((budget - fixed) / cell_bytes) & ~63u : 0u;
    if (P.cells > max_cells) P.cells = max_cells;
    if (!tune.hot) P.cells = 0;
    P.fit = RegionFit{0u, P.log_np, 0u, cell_bytes, direct ? 2u : 1u, tune.hot ? max_cells : 0u};
    if (direct) {
        // regions + windows share what the fixed parts leave of the budget (the plan decides how) -- or, when the name
        // table leaves less than the regions' upper bound (8 192 names in half a CU's LDS), exactly that bound
        const size_t fixed0 = P.lds_fixed + nt_bytes, bound = (size_t)P.region_recs * sizeof(rec16_t);
        const size_t avail = std::max(budget > fixed0 + 256 ? budget - 256 - fixed0 : 0, bound);
        P.fit.tile = P.tile * SC3_TILES_PER_FLUSH;
        P.fit.avail_bytes = (uint32_t)avail;
        P.lds_dyn = fixed0 + avail;
    } else {
        P.lds_dyn = fixed - 256 + (size_t)P.cells * 4;
    }
    const size_t ntiles = (n + P.tile - 1) / P.tile;
    size_t g1 = (size_t)num_cus * wgs_per_cu; // the workgroups of a CU own its LDS between them
    if (g1 > (ntiles + 3) / 4) g1 = (ntiles + 3) / 4;
    if (g1 < 1) g1 = 1;
    P.g1 = (uint32_t)g1;
    const size_t tiles_per_wg = (ntiles + g1 - 1) / g1;
    P.chunks_per_wg = (uint32_t)(tiles_per_wg * (P.tile / CHUNK) + P.np + 1);
    P.nchunks = P.g1 * P.chunks_per_wg;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~size_t(255); return at; };
    // the survey's tables come first: their offsets depend on the name count only, so the sub-launches of one
    // call (whose other regions shrink with n) all find the survey of the first one
    P.off_stat = take((size_t)nmetrics * 20 + 8 + 32); // (+ the six words of k_survey_mass2)
    P.off_nt = take((size_t)nmetrics * sizeof(NameEntry));
    P.off_hs = take((size_t)V2_MAX_SLOTS * sizeof(pu4_t));
    P.off_hdr = take(64);
    P.off_pt = take(512 * sizeof(pu2_t));
    P.off_hot = take(direct ? (size_t)P.fit.max_cells * 2 * ((size_t)num_cus * wgs_per_cu) : 0); // k_scatter3's windows, one copy per workgroup
    P.off_resume = take((size_t)num_cus * wgs_per_cu * 4); // k_scatter3: the first tile each workgroup left to k_scatter_clustered
    P.off_rec = take((size_t)P.nchunks * CHUNK * sizeof(rec16_t));
    P.off_cd = take((size_t)P.nchunks * sizeof(uint32_t));
    P.off_sorted = take((size_t)P.nchunks * sizeof(uint32_t));
    P.off_small = take(small_words(P.np, P2V2_SLOT_EXTRA) * sizeof(uint32_t));
    P.total = o;
    return true;
}

size_t part2_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune)
{
    Part2Plan P;
    return make_plan2(n, nmetrics, num_cus, tune, P) ? P.total : 0;
}

// survey_n > 0: first sub-launch of a call -- survey pairs [0, survey_n) of the same arrays (the whole call) before
// the scatter; survey_n == 0: a later sub-launch, the tables of the first one are still in the scratch block.
template <typename IDT>
static hipError_t launch_part2_t(const IDT *d_ids, const double *d_v, size_t n, size_t survey_n,
                                 uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                 uint32_t *d_err, void *scratch, size_t scratch_bytes, int num_cus,
                                 const PartTuning &tune, unsigned long long *region_stat, hipStream_t s)
{
    Part2Plan P;
    if (!make_plan2(n, nmetrics, num_cus, tune, P) || scratch_bytes < P.total || !scratch) return hipErrorInvalidValue;
    if (!part_aligned(d_ids, d_v)) return hipErrorInvalidValue;
    const size_t p1_dyn = P.lds_dyn;
    const size_t sv_dyn = (size_t)nmetrics * 16;
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attributes
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_hist2),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P2V2_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter2<1024, 256, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)V2_LDS_TOTAL);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter2<512, 128, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V2_LDS_TOTAL / 2));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter3<1024, 256, SC3_BATCH, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)V2_LDS_TOTAL);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter3<512, 128, SC3_BATCH, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V2_LDS_TOTAL / 2));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter3<1024, 512, SC3_BATCH, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)V2_LDS_TOTAL);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_clustered<IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)CL_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_survey_count<IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(V2_MAX_NAMES * 16));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_count),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_scatter),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    unsigned char *base = static_cast<unsigned char *>(scratch);
    LevelPtrs L1 = level_ptrs(base, P.off_rec, P.off_cd, P.off_sorted, P.off_small, P.np, P2V2_SLOT_EXTRA);
    rec16_t *records = reinterpret_cast<rec16_t *>(base + P.off_rec);
    uint32_t *g_cnt = reinterpret_cast<uint32_t *>(base + P.off_stat);
    uint32_t *g_mninv = g_cnt + nmetrics, *g_mx = g_mninv + nmetrics;
    unsigned long long *g_sum = reinterpret_cast<unsigned long long *>(g_mx + nmetrics + (nmetrics & 1u));
    NameEntry *g_nt = reinterpret_cast<NameEntry *>(base + P.off_nt);
    pu4_t *g_hs = reinterpret_cast<pu4_t *>(base + P.off_hs);
    uint32_t *g_hdr = reinterpret_cast<uint32_t *>(base + P.off_hdr);
    pu2_t *g_pt = reinterpret_cast<pu2_t *>(base + P.off_pt);
    uint32_t *g_hot = reinterpret_cast<uint32_t *>(base + P.off_hot);
    uint32_t *g_resume = reinterpret_cast<uint32_t *>(base + P.off_resume);

    hipError_t e = hipMemsetAsync(L1.cdesc, 0xff, (size_t)P.nchunks * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(L1.pc, 0, small_words(P.np, P2V2_SLOT_EXTRA) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (survey_n) {
        e = hipMemsetAsync(g_cnt, 0, (size_t)nmetrics * 20 + 8 + 32, s);
        if (e != hipSuccess) return e;
        const size_t sv_tiles = (survey_n / 2 + 2047) / 2048;
        const unsigned sv_grid = (unsigned)std::min<size_t>(SV_GRID, std::max<size_t>(1, sv_tiles));
        hipLaunchKernelGGL(k_survey_count<IDT>, dim3(sv_grid), dim3(V2_BLOCK), sv_dyn, s, d_ids, d_v, survey_n, nmetrics,
                           d_Tx, g_cnt, g_mninv, g_mx, g_sum);
        uint32_t *g_mass = reinterpret_cast<uint32_t *>(g_sum + nmetrics);
        hipLaunchKernelGGL(k_survey_mass2<IDT>, dim3(sv_grid), dim3(V2_BLOCK), 0, s, d_ids, d_v, survey_n, nmetrics, d_Tx, g_cnt,
                           g_sum, g_mass);
        hipLaunchKernelGGL(k_survey_plan, dim3(1), dim3(V2_BLOCK), 0, s, g_cnt, g_mninv, g_mx, g_sum, g_mass, nmetrics,
                           P.log_w, P.cells, P.fit, g_nt, g_hs, g_hdr, g_pt,
                           region_stat ? reinterpret_cast<uint32_t *>(region_stat + 1) : nullptr);
    }
    if (P.shape & 2u) { // whole tiles through the region kernel, the last n % tile pairs through the plain kernel
        const size_t nt_full = n / P.tile, done = nt_full * P.tile;
        if (P.shape == 3)
            hipLaunchKernelGGL((k_scatter3<512, 128, SC3_BATCH, IDT>), dim3(P.g1), dim3(512), p1_dyn, s, d_ids, d_v, nt_full, nmetrics,
                               P.log_np, P.log_w, d_Tx, g_nt, g_hs, g_hdr, g_pt, g_hot, records,
                               L1.cdesc, P.chunks_per_wg, counts, ranges, d_err, region_stat, g_resume);
        else if (P.shape & 4u)
            hipLaunchKernelGGL((k_scatter3<1024, 512, SC3_BATCH, IDT>), dim3(P.g1), dim3(1024), p1_dyn, s, d_ids, d_v, nt_full,
                               nmetrics, P.log_np, P.log_w, d_Tx, g_nt, g_hs, g_hdr, g_pt, g_hot,
                               records, L1.cdesc, P.chunks_per_wg, counts, ranges, d_err, region_stat, g_resume);
        else
            hipLaunchKernelGGL((k_scatter3<1024, 256, SC3_BATCH, IDT>), dim3(P.g1), dim3(1024), p1_dyn, s, d_ids, d_v, nt_full,
                               nmetrics, P.log_np, P.log_w, d_Tx, g_nt, g_hs, g_hdr, g_pt, g_hot,
                               records, L1.cdesc, P.chunks_per_wg, counts, ranges, d_err, region_stat, g_resume);
        if (P.shape != 3)
            hipLaunchKernelGGL(k_scatter_clustered<IDT>, dim3(P.g1), dim3(1024), CL_LDS_BYTES, s, d_ids, d_v, nt_full, nmetrics,
                               d_Tx, g_resume, counts, ranges, d_err, g_hdr + HDR_OVF);
        hipLaunchKernelGGL(k_hot_reduce, dim3(V2_MAX_SLOTS), dim3(1024), 0, s, g_hot, P.g1, g_hs, g_hdr, counts, ranges, P.tile,
                           region_stat);
        if (done < n) {
            e = launch_ingest_pairs(d_ids + done, d_v + done, n - done, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s);
            if (e != hipSuccess) return e;
        }
    }
    else if (P.shape == 1)
        hipLaunchKernelGGL((k_scatter2<512, 128, IDT>), dim3(P.g1), dim3(512), p1_dyn, s, d_ids, d_v, n, nmetrics, P.log_np,
                           P.log_w, d_Tx, g_nt, g_hs, g_hdr, P.cells, records, L1.cdesc, P.chunks_per_wg, counts,
                           ranges, d_err);
    else
        hipLaunchKernelGGL((k_scatter2<1024, 256, IDT>), dim3(P.g1), dim3(1024), p1_dyn, s, d_ids, d_v, n, nmetrics,
                           P.log_np, P.log_w, d_Tx, g_nt, g_hs, g_hdr, P.cells, records, L1.cdesc, P.chunks_per_wg,
                           counts, ranges, d_err);
    e = run_plan(L1, P.nchunks, P.np, 0u, P2V2_SLOT_EXTRA, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_part_hist2, dim3(P.np + P2V2_SLOT_EXTRA), dim3(P2_BLOCK), P2V2_LDS_BYTES, s, records, L1.cdesc,
                       L1.sorted, L1.part_start, L1.slots, L1.nslots, P.log_np, P.mpp, P.log_w, nmetrics, g_nt, counts,
                       ranges);
    return hipGetLastError();
}

hipError_t launch_ingest_pairs_part2(Ids d_ids, const double *d_v, size_t n, size_t survey_n,
                                     uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, const double *d_Tx,
                                     uint32_t *d_err, void *scratch, size_t scratch_bytes, int num_cus,
                                     const PartTuning &tune, unsigned long long *region_stat, hipStream_t s)
{
    return d_ids.width == 2 ? launch_part2_t(d_ids.u16(), d_v, n, survey_n, counts, ranges, nmetrics, d_Tx, d_err, scratch,
                                             scratch_bytes, num_cus, tune, region_stat, s)
                            : launch_part2_t(d_ids.u32(), d_v, n, survey_n, counts, ranges, nmetrics, d_Tx, d_err, scratch,
                                             scratch_bytes, num_cus, tune, region_stat, s);
}
