// lh_kernels_part.hip -- mixed (id, value) ingest for large name spaces, gfx950.
//
// Reference semantics: Histogram(name, v) = histogramCache[name][compress(v)] += 1
// (metrics.go:273-295, 316-322) for a stream that interleaves many names.
//
// Why not one kernel: 1e9 samples over 1 024 Zipf names land in ~7e5 occupied
// (name, bucket) cells.  That working set does not fit 160 KiB of LDS, and global
// uint64 atomics sustain only ~8 G/s on this part (same-address atomics serialise at
// ~12 ns), i.e. ~1 % of the HBM roofline.  So the stream is PARTITIONED BY NAME first
// and each partition is then reduced in LDS:
//
//   P1  k_scatter_samples : read (id, v) [12 B], compress on the fly, emit a 4-byte record
//                        (partition << 24 | local name << 16 | bin) into the chunk list of
//                        partition id % 256.  Tile-local counting sort in LDS so that every
//                        partition's records leave the CU as contiguous, line-buffered runs;
//                        chunks (1 024 records) come from a workgroup-private pool, so the
//                        hot loop has no global atomics at all.
//   plan k_plan_*      : group the chunk descriptors by partition and cut them into equal
//                        work slots (three tiny kernels).
//   P1b k_scatter_records (only above 16 384 names): the same scatter once more, on the 4-byte
//                        records of each level-1 slot, by the next bits of the name id -- up
//                        to 64 sub-partitions, 16 384 partitions in all -- so that P2 again
//                        sees at most 4 names per partition.  +8 B/sample of traffic.
//   P2  k_part_hist    : one workgroup per slot: LDS windows (uint32) for the few names
//                        of that partition, ds_add per record, then ONE uint64 global
//                        atomic per occupied cell at flush.
//
// HBM traffic: 12 (read) + 4 (write) + 4 (read) = 20 B/sample against 12 B algorithmic
// (28 B/sample with the second level).  Everything is exact: records carry the exact bin;
// out-of-window records and small launches fall back to direct global atomics.
#include "lh_kernels.h"
#include "lh_codec.h"
#include "lh_ids.h"
#include "lh_windows.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace lh {

constexpr int NPMAX = 256;             // partitions per scatter pass (power of two, <= 256)
constexpr uint32_t NQMAX = 16384;      // partitions after the second level
constexpr uint32_t CHUNK = 1024;       // records per chunk (4 KiB)
constexpr int P1_BLOCK = 512;          // 8 waves; three workgroups per CU (measured: 1024x4 is 10 % slower)
constexpr int P1_SPT = 8;              // samples per thread per tile
constexpr int P1_TILE = P1_BLOCK * P1_SPT;
// tuning knobs (profiles/r01c): records per staged line and P1 workgroups per CU
#ifndef LH_LINE
#define LH_LINE 16
#endif
#ifndef LH_P1_WGS_PER_CU
#define LH_P1_WGS_PER_CU 3
#endif
constexpr uint32_t INVALID = 0xffffffffu;
constexpr int P2_BLOCK = 1024;         // 16 waves: two workgroups (64 KiB windows each) fill a CU
constexpr uint32_t P2_WINWORDS = 16384; // 64 KiB of uint32 windows per workgroup
constexpr uint32_t SLOT_EXTRA = 1024;  // work slots beyond one per partition (measured: 512 is 25 % slower)
// Smallest launch that takes a partitioned path by default (round 6, profiles/r06_small_calls.txt: below it the cell-table
// kernel -- launch_ingest_pairs_cells, no scratch, no survey -- is faster): 2^20 pairs up to 8 192 names, 3 * 2^20 above
// (two scatter levels: ~0.15 ms of fixed work per launch).  Until round 6: 131 072 for both.
constexpr size_t PART_MIN_SAMPLES = size_t(1) << 20, PART_MIN_SAMPLES_2L = size_t(3) << 20;
constexpr uint32_t PART_MAX_MPP = 256;
// chunk descriptor: partition tag << 11 | records in the chunk (1..1024); INVALID = unused
constexpr uint32_t CD_SHIFT = 11, CD_MASK = (1u << CD_SHIFT) - 1;

struct PartPlan {
    uint32_t log_np, np, mpp;          // level 1: partitions, names per partition
    uint32_t log_ns, ns;               // level 2: sub-partitions per partition (1 = no second level)
    uint32_t log_nq, nq, mpp2, log_w;  // what P2 sees
    uint32_t g1, chunks_per_wg, nchunks1, nchunks2;
    uint32_t extra1;                   // work slots beyond one per partition in the level-1 plan
    bool hot;                          // P1 keeps LDS windows for the hottest names (k_scatter_samples<true>)
    size_t off_rec1, off_cd1, off_sorted1, off_small1, off_rec2, off_cd2, off_sorted2, off_small2, total;
};

static uint32_t ilog2_ceil(uint32_t x)
{
    uint32_t l = 0;
    while ((1u << l) < x) l++;
    return l;
}

// words of the "small" region of one level: pc, part_start, cursor [nq each], slots [3 x (nq + extra)],
// nslots [1], pool_start [nq + extra + 1]   (extra = slots beyond one per partition)
static size_t small_words(uint32_t nq, uint32_t extra) { return (size_t)3 * nq + 3 * (nq + extra) + 1 + (nq + extra + 1) + 16; }

// Smallest launch worth partitioning (below it: one global atomic per sample, k_ingest_pairs).
static size_t part_min_samples(const PartTuning &tune, uint32_t nmetrics)
{
    return tune.part_min_samples ? tune.part_min_samples : nmetrics > 8192u ? PART_MIN_SAMPLES_2L : PART_MIN_SAMPLES;
}

static bool make_plan(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune, PartPlan &P)
{
    if (n < part_min_samples(tune, nmetrics) || n > (size_t(1) << 31) || nmetrics < 2) return false;
    // names per partition: 4 gives every name a 4 096-bin LDS window in P2.  Fewer partitions mean longer
    // contiguous runs in P1 but narrower windows in P2 (measured at 1 024 names, profiles/r01c: 4 is the best
    // overall).
    const uint32_t names_per_part = std::max(1u, tune.names_per_part);
    // Second level only when a level-1 partition holds more than 32 names (> 8 192 names in all).
    // Measured at 1e9 samples (profiles/r01j): 8 192 names one level 139 vs two levels 105 G samples/s (512-bin
    // windows plus the overflow table still catch most records, and the extra pass costs 8 B/sample);
    // 12 288 names 85 vs 100; 16 384 names 73 vs 94; 65 536 names 27 vs 77.  lh_set_option(LH_OPT_TWO_LEVEL_ABOVE)
    // overrides the threshold, in names per level-1 partition (tests force the second level at small name counts).
    const uint32_t two_level_above = tune.two_level_above;
    const uint32_t want_np = (nmetrics + names_per_part - 1) / names_per_part;
    P.log_np = std::min(8u, ilog2_ceil(want_np));
    P.np = 1u << P.log_np;
    P.mpp = (nmetrics + P.np - 1) >> P.log_np;
    if (P.mpp > PART_MAX_MPP) return false;
    P.log_ns = 0;
    if (P.mpp > names_per_part && P.mpp > two_level_above)
        P.log_ns = std::min(6u, ilog2_ceil((P.mpp + names_per_part - 1) / names_per_part));
    P.ns = 1u << P.log_ns;
    P.log_nq = P.log_np + P.log_ns;
    P.nq = 1u << P.log_nq;
    P.mpp2 = (P.mpp + P.ns - 1) >> P.log_ns;
    uint32_t lw = 0;
    while ((P.mpp2 << (lw + 1)) <= P2_WINWORDS) lw++;
    P.log_w = lw; // window = 2^log_w bins per name, mpp2 * window <= 16384
    const size_t ntiles = (n + P1_TILE - 1) / P1_TILE;
    // Hot-name windows inside P1 (k_scatter_samples<true>, +34 KiB of LDS: two workgroups per CU) when the launch
    // is long enough (>= 32 tiles per workgroup) to amortise the per-workgroup selection of the hot names
    // (lh_set_option(LH_OPT_HOT_MIN_TILES / LH_OPT_HOT_WINDOWS): tests exercise the path on small inputs).
    const size_t min_tiles = std::max<size_t>(1, tune.hot_min_tiles);
    P.hot = tune.hot && ntiles >= (size_t)num_cus * 2 * min_tiles;
    // ~44 KiB LDS per workgroup: three 512-thread workgroups per CU (two with the hot windows)
    size_t g1 = (size_t)num_cus * (P.hot ? 2 : LH_P1_WGS_PER_CU);
    // every workgroup strands up to NP partially filled chunks (1 MiB at NP = 256): give a workgroup
    // at least 8 tiles so that a lane-sized launch (1M samples) needs ~33 MB of scratch, not ~270 MB
    if (g1 > (ntiles + 7) / 8) g1 = (ntiles + 7) / 8;
    if (g1 < 1) g1 = 1;
    P.g1 = (uint32_t)g1;
    const size_t tiles_per_wg = (ntiles + g1 - 1) / g1;
    P.chunks_per_wg = (uint32_t)(tiles_per_wg * (P1_TILE / CHUNK) + P.np + 1);
    P.nchunks1 = P.g1 * P.chunks_per_wg;
    // level 2: every level-1 slot is one P1b workgroup and re-scatters its chunks into <= cnt + ns + 1 chunks.
    // Measured at 65 536 names: with 1 024 extra slots P1b ran 1 280 workgroups in 1.67 uneven waves (4.6 ms for
    // 8 GB); 4 096 extra slots give ~5.7 waves of shorter workgroups.
    P.extra1 = P.log_ns ? 4096u : SLOT_EXTRA;
    P.nchunks2 = P.log_ns ? P.nchunks1 + (P.np + P.extra1) * (P.ns + 1) : 0;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~size_t(255); return at; };
    P.off_rec1 = take((size_t)P.nchunks1 * CHUNK * sizeof(uint32_t));
    P.off_cd1 = take((size_t)P.nchunks1 * sizeof(uint32_t));
    P.off_sorted1 = take((size_t)P.nchunks1 * sizeof(uint32_t));
    P.off_small1 = take(small_words(P.np, P.extra1) * sizeof(uint32_t));
    P.off_rec2 = take((size_t)P.nchunks2 * CHUNK * sizeof(uint32_t));
    P.off_cd2 = take((size_t)P.nchunks2 * sizeof(uint32_t));
    P.off_sorted2 = take((size_t)P.nchunks2 * sizeof(uint32_t));
    P.off_small2 = take(P.log_ns ? small_words(P.nq, SLOT_EXTRA) * sizeof(uint32_t) : 0);
    P.total = o;
    return true;
}

bool part_aligned(Ids d_ids, const double *d_v)
{
    return (((uintptr_t)d_v & 15) == 0) && d_ids.pair_aligned();
}

size_t part_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus, const PartTuning &tune)
{
    PartPlan P;
    return make_plan(n, nmetrics, num_cus, tune, P) ? P.total : 0;
}

// ---------------------------------------------------------------------------
// Scatter machinery shared by P1 (samples) and P1b (records)
// ---------------------------------------------------------------------------
typedef double pd2_t __attribute__((ext_vector_type(2)));
typedef uint32_t pu2_t __attribute__((ext_vector_type(2)));
typedef uint32_t pu4_t __attribute__((ext_vector_type(4)));

// Write-out is LINE-BUFFERED: measured, letting each tile emit its ~16-record (64 B) runs directly
// costs 2.8 ms of a 5.0 ms kernel and 1.6x write amplification (partial lines evicted before their
// other half arrives).  So every partition owns one staging line in LDS; a tile emits only whole,
// aligned lines (staged leftovers + new records) and keeps the remainder staged.
constexpr uint32_t LINE = LH_LINE; // records per staged line (16 = 64 B, one aligned HBM sector pair)

struct ScatterLds {
    uint32_t cnt[NPMAX], off[NPMAX], cfill[NPMAX], cbase[NPMAX];
    uint32_t sf[NPMAX];             // records currently staged per partition (< LINE)
    uint32_t d1[NPMAX], n1[NPMAX];  // this tile: global index / count of the staged records to emit
    // per partition, for sorted position i:  i < E ? global[(i < T ? A : B) + i] : stage[i - E + sf0]
    // packed as {A, B, T, E | sf0 << 16}
    pu4_t tbl[NPMAX];
    uint32_t sorted[P1_TILE];
    uint32_t stage[NPMAX * LINE];
    uint32_t wsum[4];
    uint32_t pool_next, total;
};

__device__ __forceinline__ void scatter_init(ScatterLds &L, uint32_t tid)
{
    if (tid < NPMAX) { L.cnt[tid] = 0; L.cfill[tid] = CHUNK; L.cbase[tid] = INVALID; L.sf[tid] = 0; L.d1[tid] = INVALID; }
    if (tid == 0) L.pool_next = 0;
}

// Phases after the rank atomics of a tile: scan + chunk bookkeeping, tile-local counting sort, copy-out.
// pr[j] = partition | rank << 8 (INVALID for an absent sample); rec[j] carries the partition in bits 24..31.
// tag(p) = (p << tag_shift) | tag_base is the partition id written into chunk descriptors.
// `prefetch` issues the next tile's loads; it runs right after the first barrier.
template <class Prefetch>
__device__ __forceinline__ void scatter_tile(ScatterLds &L, const uint32_t (&rec)[P1_SPT], const uint32_t (&pr)[P1_SPT],
                                             uint32_t *__restrict__ records, uint32_t *__restrict__ cdesc,
                                             uint32_t pool_base, uint32_t tag_shift, uint32_t tag_base, uint32_t tid,
                                             Prefetch prefetch)
{
    const uint32_t lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    prefetch();

    // exclusive scan of the per-partition counts + chunk bookkeeping (threads 0..255)
    // (no barrier between the wave scans and their bases: every scanning wave sums the counts of the waves
    // before it itself, three extra LDS reads at most, instead of meeting the others at a barrier)
    if (tid < NPMAX) {
        const uint32_t c = L.cnt[tid];
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += y;
        }
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < wave; w++) { // wave-uniform trip count
            uint32_t x = L.cnt[w * 64 + lane];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
            wbase += x;
        }
        const uint32_t off = wbase + inc - c;
        L.off[tid] = off;
        if (tid == NPMAX - 1) L.total = wbase + inc;
        L.d1[tid] = INVALID;
        if (c) {
            const uint32_t p = tid, tag = ((p << tag_shift) | tag_base) << CD_SHIFT;
            const uint32_t sf = L.sf[p], cf = L.cfill[p], cb = L.cbase[p];
            const uint32_t total = sf + c, nfull = total / LINE, out = nfull * LINE;
            const uint32_t room = CHUNK - cf; // multiple of LINE (0 when there is no open chunk)
            uint32_t first = 0;
            if (out > room) {
                const uint32_t over = out - room;
                const uint32_t k = (over + CHUNK - 1) / CHUNK;
                first = pool_base + atomicAdd(&L.pool_next, k);   // k consecutive chunks
                if (cb != INVALID) cdesc[cb] = tag | CHUNK;        // the old chunk is now full
                for (uint32_t q = 0; q + 1 < k; q++) cdesc[first + q] = tag | CHUNK;
                L.cbase[p] = first + k - 1;
                L.cfill[p] = over - (k - 1) * CHUNK;
            } else {
                L.cfill[p] = cf + out;
            }
            L.sf[p] = total - out;
            // emitted element u = sf + (i - off) of this partition goes to
            //   u < room : cb*CHUNK + cf + u            = A + i
            //   else     : first*CHUNK + (u - room)      = B + i     (new chunks are consecutive)
            pu4_t t;
            t.x = cb * CHUNK + cf + sf - off;
            t.y = first * CHUNK + sf - off - room;
            t.z = room ? off + room - sf : off;
            const uint32_t E = nfull ? off + out - sf : off;
            t.w = E | ((nfull ? 0u : sf) << 16);
            L.tbl[p] = t;
            if (nfull && sf) {
                L.d1[p] = room ? cb * CHUNK + cf : first * CHUNK;
                L.n1[p] = sf;
            }
        }
    }
    __syncthreads();

    // the counts have been consumed by every scanning wave (ranks are in registers); the next tile's rank
    // atomics come after the barrier below
    if (tid < NPMAX) L.cnt[tid] = 0;
    // tile-local counting sort into LDS ...
#pragma unroll
    for (int j = 0; j < P1_SPT; j++) {
        if (pr[j] != INVALID) L.sorted[L.off[pr[j] & 0xffu] + (pr[j] >> 8)] = rec[j];
    }
    // ... and, independently, the previously staged records of partitions that emit a line.  One thread per
    // eight staged slots writes them as two 16-byte stores when any of them is live: the line's start d1 is
    // 64-byte aligned, and the slots at or beyond n1 belong to the same emitted line, which the copy-out below
    // (after the barrier, so ordered after these stores) fills with this tile's records.
    static_assert(P1_BLOCK * 8 == NPMAX * LINE || LINE != 16, "one thread per 8 staged records");
    if (LINE == 16) {
        const uint32_t p = tid >> 1, u0 = (tid & 1u) * 8u;
        const uint32_t d = L.d1[p];
        if (d != INVALID && u0 < L.n1[p]) {
            const pu4_t a = *reinterpret_cast<const pu4_t *>(&L.stage[p * LINE + u0]);
            const pu4_t b = *reinterpret_cast<const pu4_t *>(&L.stage[p * LINE + u0 + 4]);
            *reinterpret_cast<pu4_t *>(&records[d + u0]) = a;
            *reinterpret_cast<pu4_t *>(&records[d + u0 + 4]) = b;
        }
    } else {
        for (uint32_t e = tid; e < NPMAX * LINE; e += P1_BLOCK) {
            const uint32_t p = e / LINE, u = e % LINE;
            const uint32_t d = L.d1[p];
            if (d != INVALID && u < L.n1[p]) records[d + u] = L.stage[e];
        }
    }
    __syncthreads();

    // copy out: whole lines to HBM, the remainder of each partition into its staging line
    const uint32_t tile_total = L.total;
    for (uint32_t i = tid; i < tile_total; i += P1_BLOCK) {
        const uint32_t r = L.sorted[i];
        const pu4_t t = L.tbl[r >> 24];
        const uint32_t E = t.w & 0xffffu;
        if (i < E) {
            records[(i < t.z ? t.x : t.y) + i] = r;
        } else {
            L.stage[(r >> 24) * LINE + (i - E) + (t.w >> 16)] = r;
        }
    }
    // (the next tile's first barrier separates this copy-out from the next bookkeeping)
}

// Drain: the staged remainders (one partial line per partition) and the open chunks' descriptors.
__device__ __forceinline__ void scatter_drain(ScatterLds &L, uint32_t *__restrict__ records,
                                              uint32_t *__restrict__ cdesc, uint32_t pool_base, uint32_t np,
                                              uint32_t tag_shift, uint32_t tag_base, uint32_t tid)
{
    __syncthreads();
    if (tid < NPMAX) {
        L.d1[tid] = INVALID;
        const uint32_t p = tid, sf = L.sf[p];
        if (sf) {
            uint32_t cf = L.cfill[p], cb = L.cbase[p];
            if (cf == CHUNK) { // no open chunk, or it is exactly full
                if (cb != INVALID) cdesc[cb] = (((p << tag_shift) | tag_base) << CD_SHIFT) | CHUNK;
                cb = pool_base + atomicAdd(&L.pool_next, 1u);
                cf = 0;
                L.cbase[p] = cb;
            }
            L.d1[p] = cb * CHUNK + cf;
            L.n1[p] = sf;
            L.cfill[p] = cf + sf;
        }
    }
    __syncthreads();
    for (uint32_t e = tid; e < NPMAX * LINE; e += P1_BLOCK) {
        const uint32_t p = e / LINE, u = e % LINE;
        const uint32_t d = L.d1[p];
        if (d != INVALID && u < L.n1[p]) records[d + u] = L.stage[e];
    }
    if (tid < np && L.cbase[tid] != INVALID)
        cdesc[L.cbase[tid]] = (((tid << tag_shift) | tag_base) << CD_SHIFT) | L.cfill[tid];
}

// ---------------------------------------------------------------------------
// P1: compress + partition scatter of (id, value) samples
// Record format (4 bytes): partition << 24 | local name id << 16 | bin.
// ids must be 8-byte and v 16-byte aligned (the launcher checks).
// ---------------------------------------------------------------------------
// Hot names (HOT = true).  Metric streams are skewed: under BASELINE config 3's Zipf(1.0) over 1 024 names
// the 16 most frequent names carry 45 % of the samples.  Each workgroup picks the HOT_NAMES names that are
// most frequent in its own first tile and keeps a HOT_W-bin LDS window for each; a sample of a hot name that
// falls inside its window is counted right there (one LDS atomic, like the single-pass kernels) and never
// becomes a record, so it costs no record store, no sort and no P2 work.  Everything else -- cold names and
// hot samples outside their window -- takes the scatter path unchanged, so the result stays exact.  The
// windows are flushed with one uint64 atomic per occupied bin when the workgroup ends.
constexpr uint32_t HOT_NAMES = 16, HOT_LOGW = 9, HOT_W = 1u << HOT_LOGW;
// Selection works on a 2 048-entry (tag, count) table in ScatterLds::sorted: exact for <= 2 048 names, a hashed
// heavy-hitter sketch above (an entry belongs to the first name that claims it; names that lose the race for an
// entry are simply not candidates).  The name -> slot map is direct-mapped on the low 8 bits of the id; a
// candidate whose map entry is taken is skipped.  Both only decide WHICH names get a window, never a count.
constexpr uint32_t HOT_TAGS = 2048, HOT_MAPW = 256, HOT_ROUNDS = 24;
static_assert(2 * HOT_TAGS <= (uint32_t)P1_TILE, "the selection table lives in ScatterLds::sorted");
constexpr uint32_t HOT_EMPTY = 0xffffffffu;
struct HotLds {
    uint32_t win[HOT_NAMES * HOT_W];
    uint32_t org[HOT_NAMES], name[HOT_NAMES], mn[HOT_NAMES], mx[HOT_NAMES];
    uint32_t wmax[P1_BLOCK / 64];
    uint32_t pick, nsel;
    uint32_t hmap[HOT_MAPW]; // id << 8 | slot, HOT_EMPTY when free
};
constexpr size_t HOT_LDS_BYTES = sizeof(HotLds);

template <bool HOT, typename IDT>
__global__ __launch_bounds__(P1_BLOCK, HOT ? 4 : 6) void k_scatter_samples(const IDT *__restrict__ ids,
                                                                           const double *__restrict__ v, size_t n,
                                                                           uint32_t nmetrics, uint32_t log_np,
                                                                           const double *__restrict__ Tx,
                                                                           uint32_t *__restrict__ records,
                                                                           uint32_t *__restrict__ cdesc,
                                                                           uint32_t chunks_per_wg,
                                                                           uint64_t *__restrict__ counts,
                                                                           uint32_t *__restrict__ ranges,
                                                                           uint32_t *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) ScatterLds L;
    extern __shared__ __attribute__((aligned(16))) unsigned char hot_smem[];
    HotLds &H = *reinterpret_cast<HotLds *>(hot_smem); // only touched when HOT
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t np = 1u << log_np, pmask = np - 1;
    const uint32_t pool_base = blockIdx.x * chunks_per_wg;
    scatter_init(L, tid);
    __syncthreads();

    const size_t ntiles = (n + P1_TILE - 1) / P1_TILE;
    const size_t npairs = (n + 1) / 2;
    const pd2_t *vp = reinterpret_cast<const pd2_t *>(v);
    typedef IdStream<IDT> IS;
    const IS ip(ids);
    constexpr int NPAIR = P1_SPT / 2;
    typename IS::raw_t idv[NPAIR];
    pd2_t val[NPAIR];
    // software pipeline: the loads of tile t+1 are issued right after tile t's samples have been
    // consumed, so their latency hides behind tile t's scan / LDS sort / copy-out phases.
    // Lane layout: pair index = tile*2048 + j*P1_BLOCK + tid, 16 B of values + 8 B of ids per lane.
    auto load_tile = [&](size_t tile) {
        const size_t pbase = tile * (P1_TILE / 2);
#pragma unroll
        for (int j = 0; j < NPAIR; j++) {
            const size_t i = pbase + (size_t)j * P1_BLOCK + tid;
            if (tile < ntiles && i < npairs) {
                // the last pair of an odd-length stream reads one element past n inside the same
                // 16-byte granule; it is masked below
                idv[j] = ip.ld_nt(i);
                val[j] = __builtin_nontemporal_load(vp + i);
            } else {
                idv[j] = typename IS::raw_t{}; // (pairs beyond the stream: every use checks the sample's index)
                val[j] = (pd2_t){0.0, 0.0};
            }
        }
    };
    load_tile(blockIdx.x);

    if (HOT) {
        // ---- pick the hot names from this workgroup's first tile (it is already in registers).
        // The selection table lives in L.sorted, which the scatter does not use before the first tile.
        uint32_t *tag = L.sorted, *cnt = L.sorted + HOT_TAGS;
        for (uint32_t i = tid; i < 2 * HOT_TAGS; i += P1_BLOCK) L.sorted[i] = 0;
        for (uint32_t i = tid; i < HOT_MAPW; i += P1_BLOCK) H.hmap[i] = HOT_EMPTY;
        for (uint32_t i = tid; i < HOT_NAMES * HOT_W; i += P1_BLOCK) H.win[i] = 0;
        if (tid < HOT_NAMES) { H.name[tid] = INVALID; H.mn[tid] = INVALID; H.mx[tid] = 0; H.org[tid] = 0; }
        if (tid == 0) H.nsel = 0;
        __syncthreads();
        const size_t pbase0 = (size_t)blockIdx.x * (P1_TILE / 2);
        const bool exact = nmetrics <= HOT_TAGS;
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            const size_t i = 2 * (pbase0 + (size_t)(j >> 1) * P1_BLOCK + tid) + (j & 1);
            const uint32_t id = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
            if (i < n && id < nmetrics) {
                const uint32_t h = exact ? id : (id * 2654435761u) >> 21;
                const uint32_t old = atomicCAS(&tag[h], 0u, id + 1u);
                if (old == 0u || old == id + 1u) atomicAdd(&cnt[h], 1u);
            }
        }
        __syncthreads();
        // thread t owns entries t, t + 512, ...: (count << 11 | entry) packs into 24 bits (count <= 4 096)
        constexpr int OWN = HOT_TAGS / P1_BLOCK;
        uint32_t own[OWN];
#pragma unroll
        for (int k = 0; k < OWN; k++) {
            const uint32_t e = tid + (uint32_t)k * P1_BLOCK;
            own[k] = cnt[e] ? (cnt[e] << 11) | e : 0u;
        }
        for (uint32_t r = 0; r < HOT_ROUNDS; r++) {
            uint32_t best = 0;
#pragma unroll
            for (int k = 0; k < OWN; k++) best = own[k] > best ? own[k] : best;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const uint32_t o = __shfl_xor(best, d, 64);
                best = o > best ? o : best;
            }
            if (lane == 0) H.wmax[wave] = best;
            __syncthreads();
            if (tid == 0) {
                uint32_t b = 0;
                for (int w = 0; w < P1_BLOCK / 64; w++) b = H.wmax[w] > b ? H.wmax[w] : b;
                H.pick = b;
                if (b) {
                    const uint32_t id = tag[b & (HOT_TAGS - 1)] - 1u, m = id & (HOT_MAPW - 1), slot = H.nsel;
                    if (H.hmap[m] == HOT_EMPTY) { // else: skipped, the next candidate gets its turn
                        H.hmap[m] = (id << 8) | slot;
                        H.name[slot] = id;
                        H.nsel = slot + 1;
                    }
                }
            }
            __syncthreads();
            const uint32_t b = H.pick;
            if (b == 0 || H.nsel == HOT_NAMES) break; // uniform: no candidates left, or all windows taken
#pragma unroll
            for (int k = 0; k < OWN; k++)
                if (own[k] == b) own[k] = 0;
        }
        __syncthreads();
        // window origins: centre of the hot name's sampled bins (as k_part_hist does)
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            const size_t i = 2 * (pbase0 + (size_t)(j >> 1) * P1_BLOCK + tid) + (j & 1);
            const uint32_t id = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
            const double x = (j & 1) ? val[j >> 1].y : val[j >> 1].x;
            if (i < n && id < nmetrics) {
                const uint32_t e = H.hmap[id & (HOT_MAPW - 1)], hs = e & 0xffu;
                if ((e >> 8) == id) {
                    const uint32_t bin = lh_bin_of(x, Tx);
                    if (bin < H.mn[hs]) atomicMin(&H.mn[hs], bin);
                    if (bin > H.mx[hs]) atomicMax(&H.mx[hs], bin);
                }
            }
        }
        __syncthreads();
        if (tid < HOT_NAMES) {
            uint32_t org = 32768u - HOT_W / 2;
            if (H.mn[tid] != INVALID) {
                const uint32_t centre = (H.mn[tid] + H.mx[tid] + 1) >> 1;
                org = centre > HOT_W / 2 ? centre - HOT_W / 2 : 0u;
            }
            if (org > 65536u - HOT_W) org = 65536u - HOT_W;
            H.org[tid] = org;
            H.mn[tid] = INVALID; // reused as the flush ranges
            H.mx[tid] = 0;
        }
        __syncthreads();
    }

    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t pbase = tile * (P1_TILE / 2);
        // every tile but the last is full: no per-sample bounds arithmetic there (workgroup-uniform branch,
        // two copies of the classification code)
        const bool full_tile = (tile + 1) * (size_t)P1_TILE <= n;
        uint32_t rec[P1_SPT], pr[P1_SPT];
        auto classify = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int j = 0; j < P1_SPT; j++) {
                const uint32_t id = (j & 1) ? IS::second(idv[j >> 1]) : IS::first(idv[j >> 1]);
                const double x = (j & 1) ? val[j >> 1].y : val[j >> 1].x;
                pr[j] = INVALID;
                rec[j] = 0;
                bool live = true; // (load_tile pads pairs beyond the stream: not an error)
                if (!FULL) live = 2 * (pbase + (size_t)(j >> 1) * P1_BLOCK + tid) + (j & 1) < n;
                if (!live) continue;
                if (id >= nmetrics) {
                    atomicOr(err, 1u); // reported by lh_sync / lh_extract
                    continue;
                }
                const uint32_t bin = lh_bin_of(x, Tx);
                if (HOT) {
                    const uint32_t e = H.hmap[id & (HOT_MAPW - 1)], hs = e & 0xffu;
                    if ((e >> 8) == id) {
                        const uint32_t rel = bin - H.org[hs];
                        if (rel < HOT_W) {
                            atomicAdd(&H.win[(hs << HOT_LOGW) + rel], 1u);
                            continue;
                        }
                    }
                }
                const uint32_t p = id & pmask;
                rec[j] = (p << 24) | ((id >> log_np) << 16) | bin;
                pr[j] = p | (atomicAdd(&L.cnt[p], 1u) << 8);
            }
        };
        if (full_tile) classify(std::true_type{});
        else classify(std::false_type{});
        scatter_tile(L, rec, pr, records, cdesc, pool_base, 0u, 0u, tid,
                     [&] { load_tile(tile + gridDim.x); });
    }
    scatter_drain(L, records, cdesc, pool_base, np, 0u, 0u, tid);

    if (HOT) {
        __syncthreads();
        // flush the hot windows: one uint64 atomic per occupied bin
        for (uint32_t i = tid; i < HOT_NAMES * HOT_W; i += P1_BLOCK) {
            const uint32_t c = H.win[i];
            if (c) {
                const uint32_t hs = i >> HOT_LOGW, b = H.org[hs] + (i & (HOT_W - 1));
                lh::cell_add(counts, (size_t)H.name[hs] * LH_ROW_STRIDE + b, c);
                atomicMin(&H.mn[hs], b);
                atomicMax(&H.mx[hs], b);
            }
        }
        __syncthreads();
        if (tid < HOT_NAMES && H.mn[tid] != INVALID) {
            uint32_t *r = ranges + 2 * (size_t)H.name[tid];
            if (H.mn[tid] < r[0]) atomicMin(&r[0], H.mn[tid]);
            if (H.mx[tid] > r[1]) atomicMax(&r[1], H.mx[tid]);
        }
    }
}

// ---------------------------------------------------------------------------
// P1b: second partition level.  One workgroup per level-1 work slot re-scatters that slot's records by
// the next log_ns bits of the local name id.  Output partition q = sub << log_np | p, so that the name
// is again (local >> log_ns) << log_nq | q, i.e. P2 works unchanged with log_nq in place of log_np.
// (Hot-name windows were tried here too -- the names of one level-1 partition are skewed as well -- and measured
// slower: 65 536 names 7.47 -> 8.00 ms, one rank's config-4 slice 1.49 -> 1.79 ms.  The per-slot selection and
// the drop to two workgroups per CU cost more than the records saved.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(P1_BLOCK, 6) void k_scatter_records(const uint32_t *__restrict__ in_records,
                                                                 const uint32_t *__restrict__ in_cdesc,
                                                                 const uint32_t *__restrict__ in_sorted,
                                                                 const uint32_t *__restrict__ in_part_start,
                                                                 const uint32_t *__restrict__ in_slots,
                                                                 const uint32_t *__restrict__ in_nslots,
                                                                 const uint32_t *__restrict__ pool_start,
                                                                 uint32_t log_np, uint32_t log_ns,
                                                                 uint32_t *__restrict__ records,
                                                                 uint32_t *__restrict__ cdesc)
{
    __shared__ __attribute__((aligned(16))) ScatterLds L;
    const uint32_t slot = blockIdx.x;
    if (slot >= *in_nslots) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t p1 = in_slots[3 * slot], first = in_slots[3 * slot + 1], cnt = in_slots[3 * slot + 2];
    const uint32_t *list = in_sorted + in_part_start[p1] + first;
    const uint32_t ns = 1u << log_ns, smask = ns - 1;
    const uint32_t pool_base = pool_start[slot];
    scatter_init(L, tid);
    __syncthreads();

    constexpr uint32_t CPT = P1_TILE / CHUNK; // chunks per tile
    const uint32_t ntiles = (cnt + CPT - 1) / CPT;
    uint32_t in[P1_SPT];
    uint32_t okmask = 0; // bit j: in[j] holds a record (a record may legitimately be 0xffffffff: no sentinel)
    // record j of a thread: chunk (tile*CPT + j/2), offset (j&1)*512 + tid
    auto load_tile = [&](uint32_t tile) {
        okmask = 0;
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            const uint32_t ci = tile * CPT + (uint32_t)(j >> 1);
            in[j] = 0;
            if (tile < ntiles && ci < cnt) {
                const uint32_t cid = list[ci];
                const uint32_t cn = in_cdesc[cid] & CD_MASK;
                const uint32_t o = (uint32_t)(j & 1) * P1_BLOCK + tid;
                if (o < cn) {
                    in[j] = __builtin_nontemporal_load(in_records + (size_t)cid * CHUNK + o);
                    okmask |= 1u << j;
                }
            }
        }
    };
    load_tile(0);
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        uint32_t rec[P1_SPT], pr[P1_SPT];
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            pr[j] = INVALID;
            rec[j] = 0;
            if (okmask & (1u << j)) {
                const uint32_t local = (in[j] >> 16) & 0xffu, bin = in[j] & 0xffffu;
                const uint32_t sub = local & smask;
                rec[j] = (sub << 24) | ((local >> log_ns) << 16) | bin;
                pr[j] = sub | (atomicAdd(&L.cnt[sub], 1u) << 8);
            }
        }
        scatter_tile(L, rec, pr, records, cdesc, pool_base, log_np, p1, tid, [&] { load_tile(tile + 1); });
    }
    scatter_drain(L, records, cdesc, pool_base, ns, log_np, p1, tid);
}

// ---------------------------------------------------------------------------
// plan: group chunk descriptors by partition, cut into work slots
// ---------------------------------------------------------------------------
constexpr int PL_BLOCK = 256;
constexpr int PL_PER_WG = 2048; // descriptors per workgroup

__global__ __launch_bounds__(PL_BLOCK) void k_plan_count(const uint32_t *__restrict__ cdesc, uint32_t nchunks,
                                                         uint32_t *__restrict__ pc, uint32_t nq)
{
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_h = s_dyn; // [nq]
    for (uint32_t i = threadIdx.x; i < nq; i += PL_BLOCK) s_h[i] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * PL_PER_WG;
    for (uint32_t i = lo + threadIdx.x; i < lo + PL_PER_WG && i < nchunks; i += PL_BLOCK) {
        const uint32_t d = cdesc[i];
        if (d != INVALID) atomicAdd(&s_h[d >> CD_SHIFT], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nq; i += PL_BLOCK)
        if (s_h[i]) atomicAdd(&pc[i], s_h[i]);
}

constexpr int PS_BLOCK = 1024;

// exclusive block scan over PS_BLOCK threads; returns the exclusive prefix, *total receives the sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t val, uint32_t *s_w /*[17]*/, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = val;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += y;
    }
    __syncthreads(); // s_w reuse
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (uint32_t w = 0; w < PS_BLOCK / 64; w++) {
        if (w < wave) wbase += s_w[w];
        tot += s_w[w];
    }
    *total = tot;
    return wbase + inc - val;
}

// One workgroup.  Thread t owns partitions [t*E, (t+1)*E), E = ceil(nq / PS_BLOCK).
// pool_extra > 0: also lay out private chunk pools for a following scatter level (slot s gets
// cnt + pool_extra chunks starting at pool_start[s]).
__global__ __launch_bounds__(PS_BLOCK) void k_plan_scan(const uint32_t *__restrict__ pc,
                                                        uint32_t *__restrict__ part_start,
                                                        uint32_t *__restrict__ cursor,
                                                        uint32_t *__restrict__ slots,
                                                        uint32_t *__restrict__ nslots,
                                                        uint32_t *__restrict__ pool_start, uint32_t nq,
                                                        uint32_t pool_extra, uint32_t slot_extra)
{
    __shared__ uint32_t s_w[17];
    constexpr uint32_t EMAX = NQMAX / PS_BLOCK;
    const uint32_t E = (nq + PS_BLOCK - 1) / PS_BLOCK;
    const uint32_t q0 = threadIdx.x * E;
    uint32_t c[EMAX];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        c[e] = (e < E && q0 + e < nq) ? pc[q0 + e] : 0u;
        mine += c[e];
    }
    uint32_t total = 0;
    uint32_t run = block_excl_scan(mine, s_w, &total);
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        if (e < E && q0 + e < nq) {
            part_start[q0 + e] = run;
            cursor[q0 + e] = run;
            run += c[e];
        }
    }
    uint32_t target = (total + slot_extra - 1) / slot_extra; // chunks per slot
    if (target == 0) target = 1;
    uint32_t kmine = 0;
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) kmine += (c[e] + target - 1) / target;
    uint32_t nsl = 0;
    uint32_t s0 = block_excl_scan(kmine, s_w, &nsl);
    if (threadIdx.x == 0) *nslots = nsl; // <= total/target + nq <= slot_extra + nq
#pragma unroll
    for (uint32_t e = 0; e < EMAX; e++) {
        const uint32_t k = (c[e] + target - 1) / target;
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t first = j * target;
            slots[3 * (s0 + j) + 0] = q0 + e;
            slots[3 * (s0 + j) + 1] = first;
            slots[3 * (s0 + j) + 2] = (c[e] - first) < target ? (c[e] - first) : target;
        }
        s0 += k;
    }
    if (pool_extra) {
        __threadfence_block();
        __syncthreads();
        uint32_t carry = 0; // prefix of (chunks + pool_extra) over the slots, PS_BLOCK slots per round
        for (uint32_t base = 0; base < nsl; base += PS_BLOCK) {
            const uint32_t sidx = base + threadIdx.x;
            const uint32_t val = sidx < nsl ? slots[3 * sidx + 2] + pool_extra : 0u;
            uint32_t round_total = 0;
            const uint32_t excl = block_excl_scan(val, s_w, &round_total);
            if (sidx < nsl) pool_start[sidx] = carry + excl;
            carry += round_total;
        }
        if (threadIdx.x == 0) pool_start[nsl] = carry;
    }
}

__global__ __launch_bounds__(PL_BLOCK) void k_plan_scatter(const uint32_t *__restrict__ cdesc, uint32_t nchunks,
                                                           uint32_t *__restrict__ cursor,
                                                           uint32_t *__restrict__ sorted, uint32_t nq)
{
    extern __shared__ uint32_t s_dyn[];
    uint32_t *s_h = s_dyn; // [nq]: counts, then bases
    for (uint32_t i = threadIdx.x; i < nq; i += PL_BLOCK) s_h[i] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * PL_PER_WG;
    uint32_t q[PL_PER_WG / PL_BLOCK], rank[PL_PER_WG / PL_BLOCK];
#pragma unroll
    for (int j = 0; j < PL_PER_WG / PL_BLOCK; j++) {
        const uint32_t i = lo + j * PL_BLOCK + threadIdx.x;
        q[j] = INVALID;
        rank[j] = 0;
        if (i < nchunks) {
            const uint32_t d = cdesc[i];
            if (d != INVALID) {
                q[j] = d >> CD_SHIFT;
                rank[j] = atomicAdd(&s_h[q[j]], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nq; i += PL_BLOCK) {
        const uint32_t c = s_h[i];
        if (c) s_h[i] = atomicAdd(&cursor[i], c); // count -> base of this workgroup's run
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PL_PER_WG / PL_BLOCK; j++) {
        if (q[j] != INVALID) sorted[s_h[q[j]] + rank[j]] = lo + j * PL_BLOCK + threadIdx.x;
    }
}

// ---------------------------------------------------------------------------
// P2: per-partition LDS histogram
// ---------------------------------------------------------------------------
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
constexpr size_t P2_LDS_BYTES = (P2_WINWORDS + 3 * PART_MAX_MPP + 2 * OV_SLOTS) * sizeof(uint32_t) + 16;

__device__ __forceinline__ void p2_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                              uint32_t m, uint32_t bin, uint64_t c)
{
    lh::cell_add(counts, (size_t)m * LH_ROW_STRIDE + bin, c);
    uint32_t *r = ranges + 2 * (size_t)m;
    if (bin < r[0]) atomicMin(&r[0], bin);
    if (bin > r[1]) atomicMax(&r[1], bin);
}

// log_nq: name id = local << log_nq | partition.  mpp: names per partition.
__global__ __launch_bounds__(P2_BLOCK, 8) void k_part_hist(const uint32_t *__restrict__ records,
                                                           const uint32_t *__restrict__ cdesc,
                                                           const uint32_t *__restrict__ sorted,
                                                           const uint32_t *__restrict__ part_start,
                                                           const uint32_t *__restrict__ slots,
                                                           const uint32_t *__restrict__ nslots, uint32_t log_nq,
                                                           uint32_t mpp, uint32_t log_w,
                                                           uint64_t *__restrict__ counts,
                                                           uint32_t *__restrict__ ranges)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_org = h + P2_WINWORDS;
    uint32_t *s_mn = s_org + PART_MAX_MPP;
    uint32_t *s_mx = s_mn + PART_MAX_MPP;
    uint32_t *ov_key = s_mx + PART_MAX_MPP; // out-of-window records, aggregated per slot (lh_windows.h)
    uint32_t *ov_cnt = ov_key + OV_SLOTS;

    const uint32_t slot = blockIdx.x;
    if (slot >= *nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = slots[3 * slot], first = slots[3 * slot + 1], cnt = slots[3 * slot + 2];
    const uint32_t *list = sorted + part_start[p] + first;
    const uint32_t W = 1u << log_w, words = mpp << log_w;

    // ---- window placement from a sample of the slot: its first chunk (<= 1 024 records = one per thread).
    // The sample's loads are issued first so that their latency hides behind the zeroing of the windows.
    const uint32_t c0 = list[0];
    const uint32_t n0 = cdesc[c0] & CD_MASK;
    const uint32_t srec = tid < n0 ? records[(size_t)c0 * CHUNK + tid] : INVALID;
    for (uint32_t i = tid; i < words; i += P2_BLOCK) h[i] = 0;
    ov_init(ov_key, ov_cnt, tid, P2_BLOCK);
    if (tid < mpp) { s_mn[tid] = INVALID; s_mx[tid] = 0; }
    __syncthreads();
    // Fast path: per-name min / max bin of the sample.  When every name's sampled span fits its window
    // (the usual case: a 4 096-bin window is 41 e-folds wide) the window is centred on the span.
    if (tid < n0) {
        const uint32_t l = (srec >> 16) & 0xffu, b = srec & 0xffffu;
        if (b < s_mn[l]) atomicMin(&s_mn[l], b);
        if (b > s_mx[l]) atomicMax(&s_mx[l], b);
    }
    __syncthreads();
    const bool fits = tid >= mpp || s_mn[tid] == INVALID || s_mx[tid] - s_mn[tid] < W - W / 4;
    if (__syncthreads_and(fits)) {
        if (tid < mpp) {
            uint32_t org = 32768u - W / 2; // name absent from the sample: centre on key 0
            if (s_mn[tid] != INVALID) {
                const uint32_t centre = (s_mn[tid] + s_mx[tid] + 1) >> 1;
                org = centre > W / 2 ? centre - W / 2 : 0u;
            }
            if (org > 65536u - W) org = 65536u - W;
            s_org[tid] = org;
            s_mn[tid] = INVALID; // reused as the flush ranges
            s_mx[tid] = 0;
        }
        __syncthreads();
    } else {
        // Wide or multi-modal sample (lh_windows.h): bucket it coarsely into h[name][bin >> log_cw];
        // choose_windows picks the max-mass run per name.
        const uint32_t log_cw = 16 - log_w;
        if (tid < n0) atomicAdd(&h[(((srec >> 16) & 0xffu) << log_w) + ((srec & 0xffffu) >> log_cw)], 1u);
        __syncthreads();
        choose_windows(h, s_org, s_mn, s_mx, mpp, log_w, wave, lane, P2_BLOCK / 64);
        __syncthreads();
        for (uint32_t i = tid; i < words; i += P2_BLOCK) h[i] = 0;
        __syncthreads();
    }

    // one chunk per wave per iteration: 1 024 records = 4 x (64 lanes x 16 B).  Double-buffered: the
    // next chunk's descriptor and its four 16-B loads are in flight while this chunk is reduced.
    auto load_chunk = [&](uint32_t cidx, u4_t (&dst)[CHUNK / 256]) {
        const u4_t *src = reinterpret_cast<const u4_t *>(records + (size_t)cidx * CHUNK) + lane;
#pragma unroll
        for (uint32_t q = 0; q < CHUNK / 256; q++) dst[q] = __builtin_nontemporal_load(src + q * 64);
    };
    auto reduce_chunk = [&](const u4_t (&r4)[CHUNK / 256], uint32_t cn) {
        const bool full = cn == CHUNK; // wave-uniform
#pragma unroll
        for (uint32_t q = 0; q < CHUNK / 256; q++) {
            const uint32_t rr[4] = {r4[q].x, r4[q].y, r4[q].z, r4[q].w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t rec = rr[t];
                const uint32_t l = (rec >> 16) & 0xffu, b = rec & 0xffffu;
                const uint32_t rel = b - s_org[l];
                if (full) {
                    // constant streams: the whole wave carries one record value -> one lane adds 64
                    const uint32_t f0 = __builtin_amdgcn_readfirstlane(rec);
                    if (__builtin_amdgcn_ballot_w64(rec != f0) == 0ull) {
                        if (lane == 0) {
                            if (rel < W) atomicAdd(&h[(l << log_w) + rel], 64u);
                            else if (!ov_add(ov_key, ov_cnt, (l << 16) | b, 64u))
                                p2_global_add(counts, ranges, (l << log_nq) | p, b, 64);
                        }
                    } else {
                        if (rel < W) atomicAdd(&h[(l << log_w) + rel], 1u);
                        else if (!ov_add(ov_key, ov_cnt, (l << 16) | b, 1u))
                            p2_global_add(counts, ranges, (l << log_nq) | p, b, 1);
                    }
                } else if (q * 256 + lane * 4 + t < cn) {
                    if (rel < W) atomicAdd(&h[(l << log_w) + rel], 1u);
                    else if (!ov_add(ov_key, ov_cnt, (l << 16) | b, 1u))
                        p2_global_add(counts, ranges, (l << log_nq) | p, b, 1);
                }
            }
        }
    };
    constexpr uint32_t WSTEP = P2_BLOCK / 64;
    u4_t bufA[CHUNK / 256], bufB[CHUNK / 256];
    uint32_t j = wave;
    uint32_t cnA = 0, cnB = 0;
    if (j < cnt) {
        const uint32_t cid = __builtin_amdgcn_readfirstlane(list[j]);
        cnA = __builtin_amdgcn_readfirstlane(cdesc[cid] & CD_MASK);
        load_chunk(cid, bufA);
    }
    while (j < cnt) {
        // A holds chunk j; fetch chunk j+WSTEP into B, reduce A
        if (j + WSTEP < cnt) {
            const uint32_t cid = __builtin_amdgcn_readfirstlane(list[j + WSTEP]);
            cnB = __builtin_amdgcn_readfirstlane(cdesc[cid] & CD_MASK);
            load_chunk(cid, bufB);
        }
        reduce_chunk(bufA, cnA);
        j += WSTEP;
        if (j >= cnt) break;
        // B holds chunk j; fetch chunk j+WSTEP into A, reduce B
        if (j + WSTEP < cnt) {
            const uint32_t cid = __builtin_amdgcn_readfirstlane(list[j + WSTEP]);
            cnA = __builtin_amdgcn_readfirstlane(cdesc[cid] & CD_MASK);
            load_chunk(cid, bufA);
        }
        reduce_chunk(bufB, cnB);
        j += WSTEP;
    }
    __syncthreads();

    // flush: one uint64 atomic per occupied cell
    // (a wave's 64 cells are consecutive bins of ONE name when W >= 64: the name's range is updated once per wave, from
    // the first and the last lane that found a count, not once per occupied cell)
    for (uint32_t i = tid; i < words; i += P2_BLOCK) {
        const uint32_t c = h[i];
        const uint32_t l = i >> log_w, b = s_org[l] + (i & (W - 1));
        if (c)
            lh::cell_add(counts, (size_t)((l << log_nq) | p) * LH_ROW_STRIDE + b, c);
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
    for (uint32_t i = tid; i < OV_SLOTS; i += P2_BLOCK)
        if (ov_key[i] != OV_EMPTY)
            p2_global_add(counts, ranges, ((ov_key[i] >> 16) << log_nq) | p, ov_key[i] & 0xffffu, ov_cnt[i]);
    __syncthreads();
    if (tid < mpp && s_mn[tid] != INVALID) {
        uint32_t *r = ranges + 2 * (size_t)((tid << log_nq) | p);
        if (s_mn[tid] < r[0]) atomicMin(&r[0], s_mn[tid]);
        if (s_mx[tid] > r[1]) atomicMax(&r[1], s_mx[tid]);
    }
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------
struct LevelPtrs {
    uint32_t *records, *cdesc, *sorted, *pc, *part_start, *cursor, *slots, *nslots, *pool_start;
};

static LevelPtrs level_ptrs(unsigned char *base, size_t off_rec, size_t off_cd, size_t off_sorted, size_t off_small,
                            uint32_t nq, uint32_t extra)
{
    LevelPtrs L;
    L.records = reinterpret_cast<uint32_t *>(base + off_rec);
    L.cdesc = reinterpret_cast<uint32_t *>(base + off_cd);
    L.sorted = reinterpret_cast<uint32_t *>(base + off_sorted);
    uint32_t *small = reinterpret_cast<uint32_t *>(base + off_small);
    L.pc = small;
    L.part_start = small + nq;
    L.cursor = small + 2 * (size_t)nq;
    L.slots = small + 3 * (size_t)nq;
    L.nslots = L.slots + 3 * (size_t)(nq + extra);
    L.pool_start = L.nslots + 1;
    return L;
}

static hipError_t run_plan(const LevelPtrs &L, uint32_t nchunks, uint32_t nq, uint32_t pool_extra, uint32_t slot_extra,
                           hipStream_t s)
{
    const unsigned grid = (nchunks + PL_PER_WG - 1) / PL_PER_WG;
    const size_t lds = (size_t)nq * sizeof(uint32_t);
    hipLaunchKernelGGL(k_plan_count, dim3(grid), dim3(PL_BLOCK), lds, s, L.cdesc, nchunks, L.pc, nq);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PS_BLOCK), 0, s, L.pc, L.part_start, L.cursor, L.slots, L.nslots,
                       L.pool_start, nq, pool_extra, slot_extra);
    hipLaunchKernelGGL(k_plan_scatter, dim3(grid), dim3(PL_BLOCK), lds, s, L.cdesc, nchunks, L.cursor, L.sorted, nq);
    return hipGetLastError();
}

hipError_t launch_ingest_pairs_part(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                                    uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                                    void *scratch, size_t scratch_bytes, int num_cus, const PartTuning &tune,
                                    hipStream_t s)
{
    PartPlan P;
    if (!make_plan(n, nmetrics, num_cus, tune, P) || scratch_bytes < P.total || !scratch) return hipErrorInvalidValue;
    if (!part_aligned(d_ids, d_v)) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attributes
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_hist),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P2_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_count),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_plan_scatter),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NQMAX * sizeof(uint32_t)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_samples<true, uint32_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)HOT_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter_samples<true, uint16_t>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)HOT_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    unsigned char *base = static_cast<unsigned char *>(scratch);
    const LevelPtrs L1 = level_ptrs(base, P.off_rec1, P.off_cd1, P.off_sorted1, P.off_small1, P.np, P.extra1);


    hipError_t e = hipMemsetAsync(L1.cdesc, 0xff, (size_t)P.nchunks1 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(L1.pc, 0, small_words(P.np, P.extra1) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    auto scatter = [&](auto idp) { // idp: the ids as const uint32_t * or const uint16_t *
        typedef std::remove_cv_t<std::remove_pointer_t<decltype(idp)>> IDT;
        if (P.hot)
            hipLaunchKernelGGL((k_scatter_samples<true, IDT>), dim3(P.g1), dim3(P1_BLOCK), HOT_LDS_BYTES, s, idp, d_v, n,
                               nmetrics, P.log_np, d_Tx, L1.records, L1.cdesc, P.chunks_per_wg, counts, ranges, d_err);
        else
            hipLaunchKernelGGL((k_scatter_samples<false, IDT>), dim3(P.g1), dim3(P1_BLOCK), 0, s, idp, d_v, n, nmetrics,
                               P.log_np, d_Tx, L1.records, L1.cdesc, P.chunks_per_wg, counts, ranges, d_err);
    };
    if (d_ids.width == 2) scatter(d_ids.u16()); else scatter(d_ids.u32());
    e = run_plan(L1, P.nchunks1, P.np, P.log_ns ? P.ns + 1 : 0u, P.extra1, s);
    if (e != hipSuccess) return e;

    const LevelPtrs *last = &L1;
    LevelPtrs L2;
    if (P.log_ns) {
        L2 = level_ptrs(base, P.off_rec2, P.off_cd2, P.off_sorted2, P.off_small2, P.nq, SLOT_EXTRA);
        e = hipMemsetAsync(L2.cdesc, 0xff, (size_t)P.nchunks2 * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(L2.pc, 0, small_words(P.nq, SLOT_EXTRA) * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_scatter_records, dim3(P.np + P.extra1), dim3(P1_BLOCK), 0, s, L1.records, L1.cdesc,
                           L1.sorted, L1.part_start, L1.slots, L1.nslots, L1.pool_start, P.log_np, P.log_ns,
                           L2.records, L2.cdesc);
        e = run_plan(L2, P.nchunks2, P.nq, 0u, SLOT_EXTRA, s);
        if (e != hipSuccess) return e;
        last = &L2;
    }
    hipLaunchKernelGGL(k_part_hist, dim3((P.log_ns ? P.nq : P.np) + SLOT_EXTRA), dim3(P2_BLOCK), P2_LDS_BYTES, s, last->records,
                       last->cdesc, last->sorted, last->part_start, last->slots, last->nslots, P.log_nq, P.mpp2, P.log_w,
                       counts, ranges);
    return hipGetLastError();
}

#include "lh_kernels_part2.h"
#include "lh_kernels_part3.h"

} // namespace lh
