// lh_kernels.hip -- gfx950 kernels of the loghisto hot path.
//
//   K1 k_ingest_single / k_ingest_pairs : compress + fan-in
//        reference: compress (metrics.go:316-322) + Histogram (metrics.go:273-295)
//   K2 k_extract                         : processHistograms + percentile
//        reference: metrics.go:336-387, 391-418
//   K3 k_clear_spans / k_init_ranges     : epoch buffer recycle
//        reference: the `make(map...)` half of the flip, metrics.go:461-462
//   table generation + codec-only kernels for parity tests.
//
// Built with -ffp-contract=off: the Go-math restatements in lh_codec.h must be
// evaluated operation by operation.  Where an FMA is wanted it is explicit.
//
// HBM-bound integer/indexing work: no MFMA anywhere.  The design points are
// 16-B/lane coalesced streaming loads, LDS-private u32 histograms per workgroup,
// a wave-uniform shortcut for constant streams, and one u64 global atomic per
// occupied (workgroup, bin) at flush.
#include "lh_kernels.h"
#include "lh_codec.h"
#include "lh_ids.h"

#include <algorithm>
#include <atomic>

namespace lh {

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Table generation
// ---------------------------------------------------------------------------
__global__ void k_gen_thresholds(double *__restrict__ Tx)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= LH_NTHRESH) return;
    if (j == 0) { Tx[0] = 1.0; return; }
    const uint64_t one_bits = 0x3ff0000000000000ull, max_bits = 0x7fefffffffffffffull;
    if ((int)j > d_kext_golog(__longlong_as_double((long long)max_bits))) {
        Tx[j] = __longlong_as_double(0x7ff0000000000000ll); // +Inf: unreachable
        return;
    }
    // invariant: kext(lo) < j <= kext(hi); kext(1.0) == 0.
    uint64_t lo = one_bits, hi = max_bits;
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (d_kext_golog(__longlong_as_double((long long)mid)) >= (int)j) hi = mid; else lo = mid;
    }
    Tx[j] = __longlong_as_double((long long)hi);
}

__global__ void k_gen_decompress(double *__restrict__ D)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < LH_NKEYS) D[b] = d_decompress_bin(b);
}

hipError_t launch_gen_tables(double *d_Tx, double *d_D, hipStream_t s)
{
    hipLaunchKernelGGL(k_gen_thresholds, dim3((LH_NTHRESH + 255) / 256), dim3(256), 0, s, d_Tx);
    hipLaunchKernelGGL(k_gen_decompress, dim3(LH_NKEYS / 256), dim3(256), 0, s, d_D);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1 single-metric ingest
// ---------------------------------------------------------------------------
#ifndef LH_K1_BLOCK
#define LH_K1_BLOCK 512
#endif
constexpr int K1_BLOCK = LH_K1_BLOCK;              // 8 waves; 2 workgroups per CU
#ifndef LH_K1_UNROLL
#define LH_K1_UNROLL 8   // measured on MI355X: 8 is 3-5 % faster than 4; register prefetch adds nothing (tools/k1_variants.sh)
#endif
constexpr int K1_UNROLL = LH_K1_UNROLL;             // 16-B loads in flight per lane
// LDS: K1_COPIES private histograms of K1_WIN uint32 bins each (64 KiB together).  A lane adds to copy lane % K1_COPIES:
// same-address LDS atomics serialise, and what a wave-instruction costs is the largest number of lanes on ONE word per
// 32-lane group.  Measured on few-valued streams (quantised timers, status codes, queue depths -- the ordinary input
// of TimerToken.Stop, metrics.go:242-246; profiles/r04_fewvalued.jsonl): with one copy, 8 lanes per word and group
// (k = 4 distinct values) are free next to the HBM stream, 11 (k = 3) cost 0.3 ms per 1e9 samples, 16 (k = 2) 0.7 ms.
// Two copies halve every multiplicity (k = 2 becomes k = 4's pattern) for half the window: 8 192 bins, by default the
// keys -4096 .. 4095, i.e. |v| < 6.1e17 -- nanosecond timers up to 19 years, either sign.
//
// Streams that do not live there are still bucketed in LDS:
//  * every workgroup looks at the first three samples of its share; if all three lie beyond the same end of the
//    default window the workgroup centres its main window on their median key instead (|v| ~ 1e30: the whole stream
//    at full speed);
//  * samples outside the main window go to one of two FLOATING windows of K1_OVF bins (one copy each; one for the keys
//    above the main window, one for those below) that the workgroup anchors where its first such sample falls:
//    adjacent to the main window when the sample is within K1_OVF bins of it (a stream a little wider than the
//    window: loguniform[1e-3, 1e18] reaches key 4146), centred on the sample otherwise (a far tail, a second mode);
//  * only what misses these too goes straight to the global row, and that path reads the row's range before it widens
//    it.
// Measured before any of this existed (profiles/r04_k1_lds_counters.jsonl, the first run of tools/r4_counters.sh):
// the 1.2 % of loguniform[1e-3, 1e18] above key 4095 -- 12 M samples on 50 cells plus two unconditional atomics each
// on the row's ONE range pair -- took a 1e9-sample launch from 1.25 ms to 170 ms; now 1.38 ms
// (profiles/r04_k1_wide_streams.txt).
#ifndef LH_K1_COPIES
#define LH_K1_COPIES 2
#endif
constexpr uint32_t K1_COPIES = LH_K1_COPIES;
constexpr uint32_t K1_WIN = 16384 / K1_COPIES;     // bins per copy
// copy k starts K1_PAD words past a multiple of the 32 banks: the same bin of two copies must not share a bank, or
// the copies would serialise on the bank what they no longer serialise on the word
constexpr uint32_t K1_PAD = 16;
constexpr uint32_t K1_STRIDE = K1_WIN + (K1_COPIES > 1 ? K1_PAD : 0);
#ifndef LH_K1_OVF
#define LH_K1_OVF 1024
#endif
constexpr uint32_t K1_OVF = LH_K1_OVF;             // bins of each floating window
constexpr uint32_t K1_OVF_NONE = 0xffffffffu;      // not anchored yet
constexpr uint32_t K1_MAIN_WORDS = K1_STRIDE * K1_COPIES;
// layout: main copies | floating window below | floating window above | [0] min bin [1] max bin of the flush
//         [2] anchor of the window below [3] anchor of the window above [4] first bin of the main window
constexpr uint32_t K1_CTL_WORDS = 8;
constexpr size_t K1_LDS_BYTES = (size_t)(K1_MAIN_WORDS + 2 * K1_OVF + K1_CTL_WORDS) * sizeof(uint32_t);
static_assert(2 * K1_LDS_BYTES + 2048 <= 160 * 1024, "two workgroups per CU");
static_assert(K1_OVF >= 64 && K1_OVF <= K1_WIN, "anchor arithmetic");

// A cell outside every LDS window: straight to the global row, with an atomic that returns nothing, issued from inline asm
// (lh_cells.h); its bin widens the workgroup's flush range in LDS (s_rng = s_ctl[0 .. 1]), which the end of the kernel
// folds into the row's range with everything else.  Until round 6 this path LOOKED at the row's range in HBM first: a
// global load in the middle of the tile loop, for which the compiler drains the tile's loads in flight -- one memory round
// trip per tile and wave that had a miss.  A 0.1 % tail of far outliers (a miss in every tile) cost 1.46 ms per 1e9 samples.
__device__ __forceinline__ void global_cell_add(uint64_t *row, uint32_t *s_rng, uint32_t bin, uint32_t c)
{
    lh::cell_add_hidden(row, bin, c);
    if (bin < s_rng[0]) atomicMin(&s_rng[0], bin);
    if (bin > s_rng[1]) atomicMax(&s_rng[1], bin);
}

// Where a workgroup puts a floating window when `bin` is its first sample on that side of the main window
// [win_lo, + K1_WIN): ADJACENT to the main window when the sample is within K1_OVF bins of it -- the bins right next to
// the main window are the dense ones of a stream slightly wider than it, and a window centred on a first miss 512 ..
// 1 023 bins out would leave a gap of them on the global row -- centred on the sample only beyond that.
__device__ __forceinline__ uint32_t k1_anchor(uint32_t bin, uint32_t win_lo, uint32_t win_len)
{
    const uint32_t centred = bin >= K1_OVF / 2 ? bin - K1_OVF / 2 : 0u, win_hi = win_lo + win_len;
    if (bin >= win_hi) return min(bin < win_hi + K1_OVF ? win_hi : centred, (uint32_t)LH_NKEYS - K1_OVF);
    const uint32_t below = win_lo >= K1_OVF ? win_lo - K1_OVF : 0u;
    return bin >= below ? below : centred;
}

// the sample missed the main window.  h0: the workgroup's LDS block.
__device__ __forceinline__ void k1_miss(uint32_t *h0, uint32_t win_lo, uint32_t win_len, uint64_t *row, uint32_t *s_rng,
                                        uint32_t bin, uint32_t c)
{
    const uint32_t side = bin >= win_lo + win_len ? 1u : 0u;
    uint32_t *anchor = h0 + K1_MAIN_WORDS + 2 * K1_OVF + 2 + side;
    uint32_t lo = __atomic_load_n(anchor, __ATOMIC_RELAXED);
    if (lo == K1_OVF_NONE) {
        const uint32_t want = k1_anchor(bin, win_lo, win_len);
        const uint32_t seen = atomicCAS(anchor, K1_OVF_NONE, want);
        lo = seen == K1_OVF_NONE ? want : seen;
    }
    const uint32_t rel = bin - lo;
    if (rel < K1_OVF) atomicAdd(&h0[K1_MAIN_WORDS + side * K1_OVF + rel], c);
    else global_cell_add(row, s_rng, bin, c);
}

// the workgroup's window state, wave-uniform: h0 = its LDS block, win_lo = first bin of its main window, win_len = its
// length: K1_WIN bins in K1_COPIES copies, or -- WIDE mode -- K1_WIDE bins in ONE copy that takes the copies' room
struct K1Win {
    uint32_t *h0;
    uint32_t win_lo, win_len;
};
constexpr uint32_t K1_WIDE = K1_WIN * K1_COPIES; // 16 384 bins
static_assert(K1_WIDE <= K1_MAIN_WORDS, "the wide window lies in the copies' words");

// hc: this lane's copy of the main window (h0 for every lane in wide mode)
__device__ __forceinline__ void k1_add(const K1Win w, uint32_t *hc, uint64_t *row, uint32_t *s_rng, uint32_t bin, uint32_t c = 1u)
{
    const uint32_t rel = bin - w.win_lo;
    if (rel < w.win_len) atomicAdd(&hc[rel], c);
    else k1_miss(w.h0, w.win_lo, w.win_len, row, s_rng, bin, c);
}

// All 64 lanes active.  A wave whose samples mostly share ONE bucket (a constant stream; a stream dominated by one
// value) aggregates that group before the atomic: if lane 0's bucket is shared by >= K1_AGG_MIN lanes, lane 0 adds
// the group's size, the other lanes of the group add nothing, everybody else adds 1 -- one ds_add with a per-lane
// count, no scalar round trip beyond the ballot.  (Measured and dropped: peeling up to four groups leader by leader --
// its readlane / ballot round trips cost more than the conflicts they removed, k = 4: 3.9 ms per 1e9 samples -- and
// a second group led by the first lane outside the first: constant streams 1.29 -> 1.55 ms.)  Streams with many
// buckets never enter: a lognormal stream's hottest bucket holds 0.4 % of the samples.
#ifndef LH_K1_AGG_MIN
#define LH_K1_AGG_MIN 24
#endif
__device__ __forceinline__ void k1_add_fullwave(const K1Win w, uint32_t *hc, uint64_t *row, uint32_t *s_rng, uint32_t bin)
{
    const uint32_t first = __builtin_amdgcn_readfirstlane(bin);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(bin == first);
    const uint32_t nsame = (uint32_t)__builtin_popcountll(same);
    if (nsame >= LH_K1_AGG_MIN) { // wave-uniform
        const bool leader = __lane_id() == 0;
        if (leader || bin != first) k1_add(w, hc, row, s_rng, bin, leader ? nsame : 1u);
    } else {
        k1_add(w, hc, row, s_rng, bin);
    }
}

__global__ __launch_bounds__(K1_BLOCK) void k_ingest_single(const double *__restrict__ v, size_t n,
                                                            uint64_t *__restrict__ row,
                                                            uint32_t *__restrict__ range,
                                                            const double *__restrict__ Tx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h0 = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_ctl = h0 + K1_MAIN_WORDS + 2 * K1_OVF; // [0] min bin, [1] max bin of the flush, [2] [3] floating anchors, [4] win_lo
    const uint32_t tid = threadIdx.x;

    // 16-B alignment: at most one scalar head sample, then pairs, then an odd tail.
    const size_t head = (((uintptr_t)v & 8) && n) ? 1 : 0;
    const d2_t *vp = reinterpret_cast<const d2_t *>(v + head);
    const size_t npair = (n - head) / 2;
    const size_t tile = (size_t)K1_BLOCK * K1_UNROLL; // pairs per workgroup iteration
    const size_t nfull = npair / tile;

    for (uint32_t i = tid; i < K1_MAIN_WORDS + 2 * K1_OVF; i += K1_BLOCK) h0[i] = 0;
    // The main window follows the stream (round 6; rounds 4 - 5 looked at three samples and only moved the window when all
    // three lay beyond the same end of the default one): wave 0 buckets 64 samples spread over the workgroup's first tile.
    //   * their span fits a copy with room to spare (<= 7/8 K1_WIN bins): two copies of K1_WIN bins centred on the span --
    //     every stream of the sweep but the two below, wherever it lives;
    //   * it does not (a stream over 21 decades of one sign ends a few bins past the default window; one of both signs
    //     to +-1e20 spans 9 211 bins): WIDE mode, ONE copy of 16 384 bins centred on the span.  The second copy is there
    //     for few-valued streams (same-address atomics); a stream this wide has no such lanes to separate.
    // What still falls outside goes to the floating windows as before.  loguniform[1e-3, 1e18] and +-10^U(-3, 20) ran at
    // 1.58 ms per 1e9 samples (0.63 of peak) through the floating windows' branch; profiles/r06_k1_wide.txt.
    if (tid < 64) {
        const size_t i0 = min(head + (size_t)blockIdx.x * tile * 2 + (size_t)tid * (tile * 2 / 64), n - 1);
        const uint32_t b = lh_bin_of(v[i0], Tx);
        uint32_t lowest = b, highest = b;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lowest = min(lowest, (uint32_t)__shfl_xor(lowest, d, 64));
            highest = max(highest, (uint32_t)__shfl_xor(highest, d, 64));
        }
        if (tid == 0) {
            s_ctl[0] = 0xffffffffu;
            s_ctl[1] = 0;
            s_ctl[2] = s_ctl[3] = K1_OVF_NONE;
            const uint32_t len = highest - lowest < K1_WIN - K1_WIN / 8 ? K1_WIN : K1_WIDE;
            const uint32_t mid = lowest + (highest - lowest) / 2;
            s_ctl[4] = min(mid >= len / 2 ? mid - len / 2 : 0u, (uint32_t)LH_NKEYS - len);
            s_ctl[5] = len;
        }
    }
    __syncthreads();
    const K1Win w = {h0, (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[4]),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[5])};
    uint32_t *h = h0 + (w.win_len == K1_WIN ? (tid % K1_COPIES) * K1_STRIDE : 0u); // this lane's copy

    // (register double buffering -- tile t + grid in flight while tile t is bucketed -- measured slower,
    // profiles/r02_k1_variants.txt: two workgroups per CU already overlap each other's loads; a rolling refill of the
    // same eight registers -- slot u reloaded as soon as it has been consumed -- and 1 024-thread workgroups likewise,
    // profiles/r04_fewvalued.jsonl.  Keeping the out-of-window handling out of the sixteen unrolled slots -- a first
    // pass that only notes misses, a second over the same registers when any lane had one -- was no faster on streams
    // without misses and up to 1.6 x slower on streams with them: profiles/r04_k1_wide_streams.txt.)
    for (size_t t = blockIdx.x; t < nfull; t += gridDim.x) {
        const d2_t *p = vp + t * tile + tid;
        d2_t r[K1_UNROLL];
#pragma unroll
        for (int u = 0; u < K1_UNROLL; u++) r[u] = __builtin_nontemporal_load(p + u * K1_BLOCK);
#pragma unroll
        for (int u = 0; u < K1_UNROLL; u++) {
            k1_add_fullwave(w, h, row, s_ctl, lh_bin_of(r[u].x, Tx));
            k1_add_fullwave(w, h, row, s_ctl, lh_bin_of(r[u].y, Tx));
        }
    }
    // remainder pairs (guarded), owned by the workgroup next in the rotation
    if (blockIdx.x == nfull % gridDim.x) {
        for (size_t i = nfull * tile + tid; i < npair; i += K1_BLOCK) {
            const d2_t r = vp[i];
            k1_add(w, h, row, s_ctl, lh_bin_of(r.x, Tx));
            k1_add(w, h, row, s_ctl, lh_bin_of(r.y, Tx));
        }
        if (tid == 0 && head) k1_add(w, h, row, s_ctl, lh_bin_of(v[0], Tx));
        if (tid == 1 && ((n - head) & 1)) k1_add(w, h, row, s_ctl, lh_bin_of(v[n - 1], Tx));
    }
    __syncthreads();

    // flush: one u64 atomic per occupied bin (the copies summed), then the floating windows that were anchored
    uint32_t lmin = 0xffffffffu, lmax = 0;
    for (uint32_t i = tid; i < w.win_len; i += K1_BLOCK) {
        uint32_t c = h0[i];
        if (w.win_len == K1_WIN) { // (wave-uniform; the wide window is one array)
#pragma unroll
            for (uint32_t k = 1; k < K1_COPIES; k++) c += h0[k * K1_STRIDE + i];
        }
        if (c) {
            lh::cell_add(row, w.win_lo + i, c);
            lmin = min(lmin, w.win_lo + i);
            lmax = max(lmax, w.win_lo + i);
        }
    }
    for (uint32_t side = 0; side < 2; side++) {
        const uint32_t ovf_lo = s_ctl[2 + side];
        if (ovf_lo == K1_OVF_NONE) continue;
        for (uint32_t i = tid; i < K1_OVF; i += K1_BLOCK) {
            const uint32_t c = h0[K1_MAIN_WORDS + side * K1_OVF + i];
            if (c) {
                lh::cell_add(row, ovf_lo + i, c);
                lmin = min(lmin, ovf_lo + i);
                lmax = max(lmax, ovf_lo + i);
            }
        }
    }
    if (lmin != 0xffffffffu) { atomicMin(&s_ctl[0], lmin); atomicMax(&s_ctl[1], lmax); }
    __syncthreads();
    if (tid == 0 && s_ctl[0] != 0xffffffffu) {
        atomicMin(&range[0], s_ctl[0]);
        atomicMax(&range[1], s_ctl[1]);
    }
}

hipError_t launch_ingest_single(const double *d_v, size_t n, uint64_t *row, uint32_t *range,
                                const double *d_Tx, int num_cus, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attribute
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ingest_single),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)K1_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t tile_samples = (size_t)K1_BLOCK * K1_UNROLL * 2;
    size_t want = (n + tile_samples - 1) / tile_samples;
    size_t cap = (size_t)num_cus * 2;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_ingest_single, dim3(grid), dim3(K1_BLOCK), K1_LDS_BYTES, s, d_v, n, row, range, d_Tx);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1 mixed (id, value) ingest -- v1: straight global u64 atomics.
// ---------------------------------------------------------------------------
constexpr int KP_BLOCK = 256;

__device__ __forceinline__ void kp_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                       uint32_t nmetrics, uint32_t *__restrict__ err, uint32_t id,
                                       double v, const double *__restrict__ Tx)
{
    if (id >= nmetrics) { atomicOr(err, 1u); return; }
    const uint32_t bin = lh_bin_of(v, Tx);
    lh::cell_add(counts, (size_t)id * LH_ROW_STRIDE + bin, 1ull);
    // ranges only widen: a stale read can cost a redundant atomic, never miss one
    uint32_t *r = ranges + 2 * (size_t)id;
    if (bin < r[0]) atomicMin(&r[0], bin);
    if (bin > r[1]) atomicMax(&r[1], bin);
}

template <typename IDT>
__global__ __launch_bounds__(KP_BLOCK) void k_ingest_pairs(const IDT *__restrict__ ids,
                                                           const double *__restrict__ v, size_t n,
                                                           uint64_t *__restrict__ counts,
                                                           uint32_t *__restrict__ ranges, uint32_t nmetrics,
                                                           const double *__restrict__ Tx,
                                                           uint32_t *__restrict__ err, int vec)
{
    const size_t gtid = (size_t)blockIdx.x * KP_BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * KP_BLOCK;
    if (vec) {
        const size_t npair = n / 2;
        const d2_t *vp = reinterpret_cast<const d2_t *>(v);
        const IdStream<IDT> ip(ids);
        for (size_t i = gtid; i < npair; i += gsz) {
            const d2_t r = __builtin_nontemporal_load(vp + i);
            const typename IdStream<IDT>::raw_t m = ip.ld_nt(i);
            kp_add(counts, ranges, nmetrics, err, IdStream<IDT>::first(m), r.x, Tx);
            kp_add(counts, ranges, nmetrics, err, IdStream<IDT>::second(m), r.y, Tx);
        }
        if (gtid == 0 && (n & 1)) kp_add(counts, ranges, nmetrics, err, ids[n - 1], v[n - 1], Tx);
    } else {
        for (size_t i = gtid; i < n; i += gsz) kp_add(counts, ranges, nmetrics, err, ids[i], v[i], Tx);
    }
}

hipError_t launch_ingest_pairs(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                               uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                               int num_cus, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const int vec = (((uintptr_t)d_v & 15) == 0 && d_ids.pair_aligned()) ? 1 : 0;
    size_t want = (n / 2 + KP_BLOCK - 1) / KP_BLOCK;
    size_t cap = (size_t)num_cus * 8;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    if (grid == 0) grid = 1;
    if (d_ids.width == 2)
        hipLaunchKernelGGL(k_ingest_pairs<uint16_t>, dim3(grid), dim3(KP_BLOCK), 0, s, d_ids.u16(), d_v, n, counts, ranges,
                           nmetrics, d_Tx, d_err, vec);
    else
        hipLaunchKernelGGL(k_ingest_pairs<uint32_t>, dim3(grid), dim3(KP_BLOCK), 0, s, d_ids.u32(), d_v, n, counts, ranges,
                           nmetrics, d_Tx, d_err, vec);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Counters: counterCache[name] += amount (metrics.go:251-269) for a batch of (id, amount) events.
// cur[id] collects the interval's amounts (the reference's Rates, metrics.go:430-433); flag[id] records that
// the name was touched this interval (Counter(name, 0) still creates the entry and exports a rate of 0).
// Up to CNT_LDS_NAMES counters are first summed in LDS (uint64 LDS atomics), one global atomic per touched
// name per workgroup; beyond that every event is one global atomic.
// ---------------------------------------------------------------------------
constexpr int CNT_BLOCK = 256;
constexpr uint32_t CNT_LDS_NAMES = 4096;

__global__ __launch_bounds__(CNT_BLOCK) void k_count_add(const uint32_t *__restrict__ ids,
                                                         const unsigned long long *__restrict__ amounts, size_t n,
                                                         unsigned long long *__restrict__ cur,
                                                         uint32_t *__restrict__ flag, uint32_t ncounters,
                                                         uint32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char cnt_smem[];
    unsigned long long *s_sum = reinterpret_cast<unsigned long long *>(cnt_smem);
    uint32_t *s_flag = reinterpret_cast<uint32_t *>(s_sum + CNT_LDS_NAMES);
    const bool lds = ncounters <= CNT_LDS_NAMES;
    if (lds) {
        for (uint32_t i = threadIdx.x; i < ncounters; i += CNT_BLOCK) { s_sum[i] = 0; s_flag[i] = 0; }
        __syncthreads();
    }
    const size_t gsz = (size_t)gridDim.x * CNT_BLOCK;
    for (size_t i = (size_t)blockIdx.x * CNT_BLOCK + threadIdx.x; i < n; i += gsz) {
        const uint32_t id = ids[i];
        if (id >= ncounters) { atomicOr(err, 1u); continue; }
        const unsigned long long a = amounts[i];
        if (lds) {
            atomicAdd(&s_sum[id], a);
            s_flag[id] = 1u;
        } else {
            atomicAdd(&cur[id], a);
            if (!flag[id]) atomicOr(&flag[id], 1u);
        }
    }
    if (lds) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < ncounters; i += CNT_BLOCK) {
            if (s_flag[i]) {
                if (s_sum[i]) atomicAdd(&cur[i], s_sum[i]);
                if (!flag[i]) atomicOr(&flag[i], 1u);
            }
        }
    }
}

// The fold at the epoch boundary (metrics.go:435-458): counterStore[name] += the interval's amount for every
// name touched this interval; flag bit 1 = the name exists in the store.  Applied once per snapshot.
__global__ void k_count_fold(const unsigned long long *__restrict__ cur, const uint32_t *__restrict__ flag,
                             unsigned long long *__restrict__ life, uint32_t *__restrict__ known, uint32_t ncounters)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncounters && flag[i]) {
        life[i] += cur[i];
        known[i] = 1u;
    }
}

hipError_t launch_count_add(const uint32_t *d_ids, const uint64_t *d_amounts, size_t n, uint64_t *cur, uint32_t *flag,
                            uint32_t ncounters, uint32_t *d_err, int num_cus, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    const size_t lds = ncounters <= CNT_LDS_NAMES ? (size_t)CNT_LDS_NAMES * 12 : 0;
    size_t want = (n + CNT_BLOCK * 8 - 1) / (CNT_BLOCK * 8);
    const size_t cap = (size_t)num_cus * 4;
    const unsigned grid = (unsigned)std::max<size_t>(1, std::min(want, cap));
    hipLaunchKernelGGL(k_count_add, dim3(grid), dim3(CNT_BLOCK), lds, s, d_ids,
                       reinterpret_cast<const unsigned long long *>(d_amounts), n,
                       reinterpret_cast<unsigned long long *>(cur), flag, ncounters, d_err);
    return hipGetLastError();
}

hipError_t launch_count_fold(const uint64_t *cur, const uint32_t *flag, uint64_t *life, uint32_t *known,
                             uint32_t ncounters, hipStream_t s)
{
    if (!ncounters) return hipSuccess;
    hipLaunchKernelGGL(k_count_fold, dim3((ncounters + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const unsigned long long *>(cur), flag,
                       reinterpret_cast<unsigned long long *>(life), known, ncounters);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K2 extract
// ---------------------------------------------------------------------------
constexpr int K2_BLOCK = 256;
constexpr int K2_WAVES = K2_BLOCK / 64;
constexpr int K2_PER_THREAD = 4;
constexpr int K2_TILE = K2_BLOCK * K2_PER_THREAD;
constexpr int K2_MAXP = 32;

// uint64(float64) as Go compiles it for amd64 (metrics.go:374).
__device__ inline uint64_t d_f64_to_u64_amd64(double f)
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

__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t x, int d)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl_up(lo, d, 64);
    hi = __shfl_up(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t x, int d)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl_xor(lo, d, 64);
    hi = __shfl_xor(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t x, int d)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl_down(lo, d, 64);
    hi = __shfl_down(hi, d, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t x, int src)
{
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_f64(double x, int src)
{
    return __longlong_as_double((long long)shfl_u64((uint64_t)__double_as_longlong(x), src));
}
__device__ __forceinline__ double shfl_down_f64(double x, int d)
{
    return __longlong_as_double((long long)shfl_down_u64((uint64_t)__double_as_longlong(x), d));
}

struct PctArgs { double p[K2_MAXP]; }; // percentile list by value: no H2D copy on the extract path

template <typename CELL> // the store's cell: uint64_t, or uint32_t (lh_cells.h)
__global__ __launch_bounds__(K2_BLOCK) void k_extract(const CELL *__restrict__ counts,
                                                      const uint32_t *__restrict__ ranges,
                                                      const PctArgs pa, uint32_t np,
                                                      const double *__restrict__ D,
                                                      ExtractOut *__restrict__ out,
                                                      double *__restrict__ pvals, int16_t *__restrict__ pkeys,
                                                      uint8_t *__restrict__ pvalid,
                                                      const uint32_t *__restrict__ err_in,
                                                      uint32_t *__restrict__ err_out, const ExtractNotify nt,
                                                      const ExtractCompact cx)
{
    __shared__ uint64_t s_cnt[K2_WAVES];
    __shared__ double s_sum[K2_WAVES];
    __shared__ uint32_t s_nb[K2_WAVES];
    __shared__ uint64_t s_wtot[K2_WAVES];
    __shared__ uint32_t s_found[K2_MAXP];
    __shared__ double s_p[K2_MAXP];

    const uint32_t m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];

    if (tid < K2_MAXP) {
        s_found[tid] = 0xffffffffu;
        s_p[tid] = tid < np ? pa.p[tid] : 2.0;
    }
    if (blockIdx.x == 0 && tid == 0) { // sticky bad-id flag + small-kernel fallback counter ride along with the results
        err_out[0] = err_in[0];
        err_out[1] = err_in[1];
    }

    // ---- pass 1: totalCount, totalSum, occupied buckets (metrics.go:342-347)
    uint64_t cnt = 0;
    double sum = 0;
    uint32_t nb = 0;
    if (lo <= hi) {
        for (uint32_t b = lo + tid; b <= hi; b += K2_BLOCK) {
            const uint64_t c = row[b];
            if (c) {
                cnt += c;
                sum += D[b] * (double)c; // value * float64(*count), metrics.go:344
                nb++;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        cnt += shfl_down_u64(cnt, d);
        sum += shfl_down_f64(sum, d);
        nb += __shfl_down(nb, d, 64);
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_sum[wave] = sum; s_nb[wave] = nb; }
    __syncthreads();
    uint64_t total = 0;
    double tsum = 0;
    uint32_t tnb = 0;
#pragma unroll
    for (int w = 0; w < K2_WAVES; w++) { total += s_cnt[w]; tsum += s_sum[w]; tnb += s_nb[w]; }

    if (tid == 0) {
        if (cx.count) { // compact form: the host derives avg, uint64(sum) and present
            cx.count[m] = total;
            cx.sum[m] = tsum;
            cx.nbuckets[m] = tnb;
        } else {
            ExtractOut o;
            o.count = total;
            o.sum = tsum;
            o.avg = tsum / (double)total; // metrics.go:356 (0/0 = NaN when empty)
            o.agg_sum_add = d_f64_to_u64_amd64(tsum);
            o.nbuckets = tnb;
            o.present = total ? 1u : 0u;
            out[m] = o;
        }
    }

    // ---- pass 2: percentile (metrics.go:406-418) as a prefix scan in bin order
    if (total && np) {
        const double ftotal = (double)total;
        uint64_t carry = 0;
        for (uint32_t base = lo; base <= hi; base += K2_TILE) {
            const uint32_t b0 = base + tid * K2_PER_THREAD;
            uint64_t c[K2_PER_THREAD];
#pragma unroll
            for (int k = 0; k < K2_PER_THREAD; k++) c[k] = (b0 + k <= hi) ? row[b0 + k] : 0;
            uint64_t tsumc = 0;
#pragma unroll
            for (int k = 0; k < K2_PER_THREAD; k++) tsumc += c[k];
            // inclusive wave scan of thread totals
            uint64_t inc = tsumc;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint64_t y = shfl_up_u64(inc, d);
                if ((int)lane >= d) inc += y;
            }
            __syncthreads(); // s_wtot reuse across tiles
            if (lane == 63) s_wtot[wave] = inc;
            __syncthreads();
            uint64_t wbase = 0, tile_total = 0;
#pragma unroll
            for (int w = 0; w < K2_WAVES; w++) {
                if (w < (int)wave) wbase += s_wtot[w];
                tile_total += s_wtot[w];
            }
            uint64_t sofar = carry + wbase + (inc - tsumc);
            double q[K2_PER_THREAD]; // float64(sofar)/float64(totalCount), metrics.go:413; -1 for empty buckets
#pragma unroll
            for (int k = 0; k < K2_PER_THREAD; k++) {
                sofar += c[k];
                q[k] = c[k] ? (double)sofar / ftotal : -1.0;
            }
            // q is non-decreasing in bin order and bins ascend with the lane id, so the
            // answer for percentile i inside this wave is the first lane that has a hit:
            // at most one LDS atomic per (wave, percentile).
            for (uint32_t i = 0; i < np; i++) {
                if (s_found[i] < base) continue; // settled by an earlier tile (uniform branch)
                const double pi = s_p[i];
                uint32_t hit = 0xffffffffu;
#pragma unroll
                for (int k = K2_PER_THREAD - 1; k >= 0; k--)
                    if (q[k] >= pi && q[k] >= 0.0) hit = b0 + k;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit != 0xffffffffu);
                if (mask && lane == (uint32_t)__builtin_ctzll(mask)) atomicMin(&s_found[i], hit);
            }
            carry += tile_total;
        }
    }
    __syncthreads();
    if (cx.count) {
        if (wave == 0) { // (np <= 32: the percentiles' threads are lanes of wave 0)
            const uint32_t fb = tid < np ? s_found[tid] : 0xffffffffu;
            const unsigned long long ok = __builtin_amdgcn_ballot_w64(fb != 0xffffffffu);
            if (tid < np) cx.pkeys[(size_t)m * np + tid] = fb != 0xffffffffu ? (int16_t)bin_to_key(fb) : (int16_t)0;
            if (tid == 0) cx.vbits[m] = (uint32_t)ok;
        }
    } else if (tid < np) {
        const uint32_t fb = s_found[tid];
        const size_t o = (size_t)m * np + tid;
        if (fb != 0xffffffffu) {
            pvals[o] = D[fb];
            pkeys[o] = (int16_t)bin_to_key(fb);
            pvalid[o] = 1;
        } else {
            pvals[o] = 0;
            pkeys[o] = 0;
            pvalid[o] = 0;
        }
    }
    if (nt.host_flag) { // zero-copy results: tell the host when the last workgroup is done
        __threadfence_system();
        __syncthreads();
        if (tid == 0 && atomicAdd(nt.done_ctr, 1u) == gridDim.x - 1) {
            atomicExch(nt.done_ctr, 0u);
            __threadfence_system();
            __hip_atomic_store(nt.host_flag, nt.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------
// Wave-level arithmetic that stays in the VALU (DPP): k_extract_wave's scans and reductions.  __shfl_up / __shfl_down
// compile to ds_bpermute_b32 -- a round trip through the LDS crossbar each, two per 64-bit value -- and round 4's
// wave kernel issued 216 of them per name.
// ---------------------------------------------------------------------------
#define LH_DPP32(x, ctrl, rows) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rows), 0xf, false))
template <int CTRL, int ROWS> __device__ __forceinline__ uint64_t dpp_u64(uint64_t x)
{
    const uint32_t lo = LH_DPP32((uint32_t)x, CTRL, ROWS), hi = LH_DPP32((uint32_t)(x >> 32), CTRL, ROWS);
    return ((uint64_t)hi << 32) | lo; // lanes without a source (or outside ROWS) get 0
}
// inclusive prefix sum over the 64 lanes: four steps inside the rows of 16 lanes (row_shr:1/2/4/8), then lane 15 of
// rows 0 and 2 into rows 1 and 3 (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ uint64_t wave_scan_incl_u64(uint64_t x)
{
    x += dpp_u64<0x111, 0xf>(x);
    x += dpp_u64<0x112, 0xf>(x);
    x += dpp_u64<0x114, 0xf>(x);
    x += dpp_u64<0x118, 0xf>(x);
    x += dpp_u64<0x142, 0xa>(x);
    x += dpp_u64<0x143, 0xc>(x);
    return x;
}
__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t x)
{
    x += LH_DPP32(x, 0x111, 0xf);
    x += LH_DPP32(x, 0x112, 0xf);
    x += LH_DPP32(x, 0x114, 0xf);
    x += LH_DPP32(x, 0x118, 0xf);
    x += LH_DPP32(x, 0x142, 0xa);
    x += LH_DPP32(x, 0x143, 0xc);
    return x;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, uint32_t src) // src wave-uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double x, uint32_t src)
{
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(x), src));
}
// x of lane + D for the lanes whose lane + D is in the same row of 16 (what the others get does not matter to the caller)
template <int D> __device__ __forceinline__ double row_down_f64(double x)
{
    return __longlong_as_double((long long)dpp_u64<0x100 + D, 0xf>((uint64_t)__double_as_longlong(x))); // row_shl:D
}

// metrics.go:413 as an INTEGER threshold: float64(sofar) / float64(total) >= p is monotone in sofar, so there is a
// smallest prefix count T in [1, total] that reaches percentile p, and "the first bucket that reaches p" is the first
// bin whose inclusive prefix is >= T (that bin is occupied: the prefix moves there).  One to three IEEE divides per
// (name, percentile) -- by the lane that owns the percentile -- instead of one per cell.  ~0: no prefix reaches p
// (p > 1 or NaN: the key is omitted, metrics.go:417).
__device__ __forceinline__ bool pct_reached(uint64_t s, double ft, double p) { return (double)s / ft >= p; }
__device__ inline uint64_t pct_threshold(double p, uint64_t total)
{
    if (!(1.0 >= p)) return ~0ull; // the largest quotient is float64(total) / float64(total) == 1
    if (p <= 0.0) return 1;        // the first occupied bucket
    const double ft = (double)total, est = p * ft;
    uint64_t s = est >= 18446744073709549568.0 ? total : (uint64_t)est;
    if ((double)s < est) s++; // ceil(p * total): the threshold itself unless a rounding went the other way
    s = s < 1 ? 1 : (s > total ? total : s);
    if (pct_reached(s, ft, p) && (s == 1 || !pct_reached(s - 1, ft, p))) return s; // two divides: the usual case
#pragma unroll 1
    for (int it = 0; it < 4 && s > 1 && pct_reached(s - 1, ft, p); it++) s--;
#pragma unroll 1
    for (int it = 0; it < 4 && s < total && !pct_reached(s, ft, p); it++) s++;
    if (pct_reached(s, ft, p) && (s == 1 || !pct_reached(s - 1, ft, p))) return s;
    // not settled in four steps either way (totals beyond 2^53, where float64(s) moves in steps): bisection;
    // reached(total) holds
    uint64_t lo = 0, hi = total;
#pragma unroll 1
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (pct_reached(mid, ft, p)) hi = mid; else lo = mid;
    }
    return hi;
}

// 16 bytes at an 8-byte-aligned address as ONE load (global_load_dwordx4; unaligned vector access is on for HSA)
struct __attribute__((packed, aligned(8))) u64x2_a8 { uint64_t a, b; };
struct __attribute__((packed, aligned(8))) f64x2_a8 { double a, b; };
struct __attribute__((packed, aligned(4))) u32x4_a4 { uint32_t a, b, c, d; };

// What k_extract_wave does with a span it holds in registers (32-bit cells, a total below 2^32): pass 1 of
// processHistograms, the prefix scan, the percentile search.
__device__ __forceinline__ uint32_t ew_scan(uint32_t x) { return wave_scan_incl_u32(x); }
__device__ __forceinline__ uint32_t ew_readlane(uint32_t x, uint32_t src) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)src); }

typedef uint32_t CT; // (the cells' type in the registers)
__device__ __forceinline__ void ew_reduce_and_search(const CT (&creg)[4][K2_PER_THREAD], const double *__restrict__ D,
                                                     uint32_t lo, uint32_t hi, uint32_t lane, const PctArgs &pa, uint32_t np,
                                                     uint64_t &total_out, double &tsum, uint32_t &tnb, uint32_t &found)
{
    constexpr uint32_t STEPS = 4;
    static_assert(STEPS == 4, "the percentile search compares a threshold with three step boundaries");
    // the decompressed values of the lane's bins: requested here, step by step, and consumed at once -- held beside
    // the cells they cost 32 registers and two waves per SIMD (the table is 512 KiB: L2 at worst)
    double dreg[STEPS][K2_PER_THREAD];
#pragma unroll
    for (uint32_t s = 0; s < STEPS; s++) {
        const uint32_t b0 = lo + s * K2_BLOCK + lane * K2_PER_THREAD;
        f64x2_a8 d01 = {0, 0}, d23 = {0, 0};
        if (b0 <= hi && b0 + K2_PER_THREAD <= LH_ROW_STRIDE) {
            const f64x2_a8 *dp = reinterpret_cast<const f64x2_a8 *>(D + b0);
            d01 = dp[0];
            d23 = dp[1];
        }
        dreg[s][0] = d01.a;
        dreg[s][1] = d01.b;
        dreg[s][2] = d23.a;
        dreg[s][3] = d23.b;
    }
    // pass 1 (metrics.go:342-347) from the registers: psum[k] is k_extract's thread 4 * lane + k.  An empty cell
    // adds +-0 (its table entry is finite), which leaves a partial sum that is not -0 as it is -- and none is: they
    // start at +0 and no product of an occupied cell is -0.
    uint32_t nb = 0; // occupied buckets of the span: counted on the scalar unit, a ballot per (step, k)
    double psum[K2_PER_THREAD] = {0, 0, 0, 0};
    CT stepc[STEPS]; // the lane's four cells of a step, summed
#pragma unroll
    for (uint32_t s = 0; s < STEPS; s++) {
        stepc[s] = 0;
#pragma unroll
        for (int k = 0; k < K2_PER_THREAD; k++) {
            const CT c = creg[s][k];
            stepc[s] += c;
            psum[k] += dreg[s][k] * (double)c; // value * float64(*count), metrics.go:344
            nb += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c != 0));
        }
    }
    // inclusive prefix of the lanes' step sums (kept for the percentile search) and the steps' totals
    CT inc[STEPS], stot[STEPS], total = 0;
#pragma unroll
    for (uint32_t s = 0; s < STEPS; s++) {
        inc[s] = ew_scan(stepc[s]);
        stot[s] = ew_readlane(inc[s], 63);
        total += stot[s];
    }
    total_out = total;
    tnb = nb;
    // k_extract's tree over the 64 threads of each of its four waves (threads 64 w .. 64 w + 63 live in lanes
    // 16 w .. 16 w + 15: one DPP row): distances 32, 16, 8, 4 threads are 8, 4, 2, 1 lanes; 2 and 1 stay inside the lane
#pragma unroll
    for (int k = 0; k < K2_PER_THREAD; k++) {
        psum[k] += row_down_f64<8>(psum[k]);
        psum[k] += row_down_f64<4>(psum[k]);
        psum[k] += row_down_f64<2>(psum[k]);
        psum[k] += row_down_f64<1>(psum[k]);
    }
    psum[0] += psum[2];
    psum[1] += psum[3];
    psum[0] += psum[1];
#pragma unroll
    for (int w = 0; w < K2_WAVES; w++) tsum += readlane_f64(psum[0], 16 * w); // ((w0 + w1) + w2) + w3, as k_extract

    // ---- pass 2 (metrics.go:406-418): lane i < np owns percentile i; T = the prefix count that reaches it
    if (total && np) {
        uint64_t T64 = ~0ull;
        if (lane < np) T64 = pct_threshold(pa.p[lane], (uint64_t)total);
        const bool has = T64 != ~0ull; // (np <= 32: lanes 32 .. 63 hold no percentile)
        const CT T = (CT)T64;          // (a threshold that exists is <= total)
        // the step a percentile ends in: the first whose running total reaches its threshold (one compare per boundary for
        // all percentiles at once; a step then visits only its own)
        const CT c1 = stot[0], c2 = c1 + stot[1], c3 = c2 + stot[2];
        const uint32_t mystep = (T > c1 ? 1u : 0u) + (T > c2 ? 1u : 0u) + (T > c3 ? 1u : 0u);
        CT carry = 0;
#pragma unroll
        for (uint32_t s = 0; s < STEPS; s++) {
            uint32_t todo = (uint32_t)__builtin_amdgcn_ballot_w64(has && mystep == s); // wave-uniform
            if (todo) {
                CT pre[K2_PER_THREAD];                  // inclusive prefix at the lane's four bins
                CT sofar = carry + (inc[s] - stepc[s]);
#pragma unroll
                for (int k = 0; k < K2_PER_THREAD; k++) {
                    sofar += creg[s][k];
                    pre[k] = sofar;
                }
                for (; todo; todo &= todo - 1) {
                    const uint32_t i = (uint32_t)__builtin_ctz(todo);
                    const CT Ti = ew_readlane(T, i);
                    // the first lane whose last bin reaches Ti (lane 63's does), and how many of its bins stay below
                    const unsigned long long reach = __builtin_amdgcn_ballot_w64(pre[K2_PER_THREAD - 1] >= Ti);
                    const uint32_t f = (uint32_t)__builtin_ctzll(reach);
                    const uint32_t below = (pre[0] < Ti ? 1u : 0u) + (pre[1] < Ti ? 1u : 0u) + (pre[2] < Ti ? 1u : 0u);
                    const uint32_t bin = lo + s * K2_BLOCK + f * K2_PER_THREAD +
                                         (uint32_t)__builtin_amdgcn_readlane((int)below, (int)f);
                    if (lane == i) found = bin;
                }
            }
            carry += stot[s];
        }
    }
}

// The same, one WAVE per metric (four metrics per workgroup, no workgroup barriers): for thousands of names with
// narrow spans the block-per-metric form is bound by workgroup dispatch and its six barriers, not by the scan
// (65 536 names: ~290 us).  Results are BIT-IDENTICAL to k_extract, _sum included: the wave forms the partial sums of
// the 256 k_extract threads it stands for (thread t accumulates bins lo + t, lo + t + 256, ... in that order), reduces
// them with the same tree per 64 threads and adds the four wave totals in the same order; counts and the percentile
// search are integer arithmetic.
//
// Spans of at most EW_REG bins (1 024: every window of the third generation's default width) are read ONCE: the row's
// cells stay in registers between the count / sum pass and the prefix scan, and the decompress table entries are
// requested with the cells instead of after them (round 3 read every window twice and took three dependent round
// trips per name: 139 us for 65 536 names).  Layout: lane L holds bins lo + 256 s + 4 L + k (s < 4, k < 4), i.e. the
// k_extract threads t = 4 L + k.  Wider spans take the two-pass loop.
//
// Round 5 (65 536 names: 126 - 178 us for a 73 us read of the windows).  The in-register path was bound by what it did
// with the cells, not by reading them: with the arithmetic removed the kernel takes 57 us, and every instruction of a
// wave costs four cycles of its SIMD (65 536 waves on 1 024 SIMDs: 0.1 us per instruction) -- 32 eight-byte loads, a
// branch per cell, an IEEE divide per cell (metrics.go:413 evaluated for every bucket), 216 ds_bpermute per name.
// Now: 16-byte loads per 4-bin group, only by the lanes that have a bin inside the span; 32-BIT cells in the registers
// (a span with a cell of 2^22 or more takes the two-pass loop); the table entries requested after the cells and
// consumed at once; branch-free accumulation (adding the +-0 of an empty cell leaves every partial sum as it is);
// occupied buckets counted by the scalar unit (a ballot per register); every scan and reduction in DPP; and the
// percentiles as integer thresholds (pct_threshold): two divides per (name, percentile) by the lane that owns it, then
// one ballot and three compares.  110 -> 96 us on a snapshot read repeatedly, 176 -> 128 us inside config 4's step
// (profiles/r05_extract_wave.txt).
constexpr uint32_t EW_STEPS = 4, EW_REG = EW_STEPS * K2_BLOCK;

static_assert(LH_ROW_STRIDE >= LH_NKEYS + 4, "k_extract_wave reads whole 4-bin groups inside the row's stride");
// (the decompress table is allocated LH_ROW_STRIDE entries long, zeros behind LH_NKEYS: lh_engine.cc)

template <typename CELL>
__global__ __launch_bounds__(K2_BLOCK) void k_extract_wave(const CELL *__restrict__ counts,
                                                           const uint32_t *__restrict__ ranges, uint32_t nmetrics,
                                                           const PctArgs pa, uint32_t np,
                                                           const double *__restrict__ D,
                                                           ExtractOut *__restrict__ out,
                                                           double *__restrict__ pvals, int16_t *__restrict__ pkeys,
                                                           uint8_t *__restrict__ pvalid,
                                                           const uint32_t *__restrict__ err_in,
                                                           uint32_t *__restrict__ err_out, const ExtractCompact cx)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t m = blockIdx.x * K2_WAVES + wave;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        err_out[0] = err_in[0];
        err_out[1] = err_in[1];
    }
    if (m >= nmetrics) return; // wave-uniform
    const CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    bool inreg = lo <= hi && hi - lo < EW_REG; // wave-uniform

    uint64_t total = 0;
    double tsum = 0;
    uint32_t tnb = 0;
    uint32_t found = 0xffffffffu; // lane i < np: the bin of percentile i
    uint32_t c32[EW_STEPS][K2_PER_THREAD];
    if (inreg) {
        // ---- one read of the span, 4 consecutive bins per lane and step.  Cells outside [lo, hi] are zero (the range
        // covers every cell ever written: it is all the clear kernels clear), so a 4-bin group that straddles hi needs
        // no mask.  The span stays in registers as 32-BIT cells when every cell is below 2^22 (its total then fits 32
        // bits: any interval short of ~10^9 samples in one name); a span with a larger cell takes the two-pass loop.
        uint32_t hibits = 0; // OR of every cell's bits 22 .. 63, folded into one word
#pragma unroll
        for (uint32_t s = 0; s < EW_STEPS; s++) {
            const uint32_t b0 = lo + s * K2_BLOCK + lane * K2_PER_THREAD;
            // a lane whose four bins all lie beyond hi asks for nothing: a 600-bin window is three steps = 768 bins wide,
            // and the lanes past its end were 22 % of the kernel's reads
            const bool mine = b0 <= hi && b0 + K2_PER_THREAD <= LH_ROW_STRIDE;
            if constexpr (sizeof(CELL) == 4) { // a narrow store: the four cells are ONE 16-byte load
                u32x4_a4 c = {0, 0, 0, 0};
                if (mine) c = *reinterpret_cast<const u32x4_a4 *>(row + b0);
                c32[s][0] = c.a;
                c32[s][1] = c.b;
                c32[s][2] = c.c;
                c32[s][3] = c.d;
            } else {
                u64x2_a8 c01 = {0, 0}, c23 = {0, 0};
                if (mine) {
                    const u64x2_a8 *rp = reinterpret_cast<const u64x2_a8 *>(row + b0);
                    c01 = rp[0];
                    c23 = rp[1];
                }
                c32[s][0] = (uint32_t)c01.a;
                c32[s][1] = (uint32_t)c01.b;
                c32[s][2] = (uint32_t)c23.a;
                c32[s][3] = (uint32_t)c23.b;
                hibits |= (uint32_t)(c01.a >> 32) | (uint32_t)(c01.b >> 32) | (uint32_t)(c23.a >> 32) | (uint32_t)(c23.b >> 32);
            }
            hibits |= (c32[s][0] | c32[s][1] | c32[s][2] | c32[s][3]) >> 22;
        }
        inreg = __builtin_amdgcn_ballot_w64(hibits != 0) == 0; // wave-uniform
    }
    if (inreg) {
        ew_reduce_and_search(c32, D, lo, hi, lane, pa, np, total, tsum, tnb, found);
    } else {
        // ---- wide span: every bin read here and again by the scan below
        uint64_t cnt = 0;
        double sum4[K2_WAVES] = {0, 0, 0, 0};
        uint32_t nb = 0;
        if (lo <= hi) {
            for (uint32_t base = lo; base <= hi; base += K2_BLOCK) {
#pragma unroll
                for (int v = 0; v < K2_WAVES; v++) { // virtual thread v * 64 + lane of k_extract
                    const uint32_t b = base + (uint32_t)v * 64 + lane;
                    if (b <= hi) {
                        const uint64_t c = row[b];
                        if (c) {
                            cnt += c;
                            sum4[v] += D[b] * (double)c;
                            nb++;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            cnt += shfl_down_u64(cnt, d);
            nb += __shfl_down(nb, d, 64);
#pragma unroll
            for (int v = 0; v < K2_WAVES; v++) sum4[v] += shfl_down_f64(sum4[v], d);
        }
        total = shfl_u64(cnt, 0);
#pragma unroll
        for (int v = 0; v < K2_WAVES; v++) tsum += shfl_f64(sum4[v], 0); // ((w0 + w1) + w2) + w3, as k_extract
        tnb = __shfl(nb, 0, 64);

        // pass 2, 256 bins per step, as the in-register path does it (round 6; round 4's loop evaluated the quotient of
        // metrics.go:413 with an IEEE divide at every cell and moved `found` and the hits through ds_bpermute): lane
        // i < np owns percentile i and the smallest prefix count T that reaches it (pct_threshold); a step looks only at
        // the percentiles whose threshold falls inside it -- one 64-bit compare per lane and step -- and finds each one's
        // bin with a ballot and three compares.  Cells beyond hi are zero, so a 4-bin group that straddles hi needs no mask.
        if (total && np) {
            uint64_t T64 = ~0ull;
            if (lane < np) T64 = pct_threshold(pa.p[lane], total);
            uint32_t todo = (uint32_t)__builtin_amdgcn_ballot_w64(T64 != ~0ull); // percentiles without a bin yet (np <= 32)
            uint64_t carry = 0;
            for (uint32_t base = lo; base <= hi && todo; base += K2_TILE / K2_WAVES) {
                const uint32_t b0 = base + lane * K2_PER_THREAD;
                uint64_t c[K2_PER_THREAD] = {0, 0, 0, 0};
                if (b0 <= hi && b0 + K2_PER_THREAD <= LH_ROW_STRIDE) {
                    if constexpr (sizeof(CELL) == 4) {
                        const u32x4_a4 q = *reinterpret_cast<const u32x4_a4 *>(row + b0);
                        c[0] = q.a; c[1] = q.b; c[2] = q.c; c[3] = q.d;
                    } else {
                        const u64x2_a8 *rp = reinterpret_cast<const u64x2_a8 *>(row + b0);
                        const u64x2_a8 c01 = rp[0], c23 = rp[1];
                        c[0] = c01.a; c[1] = c01.b; c[2] = c23.a; c[3] = c23.b;
                    }
                }
                const uint64_t tsumc = (c[0] + c[1]) + (c[2] + c[3]);
                const uint64_t inc = wave_scan_incl_u64(tsumc);
                const uint64_t end = carry + readlane_u64(inc, 63);
                // (every open threshold is > carry: it would have ended in an earlier step otherwise)
                uint32_t here = (uint32_t)__builtin_amdgcn_ballot_w64(T64 <= end) & todo; // wave-uniform
                todo &= ~here;
                if (here) {
                    uint64_t pre[K2_PER_THREAD], sofar = carry + (inc - tsumc);
#pragma unroll
                    for (int k = 0; k < K2_PER_THREAD; k++) {
                        sofar += c[k];
                        pre[k] = sofar;
                    }
                    for (; here; here &= here - 1) {
                        const uint32_t i = (uint32_t)__builtin_ctz(here);
                        const uint64_t Ti = readlane_u64(T64, i);
                        // the first lane whose last bin reaches Ti (lane 63's does), and how many of its bins stay below
                        const unsigned long long reach = __builtin_amdgcn_ballot_w64(pre[K2_PER_THREAD - 1] >= Ti);
                        const uint32_t f = (uint32_t)__builtin_ctzll(reach);
                        const uint32_t below = (pre[0] < Ti ? 1u : 0u) + (pre[1] < Ti ? 1u : 0u) + (pre[2] < Ti ? 1u : 0u);
                        const uint32_t bin = base + f * K2_PER_THREAD + (uint32_t)__builtin_amdgcn_readlane((int)below, (int)f);
                        if (lane == i) found = bin;
                    }
                }
                carry = end;
            }
        }
    }
    if (cx.count) { // compact form (wave-uniform): no table gather, no divide, 42 B per name at nine percentiles
        const unsigned long long ok = __builtin_amdgcn_ballot_w64(lane < np && found != 0xffffffffu);
        if (lane == 0) {
            cx.count[m] = total;
            cx.sum[m] = tsum;
            cx.nbuckets[m] = tnb;
            cx.vbits[m] = (uint32_t)ok;
        }
        if (lane < np) cx.pkeys[(size_t)m * np + lane] = found != 0xffffffffu ? (int16_t)bin_to_key(found) : (int16_t)0;
        return;
    }
    if (lane == 0) {
        ExtractOut o;
        o.count = total;
        o.sum = tsum;
        o.avg = tsum / (double)total;
        o.agg_sum_add = d_f64_to_u64_amd64(tsum);
        o.nbuckets = tnb;
        o.present = total ? 1u : 0u;
        out[m] = o;
    }
    if (lane < np) {
        const size_t o = (size_t)m * np + lane;
        if (found != 0xffffffffu) {
            pvals[o] = D[found];
            pkeys[o] = (int16_t)bin_to_key(found);
            pvalid[o] = 1;
        } else {
            pvals[o] = 0;
            pkeys[o] = 0;
            pvalid[o] = 0;
        }
    }
}

hipError_t launch_extract(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics,
                          const double *h_p, uint32_t np, const double *d_D, ExtractOut *out,
                          double *pvals, int16_t *pkeys, uint8_t *pvalid, const uint32_t *err_in,
                          uint32_t *err_out, hipStream_t s, ExtractNotify notify, ExtractCompact compact)
{
    if (nmetrics == 0) return hipSuccess;
    PctArgs pa;
    for (uint32_t i = 0; i < (uint32_t)K2_MAXP; i++) pa.p[i] = i < np ? h_p[i] : 2.0;
    // many names: one wave per metric (bit-identical results; see k_extract_wave)
    const bool narrow = cells_narrow(counts);
    const uint32_t *c32 = reinterpret_cast<const uint32_t *>(cells_base(counts));
    if (nmetrics >= 2048 && !notify.host_flag) {
        const dim3 grid((nmetrics + K2_WAVES - 1) / K2_WAVES);
        if (narrow)
            hipLaunchKernelGGL(k_extract_wave<uint32_t>, grid, dim3(K2_BLOCK), 0, s, c32, ranges, nmetrics, pa, np, d_D, out,
                               pvals, pkeys, pvalid, err_in, err_out, compact);
        else
            hipLaunchKernelGGL(k_extract_wave<uint64_t>, grid, dim3(K2_BLOCK), 0, s, counts, ranges, nmetrics, pa, np, d_D,
                               out, pvals, pkeys, pvalid, err_in, err_out, compact);
        return hipGetLastError();
    }
    if (narrow)
        hipLaunchKernelGGL(k_extract<uint32_t>, dim3(nmetrics), dim3(K2_BLOCK), 0, s, c32, ranges, pa, np, d_D, out, pvals,
                           pkeys, pvalid, err_in, err_out, notify, compact);
    else
        hipLaunchKernelGGL(k_extract<uint64_t>, dim3(nmetrics), dim3(K2_BLOCK), 0, s, counts, ranges, pa, np, d_D, out,
                           pvals, pkeys, pvalid, err_in, err_out, notify, compact);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K5 occupied-cell listing (RawMetricSet.Histograms, metrics.go:54-60): the sparse
// map[int16]*uint64 of every name, as CSR arrays, compacted on the device.
// ---------------------------------------------------------------------------
template <typename CELL>
__global__ __launch_bounds__(256) void k_count_cells(const CELL *__restrict__ counts,
                                                     const uint32_t *__restrict__ ranges,
                                                     uint32_t *__restrict__ ncells)
{
    __shared__ uint32_t s_n;
    const uint32_t m = blockIdx.x;
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t n = 0;
    if (lo <= hi) {
        const CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
        for (uint32_t b = lo + threadIdx.x; b <= hi; b += 256) n += row[b] != 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) ncells[m] = s_n;
}

template <typename CELL>
__global__ __launch_bounds__(256) void k_compact_cells(const CELL *__restrict__ counts,
                                                       const uint32_t *__restrict__ ranges,
                                                       const uint64_t *__restrict__ offsets,
                                                       int16_t *__restrict__ keys, uint64_t *__restrict__ vals)
{
    __shared__ uint32_t s_w[4];
    const uint32_t m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    if (lo > hi) return;
    const CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
    uint64_t base = offsets[m];
    for (uint32_t t0 = lo; t0 <= hi; t0 += 256) { // ascending bin == ascending key
        const uint32_t b = t0 + tid;
        const uint64_t c = b <= hi ? row[b] : 0;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(c != 0);
        const uint32_t below = (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        __syncthreads(); // s_w reuse
        if (lane == 0) s_w[wave] = (uint32_t)__builtin_popcountll(mask);
        __syncthreads();
        uint32_t wbase = 0, tile = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < (int)wave) wbase += s_w[w];
            tile += s_w[w];
        }
        if (c) {
            keys[base + wbase + below] = (int16_t)bin_to_key(b);
            vals[base + wbase + below] = c;
        }
        base += tile;
    }
}

hipError_t launch_count_cells(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics, uint32_t *ncells,
                              hipStream_t s)
{
    if (nmetrics == 0) return hipSuccess;
    if (cells_narrow(counts))
        hipLaunchKernelGGL(k_count_cells<uint32_t>, dim3(nmetrics), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t *>(cells_base(counts)), ranges, ncells);
    else
        hipLaunchKernelGGL(k_count_cells<uint64_t>, dim3(nmetrics), dim3(256), 0, s, counts, ranges, ncells);
    return hipGetLastError();
}

hipError_t launch_compact_cells(const uint64_t *counts, const uint32_t *ranges, uint32_t nmetrics,
                                const uint64_t *offsets, int16_t *keys, uint64_t *vals, hipStream_t s)
{
    if (nmetrics == 0) return hipSuccess;
    if (cells_narrow(counts))
        hipLaunchKernelGGL(k_compact_cells<uint32_t>, dim3(nmetrics), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t *>(cells_base(counts)), ranges, offsets, keys, vals);
    else
        hipLaunchKernelGGL(k_compact_cells<uint64_t>, dim3(nmetrics), dim3(256), 0, s, counts, ranges, offsets, keys, vals);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K4 helpers: pack / unpack for the multi-GPU merge (the collective itself is RCCL, lh_engine.cc)
// ---------------------------------------------------------------------------
// ranges (lo, hi) -> (lo, ~hi): one MIN all-reduce then yields min(lo) and max(hi).  `extra` rides behind the last
// row the same way (~x: the MIN all-reduce returns the complement of the largest x of any rank).
__global__ void k_ranges_flip_hi(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t nrows,
                                 uint32_t with_extra, uint32_t extra)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) { dst[2 * (size_t)i] = src[2 * (size_t)i]; dst[2 * (size_t)i + 1] = ~src[2 * (size_t)i + 1]; }
    if (i == 0 && with_extra) dst[2 * (size_t)nrows] = ~extra;
}

// The same with one more word per row for the narrow wire cells (k_merge_widths): dst[2 * nrows + 1 + r] = ~(the largest
// cell of row r on THIS rank, clipped to 2^32 - 1), so that the one MIN all-reduce also returns every row's largest
// per-rank cell.  One wave per row over the row's own dirty window.  narrow == 0: the words are written as 0 ("unknown":
// the all-reduced maximum reads 2^32 - 1 on EVERY rank, whichever rank switched the narrow cells off) and no cell is read.
template <typename CELL>
__global__ __launch_bounds__(256) void k_merge_prep(uint32_t *__restrict__ dst, const uint32_t *__restrict__ ranges,
                                                    const CELL *__restrict__ counts, uint32_t nrows,
                                                    uint32_t extra, uint32_t narrow)
{
    const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint32_t lo = ranges[2 * (size_t)r], hi = ranges[2 * (size_t)r + 1];
    unsigned long long m = 0;
    if (narrow && lo <= hi) {
        const CELL *src = counts + (size_t)r * LH_ROW_STRIDE + lo;
        for (uint32_t i = lane; i <= hi - lo; i += 64u) m = max(m, (unsigned long long)src[i]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, shfl_xor_u64(m, d));
    }
    if (lane == 0) {
        dst[2 * (size_t)r] = lo;
        dst[2 * (size_t)r + 1] = ~hi;
        const uint32_t clipped = m > 0xffffffffull ? 0xffffffffu : (uint32_t)m;
        dst[2 * (size_t)nrows + 1 + r] = narrow ? ~clipped : 0u;
        if (r == 0) dst[2 * (size_t)nrows] = ~extra;
    }
}

// ---- per-row windows, packed CSR ---------------------------------------------------------------
// After the range merge every rank holds the same [lo_r, hi_r] for every row r.  Only those cells
// travel: row r contributes w_r = hi_r - lo_r + 1 cells (0 when empty), packed back to back.  A single
// outlier sample therefore widens ONE row's window (at most 512 KiB), never the whole matrix
// (one global window made a +1e140 sample cost tens of GiB).
//
// k_merge_plan (one workgroup): P[r] = exclusive prefix of the widths.  The nblocks owner blocks of a
// reduce-scatter are CONTIGUOUS ROW RANGES OF EQUAL PACKED SIZE, not of equal row count: block k starts at the
// first row whose prefix reaches k/nblocks of the total (brow[k]; every rank computes the same boundaries from the
// same merged ranges).  RCCL's reduce-scatter wants equal blocks, so every block is padded to the largest; with
// equal row counts and names ranked by frequency (ids of a Zipf stream) the first block holds the widest windows and
// every other block pays for it, with equal cells the padding is at most one row's window.
// info = {total cells, largest block, widest row, occupied rows, largest per-rank sample count of the interval
// (complement of extra_src[0]), first row / end row of block `rank`}; when host_flag is set it is stored
// system-scope for the host, which needs the counts to size the collective.
constexpr int MP_BLOCK = 1024;
// Step 1 (one workgroup per 1 024 rows, coalesced): P[r] = exclusive prefix of the widths INSIDE the row block,
// btot / bmaxw / bocc [block] = the block's total, widest window and occupied rows.  (One workgroup walking all
// 65 536 rows with 64 rows per thread took 124 us: strided loads, twice.)
__global__ __launch_bounds__(MP_BLOCK) void k_merge_widths(const uint32_t *__restrict__ ranges, uint32_t nrows,
                                                           const uint32_t *__restrict__ extra_src,
                                                           const uint32_t *__restrict__ rowmaxc, uint32_t nranks,
                                                           uint8_t *__restrict__ cls,
                                                           unsigned long long *__restrict__ P,
                                                           unsigned long long *__restrict__ btot,
                                                           unsigned long long *__restrict__ bcells,
                                                           uint32_t *__restrict__ bmaxw, uint32_t *__restrict__ bocc,
                                                           uint32_t *__restrict__ bn8, uint32_t *__restrict__ bn16)
{
    __shared__ uint32_t s_w[MP_BLOCK / 64], s_c[MP_BLOCK / 64], s_maxw[MP_BLOCK / 64], s_occ[MP_BLOCK / 64],
        s_n8[MP_BLOCK / 64], s_n16[MP_BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = blockIdx.x * MP_BLOCK + tid;
    // the wire word: uint32 when no merged cell of the interval can reach 2^32 (nranks x the largest per-rank sample
    // count, which rides behind the ranges), else one uint64 per cell
    const unsigned long long big = extra_src ? (unsigned long long)(~extra_src[0]) : ~0ull;
    const bool words32 = big < 0xffffffffull && big * nranks < (1ull << 32);
    uint32_t w = 0, ww = 0, bits = words32 ? 32u : 64u;
    if (r < nrows) {
        const uint32_t lo = ranges[2 * (size_t)r], hi = ranges[2 * (size_t)r + 1];
        w = lo <= hi ? hi - lo + 1 : 0u;
        if (words32 && rowmaxc) {
            // no merged cell of this row exceeds nranks x (the largest cell any rank holds in it)
            const unsigned long long bound = (unsigned long long)(~rowmaxc[r]) * nranks;
            bits = bound <= 0xffull ? 8u : bound <= 0xffffull ? 16u : 32u;
        }
        ww = bits == 64u ? w : (uint32_t)(((unsigned long long)w * bits + 31u) >> 5);
        if (cls) cls[r] = (uint8_t)bits;
    }
    uint32_t inc = ww, cinc = w, maxw = w, occ = w != 0, n8 = w != 0 && bits == 8u, n16 = w != 0 && bits == 16u;
    // (a block's total is at most 1 024 x 65 536 = 2^26 words or cells)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        maxw = max(maxw, (uint32_t)__shfl_xor(maxw, d, 64));
        occ += __shfl_xor(occ, d, 64);
        cinc += __shfl_xor(cinc, d, 64);
        n8 += __shfl_xor(n8, d, 64);
        n16 += __shfl_xor(n16, d, 64);
    }
    if (lane == 63) s_w[wave] = inc;
    if (lane == 0) { s_maxw[wave] = maxw; s_occ[wave] = occ; s_c[wave] = cinc; s_n8[wave] = n8; s_n16[wave] = n16; }
    __syncthreads();
    uint32_t base = 0, total = 0, ctotal = 0, gmax = 0, gocc = 0, g8 = 0, g16 = 0;
    for (uint32_t k = 0; k < MP_BLOCK / 64; k++) {
        if (k < wave) base += s_w[k];
        total += s_w[k];
        ctotal += s_c[k];
        gmax = max(gmax, s_maxw[k]);
        gocc += s_occ[k];
        g8 += s_n8[k];
        g16 += s_n16[k];
    }
    if (r < nrows) P[r] = base + inc - ww;
    if (tid == 0) {
        btot[blockIdx.x] = total;
        bcells[blockIdx.x] = ctotal;
        bmaxw[blockIdx.x] = gmax;
        bocc[blockIdx.x] = gocc;
        bn8[blockIdx.x] = g8;
        bn16[blockIdx.x] = g16;
    }
}

// Step 2 (one workgroup; at most MP_BLOCK row blocks = 2^20 rows): the row blocks' bases, the totals, and the nblocks
// owner blocks of a reduce-scatter, which are CONTIGUOUS ROW RANGES OF EQUAL PACKED SIZE, not of equal row count:
// block k starts at the first row whose prefix reaches k/nblocks of the total (brow[k]; every rank computes the same
// boundaries from the same merged ranges).  RCCL's reduce-scatter wants equal blocks, so every block is padded to the
// largest; with equal row counts and names ranked by frequency (ids of a Zipf stream) the first block holds the
// widest windows and every other block pays for it (1.20 x on config 4's slice), with equal
// cells the padding is at most one row's window (1.0001 x).
// info = {total cells, largest block, widest row, occupied rows, largest per-rank sample count of the interval
// (complement of extra_src[0]), first row / end row of block `rank`, 0}; when host_flag is set it is stored
// system-scope for the host, which needs the counts to size the collective.  P still holds block-local prefixes:
// k_merge_finish adds bbase.
__global__ __launch_bounds__(MP_BLOCK) void k_merge_plan(uint32_t nrows, uint32_t nblocks, uint32_t rank,
                                                         const uint32_t *__restrict__ extra_src,
                                                         const unsigned long long *__restrict__ P,
                                                         const unsigned long long *__restrict__ btot,
                                                         const uint32_t *__restrict__ bmaxw,
                                                         const uint32_t *__restrict__ bocc,
                                                         const unsigned long long *__restrict__ bcells,
                                                         const uint32_t *__restrict__ bn8,
                                                         const uint32_t *__restrict__ bn16,
                                                         unsigned long long *__restrict__ bbase,
                                                         unsigned long long *__restrict__ bstart /*[nblocks+1]*/,
                                                         uint32_t *__restrict__ brow /*[nblocks+1]*/,
                                                         unsigned long long *__restrict__ info /*[16]*/,
                                                         uint32_t *__restrict__ host_flag, uint32_t seq)
{
    __shared__ unsigned long long s_base[MP_BLOCK], s_wsum[MP_BLOCK / 64], s_csum[MP_BLOCK / 64], s_bmax;
    __shared__ uint32_t s_maxw[MP_BLOCK / 64], s_occ[MP_BLOCK / 64], s_n8[MP_BLOCK / 64], s_n16[MP_BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nrb = (nrows + MP_BLOCK - 1) / MP_BLOCK; // row blocks (<= MP_BLOCK: the launcher checks)
    const unsigned long long mine = tid < nrb ? btot[tid] : 0ull;
    uint32_t maxw = tid < nrb ? bmaxw[tid] : 0u, occ = tid < nrb ? bocc[tid] : 0u;
    uint32_t n8 = tid < nrb ? bn8[tid] : 0u, n16 = tid < nrb ? bn16[tid] : 0u;
    unsigned long long inc = mine, csum = tid < nrb ? bcells[tid] : 0ull;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = shfl_up_u64(inc, d);
        if ((int)lane >= d) inc += y;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        maxw = max(maxw, (uint32_t)__shfl_xor(maxw, d, 64));
        occ += __shfl_xor(occ, d, 64);
        n8 += __shfl_xor(n8, d, 64);
        n16 += __shfl_xor(n16, d, 64);
        csum += shfl_xor_u64(csum, d);
    }
    if (lane == 63) s_wsum[wave] = inc;
    if (lane == 0) { s_maxw[wave] = maxw; s_occ[wave] = occ; s_n8[wave] = n8; s_n16[wave] = n16; s_csum[wave] = csum; }
    if (tid == 0) s_bmax = 0;
    __syncthreads();
    unsigned long long base = 0, total = 0, ctotal = 0;
    uint32_t gmax = 0, gocc = 0, g8 = 0, g16 = 0;
    for (uint32_t w = 0; w < MP_BLOCK / 64; w++) {
        if (w < wave) base += s_wsum[w];
        total += s_wsum[w];
        ctotal += s_csum[w];
        gmax = max(gmax, s_maxw[w]);
        gocc += s_occ[w];
        g8 += s_n8[w];
        g16 += s_n16[w];
    }
    s_base[tid] = base + inc - mine;
    if (tid < nrb) bbase[tid] = base + inc - mine;
    __syncthreads();
    auto prefix = [&](uint32_t r) { return r >= nrows ? total : P[r] + s_base[r / MP_BLOCK]; };
    // owner-block boundaries: thread k finds the first row whose prefix reaches k * total / nblocks
    for (uint32_t k = tid; k <= nblocks; k += MP_BLOCK) {
        uint32_t row = k == 0 ? 0u : nrows;
        if (k != 0 && k != nblocks) {
            const unsigned long long target = total / nblocks * k + total % nblocks * k / nblocks;
            uint32_t lo = 0, hi = nrows; // smallest r in [0, nrows] with prefix(r) >= target
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (prefix(mid) >= target) hi = mid; else lo = mid + 1;
            }
            row = lo;
        }
        brow[k] = row;
        bstart[k] = prefix(row);
    }
    __syncthreads();
    for (uint32_t k = tid; k < nblocks; k += MP_BLOCK) atomicMax(&s_bmax, bstart[k + 1] - bstart[k]);
    __syncthreads();
    if (tid == 0) {
        info[0] = total;
        info[1] = s_bmax;
        info[2] = gmax;
        info[3] = gocc;
        info[4] = extra_src ? (unsigned long long)(~extra_src[0]) : ~0ull;
        info[5] = brow[min(rank, nblocks)];
        info[6] = brow[min(rank + 1u, nblocks)];
        info[7] = ctotal;
        info[8] = g8;
        info[9] = g16;
        if (host_flag) {
            __threadfence_system();
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Step 3: P[r] becomes the exclusive prefix over ALL rows (what pack / unpack index with); P[nrows] = the total.
__global__ __launch_bounds__(MP_BLOCK) void k_merge_finish(unsigned long long *__restrict__ P,
                                                           const unsigned long long *__restrict__ bbase,
                                                           const unsigned long long *__restrict__ btot, uint32_t nrows)
{
    const uint32_t r = blockIdx.x * MP_BLOCK + threadIdx.x;
    if (r < nrows) P[r] += bbase[blockIdx.x];
    if (r == nrows - 1) P[nrows] = bbase[blockIdx.x] + btot[blockIdx.x];
}

// the owner block of row r: the last k with brow[k] <= r (empty blocks share a boundary with their successor)
__device__ __forceinline__ uint32_t merge_block_of(const uint32_t *__restrict__ brow, uint32_t nblocks, uint32_t r)
{
    uint32_t lo = 0, hi = nblocks; // largest k in [0, nblocks) with brow[k] <= r
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (brow[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// Row r's window <-> buf[k * bstride + (P[r] - bstart[k]) ...], k = the row's owner block.  One workgroup per row.
// WORD is the wire word: uint64_t (one cell per word), or uint32_t when no merged cell of the interval can reach 2^32.
// In uint32 words a row travels at cls[r] = 8, 16 or 32 bits per cell (k_merge_widths: nranks x the largest cell any
// rank holds in the row fits that many bits, so the collective's uint32 SUM never carries from one field of a word
// into the next): cells 4j .. 4j + 3 (8 bits) or 2j, 2j + 1 (16 bits) of the window are word j, low field first.
// Every thread loads / stores ONE cell (coalesced uint64 accesses on the row store's side); the 2 or 4 lanes of a word
// combine their fields with DPP shuffles and the first of them writes it.
// TPR threads per row: 256 (one workgroup per row: few names, wide windows) or 64 (one WAVE per row, four rows per
// workgroup: from 2 048 rows on -- 65 536 windows of ~600 cells are 2.3 cells per thread of a workgroup behind a chain
// of dependent loads (range, owner block, offsets), and the kernels ran at the latency of 32 rounds of such chains:
// pack 131 us + unpack 146 us for 157 MB of words at config 4's name count, round 4).
// four consecutive cells of a row store as vector loads / stores (uint64 cells: two 16-byte accesses at an 8-byte-aligned
// address; uint32 cells: one at a 4-byte-aligned address)
template <typename CELL> __device__ __forceinline__ void cells_load4(const CELL *p, uint64_t (&v)[4])
{
    if constexpr (sizeof(CELL) == 4) {
        const u32x4_a4 q = *reinterpret_cast<const u32x4_a4 *>(p);
        v[0] = q.a; v[1] = q.b; v[2] = q.c; v[3] = q.d;
    } else {
        const u64x2_a8 a = *reinterpret_cast<const u64x2_a8 *>(p), b = *reinterpret_cast<const u64x2_a8 *>(p + 2);
        v[0] = a.a; v[1] = a.b; v[2] = b.a; v[3] = b.b;
    }
}
template <typename CELL> __device__ __forceinline__ void cells_store4(CELL *p, const uint64_t (&v)[4])
{
    if constexpr (sizeof(CELL) == 4) {
        *reinterpret_cast<u32x4_a4 *>(p) = (u32x4_a4){(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]};
    } else {
        *reinterpret_cast<u64x2_a8 *>(p) = (u64x2_a8){v[0], v[1]};
        *reinterpret_cast<u64x2_a8 *>(p + 2) = (u64x2_a8){v[2], v[3]};
    }
}

template <typename WORD, int TPR, typename CELL>
__global__ __launch_bounds__(256) void k_pack_rows(const CELL *__restrict__ counts,
                                                   const uint32_t *__restrict__ ranges,
                                                   const uint8_t *__restrict__ cls,
                                                   const unsigned long long *__restrict__ P,
                                                   const unsigned long long *__restrict__ bstart,
                                                   const uint32_t *__restrict__ brow, uint32_t nblocks,
                                                   unsigned long long bstride, WORD *__restrict__ buf, uint32_t nrows)
{
    static_assert(TPR == 256 || TPR == 64, "a workgroup or a wave per row");
    const uint32_t t = TPR == 256 ? threadIdx.x : (threadIdx.x & 63u);
    uint32_t r = TPR == 256 ? blockIdx.x : blockIdx.x * 4u + (threadIdx.x >> 6);
    if (TPR == 64) r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r); // wave-uniform: the row's scalars by scalar loads
    if (r >= nrows) return;
    const uint32_t lo = ranges[2 * (size_t)r], hi = ranges[2 * (size_t)r + 1];
    if (lo > hi) return;
    const uint32_t k = merge_block_of(brow, nblocks, r);
    WORD *dst = buf + (size_t)k * bstride + (P[r] - bstart[k]);
    const CELL *src = counts + (size_t)r * LH_ROW_STRIDE + lo;
    const uint32_t w = hi - lo + 1;
    // whole-word cells: four consecutive cells per lane and step (16-byte accesses; the row store's side is aligned to
    // its cell, the wire's 4-byte: unaligned vector access is on for HSA), the last partial group cell by cell
    if constexpr (sizeof(WORD) == 8) {
        for (uint32_t i = 4u * t; i < w; i += 4u * TPR) {
            if (i + 4u <= w) {
                uint64_t v[4];
                cells_load4(src + i, v);
                cells_store4(dst + i, v);
            } else {
                for (uint32_t j = i; j < w; j++) dst[j] = src[j];
            }
        }
    } else {
        const uint32_t bits = cls[r];
        if (bits == 32u) {
            for (uint32_t i = 4u * t; i < w; i += 4u * TPR) {
                if (i + 4u <= w) {
                    uint64_t v[4];
                    cells_load4(src + i, v);
                    cells_store4(dst + i, v);
                } else {
                    for (uint32_t j = i; j < w; j++) dst[j] = (uint32_t)src[j];
                }
            }
        } else {
            const uint32_t log_c = bits == 8u ? 2u : 1u, c = 1u << log_c; // cells per word
            const uint32_t wpad = (w + c - 1u) & ~(c - 1u);                // (whole waves run the shuffles: TPR % c == 0)
            for (uint32_t i0 = 0; i0 < wpad; i0 += TPR) {
                const uint32_t i = i0 + t;
                uint32_t f = i < w ? (uint32_t)src[i] << ((i & (c - 1u)) * bits) : 0u;
                f |= __shfl_xor(f, 1, 64);
                if (log_c == 2u) f |= __shfl_xor(f, 2, 64);
                if (i < wpad && (i & (c - 1u)) == 0u) dst[i >> log_c] = f;
            }
        }
    }
}

// Zero the tail of every owner block of the send buffer (blocks are padded to the largest one).
template <typename WORD>
__global__ __launch_bounds__(256) void k_pack_pad(const unsigned long long *__restrict__ bstart, uint32_t nblocks,
                                                  unsigned long long bstride, WORD *__restrict__ buf)
{
    const uint32_t k = blockIdx.y;
    if (k >= nblocks) return;
    const unsigned long long used = bstart[k + 1] - bstart[k];
    WORD *dst = buf + (size_t)k * bstride;
    for (unsigned long long i = used + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < bstride;
         i += (unsigned long long)gridDim.x * 256)
        dst[i] = 0;
}

// buf holds block `kblock` (rows first_row .. first_row + nrows_out) packed from offset 0.
template <typename WORD, int TPR, typename CELL>
__global__ __launch_bounds__(256) void k_unpack_rows(CELL *__restrict__ counts,
                                                     const uint32_t *__restrict__ ranges,
                                                     const uint8_t *__restrict__ cls,
                                                     const unsigned long long *__restrict__ P,
                                                     const unsigned long long *__restrict__ bstart, uint32_t kblock,
                                                     uint32_t first_row, const WORD *__restrict__ buf, uint32_t nrows_out)
{
    static_assert(TPR == 256 || TPR == 64, "a workgroup or a wave per row (k_pack_rows)");
    const uint32_t t = TPR == 256 ? threadIdx.x : (threadIdx.x & 63u);
    uint32_t q = TPR == 256 ? blockIdx.x : blockIdx.x * 4u + (threadIdx.x >> 6);
    if (TPR == 64) q = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (q >= nrows_out) return;
    const uint32_t r = first_row + q;
    const uint32_t lo = ranges[2 * (size_t)r], hi = ranges[2 * (size_t)r + 1];
    if (lo > hi) return;
    const WORD *src = buf + (P[r] - bstart[kblock]);
    CELL *dst = counts + (size_t)r * LH_ROW_STRIDE + lo;
    const uint32_t w = hi - lo + 1;
    if constexpr (sizeof(WORD) == 8) { // (four cells per lane and step, as k_pack_rows)
        for (uint32_t i = 4u * t; i < w; i += 4u * TPR) {
            if (i + 4u <= w) {
                uint64_t v[4];
                cells_load4(src + i, v);
                cells_store4(dst + i, v);
            } else {
                for (uint32_t j = i; j < w; j++) dst[j] = (CELL)src[j];
            }
        }
    } else {
        const uint32_t bits = cls[r];
        if (bits == 32u) {
            for (uint32_t i = 4u * t; i < w; i += 4u * TPR) {
                if (i + 4u <= w) {
                    uint64_t v[4];
                    cells_load4(src + i, v);
                    cells_store4(dst + i, v);
                } else {
                    for (uint32_t j = i; j < w; j++) dst[j] = (CELL)src[j];
                }
            }
        } else {
            const uint32_t log_c = bits == 8u ? 2u : 1u, c = 1u << log_c, mask = (1u << bits) - 1u;
#pragma unroll 4
            for (uint32_t i = t; i < w; i += TPR)
                dst[i] = (CELL)((src[i >> log_c] >> ((i & (c - 1u)) * bits)) & mask);
        }
    }
}

hipError_t launch_merge_prep(uint32_t *dst, const uint32_t *ranges, const uint64_t *counts, uint32_t nrows, uint32_t extra,
                             bool narrow, hipStream_t s)
{
    if (!nrows) return hipSuccess;
    if (cells_narrow(counts))
        hipLaunchKernelGGL(k_merge_prep<uint32_t>, dim3((nrows + 3) / 4), dim3(256), 0, s, dst, ranges,
                           reinterpret_cast<const uint32_t *>(cells_base(counts)), nrows, extra, narrow ? 1u : 0u);
    else
        hipLaunchKernelGGL(k_merge_prep<uint64_t>, dim3((nrows + 3) / 4), dim3(256), 0, s, dst, ranges, counts, nrows, extra,
                           narrow ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_ranges_flip_hi(uint32_t *dst, const uint32_t *src, uint32_t nrows, bool with_extra, uint32_t extra,
                                 hipStream_t s)
{
    if (!nrows) return hipSuccess;
    hipLaunchKernelGGL(k_ranges_flip_hi, dim3((nrows + 255) / 256), dim3(256), 0, s, dst, src, nrows,
                       with_extra ? 1u : 0u, extra);
    return hipGetLastError();
}

hipError_t launch_merge_plan(const uint32_t *ranges, uint32_t nrows, uint32_t nblocks, uint32_t rank, uint32_t nranks,
                             const uint32_t *extra_src, const uint32_t *rowmaxc, uint8_t *cls, uint64_t *P,
                             uint64_t *bstart, uint32_t *brow, uint64_t *work, uint64_t *info, uint32_t *host_flag,
                             uint32_t seq, hipStream_t s)
{
    const uint32_t nrb = (nrows + MP_BLOCK - 1) / MP_BLOCK;
    if (nrows == 0 || nrb > (uint32_t)MP_BLOCK) return hipErrorInvalidValue; // more than 2^20 rows
    // work: btot[1024] | bbase[1024] | bcells[1024] (uint64), bmaxw[1024] | bocc[1024] | bn8[1024] | bn16[1024] (uint32)
    unsigned long long *btot = reinterpret_cast<unsigned long long *>(work), *bbase = btot + MP_BLOCK,
                       *bcells = bbase + MP_BLOCK;
    uint32_t *bmaxw = reinterpret_cast<uint32_t *>(bcells + MP_BLOCK), *bocc = bmaxw + MP_BLOCK, *bn8 = bocc + MP_BLOCK,
             *bn16 = bn8 + MP_BLOCK;
    unsigned long long *Pp = reinterpret_cast<unsigned long long *>(P);
    hipLaunchKernelGGL(k_merge_widths, dim3(nrb), dim3(MP_BLOCK), 0, s, ranges, nrows, extra_src, rowmaxc, nranks, cls, Pp,
                       btot, bcells, bmaxw, bocc, bn8, bn16);
    hipLaunchKernelGGL(k_merge_plan, dim3(1), dim3(MP_BLOCK), 0, s, nrows, nblocks, rank, extra_src, Pp, btot, bmaxw,
                       bocc, bcells, bn8, bn16, bbase, reinterpret_cast<unsigned long long *>(bstart), brow,
                       reinterpret_cast<unsigned long long *>(info), host_flag, seq);
    hipLaunchKernelGGL(k_merge_finish, dim3(nrb), dim3(MP_BLOCK), 0, s, Pp, bbase, btot, nrows);
    return hipGetLastError();
}

// (WORD x rows-per-workgroup x CELL; uint64 wire words carry sums that may not fit a narrow store's cells: the engine widens the
// buffer before it merges on them, lh_engine.cc)
template <typename WORD, typename CELL>
static void pack_rows_t(const CELL *counts, const uint32_t *ranges, const uint8_t *cls, const unsigned long long *Pp,
                        const unsigned long long *bs, const uint32_t *brow, uint32_t nrows, uint32_t nblocks, uint64_t bstride,
                        WORD *buf, hipStream_t s)
{
    const bool wave = nrows >= 2048; // a wave per row (k_extract_wave's and k_clear_rows_wave's switch-over)
    const dim3 grid(wave ? (nrows + 3) / 4 : nrows);
    if (wave)
        hipLaunchKernelGGL((k_pack_rows<WORD, 64, CELL>), grid, dim3(256), 0, s, counts, ranges, cls, Pp, bs, brow, nblocks,
                           (unsigned long long)bstride, buf, nrows);
    else
        hipLaunchKernelGGL((k_pack_rows<WORD, 256, CELL>), grid, dim3(256), 0, s, counts, ranges, cls, Pp, bs, brow, nblocks,
                           (unsigned long long)bstride, buf, nrows);
    if (nblocks > 1)
        hipLaunchKernelGGL(k_pack_pad<WORD>, dim3(64, nblocks), dim3(256), 0, s, bs, nblocks, (unsigned long long)bstride, buf);
}

hipError_t launch_pack_rows(const uint64_t *counts, const uint32_t *ranges, const uint8_t *cls, const uint64_t *P,
                            const uint64_t *bstart, const uint32_t *brow, uint32_t nrows, uint32_t nblocks,
                            uint64_t bstride, void *buf, bool words32, hipStream_t s)
{
    if (!nrows) return hipSuccess;
    const unsigned long long *Pp = reinterpret_cast<const unsigned long long *>(P);
    const unsigned long long *bs = reinterpret_cast<const unsigned long long *>(bstart);
    const bool narrow = cells_narrow(counts);
    const uint32_t *c32 = reinterpret_cast<const uint32_t *>(cells_base(counts));
    if (narrow && !words32) return hipErrorInvalidValue;
    if (words32 && narrow) pack_rows_t(c32, ranges, cls, Pp, bs, brow, nrows, nblocks, bstride, static_cast<uint32_t *>(buf), s);
    else if (words32) pack_rows_t(counts, ranges, cls, Pp, bs, brow, nrows, nblocks, bstride, static_cast<uint32_t *>(buf), s);
    else pack_rows_t(counts, ranges, cls, Pp, bs, brow, nrows, nblocks, bstride, static_cast<uint64_t *>(buf), s);
    return hipGetLastError();
}

template <typename WORD, typename CELL>
static void unpack_rows_t(CELL *counts, const uint32_t *ranges, const uint8_t *cls, const unsigned long long *Pp,
                          const unsigned long long *bs, uint32_t kblock, uint32_t first_row, uint32_t nrows_out,
                          const WORD *buf, hipStream_t s)
{
    const bool wave = nrows_out >= 2048;
    const dim3 grid(wave ? (nrows_out + 3) / 4 : nrows_out);
    if (wave)
        hipLaunchKernelGGL((k_unpack_rows<WORD, 64, CELL>), grid, dim3(256), 0, s, counts, ranges, cls, Pp, bs, kblock,
                           first_row, buf, nrows_out);
    else
        hipLaunchKernelGGL((k_unpack_rows<WORD, 256, CELL>), grid, dim3(256), 0, s, counts, ranges, cls, Pp, bs, kblock,
                           first_row, buf, nrows_out);
}

hipError_t launch_unpack_rows(uint64_t *counts, const uint32_t *ranges, const uint8_t *cls, const uint64_t *P,
                              const uint64_t *bstart, uint32_t kblock, uint32_t first_row, uint32_t nrows_out,
                              const void *buf, bool words32, hipStream_t s)
{
    if (!nrows_out) return hipSuccess;
    const unsigned long long *Pp = reinterpret_cast<const unsigned long long *>(P);
    const unsigned long long *bs = reinterpret_cast<const unsigned long long *>(bstart);
    const bool narrow = cells_narrow(counts);
    uint32_t *c32 = reinterpret_cast<uint32_t *>(cells_base(counts));
    if (narrow && !words32) return hipErrorInvalidValue;
    if (words32 && narrow)
        unpack_rows_t(c32, ranges, cls, Pp, bs, kblock, first_row, nrows_out, static_cast<const uint32_t *>(buf), s);
    else if (words32)
        unpack_rows_t(counts, ranges, cls, Pp, bs, kblock, first_row, nrows_out, static_cast<const uint32_t *>(buf), s);
    else
        unpack_rows_t(counts, ranges, cls, Pp, bs, kblock, first_row, nrows_out, static_cast<const uint64_t *>(buf), s);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K3 clear / ranges
// ---------------------------------------------------------------------------
constexpr int K3_SPLIT = 8; // blocks per row; each owns 8192 bins

template <typename CELL>
__global__ __launch_bounds__(256) void k_clear_spans(CELL *__restrict__ counts,
                                                     const uint32_t *__restrict__ ranges)
{
    const uint32_t m = blockIdx.x;
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    if (lo > hi) return;
    const uint32_t seg = LH_NKEYS / K3_SPLIT;
    const uint32_t a = max(lo, blockIdx.y * seg), b = min(hi, (blockIdx.y + 1) * seg - 1);
    CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
    for (uint32_t i = a + threadIdx.x; i <= b && i >= a; i += 256) row[i] = 0;
}

// Many names with narrow spans: one WAVE per row clears the span and resets the row's range (the block form above
// launches 8 workgroups per row -- 524 288 of them at 65 536 names, 135 us of dispatch for 0.3 GB of stores -- and
// needs k_init_ranges behind it).
template <typename CELL>
__global__ __launch_bounds__(256) void k_clear_rows_wave(CELL *__restrict__ counts, uint32_t *__restrict__ ranges,
                                                         uint32_t nmetrics)
{
    const uint32_t lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= nmetrics) return; // wave-uniform
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    if (lo > hi) return;       // nothing was counted and the range is already empty
    CELL *row = counts + (size_t)m * LH_ROW_STRIDE;
    for (uint32_t i = lo + lane; i <= hi; i += 64) row[i] = 0;
    if (lane == 0) { ranges[2 * (size_t)m] = LH_NKEYS; ranges[2 * (size_t)m + 1] = 0; }
}

// A narrow epoch buffer about to hold 2^32 samples moves to uint64 cells (lh_engine.cc, widen_buffer): every row's
// dirty span is copied into the wide store (zero outside the spans, like the narrow one) and zeroed behind the copy, so
// that the narrow store is clean again for the buffer's next interval.  One wave per row; the ranges stay as they are.
__global__ __launch_bounds__(256) void k_widen_rows(uint32_t *__restrict__ narrow, uint64_t *__restrict__ wide,
                                                    const uint32_t *__restrict__ ranges, uint32_t nmetrics)
{
    const uint32_t lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= nmetrics) return; // wave-uniform
    const uint32_t lo = ranges[2 * (size_t)m], hi = ranges[2 * (size_t)m + 1];
    if (lo > hi) return;
    uint32_t *src = narrow + (size_t)m * LH_ROW_STRIDE;
    uint64_t *dst = wide + (size_t)m * LH_ROW_STRIDE;
    for (uint32_t i = lo + lane; i <= hi; i += 64) {
        dst[i] = src[i];
        src[i] = 0;
    }
}

__global__ void k_init_ranges(uint32_t *__restrict__ ranges, uint32_t nmetrics)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < nmetrics) { ranges[2 * (size_t)m] = LH_NKEYS; ranges[2 * (size_t)m + 1] = 0; }
}

__global__ void k_mark_dirty(uint32_t *__restrict__ ranges, uint32_t first, uint32_t nrows, uint32_t lo, uint32_t hi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows) {
        uint32_t *r = ranges + 2 * (size_t)(first + i);
        r[0] = min(r[0], lo);
        r[1] = max(r[1], hi);
    }
}

hipError_t launch_clear(uint64_t *counts, uint32_t *ranges, uint32_t nmetrics, hipStream_t s)
{
    if (nmetrics == 0) return hipSuccess;
    const bool narrow = cells_narrow(counts);
    uint32_t *c32 = reinterpret_cast<uint32_t *>(cells_base(counts));
    if (nmetrics >= 2048) {
        if (narrow) hipLaunchKernelGGL(k_clear_rows_wave<uint32_t>, dim3((nmetrics + 3) / 4), dim3(256), 0, s, c32, ranges, nmetrics);
        else hipLaunchKernelGGL(k_clear_rows_wave<uint64_t>, dim3((nmetrics + 3) / 4), dim3(256), 0, s, counts, ranges, nmetrics);
        return hipGetLastError();
    }
    if (narrow) hipLaunchKernelGGL(k_clear_spans<uint32_t>, dim3(nmetrics, K3_SPLIT), dim3(256), 0, s, c32, ranges);
    else hipLaunchKernelGGL(k_clear_spans<uint64_t>, dim3(nmetrics, K3_SPLIT), dim3(256), 0, s, counts, ranges);
    hipLaunchKernelGGL(k_init_ranges, dim3((nmetrics + 255) / 256), dim3(256), 0, s, ranges, nmetrics);
    return hipGetLastError();
}

hipError_t launch_widen_rows(uint32_t *narrow, uint64_t *wide, const uint32_t *ranges, uint32_t nmetrics, hipStream_t s)
{
    if (nmetrics == 0) return hipSuccess;
    hipLaunchKernelGGL(k_widen_rows, dim3((nmetrics + 3) / 4), dim3(256), 0, s, narrow, wide, ranges, nmetrics);
    return hipGetLastError();
}

hipError_t launch_init_ranges(uint32_t *ranges, uint32_t nmetrics, hipStream_t s)
{
    if (nmetrics == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_ranges, dim3((nmetrics + 255) / 256), dim3(256), 0, s, ranges, nmetrics);
    return hipGetLastError();
}

hipError_t launch_mark_dirty(uint32_t *ranges, uint32_t first, uint32_t nrows, uint32_t lo, uint32_t hi, hipStream_t s)
{
    if (nrows == 0) return hipSuccess;
    hipLaunchKernelGGL(k_mark_dirty, dim3((nrows + 255) / 256), dim3(256), 0, s, ranges, first, nrows, lo, hi);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Codec-only kernels (parity tests)
// ---------------------------------------------------------------------------
__global__ void k_compress(const double *__restrict__ v, int16_t *__restrict__ keys, size_t n,
                           const double *__restrict__ Tx, int golog)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double val = v[i];
        uint32_t bin;
        if (golog) {
            const double x = 1.0 + fabs(val);
            const uint32_t eb = ((uint32_t)__double2hiint(x)) >> 20;
            const int kext = (eb < 0x7ffu) ? d_kext_golog(x) : 0;
            bin = bin_from_kext(kext, val);
        } else {
            bin = lh_bin_of(val, Tx);
        }
        keys[i] = (int16_t)bin_to_key(bin);
    }
}

hipError_t launch_compress(const double *d_v, int16_t *d_keys, size_t n, const double *d_Tx, bool golog, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    size_t want = (n + 255) / 256;
    unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(k_compress, dim3(grid), dim3(256), 0, s, d_v, d_keys, n, d_Tx, golog ? 1 : 0);
    return hipGetLastError();
}

__global__ void k_vlog_selftest(unsigned long long *__restrict__ maxerr_bits)
{
    const double inv_ln2 = 1.44269504088896338700e+00;
    double worst = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << 23); i += gridDim.x * blockDim.x) {
        const float m = __uint_as_float(0x3f800000u | i);
        const double hw = (double)__builtin_amdgcn_logf(m);
        const double ref = d_go_log((double)m) * inv_ln2;
        const double err = fabs(hw - ref);
        worst = err > worst ? err : worst;
    }
    // non-negative doubles order like their bit patterns
    atomicMax(maxerr_bits, (unsigned long long)__double_as_longlong(worst));
}

hipError_t launch_vlog_selftest(double *d_maxerr, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_maxerr, 0, sizeof(double), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_vlog_selftest, dim3(1024), dim3(256), 0, s, reinterpret_cast<unsigned long long *>(d_maxerr));
    return hipGetLastError();
}

} // namespace lh
