// lh_kernels_small.hip -- mixed (id, value) ingest when the engine has FEW names (<= 32), gfx950.
//
// Reference semantics: Histogram(name, v) = histogramCache[name][compress(v)] += 1
// (metrics.go:273-295, 316-322).  With a handful of names every workgroup can keep all of them in
// LDS, so this is a single streaming pass like K1 (lh_kernels.hip): 12 algorithmic bytes per sample
// and nothing written back but the flush -- no partition pass (lh_kernels_part.hip needs 20 B/sample).
//
// LDS: 16 384 uint32 bins (64 KiB) split evenly: 1 name -> 16 384 bins, 4 -> 4 096, 16 -> 1 024; two
// 512-thread workgroups per CU.  17 .. 32 names: 32 768 bins (128 KiB, again >= 1 024 bins per name), one
// 1 024-thread workgroup per CU (the same 16 waves): 1.96-1.99 ms per 1e9 samples against 3.4 ms through the
// partitioned path.  Not beyond 32: with 512-bin windows a lognormal(sigma = 1) stream loses 1 % of its samples
// to the global-atomic path and 64 names take 3.9 ms, more than the partitioned path with its hot windows.  Each workgroup places its windows from its own first
// tile (lh_windows.h); records outside a window go to global atomics (exact).  Grid-stride over tiles.
#include "lh_kernels.h"
#include "lh_codec.h"
#include "lh_ids.h"

#include <atomic>
#include "lh_windows.h"

namespace lh {

typedef double sd2_t __attribute__((ext_vector_type(2)));
typedef uint32_t su2_t __attribute__((ext_vector_type(2)));

constexpr int KS_UNROLL = 4;                  // (16-B values + 8-B ids) loads in flight per lane
constexpr uint32_t KS_MAXM = 32;
constexpr uint32_t KS_MAXM_SMALL = 16;        // up to here: 64 KiB windows, 512 threads
constexpr size_t ks_lds_bytes(uint32_t words) { return (words + 3 * KS_MAXM + 2 * OV_SLOTS) * sizeof(uint32_t) + 16; }
constexpr size_t KS_MIN_SAMPLES = 65536;

bool small_supported(size_t n, uint32_t nmetrics, Ids d_ids, const double *d_v)
{
    return nmetrics >= 1 && nmetrics <= KS_MAXM && n >= KS_MIN_SAMPLES && (((uintptr_t)d_v & 15) == 0) &&
           d_ids.pair_aligned();
}

__device__ __forceinline__ void ks_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                              uint32_t m, uint32_t bin, uint64_t c)
{
    lh::cell_add(counts, (size_t)m * LH_ROW_STRIDE + bin, c);
    uint32_t *r = ranges + 2 * (size_t)m;
    if (bin < r[0]) atomicMin(&r[0], bin);
    if (bin > r[1]) atomicMax(&r[1], bin);
}

template <uint32_t KS_WORDS, int KS_BLOCK, typename IDT>
__global__ __launch_bounds__(KS_BLOCK) void k_ingest_pairs_small(const IDT *__restrict__ ids,
                                                                 const double *__restrict__ v, size_t n,
                                                                 uint64_t *__restrict__ counts,
                                                                 uint32_t *__restrict__ ranges, uint32_t nmetrics,
                                                                 uint32_t log_w, const double *__restrict__ Tx,
                                                                 uint32_t *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_org = h + KS_WORDS;
    uint32_t *s_mn = s_org + KS_MAXM;
    uint32_t *s_mx = s_mn + KS_MAXM;
    uint32_t *ov_key = s_mx + KS_MAXM; // out-of-window records, aggregated per workgroup (lh_windows.h)
    uint32_t *ov_cnt = ov_key + OV_SLOTS;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t W = 1u << log_w, words = nmetrics << log_w, log_cw = 16 - log_w;

    const size_t npair = n / 2; // the odd tail sample is handled by workgroup 0
    const sd2_t *vp = reinterpret_cast<const sd2_t *>(v);
    const IdStream<IDT> ip(ids);
    typedef IdStream<IDT> IS;
    const size_t tile = (size_t)KS_BLOCK * KS_UNROLL; // pairs per workgroup iteration
    const size_t ntiles = (npair + tile - 1) / tile;

    for (uint32_t i = tid; i < words; i += KS_BLOCK) h[i] = 0;
    __syncthreads();
    // window placement from this workgroup's first tile (<= 4 096 samples)
    if ((size_t)blockIdx.x < ntiles) {
        const size_t base = (size_t)blockIdx.x * tile;
        for (int u = 0; u < KS_UNROLL; u++) {
            const size_t i = base + (size_t)u * KS_BLOCK + tid;
            if (i < npair) {
                const typename IS::raw_t id = ip.ld(i);
                const sd2_t x = vp[i];
                const uint32_t ia = IS::first(id), ib = IS::second(id);
                if (ia < nmetrics) atomicAdd(&h[(ia << log_w) + (lh_bin_of(x.x, Tx) >> log_cw)], 1u);
                if (ib < nmetrics) atomicAdd(&h[(ib << log_w) + (lh_bin_of(x.y, Tx) >> log_cw)], 1u);
            }
        }
    }
    __syncthreads();
    choose_windows(h, s_org, s_mn, s_mx, nmetrics, log_w, wave, lane, KS_BLOCK / 64);
    __syncthreads();
    for (uint32_t i = tid; i < words; i += KS_BLOCK) h[i] = 0;
    ov_init(ov_key, ov_cnt, tid, KS_BLOCK);
    __syncthreads();

    uint32_t nfall = 0; // samples this thread sent to global atomics: feeds the engine's adaptive dispatch
    auto add = [&](uint32_t id, double x, bool fullwave) {
        if (id >= nmetrics) { atomicOr(err, 1u); return; } // reported by lh_sync / lh_extract
        const uint32_t bin = lh_bin_of(x, Tx);
        const uint32_t rel = bin - s_org[id];
        auto put = [&](uint32_t c) {
            if (rel < W) atomicAdd(&h[(id << log_w) + rel], c);
            else if (!ov_add(ov_key, ov_cnt, (id << 16) | bin, c)) { ks_global_add(counts, ranges, id, bin, c); nfall += c; }
        };
        if (fullwave) {
            // few-valued streams: lanes that carry the same (name, bucket) serialise on one LDS word.  As in K1
            // (k1_add_fullwave): when lane 0's cell is shared by >= 16 lanes, lane 0 adds that group's size, the first
            // lane outside it leads a second group the same way, everybody else adds 1 -- one atomic instruction with a
            // per-lane count.  A constant stream is the one-group case.
            const uint32_t key = (id << 16) | bin;
            const uint32_t f0 = __builtin_amdgcn_readfirstlane(key);
            const unsigned long long same = __builtin_amdgcn_ballot_w64(key == f0);
            const uint32_t nsame = (uint32_t)__builtin_popcountll(same);
            if (nsame >= 16u) { // wave-uniform
                const unsigned long long rest = ~same;
                const uint32_t src2 = rest ? (uint32_t)__builtin_ctzll(rest) : 0u;
                const uint32_t leader2 = __builtin_amdgcn_readlane(key, src2);
                const unsigned long long same2 = rest ? __builtin_amdgcn_ballot_w64(key == leader2) : 0ull;
                const bool lead2 = rest && lane == src2;
                const bool grouped = key == f0 || (rest && key == leader2);
                if (!grouped || lane == 0 || lead2)
                    put(lane == 0 ? nsame : lead2 ? (uint32_t)__builtin_popcountll(same2) : 1u);
                return;
            }
        }
        put(1u);
    };

    const size_t nfull = npair / tile;
    for (size_t t = blockIdx.x; t < nfull; t += gridDim.x) {
        const size_t base = t * tile + tid;
        typename IS::raw_t idr[KS_UNROLL];
        sd2_t vr[KS_UNROLL];
#pragma unroll
        for (int u = 0; u < KS_UNROLL; u++) {
            idr[u] = ip.ld_nt(base + (size_t)u * KS_BLOCK);
            vr[u] = __builtin_nontemporal_load(vp + base + (size_t)u * KS_BLOCK);
        }
#pragma unroll
        for (int u = 0; u < KS_UNROLL; u++) {
            // a bad id breaks wave uniformity of the branch inside add(); checked there
            const uint32_t ia = IS::first(idr[u]), ib = IS::second(idr[u]);
            const bool ok = __builtin_amdgcn_ballot_w64(ia >= nmetrics || ib >= nmetrics) == 0ull;
            add(ia, vr[u].x, ok);
            add(ib, vr[u].y, ok);
        }
    }
    if (blockIdx.x == nfull % gridDim.x) { // remainder pairs + odd tail (guarded)
        for (size_t i = nfull * tile + tid; i < npair; i += KS_BLOCK) {
            const typename IS::raw_t id = ip.ld(i);
            const sd2_t x = vp[i];
            add(IS::first(id), x.x, false);
            add(IS::second(id), x.y, false);
        }
        if (tid == 0 && (n & 1)) add(ids[n - 1], v[n - 1], false);
    }
    __syncthreads();
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) nfall += __shfl_down(nfall, d, 64);
    if (lane == 0 && nfall) atomicAdd(&err[1], nfall); // err[1]: fallback counter, read back with the extract

    // flush: one uint64 atomic per occupied cell
    for (uint32_t i = tid; i < words; i += KS_BLOCK) {
        const uint32_t c = h[i];
        if (c) {
            const uint32_t l = i >> log_w, b = s_org[l] + (i & (W - 1));
            lh::cell_add(counts, (size_t)l * LH_ROW_STRIDE + b, c);
            atomicMin(&s_mn[l], b);
            atomicMax(&s_mx[l], b);
        }
    }
    for (uint32_t i = tid; i < OV_SLOTS; i += KS_BLOCK)
        if (ov_key[i] != OV_EMPTY) ks_global_add(counts, ranges, ov_key[i] >> 16, ov_key[i] & 0xffffu, ov_cnt[i]);
    __syncthreads();
    if (tid < nmetrics && s_mn[tid] != 0xffffffffu) {
        uint32_t *r = ranges + 2 * (size_t)tid;
        if (s_mn[tid] < r[0]) atomicMin(&r[0], s_mn[tid]);
        if (s_mx[tid] > r[1]) atomicMax(&r[1], s_mx[tid]);
    }
}

template <typename IDT>
static hipError_t launch_small_t(const IDT *d_ids, const double *d_v, size_t n, uint64_t *counts, uint32_t *ranges,
                                 uint32_t nmetrics, const double *d_Tx, uint32_t *d_err, int num_cus, hipStream_t s)
{
    static std::atomic<bool> attr_set{false}; // benign if two threads race: both set the same attribute
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ingest_pairs_small<16384, 512, IDT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ks_lds_bytes(16384));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_ingest_pairs_small<32768, 1024, IDT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)ks_lds_bytes(32768));
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const bool big = nmetrics > KS_MAXM_SMALL;
    const uint32_t words = big ? 32768u : 16384u;
    const int block = big ? 1024 : 512;
    uint32_t log_w = 0;
    while ((nmetrics << (log_w + 1)) <= words) log_w++; // window = 2^log_w bins per name
    const size_t tile_samples = (size_t)block * KS_UNROLL * 2;
    size_t want = (n + tile_samples - 1) / tile_samples;
    const size_t cap = (size_t)num_cus * (big ? 1 : 2);
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    if (big)
        hipLaunchKernelGGL((k_ingest_pairs_small<32768, 1024, IDT>), dim3(grid ? grid : 1), dim3(1024), ks_lds_bytes(32768), s,
                           d_ids, d_v, n, counts, ranges, nmetrics, log_w, d_Tx, d_err);
    else
        hipLaunchKernelGGL((k_ingest_pairs_small<16384, 512, IDT>), dim3(grid ? grid : 1), dim3(512), ks_lds_bytes(16384), s,
                           d_ids, d_v, n, counts, ranges, nmetrics, log_w, d_Tx, d_err);
    return hipGetLastError();
}

hipError_t launch_ingest_pairs_small(Ids d_ids, const double *d_v, size_t n, uint64_t *counts,
                                     uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                                     int num_cus, hipStream_t s)
{
    if (!small_supported(n, nmetrics, d_ids, d_v)) return hipErrorInvalidValue;
    return d_ids.width == 2 ? launch_small_t(d_ids.u16(), d_v, n, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s)
                            : launch_small_t(d_ids.u32(), d_v, n, counts, ranges, nmetrics, d_Tx, d_err, num_cus, s);
}

} // namespace lh
