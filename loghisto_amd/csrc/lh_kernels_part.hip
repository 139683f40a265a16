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
//   P1  k_part_scatter : read (id, v) [12 B], compress on the fly, emit a 4-byte record
//                        (local name id << 16 | bin) into the chunk list of partition
//                        id % NP.  Tile-local counting sort in LDS so that every
//                        partition's records leave the CU as one contiguous run;
//                        chunks (1 024 records) come from a workgroup-private pool, so
//                        the hot loop has no global atomics at all.
//   plan k_plan_*      : group the ~n/1024 chunk descriptors by partition and cut them
//                        into <= 1 024 equal work slots (three tiny kernels).
//   P2  k_part_hist    : one workgroup per slot: LDS windows (uint32) for the few names
//                        of that partition, ds_add per record, then ONE uint64 global
//                        atomic per occupied cell at flush.
//
// HBM traffic: 12 (read) + 4 (write) + 4 (read) = 20 B/sample against 12 B algorithmic.
// Everything is exact: records carry the exact bin; out-of-window records and small
// launches fall back to direct global atomics.
#include "lh_kernels.h"
#include "lh_codec.h"

namespace lh {

constexpr int NPMAX = 256;             // partitions (power of two, <= 256)
constexpr uint32_t CHUNK = 1024;       // records per chunk (4 KiB)
constexpr int P1_BLOCK = 512;
constexpr int P1_SPT = 8;              // samples per thread per tile
constexpr int P1_TILE = P1_BLOCK * P1_SPT;
constexpr uint32_t INVALID = 0xffffffffu;
constexpr int P2_BLOCK = 512;
constexpr uint32_t P2_WINWORDS = 16384; // 64 KiB of uint32 windows per workgroup
constexpr uint32_t P2_SLOTS = 1024;
constexpr size_t PART_MIN_SAMPLES = 131072;
constexpr uint32_t PART_MAX_MPP = 256;

struct PartPlan {
    uint32_t log_np, np, mpp, log_w;
    uint32_t g1, chunks_per_wg, nchunks;
    size_t off_records, off_cdesc, off_sorted, off_small, total;
};

static uint32_t ilog2_ceil(uint32_t x)
{
    uint32_t l = 0;
    while ((1u << l) < x) l++;
    return l;
}

static bool make_plan(size_t n, uint32_t nmetrics, int num_cus, PartPlan &P)
{
    if (n < PART_MIN_SAMPLES || n > (size_t(1) << 31) || nmetrics < 2) return false;
    uint32_t want_np = (nmetrics + 3) / 4;
    P.log_np = ilog2_ceil(want_np);
    if (P.log_np > 8) P.log_np = 8;
    P.np = 1u << P.log_np;
    P.mpp = (nmetrics + P.np - 1) >> P.log_np;
    if (P.mpp > PART_MAX_MPP) return false;
    uint32_t lw = 0;
    while ((P.mpp << (lw + 1)) <= P2_WINWORDS) lw++;
    P.log_w = lw; // window = 2^log_w bins per name, mpp * window <= 16384
    const size_t ntiles = (n + P1_TILE - 1) / P1_TILE;
    size_t g1 = (size_t)num_cus * 2;
    if (g1 > ntiles) g1 = ntiles;
    P.g1 = (uint32_t)g1;
    const size_t tiles_per_wg = (ntiles + g1 - 1) / g1;
    P.chunks_per_wg = (uint32_t)(tiles_per_wg * (P1_TILE / CHUNK) + P.np + 1);
    P.nchunks = P.g1 * P.chunks_per_wg;
    size_t o = 0;
    P.off_records = o; o += (size_t)P.nchunks * CHUNK * sizeof(uint32_t);
    P.off_cdesc = o; o += (size_t)P.nchunks * sizeof(uint32_t);
    P.off_sorted = o; o += (size_t)P.nchunks * sizeof(uint32_t);
    o = (o + 255) & ~size_t(255);
    P.off_small = o; o += (3 * NPMAX + 3 * P2_SLOTS + 64) * sizeof(uint32_t);
    P.total = o;
    return true;
}

size_t part_scratch_bytes(size_t n, uint32_t nmetrics, int num_cus)
{
    PartPlan P;
    return make_plan(n, nmetrics, num_cus, P) ? P.total : 0;
}

// ---------------------------------------------------------------------------
// P1: compress + partition scatter
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(P1_BLOCK) void k_part_scatter(const uint32_t *__restrict__ ids,
                                                           const double *__restrict__ v, size_t n,
                                                           uint32_t nmetrics, uint32_t log_np,
                                                           const double *__restrict__ Tx,
                                                           uint32_t *__restrict__ records,
                                                           uint32_t *__restrict__ cdesc, uint32_t chunks_per_wg,
                                                           uint32_t *__restrict__ err)
{
    __shared__ uint32_t s_cnt[NPMAX], s_off[NPMAX], s_cfill[NPMAX], s_cbase[NPMAX];
    __shared__ uint32_t s_of[NPMAX], s_obase[NPMAX], s_onew[NPMAX];
    __shared__ uint32_t s_sorted[P1_TILE];
    __shared__ uint8_t s_spart[P1_TILE];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_pool_next, s_total;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t np = 1u << log_np, pmask = np - 1;
    const uint32_t pool_base = blockIdx.x * chunks_per_wg;

    if (tid < NPMAX) { s_cnt[tid] = 0; s_cfill[tid] = CHUNK; s_cbase[tid] = INVALID; }
    if (tid == 0) s_pool_next = 0;
    __syncthreads();

    const size_t ntiles = (n + P1_TILE - 1) / P1_TILE;
    for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const size_t base = tile * P1_TILE;
        uint32_t rec[P1_SPT], pr[P1_SPT];
        uint32_t idv[P1_SPT];
        double val[P1_SPT];
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            const size_t i = base + (size_t)j * P1_BLOCK + tid;
            const bool ok = i < n;
            idv[j] = ok ? __builtin_nontemporal_load(ids + i) : INVALID;
            val[j] = ok ? __builtin_nontemporal_load(v + i) : 0.0;
        }
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            pr[j] = INVALID;
            rec[j] = 0;
            if (idv[j] < nmetrics) {
                const uint32_t bin = lh_bin_of(val[j], Tx);
                const uint32_t p = idv[j] & pmask;
                rec[j] = ((idv[j] >> log_np) << 16) | bin;
                const uint32_t rank = atomicAdd(&s_cnt[p], 1u);
                pr[j] = p | (rank << 8);
            } else if (base + (size_t)j * P1_BLOCK + tid < n) {
                atomicOr(err, 1u); // id >= nmetrics: reported by lh_sync / lh_extract
            }
        }
        __syncthreads();

        // exclusive scan of the per-partition counts + chunk bookkeeping (threads 0..255)
        uint32_t c = 0, inc = 0;
        if (tid < NPMAX) {
            c = s_cnt[tid];
            inc = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d, 64);
                if ((int)lane >= d) inc += y;
            }
            if (lane == 63) s_wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < NPMAX) {
            uint32_t wbase = 0;
            for (uint32_t w = 0; w < wave; w++) wbase += s_wsum[w];
            s_off[tid] = wbase + inc - c;
            if (tid == NPMAX - 1) s_total = wbase + inc;
            s_cnt[tid] = 0; // ranks are already in registers
            if (c) {
                const uint32_t p = tid;
                const uint32_t f = s_cfill[p], cb = s_cbase[p];
                s_of[p] = f;
                s_obase[p] = cb;
                const uint32_t room = CHUNK - f;
                if (c > room) {
                    const uint32_t over = c - room;
                    const uint32_t k = (over + CHUNK - 1) / CHUNK;
                    const uint32_t first = pool_base + atomicAdd(&s_pool_next, k);
                    s_onew[p] = first;
                    if (cb != INVALID) cdesc[cb] = (p << 16) | CHUNK; // the old chunk is now full
                    for (uint32_t q = 0; q + 1 < k; q++) cdesc[first + q] = (p << 16) | CHUNK;
                    s_cbase[p] = first + k - 1;
                    s_cfill[p] = over - (k - 1) * CHUNK;
                } else {
                    s_cfill[p] = f + c;
                }
            }
        }
        __syncthreads();

        // tile-local counting sort into LDS
#pragma unroll
        for (int j = 0; j < P1_SPT; j++) {
            if (pr[j] != INVALID) {
                const uint32_t p = pr[j] & 0xffu;
                const uint32_t pos = s_off[p] + (pr[j] >> 8);
                s_sorted[pos] = rec[j];
                s_spart[pos] = (uint8_t)p;
            }
        }
        __syncthreads();

        // copy out: every partition's records of this tile form one contiguous run
        const uint32_t tile_total = s_total;
        for (uint32_t i = tid; i < tile_total; i += P1_BLOCK) {
            const uint32_t p = s_spart[i];
            const uint32_t f = s_of[p] + (i - s_off[p]);
            size_t dst;
            if (f < CHUNK) {
                dst = (size_t)s_obase[p] * CHUNK + f;
            } else {
                const uint32_t f2 = f - CHUNK;
                dst = ((size_t)s_onew[p] + f2 / CHUNK) * CHUNK + (f2 % CHUNK);
            }
            records[dst] = s_sorted[i];
        }
        // (the next tile's first barrier separates this copy-out from the next bookkeeping)
    }
    __syncthreads();
    if (tid < np && s_cbase[tid] != INVALID) cdesc[s_cbase[tid]] = (tid << 16) | s_cfill[tid];
}

// ---------------------------------------------------------------------------
// plan: group chunk descriptors by partition, cut into work slots
// ---------------------------------------------------------------------------
constexpr int PL_BLOCK = 256;
constexpr int PL_PER_WG = 2048; // descriptors per workgroup

__global__ __launch_bounds__(PL_BLOCK) void k_plan_count(const uint32_t *__restrict__ cdesc, uint32_t nchunks,
                                                         uint32_t *__restrict__ pc)
{
    __shared__ uint32_t s_h[NPMAX];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * PL_PER_WG;
    for (uint32_t i = lo + threadIdx.x; i < lo + PL_PER_WG && i < nchunks; i += PL_BLOCK) {
        const uint32_t d = cdesc[i];
        if (d != INVALID) atomicAdd(&s_h[d >> 16], 1u);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&pc[threadIdx.x], s_h[threadIdx.x]);
}

// one workgroup of 256 threads; thread p owns partition p
__global__ __launch_bounds__(PL_BLOCK) void k_plan_scan(const uint32_t *__restrict__ pc,
                                                        uint32_t *__restrict__ part_start,
                                                        uint32_t *__restrict__ cursor,
                                                        uint32_t *__restrict__ slots,
                                                        uint32_t *__restrict__ nslots, uint32_t np)
{
    __shared__ uint32_t s_a[NPMAX], s_b[NPMAX];
    const uint32_t p = threadIdx.x;
    const uint32_t c = pc[p];
    s_a[p] = c;
    __syncthreads();
    // serial prefix by one thread: 256 entries, negligible
    if (p == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < NPMAX; i++) { const uint32_t t = s_a[i]; s_a[i] = run; run += t; }
        s_b[0] = run; // total chunks
    }
    __syncthreads();
    const uint32_t total = s_b[0];
    const uint32_t start = s_a[p];
    part_start[p] = start;
    cursor[p] = start;
    __syncthreads();
    uint32_t target = (total + (P2_SLOTS - np) - 1) / (P2_SLOTS - np);
    if (target == 0) target = 1;
    const uint32_t k = (c + target - 1) / target;
    s_a[p] = k;
    __syncthreads();
    if (p == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < NPMAX; i++) { const uint32_t t = s_a[i]; s_a[i] = run; run += t; }
        *nslots = run;
    }
    __syncthreads();
    const uint32_t s0 = s_a[p];
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t first = j * target;
        const uint32_t cnt = (c - first) < target ? (c - first) : target;
        slots[3 * (s0 + j) + 0] = p;
        slots[3 * (s0 + j) + 1] = first;
        slots[3 * (s0 + j) + 2] = cnt;
    }
}

__global__ __launch_bounds__(PL_BLOCK) void k_plan_scatter(const uint32_t *__restrict__ cdesc, uint32_t nchunks,
                                                           uint32_t *__restrict__ cursor,
                                                           uint32_t *__restrict__ sorted)
{
    __shared__ uint32_t s_h[NPMAX], s_base[NPMAX];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * PL_PER_WG;
    uint32_t pr[PL_PER_WG / PL_BLOCK];
#pragma unroll
    for (int j = 0; j < PL_PER_WG / PL_BLOCK; j++) {
        const uint32_t i = lo + j * PL_BLOCK + threadIdx.x;
        pr[j] = INVALID;
        if (i < nchunks) {
            const uint32_t d = cdesc[i];
            if (d != INVALID) {
                const uint32_t p = d >> 16;
                pr[j] = p | (atomicAdd(&s_h[p], 1u) << 8);
            }
        }
    }
    __syncthreads();
    {
        const uint32_t c = s_h[threadIdx.x];
        s_base[threadIdx.x] = c ? atomicAdd(&cursor[threadIdx.x], c) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PL_PER_WG / PL_BLOCK; j++) {
        if (pr[j] != INVALID) sorted[s_base[pr[j] & 0xffu] + (pr[j] >> 8)] = lo + j * PL_BLOCK + threadIdx.x;
    }
}

// ---------------------------------------------------------------------------
// P2: per-partition LDS histogram
// ---------------------------------------------------------------------------
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
constexpr size_t P2_LDS_BYTES = (P2_WINWORDS + 3 * PART_MAX_MPP) * sizeof(uint32_t) + 16;

__device__ __forceinline__ void p2_global_add(uint64_t *__restrict__ counts, uint32_t *__restrict__ ranges,
                                              uint32_t m, uint32_t bin, uint64_t c)
{
    atomicAdd(reinterpret_cast<unsigned long long *>(&counts[(size_t)m * LH_NKEYS + bin]), (unsigned long long)c);
    uint32_t *r = ranges + 2 * (size_t)m;
    if (bin < r[0]) atomicMin(&r[0], bin);
    if (bin > r[1]) atomicMax(&r[1], bin);
}

__global__ __launch_bounds__(P2_BLOCK) void k_part_hist(const uint32_t *__restrict__ records,
                                                        const uint32_t *__restrict__ cdesc,
                                                        const uint32_t *__restrict__ sorted,
                                                        const uint32_t *__restrict__ part_start,
                                                        const uint32_t *__restrict__ slots,
                                                        const uint32_t *__restrict__ nslots, uint32_t log_np,
                                                        uint32_t mpp, uint32_t log_w,
                                                        uint64_t *__restrict__ counts,
                                                        uint32_t *__restrict__ ranges)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *h = reinterpret_cast<uint32_t *>(smem);
    uint32_t *s_org = h + P2_WINWORDS;
    uint32_t *s_mn = s_org + PART_MAX_MPP;
    uint32_t *s_mx = s_mn + PART_MAX_MPP;

    const uint32_t slot = blockIdx.x;
    if (slot >= *nslots) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t p = slots[3 * slot], first = slots[3 * slot + 1], cnt = slots[3 * slot + 2];
    const uint32_t *list = sorted + part_start[p] + first;
    const uint32_t W = 1u << log_w, words = mpp << log_w;

    for (uint32_t i = tid; i < words; i += P2_BLOCK) h[i] = 0;
    if (tid < mpp) { s_mn[tid] = INVALID; s_mx[tid] = 0; }
    __syncthreads();

    // window origins from the slot's first chunk (streams are unimodal per name in practice;
    // a badly placed window only costs speed: out-of-window records go to global atomics)
    {
        const uint32_t c0 = list[0];
        const uint32_t n0 = cdesc[c0] & 0xffffu;
        const uint32_t *b0 = records + (size_t)c0 * CHUNK;
        for (uint32_t i = tid; i < n0; i += P2_BLOCK) {
            const uint32_t rec = b0[i];
            atomicMin(&s_mn[rec >> 16], rec & 0xffffu);
            atomicMax(&s_mx[rec >> 16], rec & 0xffffu);
        }
    }
    __syncthreads();
    if (tid < mpp) {
        uint32_t org = 32768u - W / 2; // unseen name: centre on key 0
        if (s_mn[tid] != INVALID) {
            const uint32_t mid = (s_mn[tid] + s_mx[tid]) / 2;
            org = mid > W / 2 ? mid - W / 2 : 0;
        }
        if (org > LH_NKEYS - W) org = LH_NKEYS - W;
        s_org[tid] = org;
        s_mn[tid] = INVALID; // reused for the flush ranges
        s_mx[tid] = 0;
    }
    __syncthreads();

    // one chunk per wave per iteration: 1 024 records = 4 x (64 lanes x 16 B)
    for (uint32_t j = wave; j < cnt; j += P2_BLOCK / 64) {
        const uint32_t cid = __builtin_amdgcn_readfirstlane(list[j]);
        const uint32_t cn = __builtin_amdgcn_readfirstlane(cdesc[cid] & 0xffffu);
        const u4_t *src = reinterpret_cast<const u4_t *>(records + (size_t)cid * CHUNK);
#pragma unroll
        for (uint32_t q = 0; q < CHUNK / 256; q++) {
            const uint32_t k = q * 256 + lane * 4;
            if (q * 256 >= cn) break; // wave-uniform
            const u4_t r4 = __builtin_nontemporal_load(src + (k >> 2));
            const uint32_t rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const bool ok = k + t < cn;
                const uint32_t rec = rr[t];
                // constant streams: the whole wave carries one record value
                const uint32_t f0 = __builtin_amdgcn_readfirstlane(rec);
                const unsigned long long act = __builtin_amdgcn_ballot_w64(ok);
                const unsigned long long dif = __builtin_amdgcn_ballot_w64(ok && rec != f0);
                const uint32_t l = rec >> 16, b = rec & 0xffffu;
                const uint32_t rel = b - s_org[l & (PART_MAX_MPP - 1)];
                if (act && dif == 0ull && (act & 1ull)) {
                    if (lane == 0) {
                        const uint32_t nact = (uint32_t)__builtin_popcountll(act);
                        if (rel < W) atomicAdd(&h[(l << log_w) + rel], nact);
                        else p2_global_add(counts, ranges, (l << log_np) | p, b, nact);
                    }
                } else if (ok) {
                    if (rel < W) atomicAdd(&h[(l << log_w) + rel], 1u);
                    else p2_global_add(counts, ranges, (l << log_np) | p, b, 1);
                }
            }
        }
    }
    __syncthreads();

    // flush: one uint64 atomic per occupied cell
    for (uint32_t i = tid; i < words; i += P2_BLOCK) {
        const uint32_t c = h[i];
        if (c) {
            const uint32_t l = i >> log_w, b = s_org[l] + (i & (W - 1));
            atomicAdd(reinterpret_cast<unsigned long long *>(&counts[(size_t)((l << log_np) | p) * LH_NKEYS + b]),
                      (unsigned long long)c);
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

hipError_t launch_ingest_pairs_part(const uint32_t *d_ids, const double *d_v, size_t n, uint64_t *counts,
                                    uint32_t *ranges, uint32_t nmetrics, const double *d_Tx, uint32_t *d_err,
                                    void *scratch, size_t scratch_bytes, int num_cus, hipStream_t s)
{
    PartPlan P;
    if (!make_plan(n, nmetrics, num_cus, P) || scratch_bytes < P.total || !scratch) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_part_hist),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)P2_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    unsigned char *base = static_cast<unsigned char *>(scratch);
    uint32_t *records = reinterpret_cast<uint32_t *>(base + P.off_records);
    uint32_t *cdesc = reinterpret_cast<uint32_t *>(base + P.off_cdesc);
    uint32_t *sorted = reinterpret_cast<uint32_t *>(base + P.off_sorted);
    uint32_t *small = reinterpret_cast<uint32_t *>(base + P.off_small);
    uint32_t *pc = small, *part_start = small + NPMAX, *cursor = small + 2 * NPMAX;
    uint32_t *slots = small + 3 * NPMAX, *nslots = slots + 3 * P2_SLOTS;

    hipError_t e = hipMemsetAsync(cdesc, 0xff, (size_t)P.nchunks * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(small, 0, (3 * NPMAX + 3 * P2_SLOTS + 64) * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_part_scatter, dim3(P.g1), dim3(P1_BLOCK), 0, s, d_ids, d_v, n, nmetrics, P.log_np, d_Tx,
                       records, cdesc, P.chunks_per_wg, d_err);
    const unsigned plan_grid = (P.nchunks + PL_PER_WG - 1) / PL_PER_WG;
    hipLaunchKernelGGL(k_plan_count, dim3(plan_grid), dim3(PL_BLOCK), 0, s, cdesc, P.nchunks, pc);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(PL_BLOCK), 0, s, pc, part_start, cursor, slots, nslots, P.np);
    hipLaunchKernelGGL(k_plan_scatter, dim3(plan_grid), dim3(PL_BLOCK), 0, s, cdesc, P.nchunks, cursor, sorted);
    hipLaunchKernelGGL(k_part_hist, dim3(P2_SLOTS), dim3(P2_BLOCK), P2_LDS_BYTES, s, records, cdesc, sorted,
                       part_start, slots, nslots, P.log_np, P.mpp, P.log_w, counts, ranges);
    return hipGetLastError();
}

} // namespace lh
