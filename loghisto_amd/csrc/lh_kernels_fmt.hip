// lh_kernels_fmt.hip -- K6: ProcessedMetricSet assembly and wire formatting on the device (gfx950).
//
// What the reference does per interval on one core:
//   processMetrics      /root/reference/metrics.go:483-506   name+"_count" etc. into a map[string]float64
//   addAggregates       /root/reference/metrics.go:590-608   name+"_agg_avg|_agg_count|_agg_sum"
//   GraphiteProtocol    /root/reference/graphite.go:37-48    "cockroach.<host>.<metric with _ -> .> %f %d\n"
//   OpenTSDBProtocol    /root/reference/opentsdb.go:45-58    "put <metric> %d %f host=<host>\n"
// i.e. one fmt.Sprintf per key: 65 536 names x 15 keys is ~1e6 Sprintf + map inserts per second.
//
// Here every (metric, key) pair is one thread: it takes the value from the extract results that are already in
// HBM, formats it exactly as Go's %f does (the exact decimal expansion of the float64, 6 fractional digits,
// round-half-even on the exact value, "NaN" / "+Inf" / "-Inf"), and writes the whole line.  Three launches:
// line lengths, a scan of the per-workgroup totals, and the write, which stages a workgroup's lines in LDS and
// copies them out with 16-byte stores.  Line order is metric-major, keys in the fixed order
// _count, _sum, _avg, the percentile labels, _agg_avg, _agg_count, _agg_sum (Go's map order is random, so any
// fixed order is a valid output of the reference).
#include <hip/hip_runtime.h>

#include "lh_kernels.h"

namespace lh {

namespace {

constexpr int SER_BLOCK = 256;
constexpr int SER_WAVES = SER_BLOCK / 64;
constexpr uint32_t SER_LDS = 36864;      // staging bytes of one workgroup (256 lines x 144 B)
constexpr uint32_t SER_BLOB_LDS = 2048;  // prefix / separator / suffix / key strings

// ---------------------------------------------------------------------------
// Go's %f for float64 (strconv.FormatFloat(v, 'f', 6, 64)): exact, round-half-even.
// ---------------------------------------------------------------------------
struct Dec {
    uint64_t ip;    // integer part (kind 0)
    uint32_t frac;  // 6 fractional digits as an integer 0..999999 (kind 0)
    uint32_t kind;  // 0: |v| < 2^64, 1: |v| >= 2^64 (an integer), 2: NaN, 3: Inf
    uint32_t neg;
};

__device__ __forceinline__ Dec decompose(double v)
{
    Dec d;
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    d.neg = (uint32_t)(bits >> 63);
    d.ip = 0;
    d.frac = 0;
    const uint32_t eb = (uint32_t)(bits >> 52) & 0x7ffu;
    if (eb == 0x7ffu) {
        d.kind = (bits & 0xfffffffffffffull) ? 2u : 3u;
        return d;
    }
    const double a = __longlong_as_double((long long)(bits & 0x7fffffffffffffffull));
    if (a >= 18446744073709551616.0) {
        d.kind = 1;
        return d;
    }
    d.kind = 0;
    d.ip = (uint64_t)a; // exact: a < 2^64
    if (a < 9007199254740992.0) {
        const double fp = a - (double)d.ip; // exact: the fractional part of a float64 is a float64
        if (fp != 0.0) {
            const uint64_t fb = (uint64_t)__double_as_longlong(fp);
            uint32_t fe = (uint32_t)(fb >> 52) & 0x7ffu;
            uint64_t fm = fb & 0xfffffffffffffull;
            if (fe) fm |= 1ull << 52; else fe = 1;
            const uint32_t s = 1075u - fe; // fp = fm * 2^-s, s >= 1
            if (s <= 74u) {                // else fp * 1e6 < 2^53 * 2^20 / 2^75 = 0.25: rounds to 0, no tie
                const unsigned __int128 P = (unsigned __int128)fm * 1000000u; // < 2^73
                uint64_t q = (uint64_t)(P >> s);
                const unsigned __int128 rem = P & ((((unsigned __int128)1) << s) - 1);
                const unsigned __int128 half = ((unsigned __int128)1) << (s - 1);
                if (rem > half || (rem == half && (q & 1))) q++;
                if (q == 1000000u) { q = 0; d.ip++; }
                d.frac = (uint32_t)q;
            }
        }
    }
    return d;
}

__device__ __forceinline__ uint32_t ndigits_u64(uint64_t x)
{
    if (x >= 10000000000000000000ull) return 20;
    uint32_t n = 1;
    uint64_t p = 10;
    while (x >= p) { n++; p *= 10; }
    return n;
}

__device__ __forceinline__ void put_digits(char *dst, uint64_t x, uint32_t nd)
{
    for (int i = (int)nd - 1; i >= 0; i--) {
        const uint64_t q = x / 10;
        dst[i] = (char)('0' + (uint32_t)(x - q * 10));
        x = q;
    }
}

// |v| >= 2^64: the value is the integer mant * 2^sh.  Decimal digits by repeated division by 1e9 (rare path:
// a histogram sum has to exceed 1.8e19 to get here).  Returns the digit count; writes them when dst != nullptr.
__device__ __noinline__ uint32_t big_digits(uint64_t bits, char *dst)
{
    const uint32_t eb = (uint32_t)(bits >> 52) & 0x7ffu;
    const uint64_t mant = (bits & 0xfffffffffffffull) | (1ull << 52);
    const uint32_t sh = eb - 1075u; // 11 .. 971
    uint32_t w[33];
    for (int i = 0; i < 33; i++) w[i] = 0;
    const uint32_t wi = sh >> 5, bi = sh & 31u;
    const unsigned __int128 m = (unsigned __int128)mant << bi; // < 2^84
    w[wi] = (uint32_t)m;
    w[wi + 1] = (uint32_t)(m >> 32);
    w[wi + 2] = (uint32_t)(m >> 64);
    int nw = (int)wi + 3;
    while (nw > 0 && w[nw - 1] == 0) nw--;
    uint32_t chunk[36];
    int nc = 0;
    while (nw > 0) {
        uint64_t rem = 0;
        for (int i = nw - 1; i >= 0; i--) {
            const uint64_t cur = (rem << 32) | w[i];
            const uint64_t q = cur / 1000000000ull;
            w[i] = (uint32_t)q;
            rem = cur - q * 1000000000ull;
        }
        chunk[nc++] = (uint32_t)rem;
        while (nw > 0 && w[nw - 1] == 0) nw--;
    }
    const uint32_t top = ndigits_u64(chunk[nc - 1]);
    const uint32_t nd = top + 9u * (uint32_t)(nc - 1);
    if (dst) {
        put_digits(dst, chunk[nc - 1], top);
        char *p = dst + top;
        for (int c = nc - 2; c >= 0; c--, p += 9) put_digits(p, chunk[c], 9);
    }
    return nd;
}

// Length of "%f" of v; writes the text when WRITE.
template <bool WRITE> __device__ __forceinline__ uint32_t fmt_f(double v, char *dst)
{
    const Dec d = decompose(v);
    if (d.kind == 2u) {
        if (WRITE) { dst[0] = 'N'; dst[1] = 'a'; dst[2] = 'N'; }
        return 3;
    }
    if (d.kind == 3u) {
        if (WRITE) { dst[0] = d.neg ? '-' : '+'; dst[1] = 'I'; dst[2] = 'n'; dst[3] = 'f'; }
        return 4;
    }
    uint32_t pos = 0;
    if (d.neg) {
        if (WRITE) dst[0] = '-';
        pos = 1;
    }
    uint32_t nd;
    if (d.kind == 1u) {
        nd = big_digits((uint64_t)__double_as_longlong(v), WRITE ? dst + pos : nullptr);
    } else {
        nd = ndigits_u64(d.ip);
        if (WRITE) put_digits(dst + pos, d.ip, nd);
    }
    pos += nd;
    if (WRITE) {
        dst[pos] = '.';
        put_digits(dst + pos + 1, d.frac, 6);
    }
    return pos + 7;
}

// ---------------------------------------------------------------------------
// line = prefix  key_pre name key_post  sep  %f  suffix
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool line_value(const SerArgs &a, uint32_t m, uint32_t j, double *v)
{
    if (a.flags & SER_COUNTERS) { // float64(count), metrics.go:488, 492
        const size_t c = (size_t)a.first + m;
        if (j == 0) { *v = (double)a.c_total[c]; return a.c_known[c] != 0; }
        *v = (double)a.c_rate[c];
        return a.c_present[c] != 0;
    }
    const ExtractOut &st = a.stats[m];
    if (!st.present) return false;
    if (j == 0) { *v = (double)st.count; return true; }  // metrics.go:349
    if (j == 1) { *v = st.sum; return true; }            // metrics.go:350
    if (j == 2) { *v = st.avg; return true; }            // metrics.go:351
    if (j < 3 + a.np) {                                  // metrics.go:378-385
        const size_t o = (size_t)m * a.np + (j - 3);
        if (!a.pvalid[o]) return false;
        *v = a.pvals[o];
        return true;
    }
    // metrics.go:590-608: only when the lifetime count is > 0; _agg_avg is an INTEGER division
    const uint64_t lc = a.life[2 * (size_t)(a.first + m)], ls = a.life[2 * (size_t)(a.first + m) + 1];
    if (lc == 0) return false;
    const uint32_t k = j - 3 - a.np;
    *v = k == 0 ? (double)(ls / lc) : k == 1 ? (double)lc : (double)ls;
    return true;
}

__device__ __forceinline__ uint32_t line_fixed_len(const SerArgs &a, uint32_t m, uint32_t j)
{
    const uint32_t id = a.first + m;
    return a.prefix_len + a.keys[j].pre_len + (a.name_off[id + 1] - a.name_off[id]) + a.keys[j].post_len +
           a.sep_len + a.suffix_len;
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if ((int)lane >= d) inc += y;
    }
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SER_WAVES; w++) {
        if (w < (int)wave) base += s_w[w];
        tot += s_w[w];
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SER_BLOCK) void k_ser_len(const SerArgs a, uint32_t *__restrict__ lens,
                                                       uint32_t *__restrict__ bsum)
{
    __shared__ uint32_t s_w[SER_WAVES];
    const uint64_t L = (uint64_t)blockIdx.x * SER_BLOCK + threadIdx.x;
    const uint64_t nlines = (uint64_t)a.nmetrics * a.nkeys;
    uint32_t len = 0;
    if (L < nlines) {
        const uint32_t m = (uint32_t)(L / a.nkeys), j = (uint32_t)(L - (uint64_t)m * a.nkeys);
        double v;
        if (line_value(a, m, j, &v)) len = line_fixed_len(a, m, j) + fmt_f<false>(v, nullptr);
        lens[L] = len;
    }
    uint32_t total;
    (void)block_excl_scan(len, s_w, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// exclusive scan of the workgroup totals; boff[nblocks] = total bytes
__global__ __launch_bounds__(1024) void k_ser_scan(const uint32_t *__restrict__ bsum, uint64_t *__restrict__ boff,
                                                   uint32_t nblocks)
{
    __shared__ uint64_t s_w[16];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < nblocks ? bsum[i] : 0;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t lo = (uint32_t)inc, hi = (uint32_t)(inc >> 32);
            lo = __shfl_up(lo, d, 64);
            hi = __shfl_up(hi, d, 64);
            if ((int)lane >= d) inc += ((uint64_t)hi << 32) | lo;
        }
        __syncthreads();
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        uint64_t wb = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            if (w < (int)wave) wb += s_w[w];
            tot += s_w[w];
        }
        if (i < nblocks) boff[i] = carry + wb + inc - v;
        carry += tot;
    }
    if (threadIdx.x == 0) boff[nblocks] = carry;
}

__device__ __forceinline__ char *put_bytes(char *dst, const char *src, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) dst[i] = src[i];
    return dst + n;
}

__global__ __launch_bounds__(SER_BLOCK) void k_ser_write(const SerArgs a, const uint32_t *__restrict__ lens,
                                                         const uint64_t *__restrict__ boff, char *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) char s_stage[SER_LDS + 16];
    __shared__ char s_blob[SER_BLOB_LDS];
    __shared__ uint32_t s_w[SER_WAVES];

    for (uint32_t i = threadIdx.x; i < a.blob_len; i += SER_BLOCK) s_blob[i] = a.blob[i];

    const uint64_t L = (uint64_t)blockIdx.x * SER_BLOCK + threadIdx.x;
    const uint64_t nlines = (uint64_t)a.nmetrics * a.nkeys;
    const uint32_t len = L < nlines ? lens[L] : 0;
    uint32_t total;
    const uint32_t excl = block_excl_scan(len, s_w, &total); // also orders the s_blob fill before its use
    const uint64_t base = boff[blockIdx.x];
    const uint32_t phase = (uint32_t)(base & 15u); // same 16-byte phase in LDS and in the output
    const bool staged = total + phase <= SER_LDS;
    if (len) {
        const uint32_t m = (uint32_t)(L / a.nkeys), j = (uint32_t)(L - (uint64_t)m * a.nkeys);
        const uint32_t id = a.first + m;
        char *p = staged ? s_stage + phase + excl : out + base + excl;
        p = put_bytes(p, s_blob + a.prefix_off, a.prefix_len);
        p = put_bytes(p, s_blob + a.keys[j].pre_off, a.keys[j].pre_len);
        const uint32_t n0 = a.name_off[id], n1 = a.name_off[id + 1];
        if (a.flags & SER_DOTS) { // strings.Replace(metric, "_", ".", -1), graphite.go:42
            for (uint32_t i = n0; i < n1; i++) {
                const char c = a.names[i];
                *p++ = c == '_' ? '.' : c;
            }
        } else {
            p = put_bytes(p, a.names + n0, n1 - n0);
        }
        p = put_bytes(p, s_blob + a.keys[j].post_off, a.keys[j].post_len);
        p = put_bytes(p, s_blob + a.sep_off, a.sep_len);
        double v = 0;
        (void)line_value(a, m, j, &v);
        p += fmt_f<true>(v, p);
        (void)put_bytes(p, s_blob + a.suffix_off, a.suffix_len);
    }
    if (!staged) return; // uniform: lines went straight to HBM
    __syncthreads();
    const char *src = s_stage + phase;
    char *dst = out + base;
    const uint32_t head = total < ((16u - phase) & 15u) ? total : ((16u - phase) & 15u);
    if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
    const uint32_t nvec = (total - head) >> 4;
    const uint4 *vs = reinterpret_cast<const uint4 *>(src + head);
    uint4 *vd = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t i = threadIdx.x; i < nvec; i += SER_BLOCK) vd[i] = vs[i];
    const uint32_t done = head + (nvec << 4);
    if (threadIdx.x < total - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
}

// processHistograms' lifetime side effect (metrics.go:359-376): sum store += uint64(totalSum), count store += count
__global__ __launch_bounds__(256) void k_life_add(const ExtractOut *__restrict__ stats, uint64_t *__restrict__ life,
                                                  uint32_t n)
{
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n || !stats[m].present) return;
    life[2 * (size_t)m] += stats[m].count;
    life[2 * (size_t)m + 1] += stats[m].agg_sum_add;
}

// "%f" of an array (parity tests of the formatter alone): fixed 336-byte slots
__global__ __launch_bounds__(256) void k_format_f(const double *__restrict__ v, char *__restrict__ out,
                                                  uint32_t *__restrict__ lens, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    lens[i] = fmt_f<true>(v[i], out + (size_t)i * SER_FMT_SLOT);
}

} // namespace

uint32_t ser_blocks(uint64_t nlines) { return (uint32_t)((nlines + SER_BLOCK - 1) / SER_BLOCK); }

hipError_t launch_ser_len(const SerArgs &a, uint32_t *lens, uint32_t *bsum, uint64_t *boff, hipStream_t s)
{
    const uint64_t nlines = (uint64_t)a.nmetrics * a.nkeys;
    if (nlines == 0) return hipSuccess;
    const uint32_t nb = ser_blocks(nlines);
    hipLaunchKernelGGL(k_ser_len, dim3(nb), dim3(SER_BLOCK), 0, s, a, lens, bsum);
    hipLaunchKernelGGL(k_ser_scan, dim3(1), dim3(1024), 0, s, bsum, boff, nb);
    return hipGetLastError();
}

hipError_t launch_ser_write(const SerArgs &a, const uint32_t *lens, const uint64_t *boff, char *out, hipStream_t s)
{
    const uint64_t nlines = (uint64_t)a.nmetrics * a.nkeys;
    if (nlines == 0) return hipSuccess;
    hipLaunchKernelGGL(k_ser_write, dim3(ser_blocks(nlines)), dim3(SER_BLOCK), 0, s, a, lens, boff, out);
    return hipGetLastError();
}

hipError_t launch_life_add(const ExtractOut *stats, uint64_t *life, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_life_add, dim3((n + 255) / 256), dim3(256), 0, s, stats, life, n);
    return hipGetLastError();
}

hipError_t launch_format_f(const double *d_v, char *d_out, uint32_t *d_lens, uint32_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_format_f, dim3((n + 255) / 256), dim3(256), 0, s, d_v, d_out, d_lens, n);
    return hipGetLastError();
}

} // namespace lh
