// lh_codec.h -- device-side bucket codec for gfx950.
//
// Two routes to the same int16 key (reference: compress, metrics.go:316-322):
//
//  (1) d_go_log / d_go_exp: an operation-by-operation restatement of Go's
//      math/log.go and math/exp.go in IEEE binary64 with
//      NO fma contraction (this translation unit is built with
//      -ffp-contract=off).  Used once per engine to generate the threshold table
//      Tx[] and the decompress table D[], and by the golog cross-check kernel.
//
//  (2) lh_bin_of: the hot-path index.  x = 1+|v| exactly as the reference, then
//      an approximate t = 100*ln(x)+0.5 from the exponent and v_log_f32 of the
//      mantissa.  If t is farther than LH_GUARD from an integer the bucket is
//      floor(t) with certainty; otherwise (about 1 sample in 4000) the sample is
//      within the guard band of threshold j = rint(t) and one compare against
//      the exact table entry Tx[j] decides.  Exact by construction: the only
//      requirement on the approximation is |t_approx - t_exact| < LH_GUARD/2,
//      which lh_selftest_vlog measures on the device.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define LH_NKEYS 65536
#define LH_NTHRESH 70980
#define LH_KEXT_MAX 70978

// Guard band in units of 1/16384 of a bucket: |t - rint(t)| < 2/16384 = 1.2e-4.
// Approximation error budget: mantissa truncation to fp32 (<= 2^-23/ln2 in log2)
// plus v_log_f32 (measured, ~1e-7) => < 2.5e-5 in t.  See DESIGN.md.
#define LH_GUARD_Q14 2

namespace lh {

__device__ __forceinline__ double d_from_bits(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }

// math/log.go for finite x >= 1 (the only domain compress() reaches: x = 1+|v|).
__device__ inline double d_go_log(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01;
    const double L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01;
    const double L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01;
    const double L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    const double HalfSqrt2 = 0x1.6a09e667f3bcdp-1;

    uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    int ki = (int)((hi >> 20) & 0x7ff) - 1022;               // Frexp: x = f1 * 2^ki
    double f1 = d_from_bits((hi & 0x800fffffu) | (1022u << 20), lo); // f1 in [0.5,1)
    if (f1 < HalfSqrt2) { f1 *= 2; ki--; }
    double f = f1 - 1;
    double k = (double)ki;

    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

// math/exp.go for 0 <= x <= 327.68 (decompress() reaches |c|/100 only).
__device__ inline double d_go_exp(double x)
{
    const double Ln2Hi = 6.93147180369123816490e-01;
    const double Ln2Lo = 1.90821492927058770002e-10;
    const double Log2e = 1.44269504088896338700e+00;
    const double NearZero = 1.0 / (1 << 28);
    const double P1 = 1.66666666666666657415e-01;
    const double P2 = -2.77777777770155933842e-03;
    const double P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06;
    const double P5 = 4.13813679705723846039e-08;

    if (x < NearZero) return 1 + x;
    int k = (int)(Log2e * x + 0.5);
    double hi = x - (double)k * Ln2Hi;
    double lo = (double)k * Ln2Lo;
    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    // Ldexp(y, k): y is normal, 0 <= k <= 473, the result is normal: exact
    // scaling by 2^k is an exponent-field add.
    uint32_t yh = (uint32_t)__double2hiint(y), yl = (uint32_t)__double2loint(y);
    return d_from_bits(yh + ((uint32_t)k << 20), yl);
}

// floor(100*Log(x)+0.5) for finite x >= 1 (the "extended key" before int16 wrap).
__device__ inline int d_kext_golog(double x)
{
    double t = 100.0 * d_go_log(x) + 0.5;
    return (int)t;
}

// int16 key -> dense bin and back.
__device__ __host__ __forceinline__ uint32_t key_to_bin(int k) { return ((uint32_t)k & 0xffffu) ^ 0x8000u; }
__device__ __host__ __forceinline__ int bin_to_key(uint32_t b) { return (int)(int16_t)(uint16_t)(b ^ 0x8000u); }

// Sign / wrap stage of compress (metrics.go:317-321) on an extended key:
// int16 truncation keeps the low 16 bits (amd64), `-1 * i` wraps in int16.
__device__ __forceinline__ uint32_t bin_from_kext(int kext, double v)
{
    uint32_t i = (uint32_t)kext & 0xffffu;
    uint32_t key = (v < 0) ? ((0u - i) & 0xffffu) : i;
    return key ^ 0x8000u;
}

// decompress (metrics.go:326-332) of a dense bin.
__device__ inline double d_decompress_bin(uint32_t bin)
{
    int c = bin_to_key(bin);
    double a = fabs((double)c);
    double f = d_go_exp(a / 100.0) - 1.0;
    return (c < 0) ? -1.0 * f : f;
}

// Hot-path bucket index (route 2 above).  Tx: device threshold table.
__device__ __forceinline__ uint32_t lh_bin_of(double v, const double *__restrict__ Tx)
{
    const double x = 1.0 + fabs(v);                          // metrics.go:317, exact
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const uint32_t eb = hi >> 20;                            // sign is 0: x >= 1 or NaN (fabs)
    int kext = 0;                                            // NaN / +Inf -> int16(...) == 0
    if (eb < 0x7ffu) {
        const int e = (int)eb - 1023;
        const float m = __uint_as_float(0x3f800000u | ((hi & 0xfffffu) << 3) | (lo >> 29));
        const float l2 = __builtin_amdgcn_logf(m);           // v_log_f32: log2(m), m in [1,2)
        // u = (100*ln2*(e+l2) + 0.5) * 2^14, truncated: kext in the high bits,
        // 14 fraction bits below.
        const double uq = __builtin_fma((double)e + (double)l2, 69.314718055994530942 * 16384.0, 8192.0);
        const int u = (int)uq;
        kext = u >> 14;
        if ((((uint32_t)u + LH_GUARD_Q14) & 16383u) < 2u * LH_GUARD_Q14) {
            // within the guard band of threshold j: decide exactly.
            const int j = (int)(((uint32_t)u + 8192u) >> 14);
            kext = (x >= Tx[j]) ? j : j - 1;
        }
    }
    return bin_from_kext(kext, v);
}

// The same index in two steps, for kernels that classify several samples in straight-line code: the fast
// part is branch-free (kext from the hardware log2) and reports whether the sample lies inside the guard band of
// a threshold or is not finite; only then (about 1 sample in 4 000) must the caller take lh_bin_of's exact path.
__device__ __forceinline__ uint32_t lh_bin_fast(double v, bool &uncertain)
{
    const double x = 1.0 + fabs(v);
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const float m = __uint_as_float(0x3f800000u | ((hi & 0xfffffu) << 3) | (lo >> 29));
    const float l2 = __builtin_amdgcn_logf(m);
    // (An integer / float32 form of the next three lines -- e * floor(C) as v_mad_u32_u24, the fractions in two float32
    // fmas -- was measured in round 4: float64 add / fma / converts issue at the same rate as their replacements on
    // gfx950 and the variant was 3 % slower end to end; profiles/r04_index_arithmetic.txt.)
    // (biased exponent + log2 m) * C + (8192 - 1023 C): the bias folded into the constant.  The sum is exact
    // (an integer below 2 048 plus a float), the constant's rounding (2.4e-7 of a Q14 unit) is far inside the guard.
    constexpr double C = 69.314718055994530942 * 16384.0;
    const double uq = __builtin_fma((double)(hi >> 20) + (double)l2, C, 8192.0 - 1023.0 * C);
    const int u = (int)uq;
    // NaN / Inf (x's exponent field all ones) take the exact path too: lh_bin_of gives them bucket 0
    uncertain = hi >= 0x7ff00000u || ((((uint32_t)u + LH_GUARD_Q14) & 16383u) < 2u * LH_GUARD_Q14);
    return bin_from_kext(u >> 14, v);
}

} // namespace lh
