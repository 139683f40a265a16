/*
 * lh_oracle.c -- CPU oracle for the loghisto hot path.  TEST INFRASTRUCTURE ONLY
 * (see lh_oracle.h for scope, citations and pinning status).
 *
 * Build with -ffp-contract=off (oracle/Makefile does): every operation below is
 * a separate IEEE-754 binary64 operation, as Go on amd64 evaluates it.
 */
#include "lh_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t f2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double u2f(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* Go math.Frexp: f in [0.5,1), x = f * 2^e.  Handles subnormals. */
static double go_frexp(double x, int *e)
{
    if (x == 0 || isnan(x) || isinf(x)) { *e = 0; return x; }
    int adj = 0;
    if (fabs(x) < 2.2250738585072014e-308) { /* subnormal: normalize */
        x *= 4503599627370496.0;             /* 2^52 */
        adj = -52;
    }
    uint64_t b = f2u(x);
    *e = adj + (int)((b >> 52) & 0x7ff) - 1022;
    b &= ~((uint64_t)0x7ff << 52);
    b |= (uint64_t)1022 << 52;
    return u2f(b);
}

/* Go math.Ldexp. */
static double go_ldexp(double frac, int e)
{
    if (frac == 0 || isnan(frac) || isinf(frac)) return frac;
    int fe;
    double f = go_frexp(frac, &fe);          /* f in [0.5,1) */
    long ee = (long)e + fe;                  /* frac = f * 2^fe */
    /* represent as m * 2^(ee-1) with m in [1,2) */
    long exp = ee - 1;
    if (exp < -1075) return copysign(0.0, frac);
    if (exp > 1023) return frac < 0 ? -INFINITY : INFINITY;
    double m = 1.0;
    if (exp < -1022) {                       /* denormal result */
        exp += 53;
        m = 1.0 / 9007199254740992.0;        /* 2^-53 */
    }
    uint64_t b = f2u(f);
    b &= ~((uint64_t)0x7ff << 52);
    b |= (uint64_t)(exp + 1023) << 52;
    return m * u2f(b);
}

/* math/log.go; called at metrics.go:317. */
double lho_go_log(double x)
{
    static const double Ln2Hi = 6.93147180369123816490e-01;
    static const double Ln2Lo = 1.90821492927058770002e-10;
    static const double L1 = 6.666666666666735130e-01;
    static const double L2 = 3.999999999940941908e-01;
    static const double L3 = 2.857142874366239149e-01;
    static const double L4 = 2.222219843214978396e-01;
    static const double L5 = 1.818357216161805012e-01;
    static const double L6 = 1.531383769920937332e-01;
    static const double L7 = 1.479819860511658591e-01;
    static const double HalfSqrt2 = 0x1.6a09e667f3bcdp-1; /* Sqrt2/2 */

    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;

    int ki;
    double f1 = go_frexp(x, &ki);
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

/* math/exp.go; called at metrics.go:327. */
double lho_go_exp(double x)
{
    static const double Ln2Hi = 6.93147180369123816490e-01;
    static const double Ln2Lo = 1.90821492927058770002e-10;
    static const double Log2e = 1.44269504088896338700e+00;
    static const double Overflow = 7.09782712893383973096e+02;
    static const double Underflow = -7.45133219101941108420e+02;
    static const double NearZero = 1.0 / (1 << 28);
    static const double P1 = 1.66666666666666657415e-01;
    static const double P2 = -2.77777777770155933842e-03;
    static const double P3 = 6.61375632143793436117e-05;
    static const double P4 = -1.65339022054652515390e-06;
    static const double P5 = 4.13813679705723846039e-08;

    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (isinf(x)) return 0;
    if (x > Overflow) return INFINITY;
    if (x < Underflow) return 0;
    if (-NearZero < x && x < NearZero) return 1 + x;

    long k = 0;
    if (x < 0) k = (long)(Log2e * x - 0.5);
    else if (x > 0) k = (long)(Log2e * x + 0.5);
    double hi = x - (double)k * Ln2Hi;
    double lo = (double)k * Ln2Lo;

    double r = hi - lo;
    double t = r * r;
    double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return go_ldexp(y, (int)k);
}

/* int16(float64) as the amd64 Go compiler does it: CVTTSD2SL (0x80000000 for
 * NaN/Inf/out of int32 range), keep the low 16 bits.  SURVEY.md Appendix A.3. */
static int16_t f64_to_i16_amd64(double t)
{
    int32_t w;
    if (isnan(t) || t >= 2147483648.0 || t <= -2147483649.0) w = INT32_MIN;
    else w = (int32_t)t; /* truncation toward zero */
    return (int16_t)(uint16_t)((uint32_t)w & 0xffffu);
}

uint64_t lho_f64_to_u64_amd64(double f)
{
    /* Go amd64: values < 2^63 go through CVTTSD2SQ (negative -> two's
     * complement wrap, NaN/overflow -> 0x8000000000000000); values >= 2^63
     * are converted as (f - 2^63) with the top bit set. */
    const double two63 = 9223372036854775808.0;
    if (isnan(f)) return 0x8000000000000000ull;
    if (f < two63) {
        if (f <= -two63) return 0x8000000000000000ull;
        return (uint64_t)(int64_t)f;
    }
    double g = f - two63;
    if (g >= two63) return 0; /* indefinite 0x8000.. ^ 0x8000.. */
    return (uint64_t)(int64_t)g ^ 0x8000000000000000ull;
}

/* metrics.go:316-322 */
int16_t lho_compress(double value)
{
    int16_t i = f64_to_i16_amd64(100.0 * lho_go_log(1.0 + fabs(value)) + 0.5);
    if (value < 0) return (int16_t)(-1 * i); /* int16 arithmetic wraps */
    return i;
}

/* metrics.go:326-332 */
double lho_decompress(int16_t c)
{
    double f = lho_go_exp(fabs((double)c) / 100.0) - 1.0;
    if (c < 0) return -1.0 * f;
    return f;
}

int32_t lho_kext(double x)
{
    if (isnan(x) || isinf(x)) return -1;
    double t = 100.0 * lho_go_log(x) + 0.5;
    return (int32_t)t;
}

void lho_kext_many(const double *x, size_t n, int32_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = lho_kext(x[i]);
}

void lho_compress_many(const double *v, size_t n, int16_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = lho_compress(v[i]);
}

static inline uint32_t key_to_bin(int16_t k) { return (uint32_t)((uint16_t)k ^ 0x8000u); }
static inline int16_t bin_to_key(uint32_t b) { return (int16_t)(uint16_t)(b ^ 0x8000u); }

void lho_histogram_dense(const double *v, size_t n, uint64_t *counts)
{
    for (size_t i = 0; i < n; i++) counts[key_to_bin(lho_compress(v[i]))] += 1;
}

int lho_histogram_pairs(const uint32_t *ids, const double *v, size_t n,
                        uint64_t *counts, uint32_t nmetrics)
{
    for (size_t i = 0; i < n; i++) {
        if (ids[i] >= nmetrics) return -1;
        counts[(size_t)ids[i] * LHO_NKEYS + key_to_bin(lho_compress(v[i]))] += 1;
    }
    return 0;
}

void lho_thresholds(double *Tx, size_t n)
{
    if (n == 0) return;
    Tx[0] = 1.0;
    const uint64_t lo_bits0 = f2u(1.0);
    const uint64_t max_bits = f2u(1.7976931348623157e308);
    for (size_t j = 1; j < n; j++) {
        if ((int32_t)j > lho_kext(u2f(max_bits))) { Tx[j] = INFINITY; continue; }
        /* invariant: kext(lo) < j <= kext(hi) */
        uint64_t lo = (j > 1 && isfinite(Tx[j - 1])) ? f2u(Tx[j - 1]) : lo_bits0;
        if (lho_kext(u2f(lo)) >= (int32_t)j) { Tx[j] = u2f(lo); continue; }
        uint64_t hi = max_bits;
        while (hi - lo > 1) {
            uint64_t mid = lo + (hi - lo) / 2;
            if (lho_kext(u2f(mid)) >= (int32_t)j) hi = mid; else lo = mid;
        }
        Tx[j] = u2f(hi);
    }
}

size_t lho_check_monotone(const double *Tx, size_t n, int window)
{
    size_t bad = 0;
    for (size_t j = 1; j < n; j++) {
        if (!isfinite(Tx[j])) continue;
        uint64_t c = f2u(Tx[j]);
        for (int d = -window; d < window; d++) {
            uint64_t b = c + (int64_t)d;
            if (b < f2u(1.0) || b > f2u(1.7976931348623157e308)) continue;
            int32_t k = lho_kext(u2f(b));
            if (d < 0 ? (k >= (int32_t)j) : (k < (int32_t)j)) bad++;
        }
    }
    return bad;
}

void lho_decompress_table(double *D)
{
    for (uint32_t b = 0; b < LHO_NKEYS; b++) D[b] = lho_decompress(bin_to_key(b));
}

/* metrics.go:336-387 on a dense row, canonical (ascending key) order. */
void lho_process_dense(const uint64_t *counts, const double *p, size_t np,
                       lho_stats *st, double *pvals, int16_t *pkeys,
                       uint8_t *pvalid)
{
    double total_sum = 0;
    uint64_t total_count = 0;
    uint32_t nb = 0;
    for (uint32_t b = 0; b < LHO_NKEYS; b++) {
        if (!counts[b]) continue;
        double value = lho_decompress(bin_to_key(b));
        total_sum += value * (double)counts[b];
        total_count += counts[b];
        nb++;
    }
    st->count = total_count;
    st->sum = total_sum;
    st->avg = total_sum / (double)total_count;
    st->agg_sum_add = lho_f64_to_u64_amd64(total_sum);
    st->nbuckets = nb;
    st->reserved = 0;

    /* percentile(), metrics.go:406-418: ascending by Value == ascending key
     * because decompress is strictly monotone; only occupied buckets exist. */
    for (size_t i = 0; i < np; i++) {
        uint64_t sofar = 0;
        int found = 0;
        for (uint32_t b = 0; b < LHO_NKEYS && !found; b++) {
            if (!counts[b]) continue;
            sofar += counts[b];
            if ((double)sofar / (double)total_count >= p[i]) {
                pvals[i] = lho_decompress(bin_to_key(b));
                if (pkeys) pkeys[i] = bin_to_key(b);
                found = 1;
            }
        }
        pvalid[i] = (uint8_t)found;
        if (!found) { pvals[i] = 0; if (pkeys) pkeys[i] = 0; }
    }
}

typedef struct { double v; uint64_t c; } prop_t;
static int prop_cmp(const void *a, const void *b)
{
    double x = ((const prop_t *)a)->v, y = ((const prop_t *)b)->v;
    return (x < y) ? -1 : (x > y);
}

/* metrics.go:406-418 */
int lho_percentile(uint64_t total, const double *values, const uint64_t *counts,
                   size_t n, double p, double *out)
{
    prop_t *a = (prop_t *)malloc((n ? n : 1) * sizeof(prop_t));
    for (size_t i = 0; i < n; i++) { a[i].v = values[i]; a[i].c = counts[i]; }
    qsort(a, n, sizeof(prop_t), prop_cmp);
    uint64_t sofar = 0;
    int rc = -1;
    for (size_t i = 0; i < n; i++) {
        sofar += a[i].c;
        if ((double)sofar / (double)total >= p) { *out = a[i].v; rc = 0; break; }
    }
    if (rc) *out = 0;
    free(a);
    return rc;
}

/* fmt.Sprintf("%f", v), graphite.go:40 / opentsdb.go:48 */
int lho_format_f(double v, char *dst)
{
    if (isnan(v)) { memcpy(dst, "NaN", 4); return 3; }
    if (isinf(v)) { memcpy(dst, v < 0 ? "-Inf" : "+Inf", 5); return 4; }
    return snprintf(dst, 336, "%.6f", v);
}
