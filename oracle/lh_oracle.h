/*
 * lh_oracle.h -- CPU oracle for the loghisto hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the one path
 * this repository accelerates:
 *
 *   compress / decompress           /root/reference/metrics.go:316-332
 *   Histogram fan-in (+1 per cell)  /root/reference/metrics.go:273-295
 *   processHistograms               /root/reference/metrics.go:336-387
 *   percentile                      /root/reference/metrics.go:391-418
 *
 * plus the Go standard-library arithmetic those lines call (math.Log, math.Exp,
 * math.Abs, float64->int16 and float64->uint64 conversions with amd64
 * semantics).  The Go standard library is NOT under /root/reference and the
 * reference pins no Go version (no go.mod; .travis.yml lists go 1.4 and tip);
 * the algorithms restated here are the portable math/log.go and math/exp.go
 * (FreeBSD msun derived), evaluated without FMA contraction, as SURVEY.md
 * Appendix A records them.
 *
 * Pinning status (see also DESIGN.md "Oracle"):
 *   - decompress: PINNED bit-for-bit by the 15 full-precision values the
 *     reference prints in readme.md:35-43 and print_benchmark.go:34-39
 *     (tests/golden/decompress_doc_goldens.json).
 *   - percentile: PINNED by TestPercentile (metrics_test.go:111-149).
 *   - processHistograms: PINNED at integer granularity by
 *     TestProcessedBroadcast (metrics_test.go:289-319).
 *   - compress: pinned only to 1 % by TestCompress (metrics_test.go:151-172);
 *     at threshold/ulp granularity it is PARITY UNPINNED -- no reference test,
 *     fixture or document distinguishes Go's math.Log from any other log that
 *     is good to a few ulp, and no Go toolchain exists in this environment to
 *     run the reference.  The oracle therefore DEFINES bucket parity as
 *     "math/log.go algorithm, amd64 non-fused evaluation".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this library.  The product (loghisto_amd/) never does.
 */
#ifndef LH_ORACLE_H
#define LH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LHO_NKEYS 65536          /* int16 key space, metrics.go:316 */
#define LHO_KEXT_MAX 70978       /* floor(100*ln(MaxFloat64)+0.5)   */

/* Go math.Log / math.Exp (portable algorithms, no FMA). */
double lho_go_log(double x);
double lho_go_exp(double x);

/* metrics.go:316-322.  amd64 conversion semantics for out-of-range values. */
int16_t lho_compress(double value);
/* metrics.go:326-332 */
double lho_decompress(int16_t c);

/* floor(100*Log(x)+0.5) as a wide integer for x>=1 finite ("extended key",
 * before the int16 wrap); -1 for NaN/Inf.  Used to build threshold tables. */
int32_t lho_kext(double x);

void lho_compress_many(const double *v, size_t n, int16_t *out);
void lho_kext_many(const double *x, size_t n, int32_t *out);

/* Dense fan-in: counts[bin] += 1 with bin = (uint16)key ^ 0x8000, so ascending
 * bin == ascending key.  metrics.go:273-295 (dense restatement of the map). */
void lho_histogram_dense(const double *v, size_t n, uint64_t *counts /*[65536]*/);
/* Mixed stream: counts[ids[i]*65536 + bin] += 1. ids must be < nmetrics. */
int lho_histogram_pairs(const uint32_t *ids, const double *v, size_t n,
                        uint64_t *counts, uint32_t nmetrics);

/* The same over `threads` slices of the stream adding atomically into ONE matrix (lh_cpu_baseline.cc):
 * full-size (1e9-pair) parity checks.  Integer sums commute: bit-identical to lho_histogram_pairs. */
int lho_histogram_pairs_mt(const uint32_t *ids, const double *v, size_t n, uint64_t *counts,
                           uint32_t nmetrics, int threads);

/* Threshold table in x = 1+|v| space: Tx[j] = smallest double x >= 1 with
 * lho_kext(x) >= j, for j = 0..n-1 (Tx[0] = 1.0; +Inf where unreachable).
 * Found by bisection on the bit pattern; lho_check_monotone verifies the
 * monotonicity that makes the table equivalent to the function. */
void lho_thresholds(double *Tx, size_t n);
/* Returns the number of monotonicity violations of lho_kext within +-window
 * ulps of every threshold j in [1, n). 0 == table lookup is exact. */
size_t lho_check_monotone(const double *Tx, size_t n, int window);
/* D[bin] = decompress(key(bin)) for all 65536 bins. */
void lho_decompress_table(double *D /*[65536]*/);

typedef struct {
    uint64_t count;      /* totalCount                metrics.go:345 */
    double   sum;        /* totalSum, ascending key   metrics.go:344 */
    double   avg;        /* sum / float64(count)      metrics.go:356 */
    uint64_t agg_sum_add;/* uint64(totalSum), amd64   metrics.go:374 */
    uint32_t nbuckets;   /* occupied buckets                          */
    uint32_t reserved;
} lho_stats;

/* processHistograms on a dense row.  pvals[i]/pvalid[i] receive the i-th
 * percentile (value is always some decompress(k); pvalid=0 when the reference
 * would return its "Invalid percentile" error, metrics.go:417).  pkeys (may be
 * NULL) receives the selected int16 key. */
void lho_process_dense(const uint64_t *counts /*[65536]*/, const double *p,
                       size_t np, lho_stats *st, double *pvals, int16_t *pkeys,
                       uint8_t *pvalid);

/* The reference's percentile() verbatim in behaviour, on arbitrary
 * (value,count) pairs (sorts a private copy ascending by value).
 * Returns 0 on success, -1 for the "Invalid percentile" error. */
int lho_percentile(uint64_t total, const double *values, const uint64_t *counts,
                   size_t n, double p, double *out);

/* uint64(float64) with amd64 CVTTSD2SQ semantics (metrics.go:374). */
uint64_t lho_f64_to_u64_amd64(double f);

/* Go's fmt "%f" of a float64 (graphite.go:40, opentsdb.go:48): strconv.FormatFloat(v, 'f', 6, 64) --
 * the exact decimal expansion of the binary value rounded half-even to 6 fractional digits -- with Go's
 * spellings "NaN", "+Inf", "-Inf".  C99 printf("%.6f") under round-to-nearest has the same contract and
 * glibc implements it exactly; tests/test_oracle.py cross-checks it against Python's decimal module.
 * PARITY UNPINNED by the reference: its tests hold no formatted golden (graphite_test.go and
 * opentsdb_test.go only open a socket).  dst needs 336 bytes; returns the length (no NUL counted). */
int lho_format_f(double v, char *dst);

#ifdef __cplusplus
}
#endif
#endif
