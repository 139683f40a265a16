// lh_windows.h -- placement of per-name LDS windows for the mixed-stream kernels.
//
// A kernel that keeps `mpp` names in LDS gives each name a window of W = 2^log_w uint32 bins out of
// the 65 536-bin key space.  The caller first buckets a sample of its input coarsely into
// h[name][bin >> (16 - log_w)] (so W coarse cells span the whole key space); choose_windows then
// picks, per name, the run of coarse cells with the largest mass, centred among ties, and writes
// the window origin (a bin index) to s_org[name].  A badly placed window only costs speed: records
// outside it go to global atomics and stay exact.
//
// Mid-range centring is not enough: a log-uniform stream over 21 decades puts 12 % of its mass on
// keys 0..24, which a window centred on the mid-range excludes (measured: 82 ms instead of 10 ms
// per 1e9 samples).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

// Called by every thread of the workgroup after a __syncthreads(); the caller synchronises again
// afterwards and zeroes h.  One wave handles one name at a time.
__device__ __forceinline__ void choose_windows(uint32_t *h, uint32_t *s_org, uint32_t *s_mn, uint32_t *s_mx,
                                               uint32_t mpp, uint32_t log_w, uint32_t wave, uint32_t lane,
                                               uint32_t nwaves)
{
    const uint32_t W = 1u << log_w, log_cw = 16 - log_w;
    const uint32_t wl = (W >> log_cw) ? (W >> log_cw) : 1u; // coarse cells per window
    const uint32_t S = W >= 64 ? (W >> 6) : 1u;             // coarse cells per lane
    for (uint32_t l = wave; l < mpp; l += nwaves) {
        uint32_t *hl = h + (l << log_w);
        const bool on = lane * S < W;
        // inclusive prefix sums in place (lane-serial segments + wave scan of the segment totals)
        uint32_t run = 0;
        if (on)
            for (uint32_t k = 0; k < S; k++) { run += hl[lane * S + k]; hl[lane * S + k] = run; }
        uint32_t inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d, 64);
            if ((int)lane >= d) inc += y;
        }
        const uint32_t excl = inc - run;
        const uint32_t total = __shfl(inc, 63, 64);
        if (on)
            for (uint32_t k = 0; k < S; k++) hl[lane * S + k] += excl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // best start cell: (mass << 16 | 65535 - s) maximises mass then minimises s; the mirrored
        // packing finds the largest s with the same mass.  The sample must hold < 65 536 records.
        uint32_t best_lo = 0, best_hi = 0;
        if (on)
            for (uint32_t k = 0; k < S; k++) {
                const uint32_t s = lane * S + k;
                if (s + wl <= W) {
                    const uint32_t mass = hl[s + wl - 1] - (s ? hl[s - 1] : 0u);
                    const uint32_t a = (mass << 16) | (65535u - s), b = (mass << 16) | s;
                    best_lo = a > best_lo ? a : best_lo;
                    best_hi = b > best_hi ? b : best_hi;
                }
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t a = __shfl_xor(best_lo, d, 64), b = __shfl_xor(best_hi, d, 64);
            best_lo = a > best_lo ? a : best_lo;
            best_hi = b > best_hi ? b : best_hi;
        }
        if (lane == 0) {
            uint32_t org = 32768u - W / 2; // name absent from the sample: centre on key 0
            if (total) {
                const uint32_t smin = 65535u - (best_lo & 0xffffu), smax = best_hi & 0xffffu;
                const uint32_t centre = ((smin + smax + wl) << log_cw) >> 1;
                org = centre > W / 2 ? centre - W / 2 : 0u;
            }
            if (org > 65536u - W) org = 65536u - W;
            s_org[l] = org;
            s_mn[l] = 0xffffffffu; // flush ranges
            s_mx[l] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// Overflow table: records that fall outside their name's window.
//
// Sending each of them to a global atomic is exact but serialises when a stream's span exceeds the
// window by a little: the excess lands on a handful of adjacent cells, i.e. on a few cache lines
// (measured: 1.2 % of 1e9 log-uniform samples = 15 ms of same-line atomics).  A 512-entry
// open-addressed LDS table keyed by (name << 16 | bin) aggregates them per workgroup; only a record
// that finds its four probe slots taken by other keys goes to the global atomic.  Flushed with the
// windows.
// ---------------------------------------------------------------------------
constexpr uint32_t OV_SLOTS = 512;
constexpr uint32_t OV_EMPTY = 0xffffffffu; // never a real key: names are < 2^16

__device__ __forceinline__ void ov_init(uint32_t *ov_key, uint32_t *ov_cnt, uint32_t tid, uint32_t nthreads)
{
    for (uint32_t i = tid; i < OV_SLOTS; i += nthreads) { ov_key[i] = OV_EMPTY; ov_cnt[i] = 0; }
}

// true if the record was absorbed by the table
__device__ __forceinline__ bool ov_add(uint32_t *ov_key, uint32_t *ov_cnt, uint32_t key, uint32_t c)
{
    const uint32_t h0 = (key * 2654435761u) >> 23; // 9 bits
#pragma unroll
    for (uint32_t probe = 0; probe < 4; probe++) {
        const uint32_t s = (h0 + probe) & (OV_SLOTS - 1);
        const uint32_t prev = atomicCAS(&ov_key[s], OV_EMPTY, key);
        if (prev == OV_EMPTY || prev == key) {
            atomicAdd(&ov_cnt[s], c);
            return true;
        }
    }
    return false;
}

} // namespace lh
