// lh_dispatch.cc -- the path choice of lh_dispatch.h.  Plain host C++: the only things it calls are the launch plans'
// own size functions (lh::part*_scratch_bytes, lh::small_supported, lh::part_aligned), which are arithmetic.
#include "lh_dispatch.h"

#include <algorithm>

namespace lh {

bool peel_first(uintptr_t ids, uint32_t id_width, uintptr_t vals, size_t n)
{
    return n > 1 && (ids & (2u * id_width - 1u)) == id_width && (vals & 15u) == 8u;
}

Step choose_step(const DispatchState &st, uintptr_t ids, uint32_t id_width, uintptr_t vals, size_t n, bool host_fed)
{
    Step r;
    r.tune = st.tune;
    Ids I;
    I.p = reinterpret_cast<const void *>(ids);
    I.width = id_width;
    const double *V = reinterpret_cast<const double *>(vals);
    const uint32_t M = st.max_metrics;
    size_t take = std::min(n, kMaxLaunchPairs);
    r.take = take;
    // a handful of names: every workgroup keeps all of them in LDS, one streaming pass
    if (!st.small_disabled && small_supported(take, M, I, V)) {
        r.kind = PATH_SMALL;
        return r;
    }
    const bool aligned = part_aligned(I, V);
    // a lane's half-buffer (read over PCIe in place): first generation -- no tables that outlive the launch -- in a block
    // of the lanes' own, so that its later passes run beside another lane's link-bound read
    if (host_fed && take <= kLaneBlockMaxPairs && !st.lane_blocks) {
        // A lane's half-buffer (read over PCIe in place) by default: the direct path.  Its one pass is link-bound, needs no
        // scratch and no survey, and leaves the GPU to the other lanes' launches: 0.89 - 0.90 of the link at 300 .. 65 536
        // names (round 6; the partitioned launches below held 0.84 - 0.87 up to 8 192 names, 0.75 - 0.78 above).
        r.kind = PATH_DIRECT;
        return r;
    }
    if (host_fed && aligned && take <= kLaneBlockMaxPairs && st.lane_blocks) {
        PartTuning tl = st.tune; // the lanes' own thresholds, unless the caller set them
        if (!tl.part_min_samples) tl.part_min_samples = kLanePartMinPairs;
        if (!tl.v3_min_samples) tl.v3_min_samples = kLaneV3MinPairs;
        // Above 8 192 names the first generation's two scatter levels cost a lane-sized launch more GPU time than its
        // pairs take on the link (65 536 names, 2 M pairs: 0.58 ms against 0.4): the third generation (0.36 ms) runs in the
        // lane's block too, with the survey's tables -- read-only between surveys -- shared by all lanes.
        if (st.lane_gen3 && !st.v3_disabled && !st.regions_disabled) {
            PartTuning t3 = tl;
            t3.v3_log_w = st.call_log_w;
            t3.v3_g1_cap = st.lane_g1_cap;
            const size_t need3 = part3_records_bytes(take, M, st.num_cus, t3);
            if (need3) {
                r.kind = PATH_GEN3;
                r.lane_block = true;
                r.scratch = need3;
                r.scratch_alloc = std::max(need3, part3_records_bytes(std::min(kLaneBlockMaxPairs, std::max(take, st.lane_samples)),
                                                                      M, st.num_cus, t3));
                r.tune = t3;
                return r;
            }
        }
        const size_t need1 = part_scratch_bytes(take, M, st.num_cus, tl);
        if (need1) {
            r.kind = PATH_GEN1;
            r.lane_block = true;
            r.scratch = need1;
            r.scratch_alloc = std::max(need1, part_scratch_bytes(std::min(kLaneBlockMaxPairs, std::max(take, st.lane_samples)),
                                                                 M, st.num_cus, tl));
            r.tune = tl;
            return r;
        }
    }
    if (aligned) {
        // Large launch over many names: partition by name, then reduce in LDS.  Sub-launches keep the scratch block
        // bounded: at most `sublaunch_pairs` pairs each, halved until the block fits `scratch_cap` (power-of-two cuts
        // keep both arrays on their vector alignment).  Above 8 192 names the second scatter level carries its chunk
        // pools and ~0.2 ms of fixed work per launch whatever the launch size, so cutting costs more than the bytes it
        // saves: such launches are cut only when the caller asked for a bound, and then not below 2^28 pairs.
        const bool two_level = M > 8192;
        const bool bounded = !two_level || st.scratch_cap_set || st.sublaunch_set;
        size_t sub = (bounded && take > st.sublaunch_pairs) ? st.sublaunch_pairs : take;
        PartTuning tune = st.tune;
        // wide value spans over few names: 512 partitions of two names x 16 384 bins instead of 256 x four names x 8 192
        if (st.call_wide && M <= 1024u && (tune.v2_shape & 7u) == 2u) tune.v2_shape |= 4u;
        if (st.regions_disabled) tune.v2_shape &= ~6u; // clustered stream: the exact-layout scatter
        if (st.v3_disabled) tune.v3 = false;            // skew-free names: first generation
        tune.v3_log_w = st.call_log_w;
        tune.v2_yield = st.call_yield;
        // generation 2 (survey + 2-byte records) for <= 8 192 names, generation 3 above, when the launch is large enough for
        // it (each generation has its own minimum; an option may lower one below the others'); otherwise the first
        // generation; a launch none of them takes is the direct path's
        auto scratch_need = [&](size_t m, PathKind *kind) -> size_t {
            *kind = PATH_DIRECT;
            if (tune.part_min_samples && m < tune.part_min_samples) return 0; // LH_OPT_PART_MIN_PAIRS bounds every generation
            size_t bytes = part2_scratch_bytes(m, M, st.num_cus, tune);
            *kind = PATH_GEN2;
            if (!bytes) { bytes = part3_scratch_bytes(m, M, st.num_cus, tune); *kind = PATH_GEN3; }
            if (!bytes) { bytes = part_scratch_bytes(m, M, st.num_cus, tune); *kind = PATH_GEN1; }
            return bytes;
        };
        PathKind kind = PATH_GEN1;
        size_t need = scratch_need(sub, &kind);
        const size_t floor = two_level ? (size_t(1) << 28) : (size_t(1) << 24);
        while (bounded && need > st.scratch_cap && sub > floor) {
            size_t half = floor;
            while (half * 2 < sub) half *= 2;
            sub = half;
            need = scratch_need(sub, &kind);
        }
        if (need) {
            r.kind = kind;
            r.take = sub;
            r.scratch = r.scratch_alloc = need;
            r.tune = tune;
            return r;
        }
    }
    r.kind = PATH_DIRECT;
    return r;
}

bool survey_reusable(const SurveyTables &t, int gen, uint32_t layout, uint64_t call_tune_gen, uint32_t survey_every,
                     bool healthy)
{
    if (!t.valid || t.gen != gen || t.log_w != layout || t.tune_gen != call_tune_gen || !healthy) return false;
    return t.age < survey_every;
}

bool names_without_skew(uint64_t pairs, uint64_t forwarded, bool healthy, bool same_width)
{
    return pairs >= (uint64_t(1) << 22) && forwarded * 4 > pairs * 3 && healthy && same_width;
}

} // namespace lh
