// dispatch_test.cc -- the pure decision functions of loghisto_amd/csrc/lh_dispatch.h, table-tested without a device:
// survey reuse (second and third generation), the skew-free-names switch, the health share, the peeled first sample,
// and choose_step's own invariants over a grid of states.  Built by loghisto_amd/build.py against liblhgpu.so (the
// launch plans' size functions live there); run by tests/test_dispatch.py on the CPU box.
#include "../../loghisto_amd/csrc/lh_dispatch.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_checks = 0, g_failed = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        g_checks++;                                                                   \
        if (!(cond)) { g_failed++; std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static void test_survey_reuse()
{
    // every combination of: valid, table generation, call generation, layout equal, option generation equal, healthy, age
    for (int valid = 0; valid < 2; valid++)
        for (int tgen = 2; tgen <= 3; tgen++)
            for (int gen = 2; gen <= 3; gen++)
                for (int same_layout = 0; same_layout < 2; same_layout++)
                    for (int same_tune = 0; same_tune < 2; same_tune++)
                        for (int healthy = 0; healthy < 2; healthy++)
                            for (uint32_t every : {1u, 2u, 8u, 32u})
                                for (uint32_t age : {1u, 2u, 7u, 8u, 31u, 32u, 33u}) {
                                    lh::SurveyTables t;
                                    t.valid = valid;
                                    t.gen = tgen;
                                    t.log_w = 11;
                                    t.age = age;
                                    t.tune_gen = 5;
                                    const bool got = lh::survey_reusable(t, gen, same_layout ? 11u : 12u, same_tune ? 5u : 6u, every, healthy);
                                    const bool want = valid && tgen == gen && same_layout && same_tune && healthy && age < every;
                                    CHECK(got == want);
                                }
    // a survey serves `every` calls in all: the one that ran it (age 1) and every - 1 after it
    lh::SurveyTables t;
    t.valid = true; t.gen = 3; t.log_w = 10; t.tune_gen = 0; t.age = 1;
    uint32_t served = 1;
    while (lh::survey_reusable(t, 3, 10, 0, 32, true)) { t.age++; served++; }
    CHECK(served == 32);
}

static void test_names_without_skew()
{
    const uint64_t M = uint64_t(1) << 22;
    CHECK(!lh::names_without_skew(M - 1, M - 1, true, true));          // too few pairs to judge
    CHECK(lh::names_without_skew(M, M * 3 / 4 + 1, true, true));       // more than 3/4 forwarded
    CHECK(!lh::names_without_skew(M, M * 3 / 4, true, true));          // exactly 3/4 is not "more than"
    CHECK(!lh::names_without_skew(M, M, false, true));                 // an unhealthy call (stale survey) is not judged
    CHECK(!lh::names_without_skew(M, M, true, false));                 // nor one right after a width change
    CHECK(!lh::names_without_skew(100 * M, 21 * M, true, true));       // Zipf(1): 21 % forwarded
    CHECK(lh::names_without_skew(100 * M, 88 * M, true, true));        // uniform names: 88 %
}

static void test_health_and_peel()
{
    CHECK(lh::healthy_share(0, 0) && lh::healthy_share(2, 100) && !lh::healthy_share(3, 100) && !lh::healthy_share(1, 0));
    // peel: both arrays exactly one element short of their vector alignment, and more than one pair
    for (uint32_t w : {2u, 4u})
        for (uintptr_t io = 0; io < 16; io += w)
            for (uintptr_t vo = 0; vo < 32; vo += 8)
                for (size_t n : {size_t(1), size_t(2), size_t(1000)}) {
                    const bool want = n > 1 && (io % (2 * w)) == w && (vo % 16) == 8;
                    CHECK(lh::peel_first(0x1000 + io, w, 0x2000 + vo, n) == want);
                }
}

static void test_choose_step_grid()
{
    const uint32_t names[] = {1, 2, 32, 33, 1000, 8192, 8193, 40000, 65536, 65537};
    const size_t sizes[] = {1, 1000, 65535, 65536, 131071, 131072, (size_t(1) << 18) - 1, size_t(1) << 18, (size_t(1) << 20) - 2,
                            size_t(1) << 20, (size_t(3) << 20) - 2, size_t(3) << 20, size_t(1) << 22,
                            (size_t(1) << 22) + 2, size_t(1) << 25, size_t(125000000), size_t(1000000000), size_t(3) << 30};
    int states = 0;
    for (uint32_t M : names)
        for (size_t n0 : sizes)
            for (int hf = 0; hf < 3; hf++) // device-resident; host-fed with lane blocks; host-fed without (the default)
                for (uint32_t width : {2u, 4u})
                    for (int flags = 0; flags < 8; flags++)         // small_disabled | regions_disabled << 1 | v3_disabled << 2
                        for (int bound = 0; bound < 3; bound++) {   // no bound, scratch cap 256 MiB, sub-launches of 2^24
                            if (width == 2 && M > 65536) continue;
                            const int host_fed = hf != 0;
                            lh::DispatchState st;
                            st.max_metrics = M;
                            st.lane_samples = size_t(1) << 20;
                            st.lane_blocks = hf == 1 ? 8 : 0;
                            st.small_disabled = flags & 1;
                            st.regions_disabled = flags & 2;
                            st.v3_disabled = flags & 4;
                            if (bound == 1) { st.scratch_cap = size_t(256) << 20; st.scratch_cap_set = true; }
                            if (bound == 2) { st.sublaunch_pairs = size_t(1) << 24; st.sublaunch_set = true; }
                            states++;
                            uintptr_t ids = 0x100000, vals = 0x800000;
                            size_t n = n0, steps = 0;
                            while (n) {
                                const lh::Step s = lh::choose_step(st, ids, width, vals, n, host_fed);
                                CHECK(s.take >= 1 && s.take <= n && s.take <= lh::kMaxLaunchPairs);
                                if (s.take == 0 || s.take > n) return;
                                CHECK((s.scratch != 0) == (s.kind >= lh::PATH_GEN1));
                                CHECK(s.scratch_alloc >= s.scratch);
                                if (s.kind == lh::PATH_SMALL) CHECK(M <= 32 && s.take >= 65536 && !st.small_disabled);
                                // (few names reach the partitioned paths only while adaptive dispatch has the single pass off)
                                // round 6: a device-resident launch is partitioned from 2^20 pairs (<= 8 192 names; the second
                                // generation at once) / 3 * 2^20 (above); below, the direct path's cell table is faster.  A
                                // lane's half-buffer keeps 2^17 / 2^18.
                                const size_t min12 = s.lane_block ? lh::kLanePartMinPairs : M > 8192 ? size_t(3) << 20 : size_t(1) << 20;
                                const size_t min3 = s.lane_block ? lh::kLaneV3MinPairs : size_t(3) << 20;
                                if (s.kind == lh::PATH_GEN2) CHECK(M >= 2 && M <= 8192 && (M >= 33 || st.small_disabled) && s.take >= min12 && !s.lane_block);
                                if (s.kind == lh::PATH_GEN3)
                                    CHECK(M >= 8193 && M <= 65536 && s.take >= min3 && !st.v3_disabled && !st.regions_disabled);
                                if (s.kind == lh::PATH_GEN1) CHECK(M >= 2 && M <= 65536 && (M >= 33 || st.small_disabled) && s.take >= min12);
                                // ... and takes the second generation whenever it is partitioned at all (the first only on a
                                // lane's block, or when an adaptive switch sent it there)
                                if (!s.lane_block && s.kind == lh::PATH_GEN1 && M <= 8192) CHECK(!"the first generation below 8 193 names");
                                if (!host_fed && M >= 33 && M <= 8192 && s.take >= (size_t(1) << 20)) CHECK(s.kind == lh::PATH_GEN2);
                                if (!host_fed && M >= 8193 && M <= 65536 && s.take >= (size_t(3) << 20) && !st.v3_disabled && !st.regions_disabled)
                                    CHECK(s.kind == lh::PATH_GEN3);
                                if (!host_fed && M >= 33 && M <= 65536 && s.take < (M > 8192 ? size_t(3) << 20 : size_t(1) << 20))
                                    CHECK(s.kind == lh::PATH_DIRECT);
                                if (s.kind == lh::PATH_DIRECT) CHECK(s.take == (n < lh::kMaxLaunchPairs ? n : lh::kMaxLaunchPairs));
                                // a lane's block: the first generation up to 8 192 names, the third above (while neither
                                // adaptive switch has it off)
                                if (s.lane_block)
                                    CHECK(host_fed && s.take <= lh::kLaneBlockMaxPairs &&
                                          (s.kind == lh::PATH_GEN1 || (s.kind == lh::PATH_GEN3 && M > 8192)));
                                if (s.lane_block && M > 8192 && s.take >= (size_t(1) << 18) && !st.v3_disabled && !st.regions_disabled)
                                    CHECK(s.kind == lh::PATH_GEN3);
                                if (hf == 1 && s.kind >= lh::PATH_GEN1 && s.take <= lh::kLaneBlockMaxPairs) CHECK(s.lane_block);
                                // without blocks of their own (the default since round 6) a half-buffer takes the direct path
                                if (hf == 2 && n <= lh::kLaneBlockMaxPairs) CHECK(s.kind == lh::PATH_DIRECT || s.kind == lh::PATH_SMALL);
                                if (hf == 2) CHECK(!s.lane_block);
                                // a bounded block: above the floor no sub-launch asks for more than the cap
                                if (bound == 1 && s.kind >= lh::PATH_GEN1 && !s.lane_block && s.take > (M > 8192 ? size_t(1) << 28 : size_t(1) << 24))
                                    CHECK(s.scratch <= st.scratch_cap);
                                if (bound == 2 && s.kind >= lh::PATH_GEN1 && !s.lane_block) CHECK(s.take <= st.sublaunch_pairs);
                                // what the tuning handed to the launch wrapper says about the adaptive switches
                                if (s.kind >= lh::PATH_GEN1 && !s.lane_block) {
                                    CHECK(!st.regions_disabled || !(s.tune.v2_shape & 2u));
                                    CHECK(!st.v3_disabled || !s.tune.v3);
                                }
                                ids += s.take * width;
                                vals += s.take * 8;
                                n -= s.take;
                                if (++steps > 4096) { CHECK(!"choose_step does not terminate"); return; }
                            }
                        }
    std::printf("choose_step: %d states\n", states);
    CHECK(states >= 500);
}

int main()
{
    test_survey_reuse();
    test_names_without_skew();
    test_health_and_peel();
    test_choose_step_grid();
    std::printf("dispatch_test: %d checks, %d failed\n", g_checks, g_failed);
    return g_failed ? 1 : 0;
}
