// read_ceiling.hip -- what this MI355X delivers to a kernel that ONLY reads: the ceiling K1 (k_ingest_single) and
// the level-1 scatter kernels are quoted against beside the 8 TB/s spec.
//
// Same access pattern as K1: persistent workgroups, grid-stride over tiles, UNROLL x 16-byte non-temporal loads per
// lane in flight; the values are XOR-folded and one word per workgroup is written at the end.  Prints one JSON line
// per (block, workgroups per CU, unroll) shape: average / minimum launch time over --reps launches of --bytes bytes.
//
//   hipcc -O3 --offload-arch=gfx950 tools/read_ceiling.hip -o loghisto_amd/build/read_ceiling
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef double d2_t __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void k_read(const d2_t *__restrict__ p, size_t npair, unsigned long long *__restrict__ out)
{
    const size_t tile = (size_t)BLOCK * UNROLL, nfull = npair / tile;
    unsigned long long acc = 0;
    for (size_t t = blockIdx.x; t < nfull; t += gridDim.x) {
        const d2_t *q = p + t * tile + threadIdx.x;
        d2_t r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) r[u] = NT ? __builtin_nontemporal_load(q + u * BLOCK) : q[u * BLOCK];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            acc ^= (unsigned long long)__double_as_longlong(r[u].x) ^ (unsigned long long)__double_as_longlong(r[u].y);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc ^= __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc == 0x123456789abcdefull) out[blockIdx.x] = acc; // (never true: keeps the loads)
}

template <int BLOCK, int UNROLL, bool NT>
static void run(const d2_t *d, size_t npair, unsigned long long *out, int wg_per_cu, int cus, int reps, size_t bytes)
{
    const unsigned grid = (unsigned)(cus * wg_per_cu);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int r = 0; r < reps + 3; r++) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_read<BLOCK, UNROLL, NT>), dim3(grid), dim3(BLOCK), 0, 0, d, npair, out);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        if (r >= 3) ms.push_back(t);
    }
    double avg = 0;
    for (float t : ms) avg += t;
    avg /= ms.size();
    const float mn = *std::min_element(ms.begin(), ms.end());
    printf("{\"kernel\": \"pure read\", \"block\": %d, \"workgroups_per_cu\": %d, \"loads_in_flight_per_lane\": %d, "
           "\"nontemporal\": %s, \"bytes\": %zu, \"avg_ms\": %.4f, \"min_ms\": %.4f, \"avg_GBps\": %.1f, \"best_GBps\": %.1f, "
           "\"frac_of_8TBps\": %.4f}\n",
           BLOCK, wg_per_cu, UNROLL, NT ? "true" : "false", bytes, avg, mn, bytes / avg / 1e6, bytes / mn / 1e6,
           bytes / avg / 1e6 / 8000.0);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    size_t bytes = (size_t)8e9;
    int reps = 20;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--bytes")) bytes = (size_t)atof(argv[i + 1]);
        if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
    }
    bytes &= ~(size_t)((1 << 20) - 1);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    d2_t *d;
    unsigned long long *out;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(d, 0x3f, bytes));
    CHECK(hipDeviceSynchronize());
    const size_t npair = bytes / 16;
    run<512, 8, true>(d, npair, out, 2, cus, reps, bytes);   // K1's shape
    run<512, 8, false>(d, npair, out, 2, cus, reps, bytes);
    run<512, 4, true>(d, npair, out, 2, cus, reps, bytes);
    run<512, 16, true>(d, npair, out, 2, cus, reps, bytes);
    run<256, 8, true>(d, npair, out, 4, cus, reps, bytes);
    run<256, 8, true>(d, npair, out, 8, cus, reps, bytes);
    run<1024, 8, true>(d, npair, out, 1, cus, reps, bytes);  // the scatter kernels' shape
    run<1024, 4, true>(d, npair, out, 2, cus, reps, bytes);
    run<512, 8, true>(d, npair, out, 4, cus, reps, bytes);
    CHECK(hipFree(d));
    CHECK(hipFree(out));
    return 0;
}
