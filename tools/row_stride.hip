// row_stride.hip -- does the 512-KiB row stride of the epoch buffers cost bandwidth?  (VERDICT r4 next #1a)
//
// Every kernel that walks the 65 536-name row store (k_extract_wave, k_pack_rows / k_unpack_rows, k_clear_rows_wave, the
// reduce pass's flush) touches one ~600-cell window per row.  In the product a row is uint64[65536] at a 512-KiB stride,
// and rows of names with similar value distributions keep their windows at the SAME offset inside the row: all the
// windows then share their low 19 address bits.  This tool runs the three access shapes -- read one window per wave,
// zero it, flush 43 % of its cells with uint64 atomics from a 32-row workgroup -- over the same 65 536 windows laid out
// at several strides and prints one JSON line per (shape, stride): average / minimum launch time and GB/s of window bytes.
//
// Round 6: the same shapes with uint32 CELLS (--cell 4: a 600-cell window is 38 lines instead of 75),
// plus `pack` (the merge's k_pack_rows: the window read and written out as packed uint32 words) -- would a row store of
// 32-bit cells above 8 192 names pay?  profiles/r06_cells32.txt.
//
//   hipcc -O3 --offload-arch=gfx950 tools/row_stride.hip -o loghisto_amd/build/row_stride
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// window of row r: cells [lo(r), lo(r) + width) of the row at rows + r * stride
__device__ __forceinline__ uint32_t win_lo(uint32_t r, uint32_t lo0, uint32_t jitter) { return lo0 + (r * 2654435761u >> 24) % (jitter + 1u); }

// one wave per row: the window summed (k_extract_wave's / k_pack_rows' read)
template <typename CT>
__global__ __launch_bounds__(256) void k_read(const CT *__restrict__ rows, size_t stride, uint32_t nrows,
                                               uint32_t lo0, uint32_t jitter, uint32_t width, unsigned long long *__restrict__ out)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= nrows) return;
    const CT *row = rows + (size_t)r * stride + win_lo(r, lo0, jitter);
    unsigned long long acc = 0;
    constexpr uint32_t PER = 16 / sizeof(CT); // 16 bytes per lane per step
    for (uint32_t i = lane * PER; i < width; i += 64u * PER) {
#pragma unroll
        for (uint32_t k = 0; k < PER; k++)
            if (i + k < width) acc += row[i + k];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) out[r] = acc;
}

// one wave per row: the window zeroed (k_clear_rows_wave)
template <typename CT>
__global__ __launch_bounds__(256) void k_clear(CT *__restrict__ rows, size_t stride, uint32_t nrows,
                                                uint32_t lo0, uint32_t jitter, uint32_t width)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= nrows) return;
    CT *row = rows + (size_t)r * stride + win_lo(r, lo0, jitter);
    for (uint32_t i = lane; i < width; i += 64u) row[i] = 0;
}

// one wave per row: the window read and written out as packed uint32 words (k_pack_rows)
template <typename CT>
__global__ __launch_bounds__(256) void k_pack(const CT *__restrict__ rows, size_t stride, uint32_t nrows, uint32_t lo0,
                                               uint32_t jitter, uint32_t width, uint32_t *__restrict__ packed)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= nrows) return;
    const CT *row = rows + (size_t)r * stride + win_lo(r, lo0, jitter);
    uint32_t *dst = packed + (size_t)r * width;
    for (uint32_t i = lane; i < width; i += 64u) dst[i] = (uint32_t)row[i];
}

// one 1 024-thread workgroup per 32 rows (names p, p + nslots, p + 2 nslots, ...: a fine partition of the reduce pass):
// `pct` % of the cells of every window get one update (k_part_hist3's flush).  MODE: 0 uint64 atomic, no return (the
// product's flush); 1 uint32 atomic on the low word; 2 the cell's line read by a plain load first (issued for all of a
// thread's cells before its atomics); 3 plain load + add + store (no atomic: only valid for a sole writer); 4 uint64
// atomic WITH return; 5 one uint64 atomic per LINE (lane 0 of every 8 cells adds the line's count to its first cell:
// the cost of a line transaction without the per-cell work)
template <int MODE, typename CT>
__global__ __launch_bounds__(1024) void k_flush(CT *__restrict__ rows, size_t stride, uint32_t nrows,
                                                 uint32_t lo0, uint32_t jitter, uint32_t width, uint32_t pct,
                                                 unsigned long long *__restrict__ sink)
{
    const uint32_t nslots = nrows / 32u, p = blockIdx.x;
    unsigned long long acc = 0;
    if (MODE == 2) {
        for (uint32_t i = threadIdx.x; i < 32u * 1024u; i += 1024u) {
            const uint32_t l = i >> 10, b = i & 1023u, r = l * nslots + p;
            if (b >= width || r >= nrows || (b & (64u / sizeof(CT) - 1u))) continue; // one load per line
            acc += __builtin_nontemporal_load(&rows[(size_t)r * stride + win_lo(r, lo0, jitter) + b]);
        }
    }
    for (uint32_t i = threadIdx.x; i < 32u * 1024u; i += 1024u) {
        const uint32_t l = i >> 10, b = i & 1023u, r = l * nslots + p;
        if (b >= width || r >= nrows) continue;
        const uint32_t h = (r * 1024u + b) * 2654435761u;
        CT *cell = &rows[(size_t)r * stride + win_lo(r, lo0, jitter) + b];
        if (MODE == 5) {
            if ((b & (64u / sizeof(CT) - 1u)) == 0) atomicAdd(cell, (CT)3);
            continue;
        }
        if ((h >> 16) % 100u >= pct) continue;
        if (MODE == 0 || MODE == 2) atomicAdd(cell, (CT)1);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int *>(cell), 1u);
        else if (MODE == 3) *cell = *cell + (CT)1;
        else if (MODE == 4) acc += atomicAdd(cell, (CT)1);
    }
    if (acc == 0x123456789abcdefull) sink[blockIdx.x] = acc; // (never true: keeps the loads and the returns)
}

template <typename CT>
static void run(uint32_t nrows, uint32_t width, uint32_t lo0, uint32_t jitter, uint32_t pct, int reps)
{
    struct Case { const char *name; size_t stride; uint32_t lo; };
    const size_t packed = (size_t)width + jitter + 8;
    const Case cases[] = {
        {"512KiB (product)", 65536, lo0},
        {"512KiB+256B", 65536 + 32, lo0},
        {"64KiB", 8192, lo0 & 8191u},
        {"packed", packed, 0},
    };
    unsigned long long *out;
    uint32_t *pk;
    CHECK(hipMalloc(&out, (size_t)nrows * 8));
    CHECK(hipMalloc(&pk, (size_t)nrows * width * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double win_bytes = (double)nrows * width * sizeof(CT);
    for (const Case &c : cases) {
        CT *rows;
        const size_t bytes = ((size_t)nrows * c.stride + 65536) * sizeof(CT);
        if (hipMalloc(&rows, bytes) != hipSuccess) { fprintf(stderr, "skip %s: no memory\n", c.name); (void)hipGetLastError(); continue; }
        CHECK(hipMemset(rows, 0, bytes));
        for (int shape = 0; shape < 9; shape++) {
            if (shape == 3 && sizeof(CT) == 4) continue; // (the uint32 atomic on a uint64 cell's low word)
            std::vector<float> ms;
            for (int r = 0; r < reps + 3; r++) {
                CHECK(hipEventRecord(a, 0));
                const dim3 gw((nrows + 3) / 4), gf(nrows / 32);
                if (shape == 0) hipLaunchKernelGGL(k_read<CT>, gw, dim3(256), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, out);
                else if (shape == 1) hipLaunchKernelGGL(k_clear<CT>, gw, dim3(256), 0, 0, rows, c.stride, nrows, c.lo, jitter, width);
                else if (shape == 2) hipLaunchKernelGGL((k_flush<0, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 3) hipLaunchKernelGGL((k_flush<1, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 4) hipLaunchKernelGGL((k_flush<2, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 5) hipLaunchKernelGGL((k_flush<3, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 6) hipLaunchKernelGGL((k_flush<4, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 7) hipLaunchKernelGGL((k_flush<5, CT>), gf, dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else hipLaunchKernelGGL(k_pack<CT>, gw, dim3(256), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pk);
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
            static const char *kShape[9] = {"read", "clear", "flush_atomic", "flush_atomic_u32_of_u64", "flush_line_read_then_atomic",
                                            "flush_plain_rmw", "flush_atomic_returning", "flush_one_atomic_per_line", "pack_to_u32"};
            const double moved = (shape >= 2 && shape <= 7) ? win_bytes * pct / 100.0 : win_bytes;
            printf("{\"tool\": \"row_stride\", \"cell_bytes\": %zu, \"shape\": \"%s\", \"layout\": \"%s\", \"stride_cells\": %zu, \"rows\": %u, "
                   "\"width\": %u, \"avg_us\": %.1f, \"min_us\": %.1f, \"window_GBs\": %.0f%s}\n",
                   sizeof(CT), kShape[shape], c.name, c.stride, nrows, width, avg * 1e3, mn * 1e3,
                   moved / (avg * 1e-3) / 1e9, (shape >= 2 && shape <= 7) ? ", \"note\": \"GB/s of the cells that get an update\"" : "");
            fflush(stdout);
        }
        CHECK(hipFree(rows));
    }
    CHECK(hipFree(out));
    CHECK(hipFree(pk));
}

int main(int argc, char **argv)
{
    uint32_t nrows = 65536, width = 600, lo0 = 33500, jitter = 16, pct = 43, cell = 8;
    int reps = 10;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--rows")) nrows = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--width")) width = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--jitter")) jitter = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--pct")) pct = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--cell")) cell = (uint32_t)atol(argv[i + 1]);
    }
    if (cell == 4) run<uint32_t>(nrows, width, lo0, jitter, pct, reps);
    else run<unsigned long long>(nrows, width, lo0, jitter, pct, reps);
    return 0;
}
