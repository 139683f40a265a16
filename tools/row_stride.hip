// row_stride.hip -- does the 512-KiB row stride of the epoch buffers cost bandwidth?  (VERDICT r4 next #1a)
//
// Every kernel that walks the 65 536-name row store (k_extract_wave, k_pack_rows / k_unpack_rows, k_clear_rows_wave, the
// reduce pass's flush) touches one ~600-cell window per row.  In the product a row is uint64[65536] at a 512-KiB stride,
// and rows of names with similar value distributions keep their windows at the SAME offset inside the row: all the
// windows then share their low 19 address bits.  This tool runs the three access shapes -- read one window per wave,
// zero it, flush 43 % of its cells with uint64 atomics from a 32-row workgroup -- over the same 65 536 windows laid out
// at several strides and prints one JSON line per (shape, stride): average / minimum launch time and GB/s of window bytes.
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
__global__ __launch_bounds__(256) void k_read(const unsigned long long *__restrict__ rows, size_t stride, uint32_t nrows,
                                               uint32_t lo0, uint32_t jitter, uint32_t width, unsigned long long *__restrict__ out)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= nrows) return;
    const unsigned long long *row = rows + (size_t)r * stride + win_lo(r, lo0, jitter);
    unsigned long long acc = 0;
    for (uint32_t i = lane * 2u; i < width; i += 128u) { // 16 bytes per lane per step
        acc += row[i];
        if (i + 1u < width) acc += row[i + 1u];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if (lane == 0) out[r] = acc;
}

// one wave per row: the window zeroed (k_clear_rows_wave)
__global__ __launch_bounds__(256) void k_clear(unsigned long long *__restrict__ rows, size_t stride, uint32_t nrows,
                                                uint32_t lo0, uint32_t jitter, uint32_t width)
{
    const uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (r >= nrows) return;
    unsigned long long *row = rows + (size_t)r * stride + win_lo(r, lo0, jitter);
    for (uint32_t i = lane; i < width; i += 64u) row[i] = 0ull;
}

// one 1 024-thread workgroup per 32 rows (names p, p + nslots, p + 2 nslots, ...: a fine partition of the reduce pass):
// `pct` % of the cells of every window get one update (k_part_hist3's flush).  MODE: 0 uint64 atomic, no return (the
// product's flush); 1 uint32 atomic on the low word; 2 the cell's line read by a plain load first (issued for all of a
// thread's cells before its atomics); 3 plain load + add + store (no atomic: only valid for a sole writer); 4 uint64
// atomic WITH return; 5 one uint64 atomic per LINE (lane 0 of every 8 cells adds the line's count to its first cell:
// the cost of a line transaction without the per-cell work)
template <int MODE>
__global__ __launch_bounds__(1024) void k_flush(unsigned long long *__restrict__ rows, size_t stride, uint32_t nrows,
                                                 uint32_t lo0, uint32_t jitter, uint32_t width, uint32_t pct,
                                                 unsigned long long *__restrict__ sink)
{
    const uint32_t nslots = nrows / 32u, p = blockIdx.x;
    unsigned long long acc = 0;
    if (MODE == 2) {
        for (uint32_t i = threadIdx.x; i < 32u * 1024u; i += 1024u) {
            const uint32_t l = i >> 10, b = i & 1023u, r = l * nslots + p;
            if (b >= width || r >= nrows || (b & 7u)) continue; // one load per line
            acc += __builtin_nontemporal_load(&rows[(size_t)r * stride + win_lo(r, lo0, jitter) + b]);
        }
    }
    for (uint32_t i = threadIdx.x; i < 32u * 1024u; i += 1024u) {
        const uint32_t l = i >> 10, b = i & 1023u, r = l * nslots + p;
        if (b >= width || r >= nrows) continue;
        const uint32_t h = (r * 1024u + b) * 2654435761u;
        unsigned long long *cell = &rows[(size_t)r * stride + win_lo(r, lo0, jitter) + b];
        if (MODE == 5) {
            if ((b & 7u) == 0) atomicAdd(cell, 3ull);
            continue;
        }
        if ((h >> 16) % 100u >= pct) continue;
        if (MODE == 0 || MODE == 2) atomicAdd(cell, 1ull);
        else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int *>(cell), 1u);
        else if (MODE == 3) *cell = *cell + 1ull;
        else if (MODE == 4) acc += atomicAdd(cell, 1ull);
    }
    if (acc == 0x123456789abcdefull) sink[blockIdx.x] = acc; // (never true: keeps the loads and the returns)
}

int main(int argc, char **argv)
{
    uint32_t nrows = 65536, width = 600, lo0 = 33500, jitter = 16, pct = 43;
    int reps = 10;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--rows")) nrows = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--width")) width = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--jitter")) jitter = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--pct")) pct = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[i + 1]);
    }
    struct Case { const char *name; size_t stride; uint32_t lo; };
    const size_t packed = (size_t)width + jitter + 8;
    const Case cases[] = {
        {"512KiB (product)", 65536, lo0},
        {"512KiB+256B", 65536 + 32, lo0},
        {"64KiB", 8192, lo0 & 8191u},
        {"packed", packed, 0},
    };
    unsigned long long *out;
    CHECK(hipMalloc(&out, (size_t)nrows * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    const double win_bytes = (double)nrows * width * 8.0;
    for (const Case &c : cases) {
        unsigned long long *rows;
        const size_t bytes = ((size_t)nrows * c.stride + 65536) * 8;
        if (hipMalloc(&rows, bytes) != hipSuccess) { fprintf(stderr, "skip %s: no memory\n", c.name); (void)hipGetLastError(); continue; }
        for (int shape = 0; shape < 8; shape++) {
            std::vector<float> ms;
            for (int r = 0; r < reps + 3; r++) {
                CHECK(hipEventRecord(a, 0));
                if (shape == 0) hipLaunchKernelGGL(k_read, dim3((nrows + 3) / 4), dim3(256), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, out);
                else if (shape == 1) hipLaunchKernelGGL(k_clear, dim3((nrows + 3) / 4), dim3(256), 0, 0, rows, c.stride, nrows, c.lo, jitter, width);
                else if (shape == 2) hipLaunchKernelGGL(k_flush<0>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 3) hipLaunchKernelGGL(k_flush<1>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 4) hipLaunchKernelGGL(k_flush<2>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 5) hipLaunchKernelGGL(k_flush<3>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else if (shape == 6) hipLaunchKernelGGL(k_flush<4>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
                else hipLaunchKernelGGL(k_flush<5>, dim3(nrows / 32), dim3(1024), 0, 0, rows, c.stride, nrows, c.lo, jitter, width, pct, out);
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
            static const char *kShape[8] = {"read", "clear", "flush_atomic_u64", "flush_atomic_u32", "flush_line_read_then_atomic",
                                            "flush_plain_rmw", "flush_atomic_u64_returning", "flush_one_atomic_per_line"};
            const double moved = shape >= 2 ? win_bytes * pct / 100.0 : win_bytes;
            printf("{\"tool\": \"row_stride\", \"shape\": \"%s\", \"layout\": \"%s\", \"stride_cells\": %zu, \"rows\": %u, \"width\": %u, "
                   "\"avg_us\": %.1f, \"min_us\": %.1f, \"window_GBs\": %.0f%s}\n",
                   kShape[shape], c.name, c.stride, nrows, width, avg * 1e3, mn * 1e3,
                   moved / (avg * 1e-3) / 1e9, shape >= 2 ? ", \"note\": \"GB/s of the cells that get an update\"" : "");
            fflush(stdout);
        }
        CHECK(hipFree(rows));
    }
    return 0;
}
