// lh_cells.h -- the cell width of an epoch buffer's rows (round 6: 32-bit cells above 8 192 names).
//
// A row is cell[65536 + skew]; the cell is uint64 (the reference's *uint64, metrics.go:278) or -- for the stores of many
// names -- uint32: every flush, clear, extract read and merge pack moves with the number of LINES a name's window takes
// (profiles/r06_cells32.txt: flush 237 -> 117 us, clear 2.5 x, read 1.33 x, pack 1.6 x on 65 536 windows of 600 cells),
// and lh_create(65 536 names) takes 16 GiB per epoch buffer instead of 32.  Exact all the same: a narrow buffer holds fewer
// than 2^32 samples (the engine counts what it enqueues and WIDENS the buffer before an enqueue could pass that:
// lh_engine.cc, widen_buffer), so no cell can wrap.
//
// The width travels IN the pointer: bit 0 of a `uint64_t *counts` kernel argument set = the store behind it is uint32
// cells (every store is at least 16-byte aligned).  The kernels that ADD to rows branch on it once per add -- the pointer is
// a kernel argument, so the branch is scalar -- and the kernels that READ rows are templates the launch wrappers pick.  A
// row pointer derived on the host (lh::row_of) carries the tag along.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace lh {

constexpr uintptr_t kCell32Tag = 1;

inline bool cells_narrow(const void *counts) { return ((uintptr_t)counts & kCell32Tag) != 0; }
inline uint32_t cell_bytes_of(const void *counts) { return cells_narrow(counts) ? 4u : 8u; }
// the store's address without the tag
template <typename T> inline T *cells_base(T *counts) { return reinterpret_cast<T *>((uintptr_t)counts & ~kCell32Tag); }
// store + width -> what the kernels take
inline uint64_t *cells_tagged(void *store, uint32_t cell_bytes)
{
    return reinterpret_cast<uint64_t *>((uintptr_t)store | (cell_bytes == 4 ? kCell32Tag : 0));
}
// cell `index` (row * stride + bin) of a tagged store, tag kept
inline uint64_t *cells_at(const uint64_t *counts, size_t index)
{
    const uintptr_t base = (uintptr_t)counts & ~kCell32Tag;
    return cells_narrow(counts) ? reinterpret_cast<uint64_t *>((base + index * 4) | kCell32Tag)
                                : reinterpret_cast<uint64_t *>(base + index * 8);
}

} // namespace lh

#ifdef __HIPCC__
#include <hip/hip_runtime.h>

namespace lh {

// (a row pointer into a narrow store may be only 4-byte aligned: bit 0 is the tag, nothing else is masked)
__device__ __forceinline__ bool d_cells_narrow(const uint64_t *counts) { return ((uintptr_t)counts & kCell32Tag) != 0; }
__device__ __forceinline__ uint32_t *d_cells32(const uint64_t *counts)
{
    return reinterpret_cast<uint32_t *>((uintptr_t)counts & ~kCell32Tag);
}

// counts[index] += c.  c < 2^32 on a narrow store (the whole buffer holds fewer samples than that).
__device__ __forceinline__ void cell_add(uint64_t *counts, size_t index, uint64_t c)
{
    if (d_cells_narrow(counts)) atomicAdd(d_cells32(counts) + index, (uint32_t)c);
    else atomicAdd(reinterpret_cast<unsigned long long *>(counts) + index, (unsigned long long)c);
}

// The same from inline asm: an atomic that returns nothing and that the compiler's s_waitcnt bookkeeping does not see
// (for use inside the tile loops of the scatter kernels; lh_kernels_part2.h explains why).
__device__ __forceinline__ void cell_add_hidden(uint64_t *counts, size_t index, uint32_t c)
{
    if (d_cells_narrow(counts)) {
        asm volatile("global_atomic_add %0, %1, off" : : "v"(d_cells32(counts) + index), "v"(c) : "memory");
    } else {
        const unsigned long long c64 = c;
        asm volatile("global_atomic_add_x2 %0, %1, off" : : "v"(counts + index), "v"(c64) : "memory");
    }
}

} // namespace lh
#endif
