// lh_ids.h -- how the mixed-ingest kernels read the ids of an (id, value) stream: two ids per load.
//
// The stream's ids are uint32 (12 B per pair with the float64 value) or uint16 (10 B per pair; legal for at most
// 65 536 names: SURVEY.md 8d "10 B if ids are uint16").  Every kernel is a template on the id type and touches it
// through IdStream only: `raw_t` is what ONE load brings (two ids) and what the software-pipelined kernels keep in
// their prefetch registers -- the ids are taken apart where they are used, not where they are loaded, so that the
// unpack of the narrow form does not wait for a load that was issued a whole tile ahead.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

typedef uint32_t idu2_t __attribute__((ext_vector_type(2)));

template <typename IDT> struct IdStream;

template <> struct IdStream<uint32_t> {
    typedef idu2_t raw_t;
    const idu2_t *p;
    __device__ __forceinline__ explicit IdStream(const uint32_t *ids) : p(reinterpret_cast<const idu2_t *>(ids)) {}
    __device__ __forceinline__ raw_t ld_nt(size_t pair) const { return __builtin_nontemporal_load(p + pair); }
    __device__ __forceinline__ raw_t ld(size_t pair) const { return p[pair]; }
    static __device__ __forceinline__ uint32_t first(const raw_t &r) { return r.x; }
    static __device__ __forceinline__ uint32_t second(const raw_t &r) { return r.y; }
};

template <> struct IdStream<uint16_t> {
    typedef uint32_t raw_t;
    const uint32_t *p;
    __device__ __forceinline__ explicit IdStream(const uint16_t *ids) : p(reinterpret_cast<const uint32_t *>(ids)) {}
    __device__ __forceinline__ raw_t ld_nt(size_t pair) const { return __builtin_nontemporal_load(p + pair); }
    __device__ __forceinline__ raw_t ld(size_t pair) const { return p[pair]; }
    static __device__ __forceinline__ uint32_t first(const raw_t &r) { return r & 0xffffu; }
    static __device__ __forceinline__ uint32_t second(const raw_t &r) { return r >> 16; }
};

} // namespace lh
