// Shared host/device helpers of libeeggan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/eeggan_hip.h"

namespace eeggan {

// Thread-local message of the last failed entry point.
void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Returns non-zero and records the message when the launch itself was rejected.
int check_launch(const char* what);

#define EEGGAN_REQUIRE(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            ::eeggan::set_error(__VA_ARGS__);     \
            return 1;                             \
        }                                         \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // a vector load from a 4-byte aligned address
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- activation storage types -----------------------------------------------------------------------
// Activations live in memory as float32 or, in the bf16 storage mode, as bfloat16; arithmetic is always
// float32.  A kernel that prefetches keeps the RAW loaded words in registers (raw4 / raw1: no instruction
// consumes the load, so nothing waits for it) and widens them where the value is used (widen).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool BF>
struct ActT {
    typedef float T;
    typedef f32x4 raw4;
    typedef float raw1;
};
template <>
struct ActT<true> {
    typedef __bf16 T;
    typedef u32x2 raw4;
    typedef unsigned short raw1;
};
__device__ __forceinline__ f32x4 ld_raw4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ u32x2 ld_raw4(const __bf16* p) { return *reinterpret_cast<const u32x2*>(p); }
__device__ __forceinline__ float ld_raw1(const float* p) { return *p; }
__device__ __forceinline__ unsigned short ld_raw1(const __bf16* p) { return *reinterpret_cast<const unsigned short*>(p); }
__device__ __forceinline__ f32x4 widen(f32x4 v) { return v; }
__device__ __forceinline__ f32x4 widen(u32x2 u) {
    f32x4 v;
    v[0] = __builtin_bit_cast(float, u[0] << 16);
    v[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, u[1] << 16);
    v[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ float widen(float v) { return v; }
__device__ __forceinline__ float widen(unsigned short u) { return __builtin_bit_cast(float, (unsigned int)u << 16); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return ld_raw4(p); }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) { return widen(ld_raw4(p)); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return widen(ld_raw1(p)); }
// stores round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(__bf16* p, f32x4 v) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4_t*>(p) = o;
}
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

// ---- streaming accesses ---------------------------------------------------------------------------
// NT = true marks a 16-byte access non-temporal.  A kernel whose tensors are far larger than the caches (the 256 MB
// memory-side cache included) neither finds its input there nor leaves anything the next kernel could use, and with
// the default policy its stores push its own unread input lines out.  Measured on [512, 64, 1920] float32 (251 MB in,
// 251 MB out, four input / output pairs taken in turn so that no launch re-reads the previous one's bytes): a plain
// copy 6.0 -> 6.5 TB/s, the per-column channel norm 5.3 -> 6.0 TB/s with both sides marked, in between with one side.
// Below the caches' size the default policy wins (63 MB tensors in a loop: 6.7 against 6.3 TB/s), so the host
// decides per launch (streams_past_caches).
template <bool NT>
__device__ __forceinline__ f32x4 ld4s(const float* p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ f32x4 ld4s(const __bf16* p) {
    const u32x2* q = reinterpret_cast<const u32x2*>(p);
    return widen(NT ? __builtin_nontemporal_load(q) : *q);
}
template <bool NT>
__device__ __forceinline__ void st4s(float* p, f32x4 v) {
    if (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else
        *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ void st4s(__bf16* p, f32x4 v) {
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
    if (NT)
        __builtin_nontemporal_store(o, reinterpret_cast<bf16x4_t*>(p));
    else
        *reinterpret_cast<bf16x4_t*>(p) = o;
}

// True when the bytes a launch reads and writes once are beyond what the caches hold between kernels: 192 MB by
// default (EEGGAN_STREAM_MIN_MB overrides it for comparison runs; a huge value switches the streaming forms off).
inline bool streams_past_caches(size_t bytes) {
    static const size_t limit = [] {
        const char* e = std::getenv("EEGGAN_STREAM_MIN_MB");
        return (size_t)(e ? std::atof(e) : 192.0) << 20;
    }();
    return bytes >= limit;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; every thread gets the result.  `red` holds >= 4 floats.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace eeggan
