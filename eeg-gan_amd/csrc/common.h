// Shared host/device helpers of libeeggan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

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
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
