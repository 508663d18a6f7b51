// Spectral normalisation of a layer's weight, torch.nn.utils.spectral_norm as the reference applies it to its convs
// (layers.py:170-171; network.py:520-521): the weight as a matrix W [rows = output channels, cols = the rest],
//   training:  v <- normalize(W^T u),  u <- normalize(W v)        (one power iteration, in place, no gradient)
//   sigma = u^T W v,   W_sn = W / sigma
// with normalize(x) = x / max(|x|_2, eps).  The backward treats u and v as constants:
//   g_W = (g_sn - <g_sn, W_sn> u v^T) / sigma.
// The iteration is two passes over one matrix of at most a few MB, a handful of times per step of a non-default
// option: ONE workgroup does it, in a fixed summation order (deterministic), with the workgroup barrier as the only
// synchronisation; the element-wise parts run over the whole chip.
#include "common.h"

namespace eeggan {
namespace {

constexpr int SN_THREADS = 1024;

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = SN_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

// u [rows], v [cols]: the module's vectors (updated in place when ``iterate``); u_out / v_out: the vectors sigma was
// computed with, for the backward (the module's own move on at the next forward); s [rows]: scratch.
__global__ __launch_bounds__(SN_THREADS) void spectral_sigma_kernel(const float* __restrict__ w, float* u, float* v,
                                                                    float* __restrict__ u_out, float* __restrict__ v_out,
                                                                    float* __restrict__ s, float* __restrict__ sigma,
                                                                    int rows, int cols, float eps, int iterate) {
    __shared__ float red[SN_THREADS];
    const int tid = threadIdx.x;
    if (iterate) {
        // v = normalize(W^T u): a thread per column, the rows in order (a wave reads runs of consecutive columns)
        float sq = 0.f;
        for (int c = tid; c < cols; c += SN_THREADS) {
            float t = 0.f;
            for (int r = 0; r < rows; ++r) t += w[(long long)r * cols + c] * u[r];
            v_out[c] = t;
            sq += t * t;
        }
        const float norm = sqrtf(block_sum_1024(sq, red));
        const float inv = 1.f / fmaxf(norm, eps);
        for (int c = tid; c < cols; c += SN_THREADS) {
            const float t = v_out[c] * inv;
            v_out[c] = t;
            v[c] = t;
        }
    } else {
        for (int c = tid; c < cols; c += SN_THREADS) v_out[c] = v[c];
    }
    __syncthreads();  // (v_out: global memory written and read by this workgroup only)
    // s = W v: a wave per row, lanes along the columns
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < rows; r += SN_THREADS / 64) {
        float t = 0.f;
        for (int c = lane; c < cols; c += 64) t += w[(long long)r * cols + c] * v_out[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) s[r] = t;
    }
    __syncthreads();
    float acc = 0.f;
    if (iterate) {
        float sq = 0.f;
        for (int r = tid; r < rows; r += SN_THREADS) sq += s[r] * s[r];
        const float norm = sqrtf(block_sum_1024(sq, red));
        const float inv = 1.f / fmaxf(norm, eps);
        for (int r = tid; r < rows; r += SN_THREADS) {
            const float t = s[r] * inv;
            u[r] = t;
            u_out[r] = t;
            acc += t * s[r];
        }
    } else {
        for (int r = tid; r < rows; r += SN_THREADS) {
            const float t = u[r];
            u_out[r] = t;
            acc += t * s[r];
        }
    }
    const float sg = block_sum_1024(acc, red);
    if (tid == 0) sigma[0] = sg;
}

__global__ __launch_bounds__(256) void spectral_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                                             float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = w[i] / sigma[0];
}

// <g, w_sn> in a fixed order, by one workgroup
__global__ __launch_bounds__(SN_THREADS) void spectral_dot_kernel(const float* __restrict__ g, const float* __restrict__ w_sn,
                                                                  float* __restrict__ out, long long n) {
    __shared__ float red[SN_THREADS];
    float acc = 0.f;
    for (long long i = threadIdx.x; i < n; i += SN_THREADS) acc += g[i] * w_sn[i];
    const float d = block_sum_1024(acc, red);
    if (threadIdx.x == 0) out[0] = d;
}

__global__ __launch_bounds__(256) void spectral_bwd_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                           const float* __restrict__ v, const float* __restrict__ sigma,
                                                           const float* __restrict__ dot, float* __restrict__ out, int cols,
                                                           long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    out[i] = (g[i] - dot[0] * u[r] * v[c]) / sigma[0];
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_sn, float* u_used, float* v_used,
                                        float* sigma, float* scratch, int rows, int cols, float eps, int power_iteration,
                                        void* stream) {
    EEGGAN_REQUIRE(w_orig && u && v && w_sn && u_used && v_used && sigma && scratch && rows > 0 && cols > 0 && eps > 0.f,
                   "spectral_norm forward: bad arguments");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(spectral_sigma_kernel, dim3(1), dim3(SN_THREADS), 0, st, w_orig, u, v, u_used, v_used, scratch, sigma, rows,
                       cols, eps, power_iteration);
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(spectral_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w_orig, sigma, w_sn, n);
    return check_launch("spectral_norm forward");
}

extern "C" int eeggan_spectral_norm_bwd(const float* g_sn, const float* w_sn, const float* u_used, const float* v_used,
                                        const float* sigma, float* scratch, float* g_orig, int rows, int cols, void* stream) {
    EEGGAN_REQUIRE(g_sn && w_sn && u_used && v_used && sigma && scratch && g_orig && rows > 0 && cols > 0,
                   "spectral_norm backward: bad arguments");
    hipStream_t st = as_stream(stream);
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(spectral_dot_kernel, dim3(1), dim3(SN_THREADS), 0, st, g_sn, w_sn, scratch, n);
    hipLaunchKernelGGL(spectral_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_sn, u_used, v_used, sigma,
                       scratch, g_orig, cols, n);
    return check_launch("spectral_norm backward");
}
