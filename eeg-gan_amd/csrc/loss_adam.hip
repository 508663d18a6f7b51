// WGAN-GP loss reductions and the multi-tensor Adam / bucket-copy kernels.
#include "common.h"

#include <vector>

namespace eeggan {
namespace {

// ---- loss ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ g, float* __restrict__ norms,
                                                       size_t per) {
    __shared__ float red[4];
    const float* row = g + (size_t)blockIdx.x * per;
    float s = 0.f;
    for (size_t i = threadIdx.x; i < per; i += 256) s += row[i] * row[i];
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(tot);
}

__global__ __launch_bounds__(256) void wgan_gp_final_kernel(const float* __restrict__ d_real,
                                                            const float* __restrict__ d_fake,
                                                            const float* __restrict__ norms, float* __restrict__ loss,
                                                            int B, float drift, float lambda, float target) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float r = d_real[b];
        const float p = norms[b] - target;
        s += d_fake[b] - r + drift * r * r + lambda / (target * target) * p * p;
    }
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = tot / (float)B;
}

__global__ __launch_bounds__(256) void wgan_gp_bwd_kernel(const float* __restrict__ gl, const float* __restrict__ d_real,
                                                          const float* __restrict__ g, const float* __restrict__ norms,
                                                          float* __restrict__ g_d_real, float* __restrict__ g_d_fake,
                                                          float* __restrict__ g_g, int B, size_t per, float drift,
                                                          float lambda, float target) {
    const float up = gl[0];
    const float invB = 1.f / (float)B;
    const size_t n = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per;
        const float nb = norms[b];
        const float k = nb > 0.f ? up * lambda / (target * target) * 2.f * (nb - target) * invB / nb : 0.f;
        g_g[i] = k * g[i];
    }
    if (blockIdx.x == 0) {
        for (int b = threadIdx.x; b < B; b += 256) {
            g_d_real[b] = up * invB * (-1.f + 2.f * drift * d_real[b]);
            g_d_fake[b] = up * invB;
        }
    }
}

__global__ __launch_bounds__(256) void mean_scale_kernel(const float* __restrict__ v, float* __restrict__ out, int n,
                                                         float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = tot * scale / (float)n;
}

__global__ __launch_bounds__(256) void mean_scale_bwd_kernel(const float* __restrict__ gl, float* __restrict__ out,
                                                             int n, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = gl[0] * scale / (float)n;
}

// ---- multi-tensor kernels ---------------------------------------------------------------------
// A workgroup owns one CHUNK of one tensor; it finds its tensor by walking the (short) table.
constexpr int64_t CHUNK = 256 * 4 * 4;

__device__ __forceinline__ bool locate(const eeggan_adam_tensor* __restrict__ table, int ntensors, int64_t blk,
                                       int* ti, int64_t* start) {
    int64_t acc = 0;
    for (int i = 0; i < ntensors; ++i) {
        const int64_t nb = (table[i].n + CHUNK - 1) / CHUNK;
        if (blk < acc + nb) {
            *ti = i;
            *start = (blk - acc) * CHUNK;
            return true;
        }
        acc += nb;
    }
    return false;
}

__global__ __launch_bounds__(256) void adam_kernel(const eeggan_adam_tensor* __restrict__ table, int ntensors,
                                                   float lr_c, float b1, float b2, float eps, float inv_sqrt_bc2,
                                                   float grad_scale) {
    int ti;
    int64_t start;
    if (!locate(table, ntensors, blockIdx.x, &ti, &start)) return;
    const eeggan_adam_tensor t = table[ti];
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    for (int64_t i = start + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i] * grad_scale;
        const float m = b1 * t.m[i] + (1.f - b1) * g;
        const float v = b2 * t.v[i] + (1.f - b2) * g * g;
        t.m[i] = m;
        t.v[i] = v;
        t.p[i] -= lr_c * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    }
}

__global__ __launch_bounds__(256) void multi_copy_kernel(const eeggan_adam_tensor* __restrict__ table, int ntensors) {
    int ti;
    int64_t start;
    if (!locate(table, ntensors, blockIdx.x, &ti, &start)) return;
    const eeggan_adam_tensor t = table[ti];
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    for (int64_t i = start + threadIdx.x; i < end; i += 256) t.p[i] = t.g[i];
}

int64_t total_blocks(const int64_t* sizes_host, int ntensors) {
    int64_t nb = 0;
    for (int i = 0; i < ntensors; ++i) nb += (sizes_host[i] + CHUNK - 1) / CHUNK;
    return nb;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_wgan_gp_loss_fwd(const float* d_real, const float* d_fake, const float* g, float* norms,
                                       float* loss, int B, size_t per, float drift, float lambda, float target,
                                       void* stream) {
    EEGGAN_REQUIRE(d_real && d_fake && g && norms && loss, "wgan_gp_loss_fwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && per > 0 && target != 0.f, "wgan_gp_loss_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(row_norm_kernel, dim3(B), dim3(256), 0, st, g, norms, per);
    if (check_launch("wgan_gp_loss_fwd norms")) return 1;
    hipLaunchKernelGGL(wgan_gp_final_kernel, dim3(1), dim3(256), 0, st, d_real, d_fake, norms, loss, B, drift, lambda,
                       target);
    return check_launch("wgan_gp_loss_fwd");
}

extern "C" int eeggan_wgan_gp_loss_bwd(const float* gl, const float* d_real, const float* g, const float* norms,
                                       float* g_d_real, float* g_d_fake, float* g_g, int B, size_t per, float drift,
                                       float lambda, float target, void* stream) {
    EEGGAN_REQUIRE(gl && d_real && g && norms && g_d_real && g_d_fake && g_g, "wgan_gp_loss_bwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && per > 0 && target != 0.f, "wgan_gp_loss_bwd: bad argument");
    size_t nb = ((size_t)B * per + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(wgan_gp_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), gl, d_real, g, norms,
                       g_d_real, g_d_fake, g_g, B, per, drift, lambda, target);
    return check_launch("wgan_gp_loss_bwd");
}

extern "C" int eeggan_mean_scale(const float* v, float* out, int n, float scale, void* stream) {
    EEGGAN_REQUIRE(v && out && n > 0, "mean_scale: bad argument");
    hipLaunchKernelGGL(mean_scale_kernel, dim3(1), dim3(256), 0, as_stream(stream), v, out, n, scale);
    return check_launch("mean_scale");
}

extern "C" int eeggan_mean_scale_bwd(const float* gl, float* out, int n, float scale, void* stream) {
    EEGGAN_REQUIRE(gl && out && n > 0, "mean_scale_bwd: bad argument");
    hipLaunchKernelGGL(mean_scale_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), gl, out, n,
                       scale);
    return check_launch("mean_scale_bwd");
}

extern "C" int eeggan_adam_step(const eeggan_adam_tensor* table, const int64_t* sizes_host, int ntensors, float lr,
                                float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    EEGGAN_REQUIRE(table && sizes_host && ntensors > 0, "adam_step: bad table");
    EEGGAN_REQUIRE(step >= 1, "adam_step: step must be >= 1 (got %d)", step);
    const int64_t nb = total_blocks(sizes_host, ntensors);
    EEGGAN_REQUIRE(nb > 0 && nb < (1ll << 31), "adam_step: bad total size");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), table, ntensors,
                       (float)(lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
    return check_launch("adam_step");
}

extern "C" int eeggan_multi_copy(const eeggan_adam_tensor* table, const int64_t* sizes_host, int ntensors,
                                 void* stream) {
    EEGGAN_REQUIRE(table && sizes_host && ntensors > 0, "multi_copy: bad table");
    const int64_t nb = total_blocks(sizes_host, ntensors);
    EEGGAN_REQUIRE(nb > 0 && nb < (1ll << 31), "multi_copy: bad total size");
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), table, ntensors);
    return check_launch("multi_copy");
}
