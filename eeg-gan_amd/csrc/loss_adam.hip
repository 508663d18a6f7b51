// WGAN-GP loss reductions and the multi-tensor Adam / bucket-copy kernels.
#include "common.h"


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
// The descriptor table travels BY VALUE in the kernel arguments (<= 64 tensors, 3 KiB): no device
// table, no host->device copy, and a captured launch keeps its pointers.  A workgroup owns one
// CHUNK of one tensor and finds it by walking the table.
constexpr int64_t CHUNK = 256 * 4 * 4;
constexpr int MAXT = 64;

struct Pack {
    eeggan_adam_tensor t[MAXT];
    int n;
};

__device__ __forceinline__ bool locate(const Pack& pk, int64_t blk, int* ti, int64_t* start) {
    int64_t acc = 0;
    for (int i = 0; i < pk.n; ++i) {
        const int64_t nb = (pk.t[i].n + CHUNK - 1) / CHUNK;
        if (blk < acc + nb) {
            *ti = i;
            *start = (blk - acc) * CHUNK;
            return true;
        }
        acc += nb;
    }
    return false;
}

// The step counters live on the device (one float per tensor) so that a captured launch advances
// them at every replay; this kernel reads the OLD count and uses count + 1, adam_tick_kernel
// (launched right behind it) stores the increment.
__global__ __launch_bounds__(256) void adam_kernel(Pack pk, float lr, float b1, float b2, float eps, float grad_scale) {
    int ti;
    int64_t start;
    if (!locate(pk, blockIdx.x, &ti, &start)) return;
    const eeggan_adam_tensor t = pk.t[ti];
    const float step = t.step[0] + 1.f;
    const float bc1 = 1.f - powf(b1, step);
    const float inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, step));
    const float lr_c = lr / bc1;
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    int64_t i0 = start;
    if (vec) {  // 16-byte accesses over the whole-float4 part of the chunk (start is a multiple of 4)
        const int64_t nv = (end - start) / 4;
        for (int64_t k = threadIdx.x; k < nv; k += 256) {
            const int64_t i = start + 4 * k;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(t.g + i);
            f32x4 m4 = *reinterpret_cast<const f32x4*>(t.m + i);
            f32x4 v4 = *reinterpret_cast<const f32x4*>(t.v + i);
            f32x4 p4 = *reinterpret_cast<const f32x4*>(t.p + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = g4[e] * grad_scale;
                m4[e] = b1 * m4[e] + (1.f - b1) * g;
                v4[e] = b2 * v4[e] + (1.f - b2) * g * g;
                p4[e] -= lr_c * m4[e] / (sqrtf(v4[e]) * inv_sqrt_bc2 + eps);
            }
            *reinterpret_cast<f32x4*>(t.m + i) = m4;
            *reinterpret_cast<f32x4*>(t.v + i) = v4;
            *reinterpret_cast<f32x4*>(t.p + i) = p4;
        }
        i0 = start + 4 * nv;
    }
    for (int64_t i = i0 + threadIdx.x; i < end; i += 256) {
        const float g = t.g[i] * grad_scale;
        const float m = b1 * t.m[i] + (1.f - b1) * g;
        const float v = b2 * t.v[i] + (1.f - b2) * g * g;
        t.m[i] = m;
        t.v[i] = v;
        t.p[i] -= lr_c * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
    }
}

__global__ void adam_tick_kernel(Pack pk) {
    const int i = threadIdx.x;
    if (i < pk.n) pk.t[i].step[0] += 1.f;
}

__global__ __launch_bounds__(256) void multi_copy_kernel(Pack pk) {
    int ti;
    int64_t start;
    if (!locate(pk, blockIdx.x, &ti, &start)) return;
    const eeggan_adam_tensor t = pk.t[ti];
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    for (int64_t i = start + threadIdx.x; i < end; i += 256) t.p[i] = t.g[i];
}

int make_pack(const char* what, const eeggan_adam_tensor* table, int first, int ntensors, Pack* pk, int64_t* blocks) {
    pk->n = ntensors - first < MAXT ? ntensors - first : MAXT;
    *blocks = 0;
    for (int i = 0; i < pk->n; ++i) {
        pk->t[i] = table[first + i];
        EEGGAN_REQUIRE(pk->t[i].n > 0 && pk->t[i].p && pk->t[i].g, "%s: bad descriptor %d", what, first + i);
        *blocks += (pk->t[i].n + CHUNK - 1) / CHUNK;
    }
    EEGGAN_REQUIRE(*blocks < (1ll << 31), "%s: too many elements", what);
    return 0;
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

extern "C" int eeggan_adam_step(const eeggan_adam_tensor* table, int ntensors, float lr, float beta1, float beta2,
                                float eps, float grad_scale, void* stream) {
    EEGGAN_REQUIRE(table && ntensors > 0, "adam_step: bad table");
    EEGGAN_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adam_step: betas outside [0, 1)");
    for (int first = 0; first < ntensors; first += MAXT) {
        Pack pk;
        int64_t nb;
        if (make_pack("adam_step", table, first, ntensors, &pk, &nb)) return 1;
        for (int i = 0; i < pk.n; ++i)
            EEGGAN_REQUIRE(pk.t[i].m && pk.t[i].v && pk.t[i].step, "adam_step: descriptor %d lacks state pointers", first + i);
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), pk, lr, beta1, beta2, eps,
                           grad_scale);
        if (check_launch("adam_step")) return 1;
        hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(MAXT), 0, as_stream(stream), pk);
        if (check_launch("adam_step tick")) return 1;
    }
    return 0;
}

extern "C" int eeggan_multi_copy(const eeggan_adam_tensor* table, int ntensors, void* stream) {
    EEGGAN_REQUIRE(table && ntensors > 0, "multi_copy: bad table");
    for (int first = 0; first < ntensors; first += MAXT) {
        Pack pk;
        int64_t nb;
        if (make_pack("multi_copy", table, first, ntensors, &pk, &nb)) return 1;
        hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), pk);
        if (check_launch("multi_copy")) return 1;
    }
    return 0;
}
