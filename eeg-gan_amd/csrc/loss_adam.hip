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
    if (per % 4 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        // 16-byte loads, eight in flight per thread: one workgroup walks a whole row (9 600 values for a five-channel
        // signal of 1 920 steps), and with one value per load the walk was 37 dependent round trips
        const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
        const size_t n4 = per / 4;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        size_t i = threadIdx.x;
        for (; i + 7 * 256 < n4; i += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = r4[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s4[e] += v[u][e] * v[u][e];
        }
        for (; i < n4; i += 256) {
            const f32x4 v = r4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] += v[e] * v[e];
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else {
        for (size_t i = threadIdx.x; i < per; i += 256) s += row[i] * row[i];
    }
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

// ---- the other score losses of the reference (losses.py:47-60, 82-101) ----------------------------------------
// One workgroup: value of the loss term and its gradient w.r.t. both score vectors (for an upstream gradient of
// one), from two score vectors a, b of length B.  As the critic's loss a = D(real), b = D(fake); the generator's
// relativistic losses are the same functions with the roles swapped (a = D(fake), b = D(real)).
//   1 hinge            mean relu(1 - a) + mean relu(1 + b)
//   2 rsgan            mean softplus(-(a - b))                                   (= BCE-with-logits(a - b, 1))
//   3 rasgan           (mean softplus(-(a - mean b)) + mean softplus(b - mean a)) / 2
//   4 rahinge          (mean relu(1 - (a - mean b)) + mean relu(1 + (b - mean a))) / 2
//   5 wgan_theirs      mean b - mean a + eps (mean a + mean b)^2                 (score part; its penalty is scaled by 6)
//   6 gain             relu(mean a - mean b)
__device__ __forceinline__ float softplus_f(float x) { return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void score_loss_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b,
                                                         int B, float eps, float* __restrict__ loss,
                                                         float* __restrict__ ga, float* __restrict__ gb) {
    __shared__ float red[4];
    const float invB = 1.f / (float)B;
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        sa += a[i];
        sb += b[i];
    }
    const float ma = block_sum_256(sa, red) * invB;
    const float mb = block_sum_256(sb, red) * invB;
    if (kind == 5 || kind == 6) {
        const float val = kind == 5 ? mb - ma + eps * (ma + mb) * (ma + mb) : fmaxf(ma - mb, 0.f);
        const float da = kind == 5 ? -1.f + 2.f * eps * (ma + mb) : (ma > mb ? 1.f : 0.f);
        const float db = kind == 5 ? 1.f + 2.f * eps * (ma + mb) : (ma > mb ? -1.f : 0.f);
        for (int i = threadIdx.x; i < B; i += 256) {
            ga[i] = da * invB;
            gb[i] = db * invB;
        }
        if (threadIdx.x == 0) loss[0] = val;
        return;
    }
    // per-element terms u_i (depends on a_i) and v_i (depends on b_i), their derivatives du_i, dv_i w.r.t. their
    // arguments, and for the relativistic-average forms the means of those derivatives
    const bool avg = kind == 3 || kind == 4;
    float sl = 0.f, sdu = 0.f, sdv = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float x = a[i], y = b[i];
        float u, v, du, dv;
        if (kind == 1) {
            u = fmaxf(1.f - x, 0.f), du = x < 1.f ? -1.f : 0.f;
            v = fmaxf(1.f + y, 0.f), dv = y > -1.f ? 1.f : 0.f;
        } else if (kind == 2) {
            u = softplus_f(-(x - y)), du = -sigmoid_f(-(x - y));
            v = 0.f, dv = -du;
        } else if (kind == 3) {
            u = 0.5f * softplus_f(-(x - mb)), du = -0.5f * sigmoid_f(-(x - mb));
            v = 0.5f * softplus_f(y - ma), dv = 0.5f * sigmoid_f(y - ma);
        } else {
            u = 0.5f * fmaxf(1.f - (x - mb), 0.f), du = 1.f - (x - mb) > 0.f ? -0.5f : 0.f;
            v = 0.5f * fmaxf(1.f + (y - ma), 0.f), dv = 1.f + (y - ma) > 0.f ? 0.5f : 0.f;
        }
        sl += u + v;
        sdu += du;
        sdv += dv;
        ga[i] = du;
        gb[i] = dv;
    }
    const float tot = block_sum_256(sl, red);
    const float mdu = block_sum_256(sdu, red) * invB;
    const float mdv = block_sum_256(sdv, red) * invB;
    // d/da_i = (du_i - mean dv) / B, d/db_i = (dv_i - mean du) / B for the average forms (the means enter every
    // element of the other vector with a minus sign); plain per-element derivatives over B otherwise
    for (int i = threadIdx.x; i < B; i += 256) {
        ga[i] = (ga[i] - (avg ? mdv : 0.f)) * invB;
        gb[i] = (gb[i] - (avg ? mdu : 0.f)) * invB;
    }
    if (threadIdx.x == 0) loss[0] = tot * invB;
}

// out = s[0] * v  (a loss gradient scaled by the upstream gradient, a device scalar)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float* __restrict__ s, const float* __restrict__ v,
                                                              float* __restrict__ out, size_t n) {
    const float k = s[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = k * v[i];
}

// Gradient penalty alone: lambda / target^2 * mean(p_b^2), p_b = |g_b| - target, or its one-sided form
// p_b = relu(|g_b| - target) (losses.py:113-118).  norms[b] = |g_b| is kept for the backward kernel.
__global__ __launch_bounds__(256) void grad_penalty_final_kernel(const float* __restrict__ norms, float* __restrict__ loss,
                                                                 int B, float lambda, float target, int one_sided) {
    __shared__ float red[4];
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float p = norms[b] - target;
        if (one_sided) p = fmaxf(p, 0.f);
        s += p * p;
    }
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) loss[0] = lambda / (target * target) * tot / (float)B;
}

__global__ __launch_bounds__(256) void grad_penalty_bwd_kernel(const float* __restrict__ gl, const float* __restrict__ g,
                                                               const float* __restrict__ norms, float* __restrict__ g_g,
                                                               int B, size_t per, float lambda, float target,
                                                               int one_sided) {
    const float up = gl[0];
    const float invB = 1.f / (float)B;
    const size_t n = (size_t)B * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t b = i / per;
        const float nb = norms[b];
        float p = nb - target;
        if (one_sided) p = fmaxf(p, 0.f);
        const float k = nb > 0.f ? up * lambda / (target * target) * 2.f * p * invB / nb : 0.f;
        g_g[i] = k * g[i];
    }
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
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    int64_t i0 = start;
    if (vec && end - start >= 4) {  // 16-byte accesses over the whole-float4 part of the chunk (start is a multiple of 4)
        // A chunk is four 16-byte items per thread: all sixteen loads go out before the first result is stored (a
        // loop of load, compute, store made four dependent round trips of it, each waiting for the stores of the one
        // before), and before the step counter is looked at.
        const int64_t nv = (end - start) / 4;
        constexpr int U = (int)(CHUNK / 4 / 256);
        f32x4 g4[U], m4[U], v4[U], p4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (items past the chunk's end read item 0 again, which exists: no branch between the loads)
            const int64_t k = threadIdx.x + 256 * u;
            const int64_t i = start + 4 * (k < nv ? k : 0);
            g4[u] = *reinterpret_cast<const f32x4*>(t.g + i);
            m4[u] = *reinterpret_cast<const f32x4*>(t.m + i);
            v4[u] = *reinterpret_cast<const f32x4*>(t.v + i);
            p4[u] = *reinterpret_cast<const f32x4*>(t.p + i);
        }
        const float step_v = t.step[0] + 1.f;
        const float bc1_v = 1.f - powf(b1, step_v);
        const float isb2_v = rsqrtf(1.f - powf(b2, step_v));
        const float lr_v = lr / bc1_v;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = threadIdx.x + 256 * u;
            if (k < nv) {
                const int64_t i = start + 4 * k;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = g4[u][e] * grad_scale;
                    m4[u][e] = b1 * m4[u][e] + (1.f - b1) * g;
                    v4[u][e] = b2 * v4[u][e] + (1.f - b2) * g * g;
                    p4[u][e] -= lr_v * m4[u][e] / (sqrtf(v4[u][e]) * isb2_v + eps);
                }
                *reinterpret_cast<f32x4*>(t.m + i) = m4[u];
                *reinterpret_cast<f32x4*>(t.v + i) = v4[u];
                *reinterpret_cast<f32x4*>(t.p + i) = p4[u];
            }
        }
        i0 = start + 4 * nv;
        if (i0 >= end) return;
    }
    const float step = t.step[0] + 1.f;
    const float bc1 = 1.f - powf(b1, step);
    const float inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, step));
    const float lr_c = lr / bc1;
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

// p = alpha * p + beta * g over a table of tensor pairs (the running average of the generator's weights,
// plugins.py:305-307: alpha = old_weight, beta = 1 - old_weight)
__global__ __launch_bounds__(256) void multi_axpby_kernel(Pack pk, float alpha, float beta) {
    int ti;
    int64_t start;
    if (!locate(pk, blockIdx.x, &ti, &start)) return;
    const eeggan_adam_tensor t = pk.t[ti];
    int64_t end = start + CHUNK;
    if (end > t.n) end = t.n;
    int64_t i0 = start;
    if (((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g)) & 15) == 0 && end - start >= 4) {
        // (all loads of the chunk before its first store, like adam_kernel)
        const int64_t nv = (end - start) / 4;
        constexpr int U = (int)(CHUNK / 4 / 256);
        f32x4 p4[U], g4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = threadIdx.x + 256 * u;
            const int64_t i = start + 4 * (k < nv ? k : 0);
            p4[u] = *reinterpret_cast<const f32x4*>(t.p + i);
            g4[u] = *reinterpret_cast<const f32x4*>(t.g + i);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = threadIdx.x + 256 * u;
            if (k < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = alpha * p4[u][e] + beta * g4[u][e];
                *reinterpret_cast<f32x4*>(t.p + start + 4 * k) = o;
            }
        }
        i0 = start + 4 * nv;
    }
    for (int64_t i = i0 + threadIdx.x; i < end; i += 256) t.p[i] = alpha * t.p[i] + beta * t.g[i];
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

namespace eeggan {
namespace {
// Feature matching (reference losses.py:62-69): diff[f] = mean_b real[b,f] - mean_b fake[b,f], loss = mean_f diff^2.
// One workgroup; thread f strides over the features and walks the batch (fixed order: deterministic).
__global__ __launch_bounds__(256) void feature_match_fwd_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                                float* __restrict__ diff, float* __restrict__ loss, int B,
                                                                int F) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int f = threadIdx.x; f < F; f += 256) {
        float sr = 0.f, sf = 0.f;
        for (int b = 0; b < B; ++b) {
            sr += real[(long long)b * F + f];
            sf += fake[(long long)b * F + f];
        }
        const float d = (sr - sf) / (float)B;
        diff[f] = d;
        acc += d * d;
    }
    const float tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) loss[0] = tot / (float)F;
}
// d loss / d fake[b,f] = -2 diff[f] / (B F), times the upstream scalar gradient
__global__ __launch_bounds__(256) void feature_match_bwd_kernel(const float* __restrict__ gl, const float* __restrict__ diff,
                                                                float* __restrict__ g_fake, int B, int F) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * F) return;
    g_fake[i] = -2.f * gl[0] * diff[i % F] / ((float)B * (float)F);
}
}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_feature_match_fwd(const float* real, const float* fake, float* diff, float* loss, int B, int F,
                                        void* stream) {
    EEGGAN_REQUIRE(real && fake && diff && loss && B > 0 && F > 0, "feature_match_fwd: bad argument");
    hipLaunchKernelGGL(feature_match_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), real, fake, diff, loss, B, F);
    return check_launch("feature_match_fwd");
}

extern "C" int eeggan_feature_match_bwd(const float* gl, const float* diff, float* g_fake, int B, int F, void* stream) {
    EEGGAN_REQUIRE(gl && diff && g_fake && B > 0 && F > 0, "feature_match_bwd: bad argument");
    const long long n = (long long)B * F;
    hipLaunchKernelGGL(feature_match_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), gl,
                       diff, g_fake, B, F);
    return check_launch("feature_match_bwd");
}

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

extern "C" int eeggan_score_loss(int kind, const float* a, const float* b, int B, float eps, float* loss, float* ga,
                                 float* gb, void* stream) {
    EEGGAN_REQUIRE(a && b && loss && ga && gb, "score_loss: null pointer");
    EEGGAN_REQUIRE(kind >= 1 && kind <= 6, "score_loss: kind %d outside 1..6", kind);
    EEGGAN_REQUIRE(B > 0, "score_loss: empty batch");
    hipLaunchKernelGGL(score_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), kind, a, b, B, eps, loss, ga, gb);
    return check_launch("score_loss");
}

extern "C" int eeggan_scale_by_scalar(const float* s, const float* v, float* out, size_t n, void* stream) {
    EEGGAN_REQUIRE(s && v && out && n > 0, "scale_by_scalar: bad argument");
    size_t nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), s, v, out, n);
    return check_launch("scale_by_scalar");
}

extern "C" int eeggan_grad_penalty_fwd(const float* g, float* norms, float* loss, int B, size_t per, float lambda,
                                       float target, int one_sided, void* stream) {
    EEGGAN_REQUIRE(g && norms && loss, "grad_penalty_fwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && per > 0 && target != 0.f, "grad_penalty_fwd: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(row_norm_kernel, dim3(B), dim3(256), 0, st, g, norms, per);
    if (check_launch("grad_penalty_fwd norms")) return 1;
    hipLaunchKernelGGL(grad_penalty_final_kernel, dim3(1), dim3(256), 0, st, norms, loss, B, lambda, target, one_sided);
    return check_launch("grad_penalty_fwd");
}

extern "C" int eeggan_grad_penalty_bwd(const float* gl, const float* g, const float* norms, float* g_g, int B,
                                       size_t per, float lambda, float target, int one_sided, void* stream) {
    EEGGAN_REQUIRE(gl && g && norms && g_g, "grad_penalty_bwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && per > 0 && target != 0.f, "grad_penalty_bwd: bad argument");
    size_t nb = ((size_t)B * per + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(grad_penalty_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), gl, g, norms, g_g,
                       B, per, lambda, target, one_sided);
    return check_launch("grad_penalty_bwd");
}

extern "C" int eeggan_multi_axpby(const eeggan_adam_tensor* table, int ntensors, float alpha, float beta, void* stream) {
    EEGGAN_REQUIRE(table && ntensors > 0, "multi_axpby: bad table");
    for (int first = 0; first < ntensors; first += MAXT) {
        Pack pk;
        int64_t nb;
        if (make_pack("multi_axpby", table, first, ntensors, &pk, &nb)) return 1;
        hipLaunchKernelGGL(multi_axpby_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), pk, alpha, beta);
        if (check_launch("multi_axpby")) return 1;
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
