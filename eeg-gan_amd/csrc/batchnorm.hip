// Batch normalisation with the generator layer's epilogue, the reference's act_norm = 'batch' (ConditionalBatchNorm,
// layers.py:93-134, inside GeneralConv, layers.py:190-213):
//   n = (c - mean_c) * rstd_c                       statistics over (batch, time) per channel, biased variance
//   o = gamma_c n + beta_c                          ('BN' mode: affine normaliser), or
//   o = n + gmod[b,c] n + bmod[b,c]                 (self modulation: per-sample scale and shift from the latent)
//   y = drop[b,c] * lrelu(o)
// Statistics are summed in double in a fixed order (deterministic); the running statistics of the module are
// updated as nn.BatchNorm1d does (momentum, unbiased variance).  First-order backward only: the layer exists in the
// generator, which the training step differentiates once.
#include "common.h"

namespace eeggan {
namespace {

__device__ __forceinline__ double block_sum_256d(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

// one workgroup per channel
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ c, float* __restrict__ mean,
                                                       float* __restrict__ rstd, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, int B, int C, int T, float eps,
                                                       float momentum) {
    __shared__ double red[256];
    const int ch = blockIdx.x;
    const long long per = (long long)B * T;
    double s = 0.0;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const int b = (int)(i / T), t = (int)(i - (long long)b * T);
        s += (double)c[((long long)b * C + ch) * T + t];
    }
    const double mu = block_sum_256d(s, red) / (double)per;
    double q = 0.0;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const int b = (int)(i / T), t = (int)(i - (long long)b * T);
        const double d = (double)c[((long long)b * C + ch) * T + t] - mu;
        q += d * d;
    }
    const double var = block_sum_256d(q, red) / (double)per;
    if (threadIdx.x == 0) {
        mean[ch] = (float)mu;
        rstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unbiased = per > 1 ? var * (double)per / (double)(per - 1) : var;
            run_mean[ch] = (float)((1.0 - momentum) * run_mean[ch] + momentum * mu);
            run_var[ch] = (float)((1.0 - momentum) * run_var[ch] + momentum * unbiased);
        }
    }
}

// evaluation mode: the statistics are the running ones
__global__ void bn_eval_stats_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                     float* __restrict__ mean, float* __restrict__ rstd, int C, float eps) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch < C) {
        mean[ch] = run_mean[ch];
        rstd[ch] = rsqrtf(run_var[ch] + eps);
    }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ c, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ gmod,
                                                       const float* __restrict__ bmod, const float* __restrict__ drop,
                                                       float* __restrict__ y, int C, int T, float slope, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long row = i / T;  // b * C + ch
    const int ch = (int)(row % C);
    float o = (c[i] - mean[ch]) * rstd[ch];
    if (gamma) o = gamma[ch] * o + beta[ch];
    if (gmod) o = o + gmod[row] * o + bmod[row];
    const float a = o > 0.f ? o : o * slope;
    y[i] = drop ? a * drop[row] : a;
}

// one workgroup per channel: sums of the gradient at o over (batch, time), and per sample where the layer is modulated
//   s1[ch] = sum g_o, s2[ch] = sum g_o n;   g_gmod[b,ch] = sum_t g_a o_plain, g_bmod[b,ch] = sum_t g_a
// with g_a = gy drop lrelu'(y), o_plain = gamma n + beta (or n), g_o = g_a (1 + gmod)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const float* __restrict__ c, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ gmod,
                                                            const float* __restrict__ drop, float* __restrict__ s1,
                                                            float* __restrict__ s2, float* __restrict__ g_gmod,
                                                            float* __restrict__ g_bmod, int B, int C, int T, float slope) {
    __shared__ double red[256];
    const int ch = blockIdx.x;
    const float mu = mean[ch], rs = rstd[ch];
    const float ga = gamma ? gamma[ch] : 1.f, be = gamma ? beta[ch] : 0.f;
    double t1 = 0.0, t2 = 0.0;
    for (int b = 0; b < B; ++b) {
        const long long row = (long long)b * C + ch;
        const float dr = drop ? drop[row] : 1.f;
        const float mod = gmod ? 1.f + gmod[row] : 1.f;
        double a1 = 0.0, a2 = 0.0, m1 = 0.0, m2 = 0.0;
        for (int t = threadIdx.x; t < T; t += 256) {
            const long long i = row * T + t;
            const float g_a = gy[i] * dr * (y[i] > 0.f ? 1.f : slope);
            const float nn = (c[i] - mu) * rs;
            const float g_o = g_a * mod;
            a1 += (double)g_o;
            a2 += (double)g_o * (double)nn;
            if (gmod) {
                m1 += (double)g_a * (double)(ga * nn + be);
                m2 += (double)g_a;
            }
        }
        t1 += a1;
        t2 += a2;
        if (gmod) {
            const double r1 = block_sum_256d(m1, red);
            const double r2 = block_sum_256d(m2, red);
            if (threadIdx.x == 0) {
                g_gmod[row] = (float)r1;
                g_bmod[row] = (float)r2;
            }
        }
    }
    const double r1 = block_sum_256d(t1, red);
    const double r2 = block_sum_256d(t2, red);
    if (threadIdx.x == 0) {
        s1[ch] = (float)r1;
        s2[ch] = (float)r2;
    }
}

// g_c = gamma rstd (g_o - s1 / N - n s2 / N)      (train = 0: statistics are constants, g_c = gamma rstd g_o)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ c, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ gmod, const float* __restrict__ drop,
                                                           const float* __restrict__ s1, const float* __restrict__ s2,
                                                           float* __restrict__ gc, int C, int T, float slope, float inv_n,
                                                           int train, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long row = i / T;
    const int ch = (int)(row % C);
    const float g_a = gy[i] * (drop ? drop[row] : 1.f) * (y[i] > 0.f ? 1.f : slope);
    const float g_o = g_a * (gmod ? 1.f + gmod[row] : 1.f);
    const float nn = (c[i] - mean[ch]) * rstd[ch];
    const float k = (gamma ? gamma[ch] : 1.f) * rstd[ch];
    gc[i] = train ? k * (g_o - s1[ch] * inv_n - nn * s2[ch] * inv_n) : k * g_o;
}

// ---- pieces of a batch norm that is differentiated twice (the critic's, under the gradient penalty): sums over
// (batch, time) per channel, a per-channel scale and shift, and a plain product.  Each is linear in each argument, so
// the backward of every one of them is made of the same three.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C,
                                                          int T, float scale) {
    __shared__ double red[256];
    const int ch = blockIdx.x;
    const long long per = (long long)B * T;
    double s = 0.0;
    for (long long i = threadIdx.x; i < per; i += 256) {
        const int b = (int)(i / T), t = (int)(i - (long long)b * T);
        s += (double)x[((long long)b * C + ch) * T + t];
    }
    const double tot = block_sum_256d(s, red);
    if (threadIdx.x == 0) out[ch] = (float)(tot * (double)scale);
}

// y = x * s[ch] + t[ch]   (x NULL: y = t[ch]; s NULL: 1; t NULL: 0)
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             const float* __restrict__ t, float* __restrict__ y, int C, int T,
                                                             long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ch = (int)((i / T) % C);
    const float v = x ? x[i] * (s ? s[ch] : 1.f) : 0.f;
    y[i] = v + (t ? t[ch] : 0.f);
}

__global__ __launch_bounds__(256) void product_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}

// ---- the same for sums over the channels per (sample, time step): the pieces of a pixel norm that is differentiated twice
__global__ __launch_bounds__(256) void position_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T,
                                                           float scale, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // over (b, t)
    if (i >= n) return;
    const long long b = i / T;
    const int t = (int)(i - b * T);
    const float* p = x + b * C * T + t;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += p[(long long)c * T];
    out[i] = s * scale;
}

// y[b, c, t] = x[b, c, t] * s[b, t]   (x NULL: y = s[b, t])
__global__ __launch_bounds__(256) void position_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             float* __restrict__ y, int C, int T, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long row = i / T;  // b * C + c
    const int t = (int)(i - row * T);
    const float v = s[(row / C) * T + t];
    y[i] = x ? x[i] * v : v;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_position_sum(const float* x, float* out, int B, int C, int T, float scale, void* stream) {
    EEGGAN_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "position_sum: bad arguments");
    const long long n = (long long)B * T;
    hipLaunchKernelGGL(position_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, out, C, T,
                       scale, n);
    return check_launch("position_sum");
}

extern "C" int eeggan_position_scale(const float* x, const float* s, float* y, int B, int C, int T, void* stream) {
    EEGGAN_REQUIRE(s && y && B > 0 && C > 0 && T > 0, "position_scale: bad arguments");
    const long long n = (long long)B * C * T;
    hipLaunchKernelGGL(position_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, s, y, C, T,
                       n);
    return check_launch("position_scale");
}

extern "C" int eeggan_channel_sum(const float* x, float* out, int B, int C, int T, float scale, void* stream) {
    EEGGAN_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "channel_sum: bad arguments");
    hipLaunchKernelGGL(channel_sum_kernel, dim3((unsigned)C), dim3(256), 0, as_stream(stream), x, out, B, C, T, scale);
    return check_launch("channel_sum");
}

extern "C" int eeggan_channel_affine(const float* x, const float* s, const float* t, float* y, int B, int C, int T,
                                     void* stream) {
    EEGGAN_REQUIRE(y && (x || t) && B > 0 && C > 0 && T > 0, "channel_affine: bad arguments");
    const long long n = (long long)B * C * T;
    hipLaunchKernelGGL(channel_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, s, t, y, C, T,
                       n);
    return check_launch("channel_affine");
}

extern "C" int eeggan_product(const float* a, const float* b, float* out, size_t count, void* stream) {
    EEGGAN_REQUIRE(a && b && out && count > 0, "product: bad arguments");
    hipLaunchKernelGGL(product_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, out,
                       (long long)count);
    return check_launch("product");
}

extern "C" int eeggan_bn_lrelu_drop_fwd(const float* c, const float* gamma, const float* beta, const float* gmod,
                                        const float* bmod, const float* drop, float* run_mean, float* run_var, float* mean,
                                        float* rstd, float* y, int B, int C, int T, float eps, float momentum, float slope,
                                        int train, void* stream) {
    EEGGAN_REQUIRE(c && y && mean && rstd && B > 0 && C > 0 && T > 0, "batch norm forward: bad arguments");
    EEGGAN_REQUIRE((gamma == nullptr) == (beta == nullptr), "batch norm forward: gamma and beta come together");
    EEGGAN_REQUIRE((gmod == nullptr) == (bmod == nullptr), "batch norm forward: the modulation's scale and shift come together");
    EEGGAN_REQUIRE(train || (run_mean && run_var), "batch norm forward: evaluation mode needs the running statistics");
    EEGGAN_REQUIRE((run_mean == nullptr) == (run_var == nullptr), "batch norm forward: running mean and variance come together");
    hipStream_t st = as_stream(stream);
    if (train)
        hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)C), dim3(256), 0, st, c, mean, rstd, run_mean, run_var, B, C, T, eps,
                           momentum);
    else
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, run_mean, run_var, mean,
                           rstd, C, eps);
    const long long n = (long long)B * C * T;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, mean, rstd, gamma, beta, gmod,
                       bmod, drop, y, C, T, slope, n);
    return check_launch("batch norm forward");
}

extern "C" int eeggan_bn_lrelu_drop_bwd(const float* gy, const float* y, const float* c, const float* mean, const float* rstd,
                                        const float* gamma, const float* beta, const float* gmod, const float* drop, float* gc,
                                        float* g_gamma, float* g_beta, float* g_gmod, float* g_bmod, int B, int C, int T,
                                        float slope, int train, void* stream) {
    EEGGAN_REQUIRE(gy && y && c && mean && rstd && gc && g_gamma && g_beta && B > 0 && C > 0 && T > 0,
                   "batch norm backward: bad arguments");
    EEGGAN_REQUIRE((gmod == nullptr) == (g_gmod == nullptr) && (gmod == nullptr) == (g_bmod == nullptr),
                   "batch norm backward: modulation gradients come with the modulation");
    EEGGAN_REQUIRE((gamma == nullptr) == (beta == nullptr), "batch norm backward: gamma and beta come together");
    hipStream_t st = as_stream(stream);
    // g_gamma / g_beta double as the channel sums s2 / s1 (they ARE the affine parameters' gradients)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)C), dim3(256), 0, st, gy, y, c, mean, rstd, gamma, beta, gmod, drop,
                       g_beta, g_gamma, g_gmod, g_bmod, B, C, T, slope);
    const long long n = (long long)B * C * T;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gy, y, c, mean, rstd, gamma, gmod,
                       drop, g_beta, g_gamma, gc, C, T, slope, 1.f / (float)((long long)B * T), train, n);
    return check_launch("batch norm backward");
}
