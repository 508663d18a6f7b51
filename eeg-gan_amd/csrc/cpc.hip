// Inference pass of the CPC feature network used for the Frechet distance (reference cpc/cpc_network.py:
// ConvEncoder 137-158, PNormPooling 161-188, AutoRNN/GRU 191-216, Network.forward(no_loss=True) 316-322) and
// the statistics of that distance (reference plugins.py:503-572: covariance, Newton-Schulz matrix square root).
//
//   cpc_encoder_kernel : the three strided convs (+ ReLU, + eval-mode BatchNorm as a per-channel scale/shift)
//                        of one sample in one workgroup; the sample and both intermediate activations live
//                        in LDS, only the input (38 KB) is read and the codes (16 KB) written: HBM-bound on
//                        paper, fp32 VALU-bound in practice (6.5 MFLOP per sample)
//   pnorm_pool_kernel  : L1 and L-infinity norms over time
//   dense_kernel       : y = epilogue(alpha * x w^T): the pooling MLPs, the GRU's input projection and the
//                        128 x 128 products of the Newton-Schulz iteration; fp32 MFMA (32x32x2), 64 x 64 tiles
//   gru_kernel         : the recurrence, 32 samples per workgroup, hidden state in LDS, W_hh streamed from L2
//   feature_cov        : column means, then centred X^T X partials per row block (fp32 MFMA), summed in double
#include "common.h"
#include "conv_args.h"

namespace eeggan {
namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

// ---- conv encoder ------------------------------------------------------------------------------------------
// Layer l: C[l] -> C[l+1] channels, stride S, kernel 2S-1, padding S-1 (reference cpc_network.py:141-151).
// Weights arrive transposed, wt[(ci * K + j) * Cout + co], so that the 8 output channels a wave accumulates for
// one (ci, j) are 32 contiguous bytes at a wave-uniform address (scalar loads).
constexpr int ENC_C1 = 32, ENC_C2 = 64, ENC_C3 = 128;
constexpr int ENC_S1 = 5, ENC_S2 = 4, ENC_S3 = 3;

template <int COUT, int STRIDE, bool UNIFORM>
__device__ __forceinline__ void enc_layer(const float* __restrict__ in, int in_ld, int cin, const float* __restrict__ wt,
                                          const float* __restrict__ bias, const float* __restrict__ scale,
                                          const float* __restrict__ shift, int tout, int group, int t, bool active,
                                          float* __restrict__ out_lds, int out_ld, int out_pad,
                                          float* __restrict__ out_g, float* __restrict__ out_gt) {
    // `in` points at the padded row start: element for output t, tap j is in[ci * in_ld + STRIDE * t + j]
    constexpr int K = 2 * STRIDE - 1;
    const int co0 = group * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = bias[co0 + i];
    const float* ip = in + STRIDE * (active ? t : 0);
    for (int ci = 0; ci < cin; ++ci) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float xin = ip[ci * in_ld + j];
            const float* wp = wt + (size_t)(ci * K + j) * COUT + co0;
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = fmaf(xin, w0[i], acc[i]);
                acc[4 + i] = fmaf(xin, w1[i], acc[4 + i]);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = fmaxf(acc[i], 0.f);
        if (scale) v = v * scale[co0 + i] + shift[co0 + i];
        if (out_lds) out_lds[(co0 + i) * out_ld + out_pad + t] = v;
        if (out_g) out_g[(size_t)(co0 + i) * tout + t] = v;
        if (out_gt) out_gt[(size_t)t * COUT + co0 + i] = v;
    }
}

__global__ __launch_bounds__(256) void cpc_encoder_kernel(const float* __restrict__ x, const float* __restrict__ w1t,
                                                          const float* __restrict__ b1, const float* __restrict__ s1,
                                                          const float* __restrict__ t1, const float* __restrict__ w2t,
                                                          const float* __restrict__ b2, const float* __restrict__ s2,
                                                          const float* __restrict__ t2, const float* __restrict__ w3t,
                                                          const float* __restrict__ b3, float* __restrict__ z,
                                                          float* __restrict__ zt, int cin, int T) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T1 = T / ENC_S1, T2 = T1 / ENC_S2, T3 = T2 / ENC_S3;
    const int ld0 = T + 2 * (ENC_S1 - 1), ld1 = T1 + 2 * (ENC_S2 - 1), ld2 = T2 + 2 * (ENC_S3 - 1);
    float* xs = sm;
    float* h1 = xs + cin * ld0;
    float* h2 = h1 + ENC_C1 * ld1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t b = blockIdx.x;
    const float* xb = x + b * (size_t)cin * T;
    // the sample, zero-padded per row; padding of the intermediate rows
    for (int i = tid; i < cin * ld0; i += 256) {
        const int ci = i / ld0, p = i - ci * ld0 - (ENC_S1 - 1);
        xs[i] = (p >= 0 && p < T) ? xb[ci * T + p] : 0.f;
    }
    for (int i = tid; i < ENC_C1 * 2 * (ENC_S2 - 1); i += 256) {
        const int c = i / (2 * (ENC_S2 - 1)), p = i - c * 2 * (ENC_S2 - 1);
        h1[c * ld1 + (p < ENC_S2 - 1 ? p : T1 + p)] = 0.f;
    }
    for (int i = tid; i < ENC_C2 * 2 * (ENC_S3 - 1); i += 256) {
        const int c = i / (2 * (ENC_S3 - 1)), p = i - c * 2 * (ENC_S3 - 1);
        h2[c * ld2 + (p < ENC_S3 - 1 ? p : T2 + p)] = 0.f;
    }
    __syncthreads();
    // layer 1: 32 channels = 4 groups of 8, one per wave; lanes over time
    for (int t0 = 0; t0 < T1; t0 += 64)
        enc_layer<ENC_C1, ENC_S1, true>(xs, ld0, cin, w1t, b1, s1, t1, T1, wave, t0 + lane, t0 + lane < T1, h1, ld1,
                                        ENC_S2 - 1, nullptr, nullptr);
    __syncthreads();
    // layer 2: 64 channels = 8 groups, two per wave
    for (int g = 0; g < 2; ++g)
        for (int t0 = 0; t0 < T2; t0 += 64)
            enc_layer<ENC_C2, ENC_S2, true>(h1, ld1, ENC_C1, w2t, b2, s2, t2, T2, wave + 4 * g, t0 + lane, t0 + lane < T2,
                                            h2, ld2, ENC_S3 - 1, nullptr, nullptr);
    __syncthreads();
    // layer 3: 128 channels = 16 groups; the codes are short (T3 = 32 at the full window), so the two halves of a
    // wave take two groups side by side
    float* zb = z + b * (size_t)ENC_C3 * T3;
    float* ztb = zt ? zt + b * (size_t)T3 * ENC_C3 : nullptr;
    const int half = lane >> 5, l31 = lane & 31;
    for (int g = 0; g < 2; ++g)
        for (int t0 = 0; t0 < T3; t0 += 32)
            enc_layer<ENC_C3, ENC_S3, false>(h2, ld2, ENC_C2, w3t, b3, nullptr, nullptr, T3, (wave * 2 + g) * 2 + half,
                                             t0 + l31, t0 + l31 < T3, nullptr, 0, 0, zb, ztb);
}

// ---- p-norm pooling: out[b, c] = sum_t |x|, out[b, C + c] = max_t |x| --------------------------------------
__global__ __launch_bounds__(256) void pnorm_pool_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         long long rows, int C, int T) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;  // r = b * C + c
    if (r >= rows) return;
    const float* p = x + r * T;
    float s = 0.f, m = 0.f;
    for (int t = 0; t < T; ++t) {
        const float a = fabsf(p[t]);
        s += a;
        m = fmaxf(m, a);
    }
    const long long b = r / C;
    const int c = (int)(r - b * C);
    out[b * 2 * C + c] = s;
    out[b * 2 * C + C + c] = m;
}

// ---- dense layer on the fp32 matrix cores ---------------------------------------------------------------------
// y[n, m] = shift[m] + scale[m] * act(bias[m] + alpha * sum_k x[n, k] * W(m, k)) + eye * (n == m)
// W(m, k) = w[m * K + k] (nn.Linear layout) or w[k * M + m] (WKN: a plain matrix product x w).
struct DenseArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* scale;
    const float* shift;
    float* y;
    long long N;
    int K, M;
    long long ldx, ldy;
    float alpha, eye;
    int relu;
};

template <bool WKN>
__global__ __launch_bounds__(256) void dense_kernel(DenseArgs a) {
    __shared__ float xs[64][33];
    __shared__ float ws[64][33];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const long long n0 = (long long)blockIdx.x * 64;
    const int m0 = blockIdx.y * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < a.K; k0 += 32) {
        {   // x tile: thread = (row tid >> 2, 8 consecutive k)
            const int row = tid >> 2, kq = (tid & 3) * 8;
            const long long n = n0 + row;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + kq + e;
                xs[row][kq + e] = (n < a.N && k < a.K) ? a.x[n * a.ldx + k] : 0.f;
            }
        }
        if (WKN) {  // w[k][m]: thread = (k tid >> 3, 8 consecutive m)
            const int kk = tid >> 3, mq = (tid & 7) * 8;
            const int k = k0 + kk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = m0 + mq + e;
                ws[mq + e][kk] = (k < a.K && m < a.M) ? a.w[(long long)k * a.M + m] : 0.f;
            }
        } else {
            const int row = tid >> 2, kq = (tid & 3) * 8;
            const int m = m0 + row;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + kq + e;
                ws[row][kq + e] = (m < a.M && k < a.K) ? a.w[(long long)m * a.K + k] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[wr * 32 + l31][2 * s + h], ws[wc * 32 + l31][2 * s + h], acc, 0, 0, 0);
        __syncthreads();
    }
    const int m = m0 + wc * 32 + l31;
    if (m >= a.M) return;
    const float bv = a.bias ? a.bias[m] : 0.f, sc = a.scale ? a.scale[m] : 1.f, sh = a.shift ? a.shift[m] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < a.N) {
            float v = a.alpha * acc[r] + bv;
            if (a.relu) v = fmaxf(v, 0.f);
            v = v * sc + sh;
            if (n == m) v += a.eye;
            a.y[n * a.ldy + m] = v;
        }
    }
}

// ---- GRU recurrence (hidden size 128) ---------------------------------------------------------------------------
// gi[(b * T + t) * 384 + g * 128 + j] = W_i{r,z,n} x_t + b_i (precomputed), whh [384][128], bhh [384]; h_0 = 0.
//   r = sigmoid(gi_r + W_hr h + b_hr), z = sigmoid(gi_z + W_hz h + b_hz), n = tanh(gi_n + r * (W_hn h + b_hn)),
//   h' = (1 - z) n + z h                      (torch.nn.GRU; reference cpc_network.py:203, 211-216)
// Output c[b][j][t] (the reference permutes the GRU output back to [B, C, T]).
constexpr int GRU_H = 128;
__global__ __launch_bounds__(256) void gru_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                  const float* __restrict__ bhh, float* __restrict__ c, int B, int T) {
    __shared__ float hbuf[2][32][GRU_H + 1];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int b0 = blockIdx.x * 32;
    for (int i = tid; i < 32 * (GRU_H + 1); i += 256) (&hbuf[0][0][0])[i] = 0.f;
    const int j = wave * 32 + l31;  // hidden unit of this lane (B operand / output column)
    const float br = bhh[j], bz = bhh[GRU_H + j], bn = bhh[2 * GRU_H + j];
    // this lane's rows of W_hh: gate g, unit j, reduction indices [64 h, 64 h + 64)
    const float* wr_p = whh + (size_t)(0 * GRU_H + j) * GRU_H + 64 * h;
    const float* wz_p = whh + (size_t)(1 * GRU_H + j) * GRU_H + 64 * h;
    const float* wn_p = whh + (size_t)(2 * GRU_H + j) * GRU_H + 64 * h;
    float pend[16][4];  // four time steps of output per (sample, unit), stored as one 16-byte piece
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float (*hc)[GRU_H + 1] = hbuf[t & 1];
        float (*hn)[GRU_H + 1] = hbuf[(t + 1) & 1];
        f32x16 ar, az, an;
#pragma unroll
        for (int r = 0; r < 16; ++r) ar[r] = az[r] = an[r] = 0.f;
        for (int s4 = 0; s4 < 16; ++s4) {
            const f32x4 vr = *reinterpret_cast<const f32x4*>(wr_p + 4 * s4);
            const f32x4 vz = *reinterpret_cast<const f32x4*>(wz_p + 4 * s4);
            const f32x4 vn = *reinterpret_cast<const f32x4*>(wn_p + 4 * s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float hv = hc[l31][64 * h + 4 * s4 + e];  // A operand: sample l31, reduction index
                ar = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, vr[e], ar, 0, 0, 0);
                az = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, vz[e], az, 0, 0, 0);
                an = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, vn[e], an, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h;  // sample of this accumulator register
            const int b = b0 + i < B ? b0 + i : B - 1;
            const float* g = gi + ((size_t)b * T + t) * (3 * GRU_H) + j;
            const float rr = sigmoidf_(g[0] + ar[r] + br);
            const float zz = sigmoidf_(g[GRU_H] + az[r] + bz);
            const float nn = tanhf(g[2 * GRU_H] + rr * (an[r] + bn));
            const float hv = (1.f - zz) * nn + zz * hc[i][j];
            hn[i][j] = hv;
            pend[r][t & 3] = hv;
            if ((t & 3) == 3 && b0 + i < B) {
                f32x4 o = {pend[r][0], pend[r][1], pend[r][2], pend[r][3]};
                *reinterpret_cast<f32x4*>(c + ((size_t)(b0 + i) * GRU_H + j) * T + (t - 3)) = o;
            }
        }
        __syncthreads();
    }
}

// ---- feature statistics -------------------------------------------------------------------------------------
constexpr int COV_ROWS = 4096;  // rows per workgroup of the covariance partials
constexpr int COV_MAXF = 256;

// column sums of X[N, F] in double: part[block * F + f]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long long N, int F,
                                                     double* __restrict__ part, int rows_per_block) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    long long r1 = r0 + rows_per_block;
    if (r1 > N) r1 = N;
    for (int f = threadIdx.x; f < F; f += 256) {
        double s = 0.0;
        for (long long r = r0; r < r1; ++r) s += (double)X[r * F + f];
        part[(size_t)blockIdx.x * F + f] = s;
    }
}

__global__ __launch_bounds__(256) void mean_final_kernel(const double* __restrict__ part, int nblocks, long long N, int F,
                                                         float* __restrict__ mean) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * F + f];
    mean[f] = (float)(s / (double)N);
}

// partial[block][i][j] = sum over the block's rows of (X[r,i] - mean[i]) (X[r,j] - mean[j]); F % 32 == 0.
// Wave w owns the output tiles (ti, tj) with ti * (F/32) + tj == w (mod 4).
__global__ __launch_bounds__(256) void cov_partial_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                          long long N, int F, float* __restrict__ partial) {
    extern __shared__ float xs[];  // [32 rows][F + 1], centred
    const int ld = F + 1;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wave = tid >> 6;
    const int nt = F / 32, ntiles = nt * nt;
    constexpr int MAXT = (COV_MAXF / 32) * (COV_MAXF / 32) / 4;  // tiles per wave at most
    f32x16 acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const long long r0 = (long long)blockIdx.x * COV_ROWS;
    for (long long rb = r0; rb < r0 + COV_ROWS && rb < N; rb += 32) {
        __syncthreads();
        for (int i = tid; i < 32 * F; i += 256) {
            const int rr = i / F, f = i - rr * F;
            xs[rr * ld + f] = (rb + rr < N) ? X[(rb + rr) * F + f] - mean[f] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MAXT; ++q) {
            const int tile = wave + 4 * q;
            if (tile < ntiles) {
                const int ti = tile / nt, tj = tile - ti * nt;
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[(2 * s + h) * ld + ti * 32 + l31],
                                                                   xs[(2 * s + h) * ld + tj * 32 + l31], acc[q], 0, 0, 0);
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * F * F;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        const int tile = wave + 4 * q;
        if (tile < ntiles) {
            const int ti = tile / nt, tj = tile - ti * nt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)i * F + tj * 32 + l31] = acc[q][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void cov_final_kernel(const float* __restrict__ partial, int nblocks, long long N, int F,
                                                        float* __restrict__ cov) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long long)F * F) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += (double)partial[(size_t)b * F * F + e];
    cov[e] = (float)(s / (double)(N - 1));
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_cpc_encoder(const float* x, const float* w1t, const float* b1, const float* s1, const float* t1,
                                  const float* w2t, const float* b2, const float* s2, const float* t2, const float* w3t,
                                  const float* b3, float* z, float* zt, int B, int Cin, int T, void* stream) {
    EEGGAN_REQUIRE(x && w1t && b1 && w2t && b2 && w3t && b3 && z, "cpc_encoder: null pointer");
    EEGGAN_REQUIRE((s1 == nullptr) == (t1 == nullptr) && (s2 == nullptr) == (t2 == nullptr),
                   "cpc_encoder: a normalisation needs both its scale and its shift");
    EEGGAN_REQUIRE(B > 0 && Cin > 0 && Cin <= 16, "cpc_encoder: batch %d, input channels %d (1..16)", B, Cin);
    EEGGAN_REQUIRE(T > 0 && T % (ENC_S1 * ENC_S2 * ENC_S3) == 0, "cpc_encoder: length %d must be a multiple of %d", T,
                   ENC_S1 * ENC_S2 * ENC_S3);
    EEGGAN_REQUIRE(((reinterpret_cast<uintptr_t>(w1t) | reinterpret_cast<uintptr_t>(w2t) | reinterpret_cast<uintptr_t>(w3t)) & 15) == 0,
                   "cpc_encoder: weights must be 16-byte aligned");
    const int T1 = T / ENC_S1, T2 = T1 / ENC_S2;
    const size_t lds = ((size_t)Cin * (T + 2 * (ENC_S1 - 1)) + (size_t)ENC_C1 * (T1 + 2 * (ENC_S2 - 1)) +
                        (size_t)ENC_C2 * (T2 + 2 * (ENC_S3 - 1))) * sizeof(float);
    EEGGAN_REQUIRE(lds <= 160 * 1024, "cpc_encoder: a window of %d samples x %d channels needs %zu bytes of LDS (160 KiB "
                   "available)", T, Cin, lds);
    static std::atomic<bool> ready_dev[MAX_DEVICES];  // opt in to > 64 KiB of dynamic LDS (once per device)
    std::atomic<bool>& ready = ready_dev[current_device()];
    if (!ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cpc_encoder_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        EEGGAN_REQUIRE(e == hipSuccess, "cpc_encoder: cannot reserve LDS: %s", hipGetErrorString(e));
        ready.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(cpc_encoder_kernel, dim3((unsigned)B), dim3(256), lds, as_stream(stream), x, w1t, b1, s1, t1, w2t,
                       b2, s2, t2, w3t, b3, z, zt, Cin, T);
    return check_launch("cpc_encoder");
}

extern "C" int eeggan_pnorm_pool(const float* x, float* out, int B, int C, int T, void* stream) {
    EEGGAN_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "pnorm_pool: bad argument");
    const long long rows = (long long)B * C;
    hipLaunchKernelGGL(pnorm_pool_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), x, out,
                       rows, C, T);
    return check_launch("pnorm_pool");
}

extern "C" int eeggan_dense(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                            float* y, size_t N, int K, int M, size_t ldx, size_t ldy, int w_is_kn, float alpha, float eye,
                            int relu, void* stream) {
    EEGGAN_REQUIRE(x && w && y, "dense: null pointer");
    EEGGAN_REQUIRE(N > 0 && K > 0 && M > 0 && ldx >= (size_t)K && ldy >= (size_t)M, "dense: bad shape");
    EEGGAN_REQUIRE((scale == nullptr) == (shift == nullptr), "dense: scale and shift come together");
    EEGGAN_REQUIRE((N + 63) / 64 < (1ull << 31) && (M + 63) / 64 < 65536, "dense: too large");
    DenseArgs a{x, w, bias, scale, shift, y, (long long)N, K, M, (long long)ldx, (long long)ldy, alpha, eye, relu};
    dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64));
    if (w_is_kn)
        hipLaunchKernelGGL(dense_kernel<true>, grid, dim3(256), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(dense_kernel<false>, grid, dim3(256), 0, as_stream(stream), a);
    return check_launch("dense");
}

extern "C" int eeggan_gru_forward(const float* gi, const float* whh, const float* bhh, float* c, int B, int H, int T,
                                  void* stream) {
    EEGGAN_REQUIRE(gi && whh && bhh && c, "gru_forward: null pointer");
    EEGGAN_REQUIRE(H == GRU_H, "gru_forward: hidden size %d not supported (%d)", H, GRU_H);
    EEGGAN_REQUIRE(B > 0 && T > 0 && T % 4 == 0, "gru_forward: batch %d, %d steps (a multiple of 4)", B, T);
    EEGGAN_REQUIRE(((reinterpret_cast<uintptr_t>(whh) | reinterpret_cast<uintptr_t>(c)) & 15) == 0,
                   "gru_forward: whh and c must be 16-byte aligned");
    hipLaunchKernelGGL(gru_kernel, dim3((unsigned)((B + 31) / 32)), dim3(256), 0, as_stream(stream), gi, whh, bhh, c, B, T);
    return check_launch("gru_forward");
}

extern "C" size_t eeggan_feature_stats_workspace(size_t N, int F) {
    if (N == 0 || F <= 0) return 0;
    const size_t nb = (N + COV_ROWS - 1) / COV_ROWS;
    return nb * (size_t)F * sizeof(double) + nb * (size_t)F * F * sizeof(float);
}

extern "C" int eeggan_feature_stats(const float* X, size_t N, int F, float* mean, float* cov, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    EEGGAN_REQUIRE(X && mean && cov && workspace, "feature_stats: null pointer");
    EEGGAN_REQUIRE(N > 1 && F > 0 && F % 32 == 0 && F <= COV_MAXF, "feature_stats: needs N > 1 rows and F a multiple of 32 up to %d", COV_MAXF);
    EEGGAN_REQUIRE(workspace_bytes >= eeggan_feature_stats_workspace(N, F), "feature_stats: workspace too small");
    EEGGAN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "feature_stats: workspace must be 8-byte aligned");
    const int nb = (int)((N + COV_ROWS - 1) / COV_ROWS);
    double* sums = static_cast<double*>(workspace);
    float* partial = reinterpret_cast<float*>(sums + (size_t)nb * F);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, st, X, (long long)N, F, sums, COV_ROWS);
    if (check_launch("feature_stats sums")) return 1;
    hipLaunchKernelGGL(mean_final_kernel, dim3((F + 255) / 256), dim3(256), 0, st, sums, nb, (long long)N, F, mean);
    if (check_launch("feature_stats mean")) return 1;
    static std::atomic<bool> ready_dev[MAX_DEVICES];
    std::atomic<bool>& ready = ready_dev[current_device()];
    const size_t lds = (size_t)32 * (F + 1) * sizeof(float);
    if (!ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cov_partial_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(32 * (COV_MAXF + 1) * sizeof(float)));
        EEGGAN_REQUIRE(e == hipSuccess, "feature_stats: cannot reserve LDS: %s", hipGetErrorString(e));
        ready.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(cov_partial_kernel, dim3(nb), dim3(256), lds, st, X, mean, (long long)N, F, partial);
    if (check_launch("feature_stats partial")) return 1;
    const long long ee = (long long)F * F;
    hipLaunchKernelGGL(cov_final_kernel, dim3((unsigned)((ee + 255) / 256)), dim3(256), 0, st, partial, nb, (long long)N, F, cov);
    return check_launch("feature_stats");
}
