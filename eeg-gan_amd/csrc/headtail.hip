// The two full-length-kernel layers of the path, which are dense matrix products rather
// than sliding convolutions:
//   G head: Conv1d(k=KT, pad=KT-1) on a length-1 signal  == [B,Cin] x [Cin, Cout*KT]
//   D tail: Conv1d(k=KT, pad=0)    on a length-KT signal == [B,Cin*KT] x [Cin*KT, Cout]
// At the path's sizes (B <= 512, Cin = Cout = 256, KT = 32) they move the 8.4 MB weight
// once and do ~0.1-2 GFLOP: bandwidth/latency-bound, so most of them are plain fp32 VALU kernels
// with the weight streamed coalesced and the small operand broadcast from LDS (the head's forward pass and the
// tail's input gradient run on the matrix cores).
#include "common.h"
#include "conv_args.h"

#include <cstdlib>

namespace eeggan {
namespace {

__device__ __forceinline__ float apply_act(float v, int act, float p) {
    if (act == EEGGAN_ACT_LRELU) return v > 0.f ? v : v * p;
    if (act == EEGGAN_ACT_TANH) return tanhf(v * p);
    return v;
}

// ---- head forward: grid (Cout, ceil(B/32)), 256 threads = 32 taps x 8 sample lanes ------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ drop,
                                                       float* __restrict__ y, int B, int Cin, int Cout, int KT,
                                                       float in_scale, int act, float act_param) {
    extern __shared__ float zs[];  // [32][Cin]
    const int co = blockIdx.x;
    const int b0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * Cin; i += 256) {
        const int bl = i / Cin, ci = i - bl * Cin;
        zs[i] = (b0 + bl < B) ? z[(long long)(b0 + bl) * Cin + ci] : 0.f;
    }
    __syncthreads();
    const int tp = threadIdx.x & 31, bq = threadIdx.x >> 5;
    const float bv = bias ? bias[co] : 0.f;
    for (int t0 = 0; t0 < KT; t0 += 32) {
        const int tw = t0 + tp;  // weight tap index; output time = KT-1-tw
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (tw < KT) {
            const float* wp = w + (long long)co * Cin * KT + tw;
            for (int ci = 0; ci < Cin; ++ci) {
                const float wv = wp[(long long)ci * KT];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += wv * zs[(bq + 8 * i) * Cin + ci];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + bq + 8 * i;
                if (b < B) {
                    float v = apply_act(acc[i] * in_scale + bv, act, act_param);
                    if (drop) v *= drop[b * Cout + co];
                    y[((long long)b * Cout + co) * KT + (KT - 1 - tw)] = v;
                }
            }
        }
    }
}

// ---- head forward on the matrix cores (KT == 32, Cin % 64 == 0) -------------------------------
// Per output channel the layer is a [32 samples] x [Cin] x [32 taps] product whose weight operand
// W[co][ci][tap] is already laid out as MFMA B-fragments: for a k-step (ci, ci+1) the 64 lanes read
// 2 x 32 consecutive taps straight from global memory (two full 128-byte lines), no LDS staging.
// grid (Cout, ceil(B/32)); the 4 waves split Cin and add their partial tiles through LDS.
__global__ __launch_bounds__(256) void head_fwd_mfma_kernel(const float* __restrict__ z, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ drop, float* __restrict__ y,
                                                            int B, int Cin, int Cout, float in_scale, int act,
                                                            float act_param) {
    extern __shared__ float sm[];  // zs [32][Cin + 1], then red [4][32][33]
    const int ld = Cin + 1;
    float* zs = sm;
    float* red = sm + 32 * ld;
    const int co = blockIdx.x;
    const int b0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int per = Cin / 4;  // input channels per wave (even)
    // this wave's weight fragments first (the long-latency loads), then the latent tile
    const float* wp = w + ((long long)co * Cin + wave * per + h) * 32 + l31;
    for (int i = threadIdx.x; i < 32 * Cin; i += 256) {
        const int bl = i / Cin, ci = i - bl * Cin;
        zs[bl * ld + ci] = (b0 + bl < B) ? z[(long long)(b0 + bl) * Cin + ci] : 0.f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* zp = zs + l31 * ld + wave * per + h;
    for (int s0 = 0; s0 < per / 2; s0 += 8) {  // per / 2 is a multiple of 8 (Cin % 64 == 0)
        float wf[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wf[u] = wp[(long long)2 * (s0 + u) * 32];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zp[2 * (s0 + u)], wf[u], acc, 0, 0, 0);
    }
    // acc: lane column = tap, register r = sample (r&3) + 8 (r>>2) + 4h
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 33 + l31] = acc[r];
    __syncthreads();
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = threadIdx.x + 256 * i;
        const int bl = e >> 5, tap = e & 31;
        const int b = b0 + bl;
        if (b < B) {
            const float tot = (red[(0 * 32 + bl) * 33 + tap] + red[(1 * 32 + bl) * 33 + tap]) +
                              (red[(2 * 32 + bl) * 33 + tap] + red[(3 * 32 + bl) * 33 + tap]);
            float v = apply_act(tot * in_scale + bv, act, act_param);
            if (drop) v *= drop[b * Cout + co];
            y[((long long)b * Cout + co) * 32 + (31 - tap)] = v;
        }
    }
}

// ---- head wgrad: grid (Cout, ceil(Cin/64)), threads = 32 taps x 8 channel lanes ------------------
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                         float* __restrict__ gw, float* __restrict__ gb, int B, int Cin,
                                                         int Cout, int KT, float in_scale) {
    __shared__ float gsh[32][33];
    __shared__ float zsh[32][64];
    __shared__ float red[4];
    const int co = blockIdx.x;
    const int ci0 = blockIdx.y * 64;
    const int tp = threadIdx.x & 31, cq = threadIdx.x >> 5;
    float bsum = 0.f;
    for (int t0 = 0; t0 < KT; t0 += 32) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int b0 = 0; b0 < B; b0 += 32) {
            __syncthreads();
            for (int i = threadIdx.x; i < 32 * 32; i += 256) {
                const int bl = i >> 5, tt = i & 31;  // tt = weight tap offset; time = KT-1-(t0+tt)
                const int tw = t0 + tt;
                float v = 0.f;
                if (b0 + bl < B && tw < KT) v = gy[((long long)(b0 + bl) * Cout + co) * KT + (KT - 1 - tw)];
                gsh[bl][tt] = v;
                bsum += v;
            }
            for (int i = threadIdx.x; i < 32 * 64; i += 256) {
                const int bl = i >> 6, cl = i & 63;
                zsh[bl][cl] = (b0 + bl < B && ci0 + cl < Cin) ? z[(long long)(b0 + bl) * Cin + ci0 + cl] : 0.f;
            }
            __syncthreads();
#pragma unroll 4
            for (int bl = 0; bl < 32; ++bl) {
                const float g = gsh[bl][tp];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += g * zsh[bl][cq + 8 * i];
            }
        }
        const int tw = t0 + tp;
        if (tw < KT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ci = ci0 + cq + 8 * i;
                if (ci < Cin) gw[((long long)co * Cin + ci) * KT + tw] = acc[i] * in_scale;
            }
        }
    }
    if (gb && blockIdx.y == 0) {
        const float tot = block_sum_256(bsum, red);
        if (threadIdx.x == 0) gb[co] = tot;
    }
}

// ---- head dgrad: grid (Cin, ceil(B/32)) ----------------------------------------------------------
__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                         float* __restrict__ gz, int B, int Cin, int Cout, int KT,
                                                         float in_scale) {
    const int ci = blockIdx.x;
    const int b0 = blockIdx.y * 32;
    const int tp = threadIdx.x & 31, bq = threadIdx.x >> 5;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < KT; t0 += 32) {
        const int tw = t0 + tp;
        if (tw >= KT) continue;
        for (int co = 0; co < Cout; ++co) {
            const float wv = w[((long long)co * Cin + ci) * KT + tw];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + bq + 8 * i;
                if (b < B) acc[i] += wv * gy[((long long)b * Cout + co) * KT + (KT - 1 - tw)];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const int b = b0 + bq + 8 * i;
        if (tp == 0 && b < B) gz[(long long)b * Cin + ci] = v * in_scale;
    }
}

// ---- tail forward: grid (ceil(Cout/4), ceil(B/8)); every thread strides the reduction axis ------
__global__ __launch_bounds__(256) void tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ drop,
                                                       float* __restrict__ y, int B, int Kd, int Cout, float in_scale,
                                                       int act, float act_param) {
    __shared__ float red[4][32];
    const int co0 = blockIdx.x * 4;
    const int b0 = blockIdx.y * 8;
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k = threadIdx.x * 4; k < Kd; k += 1024) {
        f32x4 wv[4], xv[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + i < Cout ? co0 + i : Cout - 1;
            wv[i] = *reinterpret_cast<const f32x4*>(w + (long long)co * Kd + k);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = b0 + j < B ? b0 + j : B - 1;
            xv[j] = *reinterpret_cast<const f32x4*>(x + (long long)b * Kd + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc[i][j] += (wv[i][0] * xv[j][0] + wv[i][1] * xv[j][1]) + (wv[i][2] * xv[j][2] + wv[i][3] * xv[j][3]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = wave_sum(acc[i][j]);
            if (lane == 0) red[wave][i * 8 + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int i = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int co = co0 + i, b = b0 + j;
        if (co < Cout && b < B) {
            float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            v = apply_act(v * in_scale + (bias ? bias[co] : 0.f), act, act_param);
            if (drop) v *= drop[b * Cout + co];
            y[(long long)b * Cout + co] = v;
        }
    }
}

// ---- tail dgrad: grid (ceil(Kd/256), ceil(B/8)); 4 channel groups x 64 float4 columns -----------
__global__ __launch_bounds__(256) void tail_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                         float* __restrict__ gx, int B, int Kd, int Cout,
                                                         float in_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gsh = sm;                    // [8][Cout]
    f32x4* part = reinterpret_cast<f32x4*>(sm + ((8 * Cout + 3) & ~3));  // [4][8][64]
    const int b0 = blockIdx.y * 8;
    for (int i = threadIdx.x; i < 8 * Cout; i += 256) {
        const int j = i / Cout, co = i - j * Cout;
        gsh[i] = (b0 + j < B) ? gy[(long long)(b0 + j) * Cout + co] : 0.f;
    }
    __syncthreads();
    const int cg = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int k = blockIdx.x * 256 + l * 4;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k < Kd) {
        // eight weight rows in flight per thread: the loop is bound by the latency of these loads
        int co = cg;
        for (; co + 28 < Cout; co += 32) {
            f32x4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const f32x4*>(w + (long long)(co + 4 * u) * Kd + k);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = gsh[j * Cout + co + 4 * u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] += g * wv[u][e];
                }
        }
        for (; co < Cout; co += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (long long)co * Kd + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = gsh[j * Cout + co];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] += g * wv[e];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) part[(cg * 8 + j) * 64 + l] = acc[j];
    __syncthreads();
    // 8 samples x 64 columns = 512 float4 results, 2 per thread
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int j = i >> 6, ll = i & 63;
        const int kk = blockIdx.x * 256 + ll * 4;
        if (b0 + j < B && kk < Kd) {
            f32x4 s = part[(0 * 8 + j) * 64 + ll];
#pragma unroll
            for (int g = 1; g < 4; ++g) {
                const f32x4 p = part[(g * 8 + j) * 64 + ll];
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += p[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] *= in_scale;
            *reinterpret_cast<f32x4*>(gx + (long long)(b0 + j) * Kd + kk) = s;
        }
    }
}

// ---- tail wgrad: grid (ceil(Kd/1024), ceil(Cout/8)) ------------------------------------------------
__global__ __launch_bounds__(256) void tail_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ gw, float* __restrict__ gb, int B, int Kd,
                                                         int Cout, float in_scale) {
    extern __shared__ float gsh[];  // [B][8]
    const int co0 = blockIdx.y * 8;
    for (int i = threadIdx.x; i < B * 8; i += 256) {
        const int b = i >> 3, c = i & 7;
        gsh[i] = (co0 + c < Cout) ? gy[(long long)b * Cout + co0 + c] : 0.f;
    }
    __syncthreads();
    const int k = blockIdx.x * 1024 + threadIdx.x * 4;
    if (k < Kd) {
        f32x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        int b = 0;
        for (; b + 8 <= B; b += 8) {  // eight sample rows in flight per thread (latency-bound loop)
            f32x4 xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + (long long)(b + u) * Kd + k);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float g = gsh[(b + u) * 8 + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] += g * xv[u][e];
                }
        }
        for (; b < B; ++b) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long long)b * Kd + k);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float g = gsh[b * 8 + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] += g * xv[e];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (co0 + c < Cout) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = acc[c][e] * in_scale;
                *reinterpret_cast<f32x4*>(gw + (long long)(co0 + c) * Kd + k) = o;
            }
        }
    }
    if (gb && blockIdx.x == 0 && threadIdx.x < 8 && co0 + threadIdx.x < Cout) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += gsh[b * 8 + threadIdx.x];
        gb[co0 + threadIdx.x] = s;
    }
}

// accumulator layout of the 32 x 32 matrix instructions: register r of a lane in half-wave h holds this row
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- the tail layer's weight gradient on the matrix cores, for large batches -----------------------------------------
// gw[Cout, Kd] = in_scale * gy^T[Cout, B] x[B, Kd]: the reduction runs over the samples.  Grid (Kd / 128, Cout / 32); a
// wave owns a 32 x 32 tile (channels x columns) and walks the samples two at a time: for a fixed sample both operands
// are runs of 32 consecutive floats (gy[b][co0 ..], x[b][k0 ..]), read straight from global memory.  At 32-64
// samples the vector kernel above is faster (this layout stores 128-byte pieces of 256 rows: 9.6 / 15.6 against
// 8.1 / 12.6 us); from 128 samples on the vector ALUs' 2 B Cout Kd flops dominate (113 us at 512 samples).
__global__ __launch_bounds__(256) void tail_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                              float* __restrict__ gw, float* __restrict__ gb, int B, int Kd,
                                                              int Cout, float in_scale) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int k0 = (blockIdx.x * 4 + wave) * 32, co0 = blockIdx.y * 32;
    const float* gp = gy + co0 + l31;
    const float* xp = x + k0 + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 8;  // pairs of samples in flight
    for (int b0 = 0; b0 < B; b0 += 2 * U) {
        float gv[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + 2 * u + h;
            const int bc = b < B ? b : B - 1;  // (past the batch: a valid address, dropped below)
            gv[u] = gp[(long long)bc * Cout];
            xv[u] = xp[(long long)bc * Kd];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = b0 + 2 * u + h < B;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? gv[u] : 0.f, xv[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) gw[(long long)(co0 + acc_row(r, h)) * Kd + k0 + l31] = acc[r] * in_scale;
    if (gb && blockIdx.x == 0 && wave == 0 && h == 0) {
        float sum = 0.f;
        for (int b = 0; b < B; ++b) sum += gp[(long long)b * Cout];
        gb[co0 + l31] = sum;
    }
}

// ---- the tail layer's input gradient on the matrix cores -----------------------------------------------------------
// gx[B, Kd] = gy[B, Cout] w[Cout, Kd] has a long OUTPUT axis (Kd = 8192 columns) and a short reduction (Cout): every
// workgroup owns 32 columns and streams its 32-column strip of the weights once, in full 128-byte rows, through
// v_mfma_f32_32x32x2_f32 (fp32 operands, no splitting: ~0.3 GFLOP per launch): 7-8 us against 13 us for the
// vector-ALU kernel below at 32-64 samples.  acc = mfma(a, b, acc) adds a (row i = this lane's first-operand index,
// lane & 31) times b (column j = lane & 31) for the reduction index the lane half (lane >> 5) stands for; the result
// has column j in lane j and row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in register r.  The four waves of a workgroup
// split the reduction and add their tiles through LDS.  (The forward pass as a split-reduction product follows
// below: a first form that read its operands straight from global memory lost to the vector kernel stand-alone,
// 16.1 against 13.7 us with its second launch; staged through LDS it wins inside the training step.  Measured and
// left on the vector ALUs: the weight gradient, whose accumulator layout stores 128-byte pieces of 256 rows where
// the vector kernel stores 4 KB runs: 9.6 / 15.6 against 8.1 / 12.6 us at 32 / 64 samples.)

// sum of the four waves' accumulator tiles (NT tiles each); every thread gets the sums of its 4 NT values back in
// out[]: value index v = tile * 1024 + r * 64 + lane for v = threadIdx.x + 256 i
template <int NT>
__device__ __forceinline__ void cross_wave_sum(const f32x16 (&acc)[NT], float* red, float (&out)[4 * NT]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {
        const int v = threadIdx.x + 256 * i;
        out[i] = (red[v] + red[NT * 1024 + v]) + (red[2 * NT * 1024 + v] + red[3 * NT * 1024 + v]);
    }
    __syncthreads();
}

// input gradient: gx[b][k] = in_scale * sum_co gy[b][co] w[co][k]; grid Kd / 32, a workgroup owns 32 columns k,
// its waves a quarter of the output channels each
__global__ __launch_bounds__(256) void tail_dgrad_mfma_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                              float* __restrict__ gx, int B, int Kd, int Cout,
                                                              float in_scale) {
    __shared__ float red[4 * 2 * 1024];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int k0 = blockIdx.x * 32;
    const int cq = Cout / 4, c0 = wave * cq;  // (Cout % 32 == 0)
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int ba = b0 + l31 < B ? b0 + l31 : B - 1, bb = b0 + 32 + l31 < B ? b0 + 32 + l31 : B - 1;
        const float* ga = gy + (long long)ba * Cout + c0 + 4 * h;
        const float* gb = gy + (long long)bb * Cout + c0 + 4 * h;
        const float* wp = w + (long long)(c0 + 4 * h) * Kd + k0 + l31;
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        for (int c = 0; c < cq; c += 8) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(ga + c);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(gb + c);
            float wv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) wv[e] = wp[(long long)(c + e) * Kd];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], wv[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[e], wv[e], acc[1], 0, 0, 0);
            }
        }
        float sums[8];
        cross_wave_sum<2>(acc, red, sums);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int v = threadIdx.x + 256 * i;
            const int t = v >> 10, r = (v >> 6) & 15, ln = v & 63;
            const int b = b0 + 32 * t + acc_row(r, ln >> 5);
            if (b < B) gx[(long long)b * Kd + k0 + (ln & 31)] = sums[i] * in_scale;
        }
    }
}

// Forward of the full-length layer as a product [B, Kd] x [Kd, Cout] with the reduction split over workgroups:
// grid (sample tiles x channel tiles, splits); a workgroup owns a 32 x 32 tile of (samples, channels) and Kd / splits
// reduction indices, its four waves a quarter of those each.  A lane reads 16 consecutive floats of its sample's and of
// its channel's row per step (a wave: whole 128-byte lines of 32 + 32 rows), the fp32 matrix instruction pairs element e
// of the lower half-wave with element e of the upper one.  The four waves' tiles meet in LDS; the partial tile goes to
// part[split][b][co] and tail_fwd_reduce_kernel sums the splits in order and applies the epilogue.  (The weights are
// 8 MB read once: at 256 workgroups every CU streams 32 KB of them instead of a workgroup re-reading rows through L2.)
constexpr int TFM_STEP = 32;  // reduction indices per wave and step

constexpr int TFM_CHUNK = 4 * TFM_STEP;  // reduction indices a workgroup stages per round (a wave computes a quarter)
constexpr int TFM_LD = TFM_CHUNK + 4;    // LDS row stride in floats (16-byte reads of 32 rows without bank conflicts)

__global__ __launch_bounds__(256) void tail_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ part, int B, int Kd, int Cout, int tiles_co,
                                                            int krange) {
    // Rows of x and w lie Kd floats (32 KB) apart: a matrix-instruction operand read straight from global memory
    // touches 32 of them per load.  So the tile is staged: a wave's load instruction takes two whole 512-byte row
    // segments (fully coalesced), the operands are read from LDS (one buffer, two barriers per chunk).
    __shared__ __attribute__((aligned(16))) float tile[64 * TFM_LD];  // rows 0-31: samples, 32-63: channels
    __shared__ float red[4 * 1024];
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    // Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own.  All tiles of one split read the
    // same columns of x and w: with splits % 8 == 0 the linear id is re-read as (XCD, split within the XCD, tile), so a
    // split's tiles share an L2 and its 32 + 32 row segments come over the fabric once instead of once per XCD.
    const int tiles = gridDim.x, splits = gridDim.y;
    int tile_id = blockIdx.x, split = blockIdx.y;
    if (splits % 8 == 0) {
        const int lin = blockIdx.x + tiles * blockIdx.y;
        const int xcd = lin & 7, j = lin >> 3;
        split = xcd * (splits >> 3) + j / tiles;
        tile_id = j - (j / tiles) * tiles;
    }
    const int tb = tile_id / tiles_co, tc = tile_id - tb * tiles_co;
    const int kbase = split * krange;
    // staging: thread -> (row pair, 16-byte column); wave `wave` takes rows 16 wave .. 16 wave + 15 of the 64
    const int s_col = 4 * l31;  // float offset within the 128-float segment
    const float* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 16 * wave + 2 * i + h;  // 0 .. 63
        if (row < 32) {
            const int b = tb * 32 + row < B ? tb * 32 + row : B - 1;
            src[i] = x + (long long)b * Kd + kbase + s_col;
        } else {
            src[i] = w + (long long)(tc * 32 + row - 32) * Kd + kbase + s_col;
        }
    }
    f32x16 acc[1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
    // The loads of up to four chunks (a workgroup's whole share at 64 samples) are issued before the first of them is
    // used: with one chunk in flight every round waited a full trip to memory for 0.4 us of matrix instructions.
    constexpr int DEPTH = 4;
    const int nchunk = krange / TFM_CHUNK;
    for (int c0 = 0; c0 < nchunk; c0 += DEPTH) {
        f32x4 st[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int c = c0 + d < nchunk ? c0 + d : c0;  // (past the share: a valid address, unused)
#pragma unroll
            for (int i = 0; i < 8; ++i) st[d][i] = *reinterpret_cast<const f32x4*>(src[i] + c * TFM_CHUNK);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (c0 + d < nchunk) {  // workgroup-uniform
                if (c0 + d > 0) __syncthreads();  // (the previous chunk's operand reads are done)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<f32x4*>(tile + (16 * wave + 2 * i + h) * TFM_LD + s_col) = st[d][i];
                __syncthreads();
                const float* xr = tile + l31 * TFM_LD + TFM_STEP * wave + 16 * h;
                const float* wr = tile + (32 + l31) * TFM_LD + TFM_STEP * wave + 16 * h;
                f32x4 xv[4], wv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xv[i] = *reinterpret_cast<const f32x4*>(xr + 4 * i);
                    wv[i] = *reinterpret_cast<const f32x4*>(wr + 4 * i);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[i][e], wv[i][e], acc[0], 0, 0, 0);
            }
        }
    }
    float sums[4];
    cross_wave_sum<1>(acc, red, sums);
    float* out = part + (long long)split * B * Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + 256 * i;
        const int r = (v >> 6) & 15, ln = v & 63;
        const int bo = tb * 32 + acc_row(r, ln >> 5);
        if (bo < B) out[(long long)bo * Cout + tc * 32 + (ln & 31)] = sums[i];
    }
}

__global__ __launch_bounds__(256) void tail_fwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                              const float* __restrict__ drop, float* __restrict__ y, int n,
                                                              int Cout, int splits, float in_scale, int act,
                                                              float act_param) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // (the splits in order, eight loads in flight: one at a time this loop is a chain of trips to memory)
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += part[(long long)k * n + i];
    const int co = i % Cout;
    float v = apply_act(s * in_scale + (bias ? bias[co] : 0.f), act, act_param);
    if (drop) v *= drop[i];
    y[i] = v;
}

// splits of the reduction for the matrix-core forward: enough workgroups for the chip, whole steps per wave
int tail_fwd_splits(int B, int Kd, int Cout) {
    const int tiles = ((B + 31) / 32) * (Cout / 32);
    int s = 1;
    while (s * 2 * tiles <= num_cus() && Kd % (s * 2 * TFM_CHUNK) == 0) s *= 2;
    return s;
}

// (comparison switch: EEGGAN_TAIL_MFMA=0 keeps the forward on the vector-ALU kernel)
bool tail_fwd_mfma_shape(int Kd, int Cout) {
    static const bool on = [] {
        const char* e = getenv("EEGGAN_TAIL_MFMA");
        return !(e && e[0] == '0');
    }();
    return on && Kd % TFM_CHUNK == 0 && Cout % 32 == 0;
}

bool tail_mfma_shape(int Kd, int Cout) { return Kd % 32 == 0 && Cout % 32 == 0; }

int check_ht(const char* what, int B, int Cin, int Cout, int KT) {
    EEGGAN_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && KT > 0, "%s: non-positive dimension", what);
    EEGGAN_REQUIRE(B <= 4096, "%s: batch %d above the supported 4096", what, B);
    EEGGAN_REQUIRE((long long)Cin * KT * (long long)(Cout > B ? Cout : B) < (1ll << 31), "%s: operand too large", what);
    return 0;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_head_fwd(const float* z, const float* w, const float* bias, const float* drop, float* y, int B,
                               int Cin, int Cout, int KT, float in_scale, int act, float act_param, void* stream) {
    if (check_ht("head_fwd", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(z && w && y, "head_fwd: null pointer");
    EEGGAN_REQUIRE(Cin <= 384, "head_fwd: Cin %d above the supported 384 (LDS tile)", Cin);
    if (KT == 32 && Cin % 64 == 0)
        hipLaunchKernelGGL(head_fwd_mfma_kernel, dim3(Cout, (B + 31) / 32), dim3(256),
                           (size_t)(32 * (Cin + 1) + 4 * 32 * 33) * sizeof(float), as_stream(stream), z, w, bias, drop,
                           y, B, Cin, Cout, in_scale, act, act_param);
    else
        hipLaunchKernelGGL(head_fwd_kernel, dim3(Cout, (B + 31) / 32), dim3(256), (size_t)32 * Cin * sizeof(float),
                           as_stream(stream), z, w, bias, drop, y, B, Cin, Cout, KT, in_scale, act, act_param);
    return check_launch("head_fwd");
}

extern "C" int eeggan_head_wgrad(const float* z, const float* gy, float* gw, float* gb, int B, int Cin, int Cout,
                                 int KT, float in_scale, void* stream) {
    if (check_ht("head_wgrad", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(z && gy && gw, "head_wgrad: null pointer");
    hipLaunchKernelGGL(head_wgrad_kernel, dim3(Cout, (Cin + 63) / 64), dim3(256), 0, as_stream(stream), z, gy, gw, gb,
                       B, Cin, Cout, KT, in_scale);
    return check_launch("head_wgrad");
}

extern "C" int eeggan_head_dgrad(const float* gy, const float* w, float* gz, int B, int Cin, int Cout, int KT,
                                 float in_scale, void* stream) {
    if (check_ht("head_dgrad", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(gy && w && gz, "head_dgrad: null pointer");
    hipLaunchKernelGGL(head_dgrad_kernel, dim3(Cin, (B + 31) / 32), dim3(256), 0, as_stream(stream), gy, w, gz, B, Cin,
                       Cout, KT, in_scale);
    return check_launch("head_dgrad");
}

extern "C" int eeggan_tail_fwd(const float* x, const float* w, const float* bias, const float* drop, float* y, int B,
                               int Cin, int Cout, int KT, float in_scale, int act, float act_param, void* stream) {
    if (check_ht("tail_fwd", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(x && w && y, "tail_fwd: null pointer");
    const int Kd = Cin * KT;
    EEGGAN_REQUIRE(Kd % 4 == 0, "tail_fwd: Cin*KT = %d must be a multiple of 4", Kd);
    hipLaunchKernelGGL(tail_fwd_kernel, dim3((Cout + 3) / 4, (B + 7) / 8), dim3(256), 0, as_stream(stream), x, w, bias,
                       drop, y, B, Kd, Cout, in_scale, act, act_param);
    return check_launch("tail_fwd");
}

extern "C" size_t eeggan_tail_fwd_workspace(int B, int Cin, int Cout, int KT) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || KT <= 0) return 0;
    const int Kd = Cin * KT;
    if (!tail_fwd_mfma_shape(Kd, Cout)) return 0;
    return (size_t)tail_fwd_splits(B, Kd, Cout) * B * Cout * sizeof(float);
}

extern "C" int eeggan_tail_fwd_ws(const float* x, const float* w, const float* bias, const float* drop, float* y, int B,
                                  int Cin, int Cout, int KT, float in_scale, int act, float act_param, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (check_ht("tail_fwd", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(x && w && y, "tail_fwd: null pointer");
    const int Kd = Cin * KT;
    const size_t need = eeggan_tail_fwd_workspace(B, Cin, Cout, KT);
    if (need == 0 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) != 0)
        return eeggan_tail_fwd(x, w, bias, drop, y, B, Cin, Cout, KT, in_scale, act, act_param, stream);
    EEGGAN_REQUIRE(workspace && workspace_bytes >= need, "tail_fwd: workspace of %zu bytes needed, %zu given", need,
                   workspace_bytes);
    const int splits = tail_fwd_splits(B, Kd, Cout);
    const int tiles_co = Cout / 32;
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(tail_fwd_mfma_kernel, dim3((unsigned)(((B + 31) / 32) * tiles_co), (unsigned)splits), dim3(256), 0,
                       as_stream(stream), x, w, part, B, Kd, Cout, tiles_co, Kd / splits);
    if (check_launch("tail_fwd")) return 1;
    const int n = B * Cout;
    hipLaunchKernelGGL(tail_fwd_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), part, bias,
                       drop, y, n, Cout, splits, in_scale, act, act_param);
    return check_launch("tail_fwd reduce");
}

extern "C" int eeggan_tail_dgrad(const float* gy, const float* w, float* gx, int B, int Cin, int Cout, int KT,
                                 float in_scale, void* stream) {
    if (check_ht("tail_dgrad", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(gy && w && gx, "tail_dgrad: null pointer");
    const int Kd = Cin * KT;
    EEGGAN_REQUIRE(Kd % 4 == 0, "tail_dgrad: Cin*KT = %d must be a multiple of 4", Kd);
    if (tail_mfma_shape(Kd, Cout) && (reinterpret_cast<uintptr_t>(gy) & 15) == 0) {
        hipLaunchKernelGGL(tail_dgrad_mfma_kernel, dim3(Kd / 32), dim3(256), 0, as_stream(stream), gy, w, gx, B, Kd, Cout,
                           in_scale);
        return check_launch("tail_dgrad");
    }
    EEGGAN_REQUIRE(Cout <= 1024, "tail_dgrad: Cout %d above the supported 1024 (LDS tile)", Cout);
    const size_t lds = ((size_t)((8 * Cout + 3) & ~3) + 4 * 8 * 64 * 4) * sizeof(float);
    hipLaunchKernelGGL(tail_dgrad_kernel, dim3((Kd + 255) / 256, (B + 7) / 8), dim3(256), lds, as_stream(stream), gy, w,
                       gx, B, Kd, Cout, in_scale);
    return check_launch("tail_dgrad");
}

extern "C" int eeggan_tail_wgrad(const float* x, const float* gy, float* gw, float* gb, int B, int Cin, int Cout,
                                 int KT, float in_scale, void* stream) {
    if (check_ht("tail_wgrad", B, Cin, Cout, KT)) return 1;
    EEGGAN_REQUIRE(x && gy && gw, "tail_wgrad: null pointer");
    const int Kd = Cin * KT;
    EEGGAN_REQUIRE(Kd % 4 == 0, "tail_wgrad: Cin*KT = %d must be a multiple of 4", Kd);
    if (B >= 128 && Kd % 128 == 0 && Cout % 32 == 0) {
        hipLaunchKernelGGL(tail_wgrad_mfma_kernel, dim3(Kd / 128, Cout / 32), dim3(256), 0, as_stream(stream), x, gy, gw, gb, B,
                           Kd, Cout, in_scale);
        return check_launch("tail_wgrad");
    }
    hipLaunchKernelGGL(tail_wgrad_kernel, dim3((Kd + 1023) / 1024, (Cout + 7) / 8), dim3(256),
                       (size_t)B * 8 * sizeof(float), as_stream(stream), x, gy, gw, gb, B, Kd, Cout, in_scale);
    return check_launch("tail_wgrad");
}
