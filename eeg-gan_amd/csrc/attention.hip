// Building blocks of the reference's SelfAttention layer (layers.py:41-64):
//   query = Wq x [B, d, T];  key = maxpool4(Wk x) [B, d, T/4];  value = maxpool4(Wv x) [B, C/2, T/4]
//   P = softmax over the key axis of key^T query [B, T/4, T];  out = gamma * Wf (value P) + x
// The 1x1 convs are the ordinary conv kernels.  Here: batched matrix products with either operand transposed
// (scores, weighted values and all of their gradients are such products), the softmax over dim 1 with its backward
// and the backward's own derivative (the critic's gradient penalty differentiates twice), max pooling with stored
// argmax (its backward is a scatter, the backward of that a gather: both linear once the indices are fixed), and a
// deterministic dot product (gradient of the scalar gamma).
// The products have a reduction length of 8 .. 32 (scores) or T/4 .. T with 32 .. 128 rows (values): small
// matrices, bound by writing and reading the [B, T/4, T] map, so they run on the vector ALUs with LDS tiles.
#include "common.h"

namespace eeggan {
namespace {

// C[b] (M x N) = alpha * op(A[b]) op(B[b]);  A[b] is stored M x K (TA = false) or K x M (TA = true), B[b] K x N
// (TB = false) or N x K (TB = true), all row-major and dense.  64 x 64 tile per workgroup, 16 reduction indices per
// step, a thread owns a 4 x 4 block; sums run over k in order (deterministic).
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bmm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                  float* __restrict__ C, int M, int N, int K, float alpha) {
    __shared__ float As[16][68];
    __shared__ float Bs[16][68];
    const int b = blockIdx.z;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const float* Ab = A + (long long)b * M * K;
    const float* Bb = Bm + (long long)b * K * N;
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;  // 1024 elements of each tile
            int m, k;
            if (TA) { k = e >> 6; m = e & 63; } else { m = e >> 4; k = e & 15; }
            const bool ok = m0 + m < M && k0 + k < K;
            As[k][m] = ok ? (TA ? Ab[(long long)(k0 + k) * M + m0 + m] : Ab[(long long)(m0 + m) * K + k0 + k]) : 0.f;
            int n, kb;
            if (TB) { n = e >> 4; kb = e & 15; } else { kb = e >> 6; n = e & 63; }
            const bool okb = n0 + n < N && k0 + kb < K;
            Bs[kb][n] = okb ? (TB ? Bb[(long long)(n0 + n) * K + k0 + kb] : Bb[(long long)(k0 + kb) * N + n0 + n]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&As[k][4 * ty]);
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Bs[k][4 * tx]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], v[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* Cb = C + (long long)b * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 4 * ty + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * tx + j;
            if (n < N) Cb[(long long)m * N + n] = alpha * acc[i][j];
        }
    }
}

// softmax over dim 1 of s [B, M, N]: a thread owns a column (b, n) and walks its M entries (the threads of a wave
// read neighbouring columns: coalesced).
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, float* __restrict__ p, int M, int N,
                                                          long long cols) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long b = c / N;
    const int n = (int)(c - b * N);
    const float* sc = s + b * M * N + n;
    float* pc = p + b * M * N + n;
    float mx = -INFINITY;
    for (int m = 0; m < M; ++m) mx = fmaxf(mx, sc[(long long)m * N]);
    float sum = 0.f;
    for (int m = 0; m < M; ++m) sum += expf(sc[(long long)m * N] - mx);
    const float inv = 1.f / sum;
    for (int m = 0; m < M; ++m) pc[(long long)m * N] = expf(sc[(long long)m * N] - mx) * inv;
}

// gx = y * (gy - sum_m gy y): the backward of the softmax (y its output), and also the derivative of this very
// expression with respect to gy (applied to that cotangent).
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                          float* __restrict__ gx, int M, int N, long long cols) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long b = c / N;
    const long long base = b * M * N + (c - b * N);
    float dot = 0.f;
    for (int m = 0; m < M; ++m) dot += gy[base + (long long)m * N] * y[base + (long long)m * N];
    for (int m = 0; m < M; ++m) {
        const long long i = base + (long long)m * N;
        gx[i] = y[i] * (gy[i] - dot);
    }
}

// derivative of gx = y (gy - sum gy y) with respect to y, applied to the cotangent q:
//   out = q (gy - sum gy y) - gy sum(q y)
__global__ __launch_bounds__(256) void softmax_bwd_dy_kernel(const float* __restrict__ q, const float* __restrict__ gy,
                                                             const float* __restrict__ y, float* __restrict__ out, int M,
                                                             int N, long long cols) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long b = c / N;
    const long long base = b * M * N + (c - b * N);
    float s = 0.f, r = 0.f;
    for (int m = 0; m < M; ++m) {
        const long long i = base + (long long)m * N;
        s += gy[i] * y[i];
        r += q[i] * y[i];
    }
    for (int m = 0; m < M; ++m) {
        const long long i = base + (long long)m * N;
        out[i] = q[i] * (gy[i] - s) - gy[i] * r;
    }
}

// y[r, t] = max_j x[r, P t + j], idx[r, t] = the first j that attains it (torch's choice on ties)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int Tin, int Tout, int P,
                                                          long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long r = i / Tout;
    const int t = (int)(i - r * Tout);
    const float* xr = x + r * Tin + (long long)P * t;
    float best = xr[0];
    int bj = 0;
    for (int j = 1; j < P; ++j) {
        const float v = xr[j];
        if (v > best) { best = v; bj = j; }
    }
    y[i] = best;
    idx[i] = (unsigned char)bj;
}

// gather: dst[r, t] = src[r, P t + idx[r, t]]   (dst [rows, Tout], src [rows, Tin])
__global__ __launch_bounds__(256) void maxpool_gather_kernel(const float* __restrict__ src, const unsigned char* __restrict__ idx,
                                                             float* __restrict__ dst, int Tin, int Tout, int P, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long r = i / Tout;
    const int t = (int)(i - r * Tout);
    dst[i] = src[r * Tin + (long long)P * t + idx[i]];
}

// scatter: dst[r, u] = src[r, u / P] if u % P == idx[r, u / P] (and u / P < Tout), else 0   (dst [rows, Tin])
__global__ __launch_bounds__(256) void maxpool_scatter_kernel(const float* __restrict__ src, const unsigned char* __restrict__ idx,
                                                              float* __restrict__ dst, int Tin, int Tout, int P, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long r = i / Tin;
    const int u = (int)(i - r * Tin);
    const int t = u / P;
    float v = 0.f;
    if (t < Tout && (u - t * P) == idx[r * Tout + t]) v = src[r * Tout + t];
    dst[i] = v;
}

// dot product in two deterministic stages: 256 partial sums in double, then their sum.
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          double* __restrict__ part, long long n) {
    __shared__ double red[256];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 256ll * gridDim.x) s += (double)a[i] * (double)b[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void dot_final_kernel(const double* __restrict__ part, float* __restrict__ out, int nparts) {
    __shared__ double red[256];
    red[threadIdx.x] = (int)threadIdx.x < nparts ? part[threadIdx.x] : 0.0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int trans_a,
                          int trans_b, float alpha, void* stream) {
    EEGGAN_REQUIRE(a && b && c, "bmm: null pointer");
    EEGGAN_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0, "bmm: bad shape (batch=%d M=%d N=%d K=%d)", batch, M, N, K);
    EEGGAN_REQUIRE(batch <= 65535 && (M + 63) / 64 <= 65535, "bmm: batch or row count beyond the grid limits");
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)batch);
    hipStream_t st = as_stream(stream);
    if (trans_a && trans_b) hipLaunchKernelGGL((bmm_kernel<true, true>), grid, dim3(256), 0, st, a, b, c, M, N, K, alpha);
    else if (trans_a) hipLaunchKernelGGL((bmm_kernel<true, false>), grid, dim3(256), 0, st, a, b, c, M, N, K, alpha);
    else if (trans_b) hipLaunchKernelGGL((bmm_kernel<false, true>), grid, dim3(256), 0, st, a, b, c, M, N, K, alpha);
    else hipLaunchKernelGGL((bmm_kernel<false, false>), grid, dim3(256), 0, st, a, b, c, M, N, K, alpha);
    return check_launch("bmm");
}

extern "C" int eeggan_softmax_dim1_fwd(const float* s, float* p, int B, int M, int N, void* stream) {
    EEGGAN_REQUIRE(s && p && B > 0 && M > 0 && N > 0, "softmax: bad arguments");
    const long long cols = (long long)B * N;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(blocks_for(cols)), dim3(256), 0, as_stream(stream), s, p, M, N, cols);
    return check_launch("softmax forward");
}

extern "C" int eeggan_softmax_dim1_bwd(const float* gy, const float* y, float* gx, int B, int M, int N, void* stream) {
    EEGGAN_REQUIRE(gy && y && gx && B > 0 && M > 0 && N > 0, "softmax backward: bad arguments");
    const long long cols = (long long)B * N;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(blocks_for(cols)), dim3(256), 0, as_stream(stream), gy, y, gx, M, N, cols);
    return check_launch("softmax backward");
}

extern "C" int eeggan_softmax_dim1_bwd_dy(const float* q, const float* gy, const float* y, float* out, int B, int M, int N,
                                          void* stream) {
    EEGGAN_REQUIRE(q && gy && y && out && B > 0 && M > 0 && N > 0, "softmax backward (dy): bad arguments");
    const long long cols = (long long)B * N;
    hipLaunchKernelGGL(softmax_bwd_dy_kernel, dim3(blocks_for(cols)), dim3(256), 0, as_stream(stream), q, gy, y, out, M, N,
                       cols);
    return check_launch("softmax backward (dy)");
}

extern "C" int eeggan_maxpool_fwd(const float* x, float* y, unsigned char* idx, size_t rows, int Tin, int pool, void* stream) {
    EEGGAN_REQUIRE(x && y && idx && rows > 0, "maxpool: bad arguments");
    EEGGAN_REQUIRE(pool >= 1 && pool <= 255 && Tin >= pool, "maxpool: window %d on length %d", pool, Tin);
    const int Tout = Tin / pool;
    const long long n = (long long)rows * Tout;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), x, y, idx, Tin, Tout, pool, n);
    return check_launch("maxpool forward");
}

extern "C" int eeggan_maxpool_select(const float* src, const unsigned char* idx, float* dst, size_t rows, int Tin, int pool,
                                     int scatter, void* stream) {
    EEGGAN_REQUIRE(src && dst && idx && rows > 0, "maxpool select: bad arguments");
    EEGGAN_REQUIRE(pool >= 1 && pool <= 255 && Tin >= pool, "maxpool select: window %d on length %d", pool, Tin);
    const int Tout = Tin / pool;
    if (scatter) {
        const long long n = (long long)rows * Tin;
        hipLaunchKernelGGL(maxpool_scatter_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), src, idx, dst, Tin,
                           Tout, pool, n);
    } else {
        const long long n = (long long)rows * Tout;
        hipLaunchKernelGGL(maxpool_gather_kernel, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), src, idx, dst, Tin,
                           Tout, pool, n);
    }
    return check_launch("maxpool select");
}

extern "C" size_t eeggan_dot_workspace(void) { return 256 * sizeof(double); }

extern "C" int eeggan_dot(const float* a, const float* b, float* out, size_t n, void* workspace, size_t workspace_bytes,
                          void* stream) {
    EEGGAN_REQUIRE(a && b && out && n > 0, "dot: bad arguments");
    EEGGAN_REQUIRE(workspace && workspace_bytes >= eeggan_dot_workspace(), "dot: workspace too small");
    long long blocks = ((long long)n + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    double* part = reinterpret_cast<double*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, b, part, (long long)n);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, part, out, (int)blocks);
    return check_launch("dot");
}
