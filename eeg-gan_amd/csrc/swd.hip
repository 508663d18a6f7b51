// Sliced Wasserstein distance between two sets of 1-D signals (reference plugins.py:383-395, 426-451):
// random patches -> per-component normalisation -> projection on random unit directions -> sort of every
// direction's projections -> mean absolute difference of the two sorted sets.
//
// All of it is byte moving and sorting, i.e. HBM-bound: the patches are gathered once into the reference's
// [patch, channel, time] order (which it then READS as rows of `channel` numbers, plugins.py:444-445), the
// projections are written direction-major as 64-bit keys (direction << 32 | order-preserving image of the
// float), so that ONE device-wide radix sort over 32 + log2(directions) bits sorts every direction's
// projections at once and leaves them contiguous per direction, and the distance is a two-stage sum in
// double precision in a fixed order (run-to-run deterministic).  The sort itself is rocPRIM's device radix
// sort: a tuned multi-pass sort is a library's job, like the collective library for the gradient exchange.
#include "common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

namespace eeggan {
namespace {

constexpr int SWD_MAXC = 16;
constexpr int SWD_BLOCKS = 1024;  // partial sums per reduction

// out[((j * ppi + k) * c + ch) * ps + p] = batch[(j * c + ch) * T + (starts[j * ppi + k] + p) * step]
__global__ __launch_bounds__(256) void swd_patches_kernel(const float* __restrict__ batch, const int32_t* __restrict__ starts,
                                                          float* __restrict__ out, long long total, int c, int T,
                                                          int step, int ppi, int ps) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i % ps);
        const long long r = i / ps;
        const int ch = (int)(r % c);
        const long long patch = r / c;
        const long long j = patch / ppi;
        out[i] = batch[(j * c + ch) * (long long)T + (long long)(starts[patch] + p) * step];
    }
}

// Column sums of desc[M, c] (or of squared deviations from `mean` when it is given): one partial per block
// and column in `part[block * c + col]`, in double precision.
__global__ __launch_bounds__(256) void swd_colsum_kernel(const float* __restrict__ desc, long long n, int c,
                                                         const double* __restrict__ mean_part, int nblocks_mean,
                                                         long long M, double* __restrict__ part) {
    __shared__ double sh[256];
    __shared__ double mean[SWD_MAXC];
    if (mean_part != nullptr) {
        if (threadIdx.x < c) {
            double s = 0.0;
            for (int b = 0; b < nblocks_mean; ++b) s += mean_part[b * c + threadIdx.x];
            mean[threadIdx.x] = s / (double)M;
        }
        __syncthreads();
    }
    // a thread walks elements i = t, t + stride, ... with stride a multiple of c: always the same column
    const long long stride = (long long)gridDim.x * 256 * c;
    double acc[SWD_MAXC];
    for (int k = 0; k < SWD_MAXC; ++k) acc[k] = 0.0;
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * c;
    for (long long i = base; i < n; i += stride) {
#pragma unroll 4
        for (int k = 0; k < c; ++k) {
            if (i + k < n) {
                const double v = (double)desc[i + k];
                const double d = mean_part ? v - mean[k] : v;
                acc[k] += mean_part ? d * d : d;
            }
        }
    }
    for (int k = 0; k < c; ++k) {
        sh[threadIdx.x] = acc[k];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) part[blockIdx.x * c + k] = sh[0];
        __syncthreads();
    }
}

// desc = (desc - mean) / (std + eps) per column, std unbiased (torch.std's default)
__global__ __launch_bounds__(256) void swd_apply_kernel(float* __restrict__ desc, long long n, int c,
                                                        const double* __restrict__ sum_part,
                                                        const double* __restrict__ sq_part, int nblocks, long long M,
                                                        float eps) {
    __shared__ float mean[SWD_MAXC], inv[SWD_MAXC];
    if (threadIdx.x < c) {
        double s = 0.0, q = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            s += sum_part[b * c + threadIdx.x];
            q += sq_part[b * c + threadIdx.x];
        }
        mean[threadIdx.x] = (float)(s / (double)M);
        inv[threadIdx.x] = 1.f / ((float)sqrt(q / (double)(M - 1)) + eps);
    }
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % c);
        desc[i] = (desc[i] - mean[k]) * inv[k];
    }
}

// order-preserving image of a float in an unsigned integer, and back
__device__ __forceinline__ uint32_t ordered(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// keys[d * M + m] = d << 32 | ordered(sum_k desc[m, k] * dirs[k, d]); thread = one row m, loop over directions
__global__ __launch_bounds__(256) void swd_project_kernel(const float* __restrict__ desc, const float* __restrict__ dirs,
                                                          unsigned long long* __restrict__ keys, long long M, int c,
                                                          int D) {
    extern __shared__ float dsh[];  // [c][D]
    for (int i = threadIdx.x; i < c * D; i += 256) dsh[i] = dirs[i];
    __syncthreads();
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float row[SWD_MAXC];
    for (int k = 0; k < c; ++k) row[k] = desc[m * c + k];
    for (int d = 0; d < D; ++d) {
        float s = 0.f;
        for (int k = 0; k < c; ++k) s = fmaf(row[k], dsh[k * D + d], s);
        keys[(long long)d * M + m] = ((unsigned long long)d << 32) | ordered(s);
    }
}

__global__ __launch_bounds__(256) void swd_absdiff_kernel(const unsigned long long* __restrict__ a,
                                                          const unsigned long long* __restrict__ b, long long n,
                                                          double* __restrict__ part) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        acc += (double)fabsf(unordered((uint32_t)a[i]) - unordered((uint32_t)b[i]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void swd_final_kernel(const double* __restrict__ part, int nblocks, long long n,
                                                        float* __restrict__ out) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += part[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(sh[0] / (double)n);
}

unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int key_bits(int D) {
    int bits = 32;
    while ((1ll << (bits - 32)) < D) ++bits;
    return bits;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_swd_patches(const float* batch, const int32_t* starts, float* out, int items, int c, int T,
                                  int step, int patches_per_item, int patch_size, void* stream) {
    EEGGAN_REQUIRE(batch && starts && out, "swd_patches: null pointer");
    EEGGAN_REQUIRE(items > 0 && c > 0 && T > 0 && step > 0 && patches_per_item > 0 && patch_size > 0,
                   "swd_patches: non-positive argument");
    const long long total = (long long)items * patches_per_item * c * patch_size;
    hipLaunchKernelGGL(swd_patches_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), batch, starts, out,
                       total, c, T, step, patches_per_item, patch_size);
    return check_launch("swd_patches");
}

extern "C" size_t eeggan_swd_normalize_workspace(void) { return (size_t)2 * SWD_BLOCKS * SWD_MAXC * sizeof(double); }

extern "C" int eeggan_swd_normalize(float* desc, size_t rows, int c, float eps, void* workspace, size_t workspace_bytes,
                                    void* stream) {
    EEGGAN_REQUIRE(desc && workspace, "swd_normalize: null pointer");
    EEGGAN_REQUIRE(rows > 1 && c > 0 && c <= SWD_MAXC, "swd_normalize: needs at least two rows and 1..%d columns", SWD_MAXC);
    EEGGAN_REQUIRE(workspace_bytes >= eeggan_swd_normalize_workspace(), "swd_normalize: workspace too small");
    EEGGAN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "swd_normalize: workspace must be 8-byte aligned");
    const long long n = (long long)rows * c;
    double* sums = static_cast<double*>(workspace);
    double* sqs = sums + (size_t)SWD_BLOCKS * SWD_MAXC;
    long long nb = (n + 256ll * c - 1) / (256ll * c);
    if (nb > SWD_BLOCKS) nb = SWD_BLOCKS;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(swd_colsum_kernel, dim3((unsigned)nb), dim3(256), 0, st, desc, n, c, nullptr, 0, (long long)rows, sums);
    if (check_launch("swd_normalize sums")) return 1;
    hipLaunchKernelGGL(swd_colsum_kernel, dim3((unsigned)nb), dim3(256), 0, st, desc, n, c, sums, (int)nb, (long long)rows, sqs);
    if (check_launch("swd_normalize deviations")) return 1;
    hipLaunchKernelGGL(swd_apply_kernel, dim3(grid_for(n)), dim3(256), 0, st, desc, n, c, sums, sqs, (int)nb,
                       (long long)rows, eps);
    return check_launch("swd_normalize");
}

extern "C" int eeggan_swd_project_keys(const float* desc, const float* dirs, uint64_t* keys, size_t rows, int c, int D,
                                       void* stream) {
    EEGGAN_REQUIRE(desc && dirs && keys, "swd_project_keys: null pointer");
    EEGGAN_REQUIRE(rows > 0 && c > 0 && c <= SWD_MAXC && D > 0 && D <= 4096, "swd_project_keys: bad shape");
    EEGGAN_REQUIRE((rows + 255) / 256 < (1ull << 31), "swd_project_keys: too many rows");
    hipLaunchKernelGGL(swd_project_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), (size_t)c * D * sizeof(float),
                       as_stream(stream), desc, dirs, reinterpret_cast<unsigned long long*>(keys), (long long)rows, c, D);
    return check_launch("swd_project_keys");
}

extern "C" size_t eeggan_swd_sort_workspace(size_t nkeys, int D) {
    if (nkeys == 0 || D <= 0) return 0;
    size_t bytes = 0;
    unsigned long long* none = nullptr;
    if (rocprim::radix_sort_keys(nullptr, bytes, none, none, nkeys, 0, (unsigned)key_bits(D)) != hipSuccess) return 0;
    return bytes < 256 ? 256 : bytes;
}

extern "C" int eeggan_swd_sort(const uint64_t* keys_in, uint64_t* keys_out, size_t nkeys, int D, void* workspace,
                               size_t workspace_bytes, void* stream) {
    EEGGAN_REQUIRE(keys_in && keys_out && workspace && keys_in != keys_out, "swd_sort: null or aliased pointer");
    EEGGAN_REQUIRE(nkeys > 0 && D > 0 && D <= 4096, "swd_sort: bad size");
    size_t need = workspace_bytes;
    hipError_t e = rocprim::radix_sort_keys(workspace, need, reinterpret_cast<const unsigned long long*>(keys_in),
                                            reinterpret_cast<unsigned long long*>(keys_out), nkeys, 0,
                                            (unsigned)key_bits(D), as_stream(stream));
    EEGGAN_REQUIRE(e == hipSuccess, "swd_sort: radix sort failed: %s (workspace %zu bytes)", hipGetErrorString(e),
                   workspace_bytes);
    return 0;
}

extern "C" size_t eeggan_swd_distance_workspace(void) { return (size_t)SWD_BLOCKS * sizeof(double); }

extern "C" int eeggan_swd_distance(const uint64_t* a, const uint64_t* b, size_t nkeys, float* out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    EEGGAN_REQUIRE(a && b && out && workspace, "swd_distance: null pointer");
    EEGGAN_REQUIRE(nkeys > 0 && workspace_bytes >= eeggan_swd_distance_workspace(), "swd_distance: bad size");
    EEGGAN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "swd_distance: workspace must be 8-byte aligned");
    long long nb = ((long long)nkeys + 255) / 256;
    if (nb > SWD_BLOCKS) nb = SWD_BLOCKS;
    double* part = static_cast<double*>(workspace);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(swd_absdiff_kernel, dim3((unsigned)nb), dim3(256), 0, st,
                       reinterpret_cast<const unsigned long long*>(a), reinterpret_cast<const unsigned long long*>(b),
                       (long long)nkeys, part);
    if (check_launch("swd_distance partial")) return 1;
    hipLaunchKernelGGL(swd_final_kernel, dim3(1), dim3(256), 0, st, part, (int)nb, (long long)nkeys, out);
    return check_launch("swd_distance");
}
