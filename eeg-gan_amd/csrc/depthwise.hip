// Depthwise Conv1d (groups = channels): the first half of the reference's separable layers
// (EqualizedSeparableConv1d, layers.py:137-147: a per-channel conv followed by a 1x1 conv; the 1x1 half is the
// ordinary conv kernel).  A handful of multiply-adds per element: bandwidth-bound, vector ALU.
//   y[b,c,t] = bias[c] + in_scale * sum_j w[c, flip ? K-1-j : j] * x[b,c,t+j-pad],  t in [0, Tout),
//   Tout = Tin + 2 pad - K + 1; any kernel size and padding (the full-length layers are K = 32 with pad 31 or 0).
// The input gradient is the same kernel on gy with flip toggled and pad' = K-1-pad.
#include "common.h"

namespace eeggan {
namespace {

constexpr int DW_MAXK = 64;

__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int C, int Tin,
                                                     int Tout, int K, int pad, float in_scale, int flip, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int t = (int)(i % Tout);
    const long long row = i / Tout;  // b * C + c
    const int c = (int)(row % C);
    const float* xr = x + row * Tin;
    const float* wr = w + (long long)c * K;
    float s = 0.f;
    for (int j = 0; j < K; ++j) {
        const int u = t + j - pad;
        if (u >= 0 && u < Tin) s += wr[flip ? K - 1 - j : j] * xr[u];
    }
    y[i] = s * in_scale + (bias ? bias[c] : 0.f);
}

// gw[c, flip ? K-1-j : j] = in_scale * sum_{b,t} gy[b,c,t] * x[b,c,t+j-pad];  gb[c] = sum_{b,t} gy[b,c,t].
// One workgroup per channel; a thread walks its (b, t) positions in a fixed order and keeps the K sums of a group of
// up to 8 taps in registers; the workgroup's sums are added in a fixed order too (deterministic).
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           float* __restrict__ gw, float* __restrict__ gb, int B, int C,
                                                           int Tin, int Tout, int K, int pad, float in_scale, int flip) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const long long per = (long long)B * Tout;
    for (int j0 = 0; j0 < K; j0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        for (long long i = threadIdx.x; i < per; i += 256) {
            const int b = (int)(i / Tout), t = (int)(i - (long long)b * Tout);
            const float g = gy[((long long)b * C + c) * Tout + t];
            const float* xr = x + ((long long)b * C + c) * Tin;
            bsum += g;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int u = t + j0 + e - pad;
                if (j0 + e < K && u >= 0 && u < Tin) acc[e] += g * xr[u];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float tot = block_sum_256(acc[e], red);
            if (threadIdx.x == 0 && j0 + e < K) gw[(long long)c * K + (flip ? K - 1 - (j0 + e) : j0 + e)] = tot * in_scale;
        }
        if (j0 == 0 && gb) {
            const float tot = block_sum_256(bsum, red);
            if (threadIdx.x == 0) gb[c] = tot;
        }
    }
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_dwconv1d(const float* x, const float* w, const float* bias, float* y, int B, int C, int Tin, int K,
                               int pad, float in_scale, int flip, void* stream) {
    EEGGAN_REQUIRE(x && w && y, "dwconv1d: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && Tin > 0 && K > 0 && K <= DW_MAXK && pad >= 0 && pad < K, "dwconv1d: bad shape B=%d C=%d "
                   "T=%d K=%d pad=%d", B, C, Tin, K, pad);
    const int Tout = Tin + 2 * pad - K + 1;
    EEGGAN_REQUIRE(Tout > 0, "dwconv1d: the kernel (%d, pad %d) is longer than the padded signal (%d)", K, pad, Tin);
    const long long n = (long long)B * C * Tout;
    EEGGAN_REQUIRE(n < (1ll << 40), "dwconv1d: tensor too large");
    hipLaunchKernelGGL(dwconv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, w, bias, y, C,
                       Tin, Tout, K, pad, in_scale, flip ? 1 : 0, n);
    return check_launch("dwconv1d");
}

extern "C" int eeggan_dwconv1d_wgrad(const float* x, const float* gy, float* gw, float* gb, int B, int C, int Tin, int K,
                                     int pad, float in_scale, int flip, void* stream) {
    EEGGAN_REQUIRE(x && gy && gw, "dwconv1d_wgrad: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && Tin > 0 && K > 0 && K <= DW_MAXK && pad >= 0 && pad < K, "dwconv1d_wgrad: bad shape");
    const int Tout = Tin + 2 * pad - K + 1;
    EEGGAN_REQUIRE(Tout > 0, "dwconv1d_wgrad: the kernel is longer than the padded signal");
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(C), dim3(256), 0, as_stream(stream), x, gy, gw, gb, B, C, Tin, Tout, K,
                       pad, in_scale, flip ? 1 : 0);
    return check_launch("dwconv1d_wgrad");
}
