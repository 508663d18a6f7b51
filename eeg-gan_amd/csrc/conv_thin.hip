// 1x1 convolutions with a thin side: at most 8 input channels (from_rgb, 5 -> 64) or at most 8
// output channels (to_rgb, 64 -> 5, and the input gradients of both).  They move C*T floats per
// sample for a handful of multiply-adds each, so they are bandwidth-bound and run on the vector ALU:
// a thread owns W consecutive time steps of one sample, keeps the thin side in registers and streams
// the wide side, every load and store of a wave a run of full cache lines.  Same epilogues and fused
// input mask as the matrix-core kernels (bias -> activation -> per-(sample, channel) multiplier).
#include "common.h"
#include <cstdlib>
#include "conv_args.h"

#include <cstdint>

namespace eeggan {
namespace {

constexpr int THIN = 8;

template <int W>
struct Vec {
    float v[W];
};

template <int W, typename T>
__device__ __forceinline__ Vec<W> load_vec(const T* p) {
    Vec<W> r;
    if (W == 4) {
        const f32x4 t = ld4(p);
#pragma unroll
        for (int e = 0; e < W; ++e) r.v[e] = t[e];
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e) r.v[e] = ld1(p + e);
    }
    return r;
}

template <int W, typename T>
__device__ __forceinline__ void store_vec(T* p, const Vec<W>& r) {
    if (W == 4) {
        f32x4 t;
#pragma unroll
        for (int e = 0; e < W; ++e) t[e] = r.v[e];
        st4(p, t);
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e) st1(p + e, r.v[e]);
    }
}

// x * (fused backward mask of the layer that produced the gradient x): see eeggan_conv1d
template <int W, typename XT>
__device__ __forceinline__ Vec<W> load_input(const ConvArgs& a, int b, int ci, long long off) {
    Vec<W> x = load_vec<W>(reinterpret_cast<const XT*>(a.x) + off);
    if (a.mask_y) {  // uniform
        const Vec<W> my = load_vec<W>(reinterpret_cast<const XT*>(a.mask_y) + off);
        const float md = a.mask_drop ? a.mask_drop[b * a.Cin + ci] : 1.f;
#pragma unroll
        for (int e = 0; e < W; ++e) x.v[e] *= my.v[e] > 0.f ? md : md * a.mask_slope;
    }
    return x;
}

template <int W, typename YT>
__device__ __forceinline__ void finish(const ConvArgs& a, Vec<W> v, int b, int co, long long off) {
    const float bias = a.bias ? a.bias[co] : 0.f;
#pragma unroll
    for (int e = 0; e < W; ++e) v.v[e] = v.v[e] * a.in_scale + bias;
    if (a.act == EEGGAN_ACT_LRELU) {
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] = v.v[e] > 0.f ? v.v[e] : v.v[e] * a.act_param;
    } else if (a.act == EEGGAN_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] = tanhf(v.v[e] * a.act_param);
    } else if (a.act == EEGGAN_ACT_MASK) {
        const Vec<W> m = load_vec<W>(reinterpret_cast<const YT*>(a.act_aux) + off);
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] = m.v[e] > 0.f ? v.v[e] : v.v[e] * a.act_param;
    }
    if (a.drop) {
        const float d = a.drop[b * a.Cout + co];
#pragma unroll
        for (int e = 0; e < W; ++e) v.v[e] *= d;
    }
    store_vec<W>(reinterpret_cast<YT*>(a.y) + off, v);
}

// THIN_IN: Cin <= 8, any Cout; a thread holds its Cin inputs and walks the output channels.
// !THIN_IN: Cout <= 8, any Cin; a thread holds its Cout sums and walks the input channels.
// XBF / YBF: x and mask_y / y and act_aux are bfloat16 in memory (bf16 storage mode: the thin side of these layers,
// the [B, 5, T] signals and their gradients, stays float32, so the two sides usually differ).
template <bool THIN_IN, int W, bool XBF = false, bool YBF = false>
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvArgs a) {
    typedef typename ActT<XBF>::T XT;
    typedef typename ActT<YBF>::T YT;
    const int TW = a.T / W;
    const long long items = (long long)a.B * TW;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= items) return;
    const int b = (int)(i / TW);
    const int t = (int)(i - (long long)b * TW) * W;
    const long long xin = (long long)b * a.Cin * a.T + t;
    const long long yout = (long long)b * a.Cout * a.T + t;
    if (THIN_IN) {
        Vec<W> x[THIN];
#pragma unroll
        for (int ci = 0; ci < THIN; ++ci) {
            if (ci < a.Cin) {
                x[ci] = load_input<W, XT>(a, b, ci, xin + (long long)ci * a.T);
            } else {
#pragma unroll
                for (int e = 0; e < W; ++e) x[ci].v[e] = 0.f;
            }
        }
        for (int co = 0; co < a.Cout; ++co) {
            Vec<W> s;
#pragma unroll
            for (int e = 0; e < W; ++e) s.v[e] = 0.f;
#pragma unroll
            for (int ci = 0; ci < THIN; ++ci) {
                const float wv = ci < a.Cin ? a.w[co * a.Cin + ci] : 0.f;  // uniform: scalar loads
#pragma unroll
                for (int e = 0; e < W; ++e) s.v[e] += wv * x[ci].v[e];
            }
            finish<W, YT>(a, s, b, co, yout + (long long)co * a.T);
        }
    } else {
        Vec<W> s[THIN];
#pragma unroll
        for (int co = 0; co < THIN; ++co)
#pragma unroll
            for (int e = 0; e < W; ++e) s[co].v[e] = 0.f;
        int ci = 0;
        for (; ci + 4 <= a.Cin; ci += 4) {  // four rows in flight
            Vec<W> x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = load_input<W, XT>(a, b, ci + u, xin + (long long)(ci + u) * a.T);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int co = 0; co < THIN; ++co) {
                    const float wv = co < a.Cout ? a.w[co * a.Cin + ci + u] : 0.f;
#pragma unroll
                    for (int e = 0; e < W; ++e) s[co].v[e] += wv * x[u].v[e];
                }
        }
        for (; ci < a.Cin; ++ci) {
            const Vec<W> x = load_input<W, XT>(a, b, ci, xin + (long long)ci * a.T);
#pragma unroll
            for (int co = 0; co < THIN; ++co) {
                const float wv = co < a.Cout ? a.w[co * a.Cin + ci] : 0.f;
#pragma unroll
                for (int e = 0; e < W; ++e) s[co].v[e] += wv * x.v[e];
            }
        }
#pragma unroll
        for (int co = 0; co < THIN; ++co)
            if (co < a.Cout) finish<W, YT>(a, s[co], b, co, yout + (long long)co * a.T);
    }
}

}  // namespace

// Used when 16-byte accesses are possible and the launch has enough workgroups per CU.  Below that a launch is bound
// by latency (one thread walks the whole wide side) and the matrix-core kernel, which pads the thin side to a
// 32-channel chunk but spreads the walk, is faster.  Measured per launch at T = 1920, hot caches, samples 32 / 64 /
// 96 / 128 / 160 / 256: thin OUTPUT side (to_rgb, 64 -> 5) matrix cores 13 / 18 / 27 / 35 / 42 / 65 us, this kernel
// - / 21 / 21 / 22 / 26 / 27 us: taken from half a workgroup per CU (~70 samples); thin INPUT side (from_rgb, 5 -> 64)
// matrix cores 13 / 16 / 22 / 33 / 39 / 45 us, this kernel - / 32 / 32 / 33 / 35 / 36 us: from one workgroup per CU
// (~140 samples).  At 512 samples it runs at 0.76-0.79 of the HBM peak against 0.36-0.38.
// (EEGGAN_THIN_MIN_WG_PER_CU overrides both thresholds: comparison runs.)
static long long thin_min_items(bool thin_output) {
    static const double forced = [] {
        const char* e = std::getenv("EEGGAN_THIN_MIN_WG_PER_CU");
        return e ? std::atof(e) : -1.0;
    }();
    const double per_cu = forced >= 0.0 ? forced : (thin_output ? 0.5 : 1.0);
    return (long long)(256.0 * per_cu * num_cus());
}

int launch_conv_thin(const ConvArgs& a, int K, hipStream_t st) {
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool typed = a.x_bf16 || a.y_bf16;  // (the matrix-core kernel has no bfloat16 form for these layers)
    if (K != 1 || (a.Cin > THIN && a.Cout > THIN) || a.act == ACT_LRELU_PN || !a.vec || !al16(a.x) || !al16(a.y) ||
        !al16(a.mask_y) || !al16(a.act_aux) || (!typed && (long long)a.B * (a.T / 4) < thin_min_items(a.Cout <= THIN)))
        return -1;
    const long long items = (long long)a.B * (a.T / 4);
    const unsigned grid = (unsigned)((items + 255) / 256);
    const int variant = (a.Cin <= THIN ? 4 : 0) + (a.x_bf16 ? 2 : 0) + (a.y_bf16 ? 1 : 0);
#define EEGGAN_THIN(IN, XB, YB) hipLaunchKernelGGL((conv_thin_kernel<IN, 4, XB, YB>), dim3(grid), dim3(256), 0, st, a)
    switch (variant) {
        case 0: EEGGAN_THIN(false, false, false); break;
        case 1: EEGGAN_THIN(false, false, true); break;
        case 2: EEGGAN_THIN(false, true, false); break;
        case 3: EEGGAN_THIN(false, true, true); break;
        case 4: EEGGAN_THIN(true, false, false); break;
        case 5: EEGGAN_THIN(true, false, true); break;
        case 6: EEGGAN_THIN(true, true, false); break;
        default: EEGGAN_THIN(true, true, true); break;
    }
#undef EEGGAN_THIN
    return check_launch("conv1d");
}

}  // namespace eeggan
