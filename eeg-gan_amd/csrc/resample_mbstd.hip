// Rational resampler (one sparse-row gather kernel for both directions) and the
// MinibatchStddev layer with its first- and second-order backward.
#include "common.h"

#include <mutex>

namespace eeggan {
namespace {

// ---- resample -----------------------------------------------------------------------------------
// Thread = one output position m for ROWS_PER_BLOCK consecutive rows; its taps stay in registers.
constexpr int RS_ROWS = 8;
constexpr int RS_MAXTAPS = 16;

// (TI / TO: float, or __bf16 for a side that is stored as bfloat16; sums in float)
template <int TAPS, typename TI = float, typename TO = float>
__global__ __launch_bounds__(256) void resample_kernel(const TI* __restrict__ x, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ coef, TO* __restrict__ out,
                                                       long long rows, int Tin, int Tout) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Tout) return;
    int ii[TAPS];
    float cc[TAPS];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        ii[k] = idx[m * TAPS + k];
        cc[k] = coef[m * TAPS + k];
    }
    const long long r0 = (long long)blockIdx.y * RS_ROWS;
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const long long row = r0 + r;
        if (row < rows) {
            const TI* xr = x + row * Tin;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) s += cc[k] * ld1(xr + ii[k]);
            st1(out + row * Tout + m, s);
        }
    }
}

// The same gather for tables wider than RS_MAXTAPS (the transpose of an upsampling by p has about 2 p + 2 taps per
// position: conditioning signals of a few steps stretched to a layer's length): the taps are walked in a loop
// instead of living in registers.  One thread per (output position, row).
constexpr int RS_WIDE_MAXTAPS = 1024;

template <typename TI = float, typename TO = float>
__global__ __launch_bounds__(256) void resample_wide_kernel(const TI* __restrict__ x, const int32_t* __restrict__ idx,
                                                            const float* __restrict__ coef, TO* __restrict__ out,
                                                            long long rows, int Tin, int Tout, int taps) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    const long long row = blockIdx.y;
    if (m >= Tout || row >= rows) return;
    const TI* xr = x + row * Tin;
    float s = 0.f;
    for (int k = 0; k < taps; ++k) s += coef[(long long)m * taps + k] * ld1(xr + idx[(long long)m * taps + k]);
    st1(out + row * Tout + m, s);
}

// Quad form: a thread produces 4 consecutive output positions (one 16-byte store per row).  Their taps
// fall into a span of at most 4*NV input samples starting at base[q] (a multiple of 4), which the thread
// loads as NV 16-byte vectors; the host hands the taps over as a dense 4 x (4*NV) coefficient block per
// quad (zeros where there is no tap), so no address depends on a table value except through base[q].
// Summation runs over the span in ascending input position: the order of resample_kernel's sorted taps,
// with exact zero terms in between (the two agree up to the fusing of multiply-adds, ~1 ulp).
// The coefficient table is lane-interleaved (groups of 64 quads, [e][v][quad % 64][4] inside a group): the 64 lanes
// of a wave read 1 KB in one piece per load instruction.  With one block per lane in a row every instruction touched
// 64 lines, the 60-90 KB table fell out of the vector cache between a wave's 8..12 coefficient loads, and a thread had
// to keep its block for 32 rows to hide that (0.56-0.60 of the HBM peak at 512 samples whatever the rest of the kernel
// did; a table of three repeating blocks showed 0.72).  Now a thread takes 8 rows, all loads of which are issued
// before the first is used.  NTS: streaming stores (see st4s); the loads stay cached, neighbouring lanes share
// their vectors.
constexpr int RQ_ROWS = 8;
constexpr int RQ_LANES = 64;  // quads per interleaved group of the coefficient table

template <int NV, typename TI = float, typename TO = float, bool NTS = false>
__global__ __launch_bounds__(256) void resample_quad_kernel(const TI* __restrict__ x, const int32_t* __restrict__ base,
                                                            const float* __restrict__ coef, TO* __restrict__ out,
                                                            long long rows, int Tin, int Tout, int fa, int fc, int fd) {
    // threads are dealt out over (row group, quad) without regard to workgroup borders: Tout/4 is rarely a
    // multiple of 256 (320 at Tout = 1280), and a grid with one axis per row would idle the remainder
    const int NQ = Tout / 4;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rg = id / NQ;
    const int q = (int)(id - rg * NQ);
    const long long r0 = rg * RQ_ROWS;
    if (r0 >= rows) return;
    // base == nullptr: the quad's start in closed form (an affine function of q, clamped to the row), so that the
    // loads below wait for no look-up
    int b0;
    if (base) {
        b0 = base[q];
    } else {
        const int n = fa * q + fc;
        b0 = (int)((unsigned)(n > 0 ? n : 0) / (unsigned)fd) & ~3;
        b0 = b0 < Tin - 4 * NV ? b0 : Tin - 4 * NV;
    }
    f32x4 cf[4][NV];  // cf[e][v][i]: weight of input b0 + 4v + i in output 4q + e
    const float* cq = coef + ((long long)(q / RQ_LANES) * (4 * NV) * RQ_LANES + q % RQ_LANES) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int v = 0; v < NV; ++v) cf[e][v] = *reinterpret_cast<const f32x4*>(cq + (e * NV + v) * RQ_LANES * 4);
    f32x4 xv[RQ_ROWS][NV];
#pragma unroll
    for (int r = 0; r < RQ_ROWS; ++r) {
        const long long row = r0 + r < rows ? r0 + r : rows - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[r][v] = ld4(x + row * Tin + b0 + 4 * v);
    }
#pragma unroll
    for (int r = 0; r < RQ_ROWS; ++r) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) s += cf[e][v][i] * xv[r][v][i];
            o[e] = s;
        }
        if (r0 + r < rows) st4s<NTS>(out + (r0 + r) * Tout + 4 * q, o);
    }
}

// ---- minibatch stddev --------------------------------------------------------------------------
// Group member m owns the G samples b = g*M + m.  The statistics are taken over one temporal WINDOW of the
// signal, time steps [t0, t0 + Tw) of T (reference layers.py:81-89 runs the layer once per window of
// ``stride_size`` steps; the default is one window, Tw = T): n = C*Tw elements per sample, element e being
// channel e / Tw at step t0 + e % Tw.  The element axis is cut into chunks of MB_CHUNK so that the grid
// (M x chunks) fills the chip (M is only B/4); sums over all elements go through a per-(m, chunk) partial
// buffer and a tiny finalize kernel, in a fixed order.  A launch touches only its window's columns of the
// outputs, so a tensor with several windows takes one launch (pair) per window.
constexpr int MB_MAXG = 8;
constexpr int MB_CHUNK = 256;

struct MbWin {
    int C, T, t0, Tw;
    __device__ __forceinline__ int n() const { return C * Tw; }                 // elements of the window per sample
    __device__ __forceinline__ long long xs() const { return (long long)C * T; }  // sample stride of x / gx
    __device__ __forceinline__ long long os() const { return (long long)(C + 1) * T; }  // ... of out / gout
    __device__ __forceinline__ int off(int e) const {
        const int c = e / Tw;
        return c * T + t0 + (e - c * Tw);
    }
    __device__ __forceinline__ long long extra(int t) const { return (long long)C * T + t0 + t; }  // appended channel
};

__device__ __forceinline__ float group_scale(const float* __restrict__ gout, int m, int M, int G, const MbWin& w,
                                             float* red) {
    // a[m] = (sum over the group's extra-channel gradients inside the window) / (n * G)
    float gs = 0.f;
    for (int i = threadIdx.x; i < G * w.Tw; i += 256) {
        const int g = i / w.Tw, t = i - g * w.Tw;
        gs += gout[(long long)(g * M + m) * w.os() + w.extra(t)];
    }
    return block_sum_256(gs, red) / ((float)w.n() * (float)G);
}

__global__ __launch_bounds__(256) void mbstd_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        float* __restrict__ ws, int M, int G, MbWin w) {
    __shared__ float red[4];
    const int m = blockIdx.x, chunk = blockIdx.y;
    const float invG = 1.f / (float)G;
    const int e = chunk * MB_CHUNK + threadIdx.x;
    float sd = 0.f;
    if (e < w.n()) {
        const int o = w.off(e);
        float v[MB_MAXG];
        float mu = 0.f;
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) {
                v[g] = x[(long long)(g * M + m) * w.xs() + o];
                mu += v[g];
            }
        mu *= invG;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) {
                const float d = v[g] - mu;
                var += d * d;
                out[(long long)(g * M + m) * w.os() + o] = v[g];  // channels [0, C) are a copy of x
            }
        sd = sqrtf(var * invG);
    }
    const float tot = block_sum_256(sd, red);
    if (threadIdx.x == 0) ws[m * gridDim.y + chunk] = tot;
}

// out[b, C, t0 + t] = scale * sum_chunk ws[m][chunk] for the G samples of member m
__global__ __launch_bounds__(256) void mbstd_finalize_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                             int M, int G, MbWin w, int nchunks, float scale) {
    __shared__ float red[4];
    const int m = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 256) s += ws[m * nchunks + i];
    const float tot = block_sum_256(s, red) * scale;
    for (int i = threadIdx.x; i < G * w.Tw; i += 256) {
        const int g = i / w.Tw, t = i - g * w.Tw;
        out[(long long)(g * M + m) * w.os() + w.extra(t)] = tot;
    }
}

__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                        float* __restrict__ gx, int M, int G, MbWin w) {
    __shared__ float red[4];
    const int m = blockIdx.x, chunk = blockIdx.y;
    const float invG = 1.f / (float)G;
    const float a = group_scale(gout, m, M, G, w, red);
    const int e = chunk * MB_CHUNK + threadIdx.x;
    if (e >= w.n()) return;
    const int o = w.off(e);
    float v[MB_MAXG];
    float mu = 0.f;
#pragma unroll
    for (int g = 0; g < MB_MAXG; ++g)
        if (g < G) {
            v[g] = x[(long long)(g * M + m) * w.xs() + o];
            mu += v[g];
        }
    mu *= invG;
    float var = 0.f;
#pragma unroll
    for (int g = 0; g < MB_MAXG; ++g)
        if (g < G) {
            const float d = v[g] - mu;
            var += d * d;
        }
    const float inv_sigma = 1.f / sqrtf(var * invG);
#pragma unroll
    for (int g = 0; g < MB_MAXG; ++g)
        if (g < G) {
            const long long b = g * M + m;
            gx[b * w.xs() + o] = gout[b * w.os() + o] + a * (v[g] - mu) * inv_sigma;
        }
}

__global__ __launch_bounds__(256) void mbstd_bwd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                            const float* __restrict__ q, float* __restrict__ ggout,
                                                            float* __restrict__ gx2, float* __restrict__ ws, int M,
                                                            int G, MbWin w) {
    __shared__ float red[4];
    const int m = blockIdx.x, chunk = blockIdx.y;
    const float invG = 1.f / (float)G;
    const float a = group_scale(gout, m, M, G, w, red);
    const int e = chunk * MB_CHUNK + threadIdx.x;
    float qr = 0.f;  // sum_g q_g * R_g of this element
    if (e < w.n()) {
        const int o = w.off(e);
        float v[MB_MAXG], qv[MB_MAXG];
        float mu = 0.f, qm = 0.f;
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) {
                const long long b = g * M + m;
                v[g] = x[b * w.xs() + o];
                qv[g] = q[b * w.xs() + o];
                mu += v[g];
                qm += qv[g];
            }
        mu *= invG;
        qm *= invG;
        float var = 0.f;
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) {
                const float d = v[g] - mu;
                var += d * d;
            }
        const float inv_sigma = 1.f / sqrtf(var * invG);
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) qr += qv[g] * (v[g] - mu) * inv_sigma;
        const float qrm = qr * invG;
#pragma unroll
        for (int g = 0; g < MB_MAXG; ++g)
            if (g < G) {
                const long long b = g * M + m;
                const float R = (v[g] - mu) * inv_sigma;
                gx2[b * w.xs() + o] = a * inv_sigma * ((qv[g] - qm) - R * qrm);
                ggout[b * w.os() + o] = qv[g];
            }
    }
    const float tot = block_sum_256(qr, red);
    if (threadIdx.x == 0) ws[m * gridDim.y + chunk] = tot;
}

int mb_check(const char* what, int B, int C, int T, int group, int* G, int* M) {
    EEGGAN_REQUIRE(B > 0 && C > 0 && T > 0 && group > 0, "%s: non-positive argument", what);
    *G = B < group ? B : group;
    EEGGAN_REQUIRE(*G <= MB_MAXG, "%s: group size %d above the supported %d", what, *G, MB_MAXG);
    EEGGAN_REQUIRE(B % *G == 0, "%s: batch %d not divisible by group size %d", what, B, *G);
    EEGGAN_REQUIRE((long long)C * T < (1ll << 24), "%s: C*T too large", what);
    *M = B / *G;
    return 0;
}

int win_check(const char* what, int T, int t0, int Tw) {
    EEGGAN_REQUIRE(t0 >= 0 && Tw > 0 && t0 + Tw <= T, "%s: window [%d, %d) outside the signal's %d steps", what, t0,
                   t0 + Tw, T);
    return 0;
}

// ---- built-in scratch for the entry point that allocates nothing ------------------------------------
// Partial sums of eeggan_mbstd_fwd_static: one float per (group member, element chunk).  The array lives
// in the code object, so every GPU the library runs on holds its own copy at its own address until the
// process ends.  Host code never uses the array's name as an address: the name is a host-side stand-in,
// and the per-device address comes from hipGetSymbolAddress, once per device.
constexpr size_t MB_STATIC_FLOATS = 65536;  // 256 KiB per GPU
static __device__ __align__(256) float scratch[MB_STATIC_FLOATS];

constexpr int MB_MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[MB_MAX_DEVICES] = {};

// Device address of `scratch` on the current device (resolved at the first call on that device).
int scratch_for_current_device(const char* what, float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    EEGGAN_REQUIRE(e == hipSuccess, "%s: hipGetDevice failed: %s", what, hipGetErrorString(e));
    EEGGAN_REQUIRE(dev >= 0 && dev < MB_MAX_DEVICES, "%s: device index %d outside [0, %d)", what, dev, MB_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        float* ptr = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
        EEGGAN_REQUIRE(e == hipSuccess && ptr != nullptr, "%s: cannot resolve the built-in scratch buffer on device %d: %s",
                       what, dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_resample(const float* x, const int32_t* idx, const float* coef, float* out, size_t rows, int Tin,
                               int Tout, int taps, void* stream) {
    return eeggan_resample_t(x, idx, coef, out, rows, Tin, Tout, taps, 0, 0, stream);
}

template <typename TI, typename TO>
static int resample_impl(const TI* x, const int32_t* idx, const float* coef, TO* out, size_t rows, int Tin, int Tout,
                         int taps, void* stream) {
    EEGGAN_REQUIRE(x && idx && coef && out, "resample: null pointer");
    EEGGAN_REQUIRE(rows > 0 && Tin > 0 && Tout > 0, "resample: non-positive dimension");
    EEGGAN_REQUIRE(taps >= 1 && taps <= RS_WIDE_MAXTAPS, "resample: taps %d outside [1, %d]", taps, RS_WIDE_MAXTAPS);
    hipStream_t st = as_stream(stream);
    if (taps > RS_MAXTAPS) {
        EEGGAN_REQUIRE(rows <= 65535u * 1024u, "resample: too many rows");
        hipLaunchKernelGGL((resample_wide_kernel<TI, TO>), dim3((Tout + 255) / 256, (unsigned)rows), dim3(256), 0, st, x, idx,
                           coef, out, (long long)rows, Tin, Tout, taps);
        return check_launch("resample");
    }
    const size_t gy = (rows + RS_ROWS - 1) / RS_ROWS;
    EEGGAN_REQUIRE(gy <= 65535u * 1024u, "resample: too many rows");
    dim3 grid((Tout + 255) / 256, (unsigned)gy);
#define RS_CASE(N)                                                                                              \
    case N:                                                                                                     \
        hipLaunchKernelGGL((resample_kernel<N, TI, TO>), grid, dim3(256), 0, st, x, idx, coef, out, (long long)rows, Tin, \
                           Tout);                                                                               \
        break;
    switch (taps) {
        RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
        RS_CASE(9) RS_CASE(10) RS_CASE(11) RS_CASE(12) RS_CASE(13) RS_CASE(14) RS_CASE(15) RS_CASE(16)
    }
#undef RS_CASE
    return check_launch("resample");
}

extern "C" int eeggan_resample_t(const void* x, const int32_t* idx, const float* coef, void* out, size_t rows, int Tin,
                                 int Tout, int taps, int in_bf16, int out_bf16, void* stream) {
    auto run = [&](auto* xp, auto* op) { return resample_impl(xp, idx, coef, op, rows, Tin, Tout, taps, stream); };
    if (in_bf16)
        return out_bf16 ? run(static_cast<const __bf16*>(x), static_cast<__bf16*>(out))
                        : run(static_cast<const __bf16*>(x), static_cast<float*>(out));
    return out_bf16 ? run(static_cast<const float*>(x), static_cast<__bf16*>(out))
                    : run(static_cast<const float*>(x), static_cast<float*>(out));
}

extern "C" int eeggan_resample_quads(const float* x, const int32_t* base, const float* coef, float* out, size_t rows,
                                     int Tin, int Tout, int nv, void* stream) {
    return eeggan_resample_quads_t(x, base, coef, out, rows, Tin, Tout, nv, 0, 0, stream);
}

// base == nullptr: the closed form min(max(fa q + fc, 0) / fd / 4 * 4, Tin - 4 nv)
template <typename TI, typename TO>
static int resample_quads_impl(const TI* x, const int32_t* base, const float* coef, TO* out, size_t rows, int Tin,
                               int Tout, int nv, void* stream, int fa = 0, int fc = 0, int fd = 0) {
    EEGGAN_REQUIRE(x && coef && out, "resample_quads: null pointer");
    EEGGAN_REQUIRE(base || (fd > 0 && fa >= 0 && (long long)fa * (Tout / 4) + (fc < 0 ? -(long long)fc : fc) < (1ll << 31)),
                   "resample_quads: no table and no usable closed form (A=%d C=%d D=%d)", fa, fc, fd);
    EEGGAN_REQUIRE(rows > 0 && Tin > 0 && Tout > 0, "resample_quads: non-positive dimension");
    EEGGAN_REQUIRE(nv >= 1 && nv <= 4, "resample_quads: nv %d outside [1, 4]", nv);
    EEGGAN_REQUIRE(Tin % 4 == 0 && Tout % 4 == 0 && Tin >= 4 * nv,
                   "resample_quads: lengths %d -> %d must be multiples of 4 and the input at least %d long", Tin, Tout,
                   4 * nv);
    EEGGAN_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                     reinterpret_cast<uintptr_t>(coef)) & 15) == 0,
                   "resample_quads: x, out and coef must be 16-byte aligned");
    const size_t threads = (rows + RQ_ROWS - 1) / RQ_ROWS * (size_t)(Tout / 4);
    EEGGAN_REQUIRE(threads < ((size_t)1 << 39), "resample_quads: too many rows");
    dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = as_stream(stream);
    const bool nts = streams_past_caches(rows * ((size_t)Tin * sizeof(TI) + (size_t)Tout * sizeof(TO)));
#define RQ(NV_, NTS_) \
    hipLaunchKernelGGL((resample_quad_kernel<NV_, TI, TO, NTS_>), grid, dim3(256), 0, st, x, base, coef, out, (long long)rows, Tin, Tout, fa, fc, fd)
#define RQN(NV_)                 \
    do {                         \
        if (nts) RQ(NV_, true);  \
        else RQ(NV_, false);     \
    } while (0)
    switch (nv) {
        case 1: RQN(1); break;
        case 2: RQN(2); break;
        case 3: RQN(3); break;
        default: RQN(4); break;
    }
#undef RQN
#undef RQ
    return check_launch("resample_quads");
}

extern "C" int eeggan_resample_quads_t(const void* x, const int32_t* base, const float* coef, void* out, size_t rows,
                                       int Tin, int Tout, int nv, int in_bf16, int out_bf16, void* stream) {
    auto run = [&](auto* xp, auto* op) { return resample_quads_impl(xp, base, coef, op, rows, Tin, Tout, nv, stream); };
    if (in_bf16)
        return out_bf16 ? run(static_cast<const __bf16*>(x), static_cast<__bf16*>(out))
                        : run(static_cast<const __bf16*>(x), static_cast<float*>(out));
    return out_bf16 ? run(static_cast<const float*>(x), static_cast<__bf16*>(out))
                    : run(static_cast<const float*>(x), static_cast<float*>(out));
}

extern "C" int eeggan_resample_quads_affine_t(const void* x, int A, int C, int D, const float* coef, void* out, size_t rows,
                                              int Tin, int Tout, int nv, int in_bf16, int out_bf16, void* stream) {
    EEGGAN_REQUIRE(D > 0 && A >= 0, "resample_quads_affine: closed form with A=%d D=%d", A, D);
    auto run = [&](auto* xp, auto* op) {
        return resample_quads_impl(xp, (const int32_t*)nullptr, coef, op, rows, Tin, Tout, nv, stream, A, C, D);
    };
    if (in_bf16)
        return out_bf16 ? run(static_cast<const __bf16*>(x), static_cast<__bf16*>(out))
                        : run(static_cast<const __bf16*>(x), static_cast<float*>(out));
    return out_bf16 ? run(static_cast<const float*>(x), static_cast<__bf16*>(out))
                    : run(static_cast<const float*>(x), static_cast<float*>(out));
}

extern "C" size_t eeggan_mbstd_workspace(int B, int C, int T, int group) {
    if (B <= 0 || C <= 0 || T <= 0 || group <= 0) return 0;
    const int G = B < group ? B : group;
    const size_t M = (size_t)(B / G), nch = ((size_t)C * T + MB_CHUNK - 1) / MB_CHUNK;
    return M * nch * sizeof(float);
}

extern "C" int eeggan_mbstd_fwd(const float* x, float* out, void* workspace, int B, int C, int T, int group,
                                void* stream) {
    return eeggan_mbstd_fwd_window(x, out, workspace, B, C, T, 0, T, group, stream);
}

extern "C" int eeggan_mbstd_fwd_window(const float* x, float* out, void* workspace, int B, int C, int T, int t0, int Tw,
                                       int group, void* stream) {
    int G, M;
    if (mb_check("mbstd_fwd", B, C, T, group, &G, &M) || win_check("mbstd_fwd", T, t0, Tw)) return 1;
    EEGGAN_REQUIRE(x && out && workspace, "mbstd_fwd: null pointer");
    const MbWin w{C, T, t0, Tw};
    const int n = C * Tw, nch = (n + MB_CHUNK - 1) / MB_CHUNK;
    float* ws = static_cast<float*>(workspace);
    hipLaunchKernelGGL(mbstd_fwd_kernel, dim3(M, nch), dim3(256), 0, as_stream(stream), x, out, ws, M, G, w);
    if (check_launch("mbstd_fwd")) return 1;
    hipLaunchKernelGGL(mbstd_finalize_kernel, dim3(M), dim3(256), 0, as_stream(stream), ws, out, M, G, w, nch,
                       1.f / (float)n);
    return check_launch("mbstd_fwd finalize");
}

extern "C" int eeggan_mbstd_fwd_static(const float* x, float* out, int B, int C, int T, int group, void* stream) {
    int G, M;
    if (mb_check("mbstd_fwd_static", B, C, T, group, &G, &M)) return 1;
    EEGGAN_REQUIRE(x && out, "mbstd_fwd_static: null pointer");
    const int n = C * T, nch = (n + MB_CHUNK - 1) / MB_CHUNK;
    const size_t need = (size_t)M * (size_t)nch;
    EEGGAN_REQUIRE(need <= MB_STATIC_FLOATS,
                   "mbstd_fwd_static: B=%d, C=%d, T=%d needs %zu floats of scratch, the built-in buffer holds %zu "
                   "(use eeggan_mbstd_fwd with a workspace)", B, C, T, need, MB_STATIC_FLOATS);
    float* ws = nullptr;
    if (scratch_for_current_device("mbstd_fwd_static", &ws)) return 1;
    // every partial sum the finalize kernel reads is written by the first kernel of this call: nothing of
    // an earlier call's contents is read, so nothing needs clearing
    const MbWin w{C, T, 0, T};
    hipLaunchKernelGGL(mbstd_fwd_kernel, dim3(M, nch), dim3(256), 0, as_stream(stream), x, out, ws, M, G, w);
    if (check_launch("mbstd_fwd_static")) return 1;
    hipLaunchKernelGGL(mbstd_finalize_kernel, dim3(M), dim3(256), 0, as_stream(stream), ws, out, M, G, w, nch,
                       1.f / (float)n);
    return check_launch("mbstd_fwd_static finalize");
}

extern "C" int eeggan_mbstd_bwd(const float* x, const float* gout, float* gx, int B, int C, int T, int group,
                                void* stream) {
    return eeggan_mbstd_bwd_window(x, gout, gx, B, C, T, 0, T, group, stream);
}

extern "C" int eeggan_mbstd_bwd_window(const float* x, const float* gout, float* gx, int B, int C, int T, int t0, int Tw,
                                       int group, void* stream) {
    int G, M;
    if (mb_check("mbstd_bwd", B, C, T, group, &G, &M) || win_check("mbstd_bwd", T, t0, Tw)) return 1;
    EEGGAN_REQUIRE(x && gout && gx, "mbstd_bwd: null pointer");
    const MbWin w{C, T, t0, Tw};
    const int nch = (C * Tw + MB_CHUNK - 1) / MB_CHUNK;
    hipLaunchKernelGGL(mbstd_bwd_kernel, dim3(M, nch), dim3(256), 0, as_stream(stream), x, gout, gx, M, G, w);
    return check_launch("mbstd_bwd");
}

extern "C" int eeggan_mbstd_bwd_bwd(const float* x, const float* gout, const float* q, float* ggout, float* gx2,
                                    void* workspace, int B, int C, int T, int group, void* stream) {
    return eeggan_mbstd_bwd_bwd_window(x, gout, q, ggout, gx2, workspace, B, C, T, 0, T, group, stream);
}

extern "C" int eeggan_mbstd_bwd_bwd_window(const float* x, const float* gout, const float* q, float* ggout, float* gx2,
                                           void* workspace, int B, int C, int T, int t0, int Tw, int group,
                                           void* stream) {
    int G, M;
    if (mb_check("mbstd_bwd_bwd", B, C, T, group, &G, &M) || win_check("mbstd_bwd_bwd", T, t0, Tw)) return 1;
    EEGGAN_REQUIRE(x && gout && q && ggout && gx2 && workspace, "mbstd_bwd_bwd: null pointer");
    const MbWin w{C, T, t0, Tw};
    const int n = C * Tw, nch = (n + MB_CHUNK - 1) / MB_CHUNK;
    float* ws = static_cast<float*>(workspace);
    hipLaunchKernelGGL(mbstd_bwd_bwd_kernel, dim3(M, nch), dim3(256), 0, as_stream(stream), x, gout, q, ggout, gx2, ws,
                       M, G, w);
    if (check_launch("mbstd_bwd_bwd")) return 1;
    hipLaunchKernelGGL(mbstd_finalize_kernel, dim3(M), dim3(256), 0, as_stream(stream), ws, ggout, M, G, w, nch,
                       1.f / ((float)n * (float)G));
    return check_launch("mbstd_bwd_bwd finalize");
}
