// Bandwidth-bound elementwise / per-column normalisation kernels (fp32, [B, C, T]).
// All of them read each logical input once and write each output once; rows are walked
// with 16-byte accesses whenever T (or n) is a multiple of 4.
#include "common.h"

namespace eeggan {
namespace {

constexpr int MAX_BLOCKS = 256 * 8;

inline unsigned blocks_for(size_t items) {
    size_t b = (items + 255) / 256;
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// Grid of the 16-byte streaming kernels: a capped grid whose threads stride over the tensor is quickest
// while the tensor lives in the caches (16 MB: 7.6 vs 8.2 us); from ~128 MB on one item per thread is, the
// workgroups then sweep memory in address order (5.8 vs 5.1 TB/s).
inline unsigned blocks_stream(size_t items) {
    const size_t b = (items + 255) / 256;
    if (b >= 32768 && b < ((size_t)1 << 31)) return (unsigned)b;
    return blocks_for(items);
}

// ---- leaky-relu + drop backward ------------------------------------------------------------
// (ET: float, or __bf16 in the bf16 storage mode: 8-byte items then; arithmetic in float either way)
// (NT: the streaming form of the 16-byte path, see ld4s)
template <bool VEC, typename ET = float, bool NT = false>
__global__ __launch_bounds__(256) void lrelu_drop_bwd_kernel(const ET* __restrict__ g, const ET* __restrict__ y,
                                                             const float* __restrict__ drop, ET* __restrict__ out,
                                                             size_t n, int T, float slope) {
    constexpr int W = VEC ? 4 : 1;
    const size_t items = n / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const size_t e = i * W;
        const float r = drop ? drop[e / T] : 1.f;
        if (VEC) {
            const f32x4 gv = ld4s<NT>(g + e);
            const f32x4 yv = ld4s<NT>(y + e);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = gv[k] * r * (yv[k] > 0.f ? 1.f : slope);
            st4s<NT>(out + e, o);
        } else {
            st1(out + e, ld1(g + e) * r * (ld1(y + e) > 0.f ? 1.f : slope));
        }
    }
}

// ---- pixel norm (+drop) ---------------------------------------------------------------------
// Workgroup = CG channel groups x (256 / CG) column units; a unit is W (4 or 1) time steps of one
// sample.  The host picks (W, CG) so that even the short early stages (T = 32..256) spread over the
// chip: fewer units per workgroup and more channel groups mean more workgroups and a shorter
// sequential channel loop per thread.
// CPT > 0: the channel count is CG * CPT and a thread keeps its CPT values in registers between the
// reduction and the scaling (one read of the input instead of two; all its loads are issued up front, and no
// load stands between its stores: in the CPT = 0 loops every channel costs a round trip to memory, 9-13 us for a
// 4 MB tensor of the 128- and 256-channel stages).
// NT (register-resident forms only: the others read their input twice, the second time from the caches): the
// streaming form, see ld4s.
// a unit of W time steps as a vector (components past W unused)
template <int W, bool NT, typename ET>
__device__ __forceinline__ f32x4 ldw(const ET* p) {
    if constexpr (W == 4) {
        return ld4s<NT>(p);
    } else {
        return f32x4{ld1(p), 0.f, 0.f, 0.f};
    }
}
template <int W, bool NT, typename ET>
__device__ __forceinline__ void stw(ET* p, f32x4 v) {
    if constexpr (W == 4) {
        st4s<NT>(p, v);
    } else {
        st1(p, v[0]);
    }
}

template <int W, int CG, int CPT, typename ET = float, bool NT = false>
__global__ __launch_bounds__(256) void pixelnorm_fwd_kernel(const ET* __restrict__ h, const float* __restrict__ drop,
                                                            ET* __restrict__ y, float* __restrict__ d, int B, int C,
                                                            int T, float eps) {
    constexpr int UL = 256 / CG;
    __shared__ float red[CG][UL][W];
    const int cg = threadIdx.x / UL, ul = threadIdx.x % UL;
    const int upr = T / W;  // units per row
    const long long unit = (long long)blockIdx.x * UL + ul;
    const bool ok = unit < (long long)B * upr;
    const int b = ok ? (int)(unit / upr) : 0;
    const int t = ok ? (int)(unit - (long long)b * upr) * W : 0;
    float s[W];
#pragma unroll
    for (int k = 0; k < W; ++k) s[k] = 0.f;
    f32x4 keep[CPT > 0 ? CPT : 1];
    // the drop multipliers of the thread's channels, fetched with the tile: a load between the stores would make every
    // store wait for the one before it (loads and stores retire through one counter)
    float rr[CPT > 0 ? CPT : 1];
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) rr[i] = 1.f;
        if (ok) {
            if (drop) {
#pragma unroll
                for (int i = 0; i < CPT; ++i) rr[i] = drop[b * C + cg + i * CG];
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i)
                keep[i] = ldw<W, NT>(h + ((long long)b * C + cg + i * CG) * T + t);
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int k = 0; k < W; ++k) s[k] += keep[i][k] * keep[i][k];
        }
    } else if (ok) {
        for (int c = cg; c < C; c += CG) {
            const long long off = ((long long)b * C + c) * T + t;
            if (W == 4) {
                const f32x4 v = ld4(h + off);
#pragma unroll
                for (int k = 0; k < W; ++k) s[k] += v[k] * v[k];
            } else {
                const float v = ld1(h + off);
                s[0] += v * v;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) red[cg][ul][k] = s[k];
    __syncthreads();
    if (!ok) return;
    float dv[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) tot += red[g][ul][k];
        dv[k] = rsqrtf(tot / (float)C + eps);
    }
    if (cg == 0) {
        if constexpr (W == 4) {
            *reinterpret_cast<f32x4*>(d + (long long)b * T + t) = f32x4{dv[0], dv[1], dv[2], dv[3]};
        } else {
            d[(long long)b * T + t] = dv[0];
        }
    }
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = cg + i * CG;
            const float r = rr[i];
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = keep[i][k] * dv[k] * r;
            stw<W, NT>(y + ((long long)b * C + c) * T + t, o);
        }
        return;
    }
    for (int c = cg; c < C; c += CG) {
        const long long off = ((long long)b * C + c) * T + t;
        const float r = drop ? drop[b * C + c] : 1.f;
        if (W == 4) {
            const f32x4 v = ld4(h + off);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < W; ++k) o[k] = v[k] * dv[k] * r;
            st4(y + off, o);
        } else {
            st1(y + off, ld1(h + off) * dv[0] * r);
        }
    }
}

template <int W, int CG, int CPT, typename ET = float, bool NT = false>
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const ET* __restrict__ gy, const ET* __restrict__ y,
                                                            const float* __restrict__ drop, const float* __restrict__ d,
                                                            ET* __restrict__ gc, int B, int C, int T, float slope) {
    constexpr int UL = 256 / CG;
    __shared__ float red[CG][UL][W];
    const int cg = threadIdx.x / UL, ul = threadIdx.x % UL;
    const int upr = T / W;
    const long long unit = (long long)blockIdx.x * UL + ul;
    const bool ok = unit < (long long)B * upr;
    const int b = ok ? (int)(unit / upr) : 0;
    const int t = ok ? (int)(unit - (long long)b * upr) * W : 0;
    float s[W];
#pragma unroll
    for (int k = 0; k < W; ++k) s[k] = 0.f;
    f32x4 ka[CPT > 0 ? CPT : 1], kv[CPT > 0 ? CPT : 1];
    float rr[CPT > 0 ? CPT : 1];  // (fetched with the tile, see the forward kernel)
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) rr[i] = 1.f;
        if (ok) {
            if (drop) {
#pragma unroll
                for (int i = 0; i < CPT; ++i) rr[i] = drop[b * C + cg + i * CG];
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const long long off = ((long long)b * C + cg + i * CG) * T + t;
                ka[i] = ldw<W, NT>(gy + off);
                kv[i] = ldw<W, NT>(y + off);
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i)
#pragma unroll
                for (int k = 0; k < W; ++k) s[k] += ka[i][k] * kv[i][k];
        }
    } else if (ok) {
        for (int c = cg; c < C; c += CG) {
            const long long off = ((long long)b * C + c) * T + t;
            if (W == 4) {
                const f32x4 a = ld4(gy + off);
                const f32x4 v = ld4(y + off);
#pragma unroll
                for (int k = 0; k < W; ++k) s[k] += a[k] * v[k];
            } else {
                s[0] += ld1(gy + off) * ld1(y + off);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < W; ++k) red[cg][ul][k] = s[k];
    __syncthreads();
    if (!ok) return;
    float S[W], dv[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) tot += red[g][ul][k];
        S[k] = tot / (float)C;
    }
    if constexpr (W == 4) {
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(d + (long long)b * T + t);
#pragma unroll
        for (int k = 0; k < W; ++k) dv[k] = d4[k];
    } else {
        dv[0] = d[(long long)b * T + t];
    }
    if (CPT > 0) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = cg + i * CG;
            const float r = rr[i];
            const float ir = 1.f / r;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < W; ++k)
                o[k] = dv[k] * (ka[i][k] * r - kv[i][k] * ir * S[k]) * (kv[i][k] > 0.f ? 1.f : slope);
            stw<W, NT>(gc + ((long long)b * C + c) * T + t, o);
        }
        return;
    }
    for (int c = cg; c < C; c += CG) {
        const long long off = ((long long)b * C + c) * T + t;
        const float r = drop ? drop[b * C + c] : 1.f;
        const float ir = 1.f / r;
        if (W == 4) {
            const f32x4 a = ld4(gy + off);
            const f32x4 v = ld4(y + off);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < W; ++k)
                o[k] = dv[k] * (a[k] * r - v[k] * ir * S[k]) * (v[k] > 0.f ? 1.f : slope);
            st4(gc + off, o);
        } else {
            const float a = ld1(gy + off), v = ld1(y + off);
            st1(gc + off, dv[0] * (a * r - v * ir * S[0]) * (v > 0.f ? 1.f : slope));
        }
    }
}

// (W, CG) choice: the widest accesses that still give ~one workgroup per CU.
struct PnCfg {
    int w, cg;
    unsigned grid;
};
inline PnCfg pn_config(long long BT, bool vec) {
    const auto blocks = [&](int w, int cg) { return (BT / w + 256 / cg - 1) / (256 / cg); };
    if (vec && blocks(4, 4) >= 256) return {4, 4, (unsigned)blocks(4, 4)};
    if (vec && blocks(4, 16) >= 256) return {4, 16, (unsigned)blocks(4, 16)};
    return {1, 16, (unsigned)blocks(1, 16)};
}

// The wide-tile choice keeps C / 4 channels per thread: where that count has no register-resident form (256 channels at
// a large batch: 64 per thread, 113 us for a 15 us transfer) but C / 16 has one, the 16-group shape is taken.
inline PnCfg pn_refine(PnCfg c, int C, long long BT, bool forward) {
    if (c.w != 4 || c.cg != 4) return c;
    const int cpt4 = C % 4 == 0 ? C / 4 : 0, cpt16 = C % 16 == 0 ? C / 16 : 0;
    const bool have4 = cpt4 == 16 || (forward && cpt4 == 32);
    const bool have16 = cpt16 == 4 || cpt16 == 8 || cpt16 == 16;
    if (!have4 && have16) return {4, 16, (unsigned)((BT / 4 + 15) / 16)};
    return c;
}

// ---- small flat kernels ------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                       float* __restrict__ gc, size_t n, float scale) {
    constexpr int W = VEC ? 4 : 1;
    const size_t items = n / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        if (VEC) {
            const f32x4 a = reinterpret_cast<const f32x4*>(gy)[i];
            const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = a[k] * scale * (1.f - v[k] * v[k]);
            reinterpret_cast<f32x4*>(gc)[i] = o;
        } else {
            gc[i] = gy[i] * scale * (1.f - y[i] * y[i]);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void scaled_tanh_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                          float scale) {
    constexpr int W = VEC ? 4 : 1;
    const size_t items = n / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        if (VEC) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = tanhf(v[k] * scale);
            reinterpret_cast<f32x4*>(y)[i] = o;
        } else {
            y[i] = tanhf(x[i] * scale);
        }
    }
}

// out = ca*a + cb*b.  `alpha` != nullptr: the coefficients come from the fade-in factor in device
// memory (mode 0: ca = alpha, cb = 1 - alpha; 1: ca = alpha; 2: ca = 1 - alpha) -- the same loop body as
// the by-value form, so both give bit-identical results for equal coefficients.
template <bool VEC, bool HASB, typename ET = float, bool NT = false>
__global__ __launch_bounds__(256) void axpby_kernel(const ET* __restrict__ a, float ca, const ET* __restrict__ b,
                                                    float cb, ET* __restrict__ out, size_t n,
                                                    const float* __restrict__ alpha, int mode) {
    constexpr int W = VEC ? 4 : 1;
    if (alpha) {
        const float al = alpha[0];
        ca = mode == 2 ? 1.f - al : al;
        cb = 1.f - al;
    }
    const size_t items = n / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        if (VEC) {
            const f32x4 x = ld4s<NT>(a + 4 * i);
            f32x4 o;
            if (HASB) {
                const f32x4 z = ld4s<NT>(b + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = x[k] * ca + z[k] * cb;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = x[k] * ca;
            }
            st4s<NT>(out + 4 * i, o);
        } else {
            st1(out + i, HASB ? ld1(a + i) * ca + ld1(b + i) * cb : ld1(a + i) * ca);
        }
    }
}

// (VEC: 16-byte items; a sample's length is then a multiple of 4, so an item never straddles two samples)
template <bool VEC>
__global__ __launch_bounds__(256) void mix_batch_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                        const float* __restrict__ mix, float* __restrict__ out,
                                                        size_t n, size_t per) {
    constexpr int W = VEC ? 4 : 1;
    const size_t items = n / W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256) {
        const float m = mix[i * W / per];
        if (VEC) {
            const f32x4 a = ld4(real + 4 * i), b = ld4(fake + 4 * i);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = m * a[k] + (1.f - m) * b[k];
            st4(out + 4 * i, o);
        } else {
            out[i] = m * real[i] + (1.f - m) * fake[i];
        }
    }
}

__global__ __launch_bounds__(256) void pow_base_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                       size_t n, float log_base) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = expf(x[i] * log_base);
}

// GDrop multipliers drawn on the device in one launch: out[i] = base ** z_i, z_i ~ N(0, 1) (layers.py:36, on the
// device instead of np.random).  Philox-4x32-10 (Salmon et al., SC'11) keyed by (seed, salt), counter
// (i / 4, epoch of the workgroup); four uniforms give four normals by Box-Muller.  ``state`` is LOGNORMAL_STATE_WORDS
// words of device memory owned by the caller, zero before the first launch: one epoch per workgroup, which the
// workgroup reads at its start and advances at its end.  So every launch, and every replay of a captured launch,
// draws a fresh stream with no host involvement, no second kernel, and no traffic between workgroups (a count of
// finished workgroups behind a device-scope fence made every workgroup write its XCD's L2 back; without the fence,
// 256 atomics on one word still took longer than the generator itself).
constexpr int LOGNORMAL_STATE_WORDS = 256;
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void lognormal_fill_kernel(float* __restrict__ out, size_t n, float log2_base,
                                                             unsigned key0, unsigned key1, unsigned* state) {
    const unsigned epoch = __hip_atomic_load(&state[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; 4 * i4 < n; i4 += (size_t)gridDim.x * 256) {
        unsigned c[4] = {(unsigned)i4, (unsigned)(i4 >> 32), epoch, 0u};
        unsigned k0 = key0, k1 = key1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // u1 in (0, 1], u2 in [0, 1): 24 random bits each.  The hardware's sine and cosine take their
            // argument in revolutions, and its base-2 logarithm is accurate to an ulp: no range reduction needed.
            const float u1 = ((float)(c[2 * h] >> 8) + 1.f) * (1.f / 16777216.f);
            const float u2 = (float)(c[2 * h + 1] >> 8) * (1.f / 16777216.f);
            const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u1
            z[2 * h] = r * __builtin_amdgcn_cosf(u2);
            z[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * i4 + e < n) out[4 * i4 + e] = exp2f(z[e] * log2_base);
    }
    // every thread has its epoch before it computes with it, hence before the barrier
    __syncthreads();
    if (threadIdx.x == 0) state[blockIdx.x] = epoch + 1u;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_lrelu_drop_bwd_t(const void* gv, const void* yv, const float* drop, void* outv, int B, int C,
                                       int T, float slope, int bf16, void* stream) {
    if (!bf16)
        return eeggan_lrelu_drop_bwd(static_cast<const float*>(gv), static_cast<const float*>(yv), drop,
                                     static_cast<float*>(outv), B, C, T, slope, stream);
    const __bf16* g = static_cast<const __bf16*>(gv);
    const __bf16* y = static_cast<const __bf16*>(yv);
    __bf16* out = static_cast<__bf16*>(outv);
    EEGGAN_REQUIRE(g && y && out, "lrelu_drop_bwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && T > 0, "lrelu_drop_bwd: non-positive dimension");
    EEGGAN_REQUIRE(T % 4 == 0 && aligned16(g) && aligned16(y) && aligned16(out),
                   "lrelu_drop_bwd: bfloat16 tensors need T %% 4 == 0 and 16-byte alignment");
    const size_t n = (size_t)B * C * T;
    if (streams_past_caches(3 * n * sizeof(__bf16)))
        hipLaunchKernelGGL((lrelu_drop_bwd_kernel<true, __bf16, true>), dim3(blocks_stream(n / 4)), dim3(256), 0,
                           as_stream(stream), g, y, drop, out, n, T, slope);
    else
        hipLaunchKernelGGL((lrelu_drop_bwd_kernel<true, __bf16>), dim3(blocks_stream(n / 4)), dim3(256), 0,
                           as_stream(stream), g, y, drop, out, n, T, slope);
    return check_launch("lrelu_drop_bwd");
}

extern "C" int eeggan_lrelu_drop_bwd(const float* g, const float* y, const float* drop, float* out, int B, int C,
                                     int T, float slope, void* stream) {
    EEGGAN_REQUIRE(g && y && out, "lrelu_drop_bwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && T > 0, "lrelu_drop_bwd: non-positive dimension");
    const size_t n = (size_t)B * C * T;
    const bool vec = (T % 4 == 0) && aligned16(g) && aligned16(y) && aligned16(out);
    if (vec && streams_past_caches(3 * n * sizeof(float)))
        hipLaunchKernelGGL((lrelu_drop_bwd_kernel<true, float, true>), dim3(blocks_stream(n / 4)), dim3(256), 0,
                           as_stream(stream), g, y, drop, out, n, T, slope);
    else if (vec)
        hipLaunchKernelGGL(lrelu_drop_bwd_kernel<true>, dim3(blocks_stream(n / 4)), dim3(256), 0, as_stream(stream), g, y,
                           drop, out, n, T, slope);
    else
        hipLaunchKernelGGL(lrelu_drop_bwd_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), g, y,
                           drop, out, n, T, slope);
    return check_launch("lrelu_drop_bwd");
}

template <typename ET>
static int pixelnorm_drop_fwd_impl(const ET* h, const float* drop, ET* y, float* d, int B, int C, int T, float eps,
                                   void* stream) {
    EEGGAN_REQUIRE(h && y && d, "pixelnorm_drop_fwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && T > 0, "pixelnorm_drop_fwd: non-positive dimension");
    const bool vec = (T % 4 == 0) && aligned16(h) && aligned16(y) && aligned16(d);
    PnCfg c = pn_refine(pn_config((long long)B * T, vec), C, (long long)B * T, true);
    if (sizeof(ET) == 2) {
        EEGGAN_REQUIRE(vec, "pixelnorm_drop_fwd: bfloat16 tensors need T %% 4 == 0 and 16-byte alignment");
        if (c.w != 4) c = {4, 16, (unsigned)(((long long)B * T / 4 + 15) / 16)};
    }
    hipStream_t st = as_stream(stream);
    const bool nt = streams_past_caches((size_t)2 * B * C * T * sizeof(ET));
#define PN_FWD_(W, CG, CPT, NT) \
    hipLaunchKernelGGL((pixelnorm_fwd_kernel<W, CG, CPT, ET, NT>), dim3(c.grid), dim3(256), 0, st, h, drop, y, d, B, C, T, eps)
#define PN_FWD(W, CG, CPT)                                         \
    do {                                                           \
        if (nt && CPT > 0) PN_FWD_(W, CG, CPT, (CPT > 0));         \
        else PN_FWD_(W, CG, CPT, false);                           \
    } while (0)
    if (c.w == 4 && c.cg == 4 && C == 64) {  // 8 groups x 8 channels: 0.69 against 0.65 of the HBM peak at 512 samples
        c.grid = (unsigned)(((long long)B * T / 4 + 31) / 32);
        PN_FWD(4, 8, 8);
        return check_launch("pixelnorm_drop_fwd");
    }
    const int cpt = C % c.cg == 0 ? C / c.cg : 0;  // channels per thread, register-resident when small
    if (c.w == 4 && c.cg == 4) {
        if (cpt == 16) PN_FWD(4, 4, 16);
        else if (cpt == 32) PN_FWD(4, 4, 32);
        else PN_FWD(4, 4, 0);
    } else if (c.w == 4) {
        if (cpt == 4) PN_FWD(4, 16, 4);
        else if (cpt == 8) PN_FWD(4, 16, 8);
        else if (cpt == 16) PN_FWD(4, 16, 16);
        else PN_FWD(4, 16, 0);
    } else if constexpr (sizeof(ET) == 4) {
        if (cpt == 4) PN_FWD(1, 16, 4);
        else if (cpt == 8) PN_FWD(1, 16, 8);
        else if (cpt == 16) PN_FWD(1, 16, 16);
        else PN_FWD(1, 16, 0);
    }
#undef PN_FWD
#undef PN_FWD_
    return check_launch("pixelnorm_drop_fwd");
}

extern "C" int eeggan_pixelnorm_drop_fwd(const float* h, const float* drop, float* y, float* d, int B, int C, int T,
                                         float eps, void* stream) {
    return pixelnorm_drop_fwd_impl<float>(h, drop, y, d, B, C, T, eps, stream);
}

extern "C" int eeggan_pixelnorm_drop_fwd_t(const void* h, const float* drop, void* y, float* d, int B, int C, int T,
                                           float eps, int bf16, void* stream) {
    if (bf16)
        return pixelnorm_drop_fwd_impl<__bf16>(static_cast<const __bf16*>(h), drop, static_cast<__bf16*>(y), d, B, C, T,
                                               eps, stream);
    return pixelnorm_drop_fwd_impl<float>(static_cast<const float*>(h), drop, static_cast<float*>(y), d, B, C, T, eps,
                                          stream);
}

template <typename ET>
static int pixelnorm_drop_bwd_impl(const ET* gy, const ET* y, const float* drop, const float* d, ET* gc, int B, int C,
                                   int T, float slope, void* stream) {
    EEGGAN_REQUIRE(gy && y && d && gc, "pixelnorm_drop_bwd: null pointer");
    EEGGAN_REQUIRE(B > 0 && C > 0 && T > 0, "pixelnorm_drop_bwd: non-positive dimension");
    const bool vec = (T % 4 == 0) && aligned16(gy) && aligned16(y) && aligned16(gc) && aligned16(d);
    PnCfg c = pn_refine(pn_config((long long)B * T, vec), C, (long long)B * T, false);
    if (sizeof(ET) == 2) {
        EEGGAN_REQUIRE(vec, "pixelnorm_drop_bwd: bfloat16 tensors need T %% 4 == 0 and 16-byte alignment");
        if (c.w != 4) c = {4, 16, (unsigned)(((long long)B * T / 4 + 15) / 16)};
    }
    hipStream_t st = as_stream(stream);
    const bool nt = streams_past_caches((size_t)3 * B * C * T * sizeof(ET));
#define PN_BWD_(W, CG, CPT, NT)                                                                                    \
    hipLaunchKernelGGL((pixelnorm_bwd_kernel<W, CG, CPT, ET, NT>), dim3(c.grid), dim3(256), 0, st, gy, y, drop, d, gc, B, C, \
                       T, slope)
#define PN_BWD(W, CG, CPT)                                         \
    do {                                                           \
        if (nt && CPT > 0) PN_BWD_(W, CG, CPT, (CPT > 0));         \
        else PN_BWD_(W, CG, CPT, false);                           \
    } while (0)
    if (c.w == 4 && c.cg == 4 && C == 64) {  // 8 groups x 8 channels (16 registers of operands less): 0.74 against 0.68
        c.grid = (unsigned)(((long long)B * T / 4 + 31) / 32);
        PN_BWD(4, 8, 8);
        return check_launch("pixelnorm_drop_bwd");
    }
    const int cpt = C % c.cg == 0 ? C / c.cg : 0;
    if (c.w == 4 && c.cg == 4) {
        if (cpt == 16) PN_BWD(4, 4, 16);
        else PN_BWD(4, 4, 0);
    } else if (c.w == 4) {
        if (cpt == 4) PN_BWD(4, 16, 4);
        else if (cpt == 8) PN_BWD(4, 16, 8);
        else if (cpt == 16) PN_BWD(4, 16, 16);
        else PN_BWD(4, 16, 0);
    } else if constexpr (sizeof(ET) == 4) {
        if (cpt == 4) PN_BWD(1, 16, 4);
        else if (cpt == 8) PN_BWD(1, 16, 8);
        else if (cpt == 16) PN_BWD(1, 16, 16);
        else PN_BWD(1, 16, 0);
    }
#undef PN_BWD
#undef PN_BWD_
    return check_launch("pixelnorm_drop_bwd");
}

extern "C" int eeggan_pixelnorm_drop_bwd(const float* gy, const float* y, const float* drop, const float* d, float* gc,
                                         int B, int C, int T, float slope, void* stream) {
    return pixelnorm_drop_bwd_impl<float>(gy, y, drop, d, gc, B, C, T, slope, stream);
}

extern "C" int eeggan_pixelnorm_drop_bwd_t(const void* gy, const void* y, const float* drop, const float* d, void* gc,
                                           int B, int C, int T, float slope, int bf16, void* stream) {
    if (bf16)
        return pixelnorm_drop_bwd_impl<__bf16>(static_cast<const __bf16*>(gy), static_cast<const __bf16*>(y), drop, d,
                                               static_cast<__bf16*>(gc), B, C, T, slope, stream);
    return pixelnorm_drop_bwd_impl<float>(static_cast<const float*>(gy), static_cast<const float*>(y), drop, d,
                                          static_cast<float*>(gc), B, C, T, slope, stream);
}

extern "C" int eeggan_tanh_bwd(const float* gy, const float* y, float* gc, size_t n, float scale, void* stream) {
    EEGGAN_REQUIRE(gy && y && gc && n > 0, "tanh_bwd: null pointer or empty");
    const bool vec = (n % 4 == 0) && aligned16(gy) && aligned16(y) && aligned16(gc);
    if (vec)
        hipLaunchKernelGGL(tanh_bwd_kernel<true>, dim3(blocks_stream(n / 4)), dim3(256), 0, as_stream(stream), gy, y, gc,
                           n, scale);
    else
        hipLaunchKernelGGL(tanh_bwd_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), gy, y, gc, n,
                           scale);
    return check_launch("tanh_bwd");
}

extern "C" int eeggan_scaled_tanh(const float* x, float* y, size_t n, float scale, void* stream) {
    EEGGAN_REQUIRE(x && y && n > 0, "scaled_tanh: null pointer or empty");
    const bool vec = (n % 4 == 0) && aligned16(x) && aligned16(y);
    if (vec)
        hipLaunchKernelGGL(scaled_tanh_kernel<true>, dim3(blocks_stream(n / 4)), dim3(256), 0, as_stream(stream), x, y, n,
                           scale);
    else
        hipLaunchKernelGGL(scaled_tanh_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), x, y, n,
                           scale);
    return check_launch("scaled_tanh");
}

// One launch of axpby_kernel: 16-byte items where `vec`, the streaming form when the tensors are beyond the caches.
template <typename ET>
static void launch_axpby(bool vec, const ET* a, float ca, const ET* b, float cb, ET* out, size_t n, const float* alpha,
                         int mode, hipStream_t st) {
#define AXPBY(VEC, HASB, NT, GRID) \
    hipLaunchKernelGGL((axpby_kernel<VEC, HASB, ET, NT>), dim3(GRID), dim3(256), 0, st, a, ca, b, cb, out, n, alpha, mode)
    if (!vec) {
        if constexpr (sizeof(ET) == 4) {
            if (b) AXPBY(false, true, false, blocks_for(n));
            else AXPBY(false, false, false, blocks_for(n));
        }
        return;
    }
    const bool nt = streams_past_caches((b ? 3 : 2) * n * sizeof(ET));
    if (b) {
        if (nt) AXPBY(true, true, true, blocks_stream(n / 4));
        else AXPBY(true, true, false, blocks_stream(n / 4));
    } else {
        if (nt) AXPBY(true, false, true, blocks_stream(n / 4));
        else AXPBY(true, false, false, blocks_stream(n / 4));
    }
#undef AXPBY
}

extern "C" int eeggan_axpby(const float* a, float ca, const float* b, float cb, float* out, size_t n, void* stream) {
    EEGGAN_REQUIRE(a && out && n > 0, "axpby: null pointer or empty");
    const bool vec = (n % 4 == 0) && aligned16(a) && aligned16(out) && (!b || aligned16(b));
    launch_axpby<float>(vec, a, ca, b, cb, out, n, nullptr, 0, as_stream(stream));
    return check_launch("axpby");
}

extern "C" int eeggan_blend(const float* a, const float* b, const float* alpha, int mode, float* out, size_t n,
                            void* stream) {
    EEGGAN_REQUIRE(a && alpha && out && n > 0, "blend: null pointer or empty");
    EEGGAN_REQUIRE(mode >= 0 && mode <= 2 && (mode != 0 || b), "blend: mode %d (0 needs b)", mode);
    if (mode != 0) b = nullptr;
    const bool vec = (n % 4 == 0) && aligned16(a) && aligned16(out) && (!b || aligned16(b));
    launch_axpby<float>(vec, a, 0.f, b, 0.f, out, n, alpha, mode, as_stream(stream));
    return check_launch("blend");
}

// bfloat16 forms of axpby / blend (bf16 storage mode: fade-in blends between layers); n % 4 == 0, 16-byte alignment
extern "C" int eeggan_axpby_t(const void* av, float ca, const void* bv, float cb, void* outv, size_t n, int bf16,
                              void* stream) {
    if (!bf16)
        return eeggan_axpby(static_cast<const float*>(av), ca, static_cast<const float*>(bv), cb, static_cast<float*>(outv),
                            n, stream);
    const __bf16* a = static_cast<const __bf16*>(av);
    const __bf16* b = static_cast<const __bf16*>(bv);
    __bf16* out = static_cast<__bf16*>(outv);
    EEGGAN_REQUIRE(a && out && n > 0, "axpby: null pointer or empty");
    EEGGAN_REQUIRE((n % 4 == 0) && aligned16(a) && aligned16(out) && (!b || aligned16(b)),
                   "axpby: bfloat16 tensors need n %% 4 == 0 and 16-byte alignment");
    launch_axpby<__bf16>(true, a, ca, b, cb, out, n, nullptr, 0, as_stream(stream));
    return check_launch("axpby");
}

extern "C" int eeggan_blend_t(const void* av, const void* bv, const float* alpha, int mode, void* outv, size_t n,
                              int bf16, void* stream) {
    if (!bf16)
        return eeggan_blend(static_cast<const float*>(av), static_cast<const float*>(bv), alpha, mode,
                            static_cast<float*>(outv), n, stream);
    const __bf16* a = static_cast<const __bf16*>(av);
    const __bf16* b = static_cast<const __bf16*>(bv);
    __bf16* out = static_cast<__bf16*>(outv);
    EEGGAN_REQUIRE(a && alpha && out && n > 0, "blend: null pointer or empty");
    EEGGAN_REQUIRE(mode >= 0 && mode <= 2 && (mode != 0 || b), "blend: mode %d (0 needs b)", mode);
    if (mode != 0) b = nullptr;
    EEGGAN_REQUIRE((n % 4 == 0) && aligned16(a) && aligned16(out) && (!b || aligned16(b)),
                   "blend: bfloat16 tensors need n %% 4 == 0 and 16-byte alignment");
    launch_axpby<__bf16>(true, a, 0.f, b, 0.f, out, n, alpha, mode, as_stream(stream));
    return check_launch("blend");
}

// out = in converted between float32 and bfloat16 (round to nearest even); same type on both sides: a copy
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n) {
    const size_t items = n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < items; i += (size_t)gridDim.x * 256)
        st4(out + 4 * i, ld4(in + 4 * i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) st1(out + 4 * items + threadIdx.x, ld1(in + 4 * items + threadIdx.x));
}

extern "C" int eeggan_cast(const void* in, void* out, size_t n, int in_bf16, int out_bf16, void* stream) {
    EEGGAN_REQUIRE(in && out && n > 0, "cast: null pointer or empty");
    EEGGAN_REQUIRE(aligned16(in) && aligned16(out), "cast: tensors must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const dim3 grid(blocks_stream(n / 4 + 1)), blk(256);
    if (in_bf16 && out_bf16)
        hipLaunchKernelGGL((cast_kernel<__bf16, __bf16>), grid, blk, 0, st, static_cast<const __bf16*>(in), static_cast<__bf16*>(out), n);
    else if (in_bf16)
        hipLaunchKernelGGL((cast_kernel<__bf16, float>), grid, blk, 0, st, static_cast<const __bf16*>(in), static_cast<float*>(out), n);
    else if (out_bf16)
        hipLaunchKernelGGL((cast_kernel<float, __bf16>), grid, blk, 0, st, static_cast<const float*>(in), static_cast<__bf16*>(out), n);
    else
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, st, static_cast<const float*>(in), static_cast<float*>(out), n);
    return check_launch("cast");
}

// dst[b, cd, t] = (base ? base[b, cd, t] : 0) + (cd in [d0, d0 + n) ? src[b, s0 + cd - d0, Ts == 1 ? 0 : t] : 0) over the
// WHOLE of dst [B, Cd, T]: channel slices, zero-padded placements and their sums (the skips of the `deep` blocks)
__global__ __launch_bounds__(256) void channel_window_kernel(const float* __restrict__ src, const float* base, float* dst,
                                                             int Cs, int Cd, int T, int Ts, int s0, int d0, int n,
                                                             long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int t = (int)(i % T);
        const long long r = i / T;
        const int cd = (int)(r % Cd);
        const long long b = r / Cd;
        float v = base ? base[i] : 0.f;
        const int c = cd - d0;
        if (c >= 0 && c < n) v += src[(b * Cs + s0 + c) * Ts + (Ts == 1 ? 0 : t)];
        dst[i] = v;
    }
}

extern "C" int eeggan_channel_window(const float* src, const float* base, float* dst, int B, int Cs, int Cd, int T, int Ts,
                                     int s0, int d0, int n, void* stream) {
    EEGGAN_REQUIRE(src && dst, "channel_window: null pointer");
    EEGGAN_REQUIRE(B > 0 && Cs > 0 && Cd > 0 && T > 0 && (Ts == T || Ts == 1), "channel_window: bad shape");
    EEGGAN_REQUIRE(n > 0 && s0 >= 0 && d0 >= 0 && s0 + n <= Cs && d0 + n <= Cd, "channel_window: window [%d, %d) of %d "
                   "source channels to [%d, %d) of %d", s0, s0 + n, Cs, d0, d0 + n, Cd);
    const long long total = (long long)B * Cd * T;
    hipLaunchKernelGGL(channel_window_kernel, dim3(blocks_for((size_t)total)), dim3(256), 0, as_stream(stream), src, base, dst,
                       Cs, Cd, T, Ts, s0, d0, n, total);
    return check_launch("channel_window");
}

extern "C" int eeggan_mix_batch(const float* real, const float* fake, const float* mix, float* out, int B, size_t per,
                                void* stream) {
    EEGGAN_REQUIRE(real && fake && mix && out && B > 0 && per > 0, "mix_batch: null pointer or empty");
    const size_t n = (size_t)B * per;
    if (per % 4 == 0 && aligned16(real) && aligned16(fake) && aligned16(out))
        hipLaunchKernelGGL(mix_batch_kernel<true>, dim3(blocks_stream(n / 4)), dim3(256), 0, as_stream(stream), real, fake,
                           mix, out, n, per);
    else
        hipLaunchKernelGGL(mix_batch_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, as_stream(stream), real, fake, mix,
                           out, n, per);
    return check_launch("mix_batch");
}

extern "C" int eeggan_lognormal_fill(float* out, size_t count, float base, unsigned long long seed, unsigned int salt,
                                     unsigned int* state, void* stream) {
    EEGGAN_REQUIRE(out && state && count > 0 && base > 0.f, "lognormal_fill: bad argument");
    long long blocks = ((long long)((count + 3) / 4) + 255) / 256;
    if (blocks > LOGNORMAL_STATE_WORDS) blocks = LOGNORMAL_STATE_WORDS;
    hipLaunchKernelGGL(lognormal_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, count, log2f(base),
                       (unsigned)seed ^ (salt * 0x9E3779B9u), (unsigned)(seed >> 32) ^ 0x85EBCA6Bu, state);
    return check_launch("lognormal_fill");
}

extern "C" int eeggan_pow_base(const float* x, float* out, size_t count, float base, void* stream) {
    EEGGAN_REQUIRE(x && out && count > 0 && base > 0.f, "pow_base: bad argument");
    hipLaunchKernelGGL(pow_base_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), x, out, count,
                       logf(base));
    return check_launch("pow_base");
}
