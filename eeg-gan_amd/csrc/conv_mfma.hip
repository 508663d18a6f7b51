// Equalized-lr Conv1d (stride 1, same padding, K in {1,3}) as fp32 MFMA implicit GEMM
// for gfx950: forward / input-gradient (one kernel, two weight views) and
// weight-gradient (split over the batch*time axis, deterministic two-stage sum).
//
// Data layout: activations [B, C, T] (T fastest).  The flattened column space of a
// layer is (sample, 32-wide time subtile); a workgroup owns NT consecutive subtiles.
// In LDS every subtile carries its own left/right halo sample, so subtiles are
// independent of each other and of sample boundaries (zero padding is a zero halo):
//     row = [3 unused][halo t0-1][32 samples t0..t0+31][halo t0+32][3 unused]  = 40 floats
// which keeps the 32 main samples 16-byte aligned for vector LDS stores.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32, exact fp32 FMA chain):
//   forward/dgrad:  D[t][co]  += A[t][k] * B[k][co],  k = (ci, tap)
//   wgrad:          D[co][ci] += A[co][k] * B[k][ci], k = (sample, t), one D per tap
// With time on the MFMA row axis a lane owns one output channel and its 16 accumulator
// registers are four runs of 4 consecutive time steps: 16-byte stores (4-byte-per-lane stores
// are issue-bound on this chip and cost more than the MFMAs of a 64-channel tile).
#include "common.h"
#include "conv_args.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace eeggan {
namespace {

constexpr int SUB = 40;  // LDS floats per 32-sample subtile (see header comment)
constexpr int WGRAD_W4_PER_CU = 2;  // workgroups per CU of the weight gradient's four-wave form (43 KB of LDS each)
constexpr int WGRAD_LDH = 88;  // halfwords per LDS row of the weight gradient's bfloat16-native tiles (see wgrad_body)
// LDS of that form: two buffers of two tiles, or the eight waves' closing exchange of sums, whichever is larger
constexpr size_t WGRAD_BFN_LDS = (size_t)4 * (3 * 16 + 1) * 64 * sizeof(float) > (size_t)2 * 2 * 64 * WGRAD_LDH * 2
                                     ? (size_t)4 * (3 * 16 + 1) * 64 * sizeof(float)
                                     : (size_t)2 * 2 * 64 * WGRAD_LDH * 2;

// Split products: an fp32 operand as the sum of three bf16 pieces (v = p1 + p2 + p3 up to 2^-24 |v|), see conv_ws.hip
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
// Two values at once (see conv_ws.hip: the packed conversion's result is the operand word)
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16s(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3s_pair(float a, float b, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = pack_bf16s(a, b);
    const float ra = a - __builtin_bit_cast(float, w1 << 16), rb = b - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = pack_bf16s(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w2 << 16), sb = rb - __builtin_bit_cast(float, w2 & 0xffff0000u);
    w3 = pack_bf16s(sa, sb);
}
__device__ __forceinline__ void split3s(float v, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    p3 = (__bf16)r2;
}

// ---------------------------------------------------------------------------------------
// Global -> register -> LDS staging of ROWS channels x NT subtiles of an activation.
template <int ROWS, int NT, int LD, bool HALO>
struct XStage {
    static constexpr int NV = ROWS * NT * 8;  // float4 items
    static constexpr int VPT = (NV + 255) / 256;
    static constexpr int NH = ROWS * NT * 2;  // halo items
    static constexpr int HPT = (NH + 255) / 256;
    static_assert(32 % NT == 0, "NT must divide 32");
    f32x4 v[VPT];
    float hv[HPT];

    // (TS: float or __bf16, the storage type of src and mask_y)
    template <typename TS>
    __device__ __forceinline__ void load(const TS* __restrict__ src, const TS* __restrict__ mask_y,
                                         const float* __restrict__ mask_drop, float slope, int C, int T,
                                         int ST, int NS, int sub0, int c0, bool vec) {
        const int tid = threadIdx.x;
        {
            const int q = tid & 7;
            const int n = (tid >> 3) % NT;
            const int sub = sub0 + n;
            const int b = sub / ST;
            const int t = (sub - b * ST) * 32 + 4 * q;
            const int row0 = (tid >> 3) / NT;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int row = row0 + i * (32 / NT);
                const int c = c0 + row;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (row < ROWS && c < C && sub < NS && t < T) {
                    const long long off = ((long long)b * C + c) * T + t;
                    if (vec) {
                        val = ld4(src + off);
                        if (mask_y) {
                            const f32x4 my = ld4(mask_y + off);
                            const float md = mask_drop ? mask_drop[b * C + c] : 1.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] *= md * (my[e] > 0.f ? 1.f : slope);
                        }
                    } else {
                        const float md = (mask_y && mask_drop) ? mask_drop[b * C + c] : 1.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (t + e < T) {
                                float u = ld1(src + off + e);
                                if (mask_y) u *= md * (ld1(mask_y + off + e) > 0.f ? 1.f : slope);
                                val[e] = u;
                            }
                        }
                    }
                }
                v[i] = val;
            }
        }
        if (HALO) {
            const int side = tid & 1;
            const int n = (tid >> 1) % NT;
            const int sub = sub0 + n;
            const int b = sub / ST;
            const int t = (sub - b * ST) * 32 + (side ? 32 : -1);
            const int row0 = (tid >> 1) / NT;
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                const int row = row0 + i * (128 / NT);
                const int c = c0 + row;
                float u = 0.f;
                if (row < ROWS && c < C && sub < NS && t >= 0 && t < T) {
                    const long long off = ((long long)b * C + c) * T + t;
                    u = ld1(src + off);
                    if (mask_y) {
                        const float md = mask_drop ? mask_drop[b * C + c] : 1.f;
                        u *= md * (ld1(mask_y + off) > 0.f ? 1.f : slope);
                    }
                }
                hv[i] = u;
            }
        }
    }

    __device__ __forceinline__ void store(float* lds) const {
        const int tid = threadIdx.x;
        {
            const int q = tid & 7;
            const int n = (tid >> 3) % NT;
            const int row0 = (tid >> 3) / NT;
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int row = row0 + i * (32 / NT);
                if (row < ROWS) *reinterpret_cast<f32x4*>(lds + row * LD + n * SUB + 4 + 4 * q) = v[i];
            }
        }
        if (HALO) {
            const int side = tid & 1;
            const int n = (tid >> 1) % NT;
            const int row0 = (tid >> 1) / NT;
#pragma unroll
            for (int i = 0; i < HPT; ++i) {
                const int row = row0 + i * (128 / NT);
                if (row < ROWS) lds[row * LD + n * SUB + (side ? 36 : 3)] = hv[i];
            }
        }
    }
};

// ---------------------------------------------------------------------------------------
// Forward kernel (the input gradient is the same kernel on tap-flipped, channel-transposed
// weights, see eeggan_conv1d_flip_weights).
//
// Persistent workgroups walk a flattened list of (tile, Cin-chunk) steps.  A tile is CT output
// channels x NT 32-wide subtiles; a chunk is CK = 32 input channels.  Per step the weight chunk
// [CT][CK*K] and the activation chunk [CK][NT*40] are staged global -> registers -> LDS into the
// buffer the NEXT step reads (two LDS buffers, loads issued before the MFMA block and written
// after it), so one barrier per step suffices and global latency hides under the MFMAs.
//   K taps (1|3);  CK input channels per chunk;  WM x WN wave grid over (32-channel row tiles x
//   subtiles);  VEC: T % 4 == 0;  FULL: Cin % CK == 0, Cout % CT == 0 and T % 32 == 0, so that
//   the staging needs neither predicates nor per-step address arithmetic (per-tile base pointers
//   plus wave-uniform chunk offsets);  KS (1|2): the waves split each chunk's reduction indices KS
//   ways and sum their accumulators through LDS before the epilogue -- for layers with so few
//   (b, t) positions (T = 32) that the tile has to be one subtile wide to give every CU work.
//   SPLIT (FULL, K = 3, CK = 32): products on the bf16 matrix pipe from three bf16 pieces per operand (six piece
//   products, fp32-exact to 2^-24).  The weight chunk is split once, when it is staged: three piece planes
//   [CT][k-step s = 2 tap + (ci >> 4)][lane half (ci >> 3) & 1][8 channels], one 16-byte operand fragment per
//   (row, k-step, lane half), rows WROWB bytes apart; the activations stay fp32 in LDS and are split as read.
//   MASK (-1 | 0 | 1): whether the input carries a fused backward mask; -1 decides at run time.  The two-stage
//   prefetch wants it known at compile time: a branch around the mask loads makes the load counts the waits
//   are computed from path-dependent, and the waits conservative.
//   BF (FULL, SPLIT = 1): the activations (x, mask_y, y, act_aux) are bfloat16 in memory (bf16 storage mode).
//   FULL tolerates Cout % CT != 0: the last channel tile stages the last weight row again for its missing rows and
//   stores only the rows that exist (the input gradient of a layer with 256 + 1 input channels).
//   RAGC (FULL, SPLIT = 3): Cin = 32 n + r with a few left-over channels r.  The main loop takes the 32 n channels
//   (weight rows are then not 16-byte aligned: their vector loads are declared 4-byte aligned); the r channels'
//   terms are added in the epilogue with plain fp32 arithmetic -- the layer behind MinibatchStddev has 256 + 1 input
//   channels, and the generic predicated form took 2.5x the time of the 256-channel layers beside it.
template <int K, int CT, int NT, int CK, int WM, int WN, bool VEC, bool FULL, int KS, int SPLIT = 0, int MASK = -1,
          bool BF = false, bool RAGC = false, int NTHR = 64 * WM * WN * KS>
__global__ __launch_bounds__(NTHR) void conv_fwd_kernel(ConvArgs a) {
    static_assert(!SPLIT || (FULL && K == 3 && CK == 32 && (KS == 1 || KS == 2)), "split products: predicate-free k = 3 form");
    static_assert(!BF || (FULL && VEC && SPLIT == 1), "bf16 storage: predicate-free form with bf16 products");
    static_assert(!RAGC || (FULL && VEC && SPLIT != 0 && !BF), "left-over input channels: predicate-free form, float32 storage");
    typedef typename ActT<BF>::T AT;
    typedef typename ActT<BF>::raw4 R4;
    typedef typename ActT<BF>::raw1 R1;
    const AT* const X0 = reinterpret_cast<const AT*>(a.x);
    const AT* const M0 = reinterpret_cast<const AT*>(a.mask_y);
    constexpr int PAD = (K - 1) / 2;
    constexpr int WROWB = CK * K * 2 + 16;  // SPLIT: bytes per weight row and piece (conflict-free b128 reads)
    constexpr int WLD = SPLIT ? 3 * WROWB / 4 : CK * K + 4;  // LDS floats per weight row of a chunk (fp32 form: conflict-free b128 reads)
    constexpr int XLD = NT * SUB;
    constexpr int RM = CT / 32 / WM;
    constexpr int RN = NT / WN;
    constexpr int KH = CK / 2 * K;      // k-steps per chunk (each MFMA consumes 2 reduction indices)
    constexpr int WQ = CK * K / 4;      // float4 per weight row of a chunk
    constexpr int WV = CT * WQ / NTHR;  // float4 weight items per thread
    constexpr int XV = CK * NT * 8 / NTHR;  // float4 activation items per thread
    constexpr int XROWS = NTHR / 8 / NT;    // activation rows covered per pass of the workgroup's threads
    constexpr int HROWS = NTHR / 2 / NT;    // halo rows covered per pass
    constexpr int HV = (CK + HROWS - 1) / HROWS;
    constexpr int BUF = CT * WLD + CK * XLD;
    static_assert((NTHR == 256 || NTHR == 512) && KH % (4 * KS) == 0 && (CT * WQ) % NTHR == 0 && XV >= 1 &&
                      (CK * NT * 8) % NTHR == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int wave = tid >> 6;
    const int kid = wave / (WM * WN);  // which share of every chunk's reduction indices this wave takes
    const int wm = (wave % (WM * WN)) / WN;
    const int wn = wave % WN;

    const int tiles_x = (a.NS + NT - 1) / NT;
    const int tiles_y = (a.Cout + CT - 1) / CT;
    const int ntiles = tiles_x * tiles_y;
    const int nchunks = RAGC ? a.Cin / CK : (a.Cin + CK - 1) / CK;
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nsteps = my_tiles * nchunks;
    if (nsteps <= 0) return;
    const bool wvec = ((a.Cin * K) % 4 == 0) && (a.Cin % CK == 0);
    const bool has_mask = MASK < 0 ? a.mask_y != nullptr : MASK != 0;

    // per-thread staging coordinates that do not depend on the step
    int w_row[WV], w_q[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        if (SPLIT) {  // a thread owns 4 consecutive channels of one weight row: 12 floats, one 8-byte fragment half per tap and piece
            static_assert(!SPLIT || (WV == 3 && NTHR == CT * 8), "one (row, 4-channel group) per thread");
            w_row[i] = tid >> 3;
            w_q[i] = (tid & 7) * 3 + i;
        } else {
            const int e4 = tid + NTHR * i;
            w_row[i] = e4 / WQ;
            w_q[i] = e4 - w_row[i] * WQ;
        }
    }
    const int x_q = tid & 7, x_n = (tid >> 3) % NT, x_row0 = (tid >> 3) / NT;
    const int h_side = tid & 1, h_n = (tid >> 1) % NT, h_row0 = (tid >> 1) / NT;

    // Registers of one prefetched step.  SPLIT keeps two steps in flight (DEEP): the loads of step s + 2 are issued
    // before the MFMAs of step s and written to LDS after the MFMAs of step s + 1, so they have a whole step to
    // arrive -- a chunk's MFMAs are shorter than a trip to L2 once the products run on the bf16 pipe.
    constexpr bool DEEP = SPLIT;
    struct Stage {
        f32x4 wv[WV];
        R4 xv[XV];
        R4 mv[VEC ? XV : 1];  // mask_y values of the activation items (VEC path)
        R1 hv[HV], mhv[HV];
        float mdv[XV], mdh[HV];
        unsigned okbits;         // bit i: weight item i valid; bit 8+i: activation item i; bit 16+i: halo i
        int p_lim;
        bool x_ok, h_ok;         // the tile-level validity of this thread's items when the step was prefetched
    };
    Stage sa, sb;

    // state of the step being prefetched
    int p_tile = blockIdx.x, p_chunk = 0;
    // per-tile staging state (recomputed when the prefetch moves to a new tile)
    long long x_off = 0, h_off = 0;
    int x_b = 0, h_b = 0;
    bool x_ok = false, h_ok = false;
    int p_co0 = 0;
    // FULL: per-tile base pointers of this thread's items
    const float* fw[WV];
    const AT* fx = X0;
    const AT* fh = X0;
    const AT* fmx = X0;
    const AT* fmh = X0;
    const float* fdx = a.w;
    const float* fdh = a.w;

    auto setup_tile = [&](int tile) {
        // (a tile past the end -- the two-stage prefetch runs unconditionally -- stages nothing: all of its loads
        // go to one fixed address)
        const bool live = tile < ntiles;
        if (!live) tile = 0;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        p_co0 = ty * CT;
        const int sub0 = tx * NT;
        {
            const int sub = sub0 + x_n;
            const int b = sub / a.ST;
            const int t = (sub - b * a.ST) * 32 + 4 * x_q;
            x_ok = live && sub < a.NS && t < a.T;
            x_b = b;
            x_off = ((long long)b * a.Cin + x_row0) * a.T + t;
        }
        if (K > 1) {
            const int sub = sub0 + h_n;
            const int b = sub / a.ST;
            const int t = (sub - b * a.ST) * 32 + (h_side ? 32 : -1);
            // (with few subtiles per tile the halo pass has more threads than the chunk has rows)
            h_ok = live && sub < a.NS && t >= 0 && t < a.T && h_row0 < CK;
            h_b = b;
            h_off = ((long long)b * a.Cin + h_row0) * a.T + t;
        }
        if (FULL) {
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int co = p_co0 + w_row[i] < a.Cout ? p_co0 + w_row[i] : a.Cout - 1;  // (a partly filled last tile)
                fw[i] = live ? a.w + ((long long)co * a.Cin) * K + 4 * w_q[i] : a.w;
            }
            fx = X0 + (x_ok ? x_off : 0);
            fh = X0 + (h_ok ? h_off : 0);
            if (has_mask) {
                fmx = M0 + (x_ok ? x_off : 0);
                fmh = M0 + (h_ok ? h_off : 0);
                if (a.mask_drop) {
                    fdx = a.mask_drop + (x_ok ? x_b * a.Cin + x_row0 : 0);
                    fdh = a.mask_drop + (h_ok ? h_b * a.Cin + h_row0 : 0);
                }
            }
        }
    };

    // Prefetch ONLY issues loads (from clamped, always-valid addresses) and records validity bits;
    // every consumer of the loaded values (zeroing of out-of-range items, the fused backward mask,
    // the equalized-lr scale) runs in commit(), after the MFMA block, so the loads' latency hides
    // under it.  A predicated load, or a select right behind the load, makes the compiler wait for
    // the whole prefetch before the first MFMA.
    auto prefetch = [&](Stage& s) {  // global -> registers for step (p_tile, p_chunk)
        auto& wv = s.wv;
        auto& xv = s.xv;
        auto& mv = s.mv;
        auto& hv = s.hv;
        auto& mdv = s.mdv;
        auto& mhv = s.mhv;
        auto& mdh = s.mdh;
        unsigned& okbits = s.okbits;
        int& p_lim = s.p_lim;
        s.x_ok = x_ok;
        s.h_ok = h_ok;
        const int ci0 = p_chunk * CK;
        if constexpr (FULL) {
            // every address = per-tile base + a wave-uniform offset; a whole-item predicate (tile tail,
            // sample edge) only redirects the base to a valid address, its value is dropped in commit()
            const int wo = ci0 * K;
            const long long xo = (long long)ci0 * a.T;
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                if constexpr (RAGC) wv[i] = *reinterpret_cast<const f32x4u*>(fw[i] + wo);
                else wv[i] = *reinterpret_cast<const f32x4*>(fw[i] + wo);
            }
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const long long o = x_ok ? xo + (long long)(i * XROWS) * a.T : 0;
                xv[i] = ld_raw4(fx + o);
                if (has_mask) {  // wave-uniform
                    mv[VEC ? i : 0] = ld_raw4(fmx + o);
                    mdv[i] = fdx[(x_ok && a.mask_drop) ? ci0 + i * XROWS : 0];  // (without multipliers: any valid address; selected in commit)
                }
            }
            if (K > 1) {
#pragma unroll
                for (int i = 0; i < HV; ++i) {
                    const bool ok = h_ok && h_row0 + i * HROWS < CK;
                    const long long o = ok ? xo + (long long)(i * HROWS) * a.T : 0;
                    hv[i] = ld_raw1(fh + o);
                    if (has_mask) {
                        mhv[i] = ld_raw1(fmh + o);
                        mdh[i] = fdh[(ok && a.mask_drop) ? ci0 + i * HROWS : 0];
                    }
                }
            }
            return;
        } else {
        okbits = 0;
        if (wvec) {
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int co = p_co0 + w_row[i];
                const bool ok = co < a.Cout;
                okbits |= ok ? (1u << i) : 0u;
                const long long off = ok ? ((long long)co * a.Cin + ci0) * K + 4 * w_q[i] : 0;
                wv[i] = *reinterpret_cast<const f32x4*>(a.w + off);
            }
        } else {
            p_lim = (a.Cin - ci0) * K;  // valid floats of this chunk per weight row
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                const int co = p_co0 + w_row[i];
                const bool rok = co < a.Cout;
                okbits |= rok ? (1u << i) : 0u;
                const long long base = ((long long)co * a.Cin + ci0) * K + 4 * w_q[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = rok && 4 * w_q[i] + e < p_lim;
                    wv[i][e] = a.w[ok ? base + e : 0];
                }
            }
        }
        const long long coff = (long long)ci0 * a.T;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int c = ci0 + x_row0 + i * XROWS;
            const long long off = x_off + coff + (long long)(i * XROWS) * a.T;
            const bool ok = x_ok && c < a.Cin;
            okbits |= ok ? (1u << (8 + i)) : 0u;
            if (VEC) {
                const long long o = ok ? off : 0;
                xv[i] = *reinterpret_cast<const f32x4*>(a.x + o);
                if (has_mask) {  // wave-uniform
                    mv[VEC ? i : 0] = *reinterpret_cast<const f32x4*>(a.mask_y + o);
                    mdv[i] = a.mask_drop ? a.mask_drop[ok ? x_b * a.Cin + c : 0] : 1.f;
                }
            } else {
                const int t = (int)((x_off - ((long long)x_b * a.Cin + x_row0) * a.T));
                const float md = (has_mask && a.mask_drop) ? a.mask_drop[ok ? x_b * a.Cin + c : 0] : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool oke = ok && t + e < a.T;
                    const long long o = oke ? off + e : 0;
                    float u = a.x[o];
                    if (has_mask) u *= a.mask_y[o] > 0.f ? md : md * a.mask_slope;
                    xv[i][e] = oke ? u : 0.f;  // (rare path: T % 4 != 0)
                }
            }
        }
        if (K > 1) {
#pragma unroll
            for (int i = 0; i < HV; ++i) {
                const int row = h_row0 + i * HROWS;
                const int c = ci0 + row;
                const bool ok = h_ok && row < CK && c < a.Cin;
                okbits |= ok ? (1u << (16 + i)) : 0u;
                const long long off = ok ? h_off + coff + (long long)(i * HROWS) * a.T : 0;
                hv[i] = a.x[off];
                if (has_mask) {
                    mhv[i] = a.mask_y[off];
                    mdh[i] = a.mask_drop ? a.mask_drop[ok ? h_b * a.Cin + c : 0] : 1.f;
                }
            }
        }
        }
    };

    auto commit = [&](const Stage& s, float* buf) {  // registers -> LDS buffer (all consumers of the prefetch live here)
        const auto& wv = s.wv;
        const auto& xv = s.xv;
        const auto& mv = s.mv;
        const auto& hv = s.hv;
        const auto& mdv = s.mdv;
        const auto& mhv = s.mhv;
        const auto& mdh = s.mdh;
        const unsigned okbits = s.okbits;
        const int p_lim = s.p_lim;
        const bool x_ok = s.x_ok, h_ok = s.h_ok;
        float* Ws = buf;
        float* Xs = buf + CT * WLD;
        if constexpr (FULL) {
            if constexpr (SPLIT) {
                u32x2s p[3][3];  // [tap][piece]: this thread's channels 4 (tid & 7) .. + 3 of the chunk, two per word
#pragma unroll
                for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
                        const int k0 = 6 * cp + tap, k1 = k0 + 3;  // float k of the thread is tap k % 3 of channel k / 3
                        unsigned w1, w2, w3;
                        if constexpr (SPLIT == 3) {
                            split3s_pair(wv[k0 / 4][k0 % 4] * a.in_scale, wv[k1 / 4][k1 % 4] * a.in_scale, w1, w2, w3);
                        } else {
                            w1 = pack_bf16s(wv[k0 / 4][k0 % 4] * a.in_scale, wv[k1 / 4][k1 % 4] * a.in_scale);
                            w2 = w3 = 0;
                        }
                        p[tap][0][cp] = w1;
                        p[tap][1][cp] = w2;
                        p[tap][2][cp] = w3;
                    }
                const int g4 = tid & 7;  // channels 4 g4 ..: k-step half g4 >> 2, lane half (g4 >> 1) & 1, fragment half g4 & 1
                char* Wb = reinterpret_cast<char*>(Ws) + (tid >> 3) * WROWB + (g4 >> 2) * 32 + ((g4 >> 1) & 1) * 16 + (g4 & 1) * 8;
#pragma unroll
                for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        *reinterpret_cast<u32x2s*>(Wb + k * CT * WROWB + tap * 64) = p[tap][k];
            } else {
#pragma unroll
                for (int i = 0; i < WV; ++i) {
                    f32x4 v = wv[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= a.in_scale;
                    *reinterpret_cast<f32x4*>(Ws + w_row[i] * WLD + 4 * w_q[i]) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                f32x4 v = widen(xv[i]);
                if (has_mask) {
                    const f32x4 my = widen(mv[VEC ? i : 0]);
                    const float d = a.mask_drop ? mdv[i] : 1.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= my[e] > 0.f ? d : d * a.mask_slope;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = x_ok ? v[e] : 0.f;
                *reinterpret_cast<f32x4*>(Xs + (x_row0 + i * XROWS) * XLD + x_n * SUB + 4 + 4 * x_q) = v;
            }
            if (K > 1) {
#pragma unroll
                for (int i = 0; i < HV; ++i) {
                    const int row = h_row0 + i * HROWS;
                    float u = widen(hv[i]);
                    const float d = a.mask_drop ? mdh[i] : 1.f;
                    if (has_mask) u *= widen(mhv[i]) > 0.f ? d : d * a.mask_slope;
                    if (row < CK) Xs[row * XLD + h_n * SUB + (h_side ? 36 : 3)] = h_ok ? u : 0.f;
                }
            }
            return;
        } else {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const bool ok = (okbits >> i) & 1u;
            f32x4 v = wv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool oke = ok && (wvec || 4 * w_q[i] + e < p_lim);
                v[e] = oke ? v[e] * a.in_scale : 0.f;
            }
            *reinterpret_cast<f32x4*>(Ws + w_row[i] * WLD + 4 * w_q[i]) = v;
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const bool ok = (okbits >> (8 + i)) & 1u;
            f32x4 v = xv[i];
            if (VEC) {
                if (has_mask) {
                    const f32x4 my = mv[VEC ? i : 0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= my[e] > 0.f ? mdv[i] : mdv[i] * a.mask_slope;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(Xs + (x_row0 + i * XROWS) * XLD + x_n * SUB + 4 + 4 * x_q) = v;
        }
        if (K > 1) {
#pragma unroll
            for (int i = 0; i < HV; ++i) {
                const int row = h_row0 + i * HROWS;
                const bool ok = (okbits >> (16 + i)) & 1u;
                float u = hv[i];
                if (has_mask) u *= mhv[i] > 0.f ? mdh[i] : mdh[i] * a.mask_slope;
                if (row < CK) Xs[row * XLD + h_n * SUB + (h_side ? 36 : 3)] = ok ? u : 0.f;
            }
        }
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    EEGGAN_STAMP(0, 6);
    auto advance = [&]() {
        if (++p_chunk == nchunks) {
            p_chunk = 0;
            p_tile += gridDim.x;
            setup_tile(p_tile);
        }
    };
    // DEEP prefetches unconditionally (past the last step from one fixed address, the values are never
    // committed): a branch around the loads would make the wait before each commit cover both stages.
    setup_tile(p_tile);
    prefetch(sa);
    if (DEEP) {
        advance();
        prefetch(sb);
    }
    commit(sa, smem);
    __syncthreads();
    EEGGAN_STAMP(0, 7);

    int c_tile = blockIdx.x, c_chunk = 0;  // step being computed
    // bias and multiplier of this lane's output rows, loaded at the start of a tile's LAST step: in the epilogue
    // their latency would be exposed (through an always-valid pointer, neutralised by a select at their use)
    float pre_b[RM], pre_d[RM][RN];
    auto body = [&](const int step, Stage& s_issue, const Stage& s_commit) {
        const bool more = step + 1 < nsteps;
        EEGGAN_STAMP(step, 0);
        if (c_chunk + 1 == nchunks) {
            const int ty = c_tile / tiles_x, tx = c_tile - ty * tiles_x;
            const float* bp = a.bias ? a.bias : a.w;
            const float* dp = a.drop ? a.drop : a.w;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
                const int co = ty * CT + (wm * RM + i) * 32 + l31;
                const int cc = co < a.Cout ? co : 0;
                pre_b[i] = bp[a.bias ? cc : 0];
#pragma unroll
                for (int n = 0; n < RN; ++n) {
                    const int sub = tx * NT + wn * RN + n;
                    const int b = sub < a.NS ? sub / a.ST : 0;
                    pre_d[i][n] = dp[a.drop ? b * a.Cout + cc : 0];
                }
            }
        }
        if (DEEP || more) {
            advance();
            prefetch(s_issue);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMA block
        EEGGAN_STAMP(step, 1);
        const float* buf = smem + (step & 1) * BUF;
        const float* Ap = buf + (wm * RM * 32 + l31) * WLD + h * KH;
        const float* Bp = buf + CT * WLD + h * (CK / 2) * XLD + wn * RN * SUB + (4 - PAD) + l31;
        if constexpr (SPLIT) {
            // k-step s = (tap j, channel half g): this lane's 8 channels are 16 g + 8 h + i, at time l31 + j - 1
            const char* Wb = reinterpret_cast<const char*>(buf) + (wm * RM * 32 + l31) * WROWB + h * 16;
            const float* Bq = buf + CT * WLD + h * 8 * XLD + wn * RN * SUB + (4 - PAD) + l31;
            // all LDS reads of this wave's k-steps first (a wave has at most one partner on its SIMD to hide their
            // latency behind), then the splitting and the MFMAs
            // (the weight fragments too where the registers allow it: one 32-channel row tile per wave)
            constexpr int NSTEP = 6 / KS;
            constexpr bool HOISTW = RM == 1;
            float xr[NSTEP][RN][8];
            bf16x8s wf[HOISTW ? NSTEP : 1][RM][3];
            auto read_w = [&](int u, int s) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        wf[HOISTW ? u : 0][i][k] = *reinterpret_cast<const bf16x8s*>(Wb + i * 32 * WROWB + s * 32 + k * CT * WROWB);
            };
#pragma unroll
            for (int u = 0; u < NSTEP; ++u) {
                const int s = kid * NSTEP + u, j = s >> 1, g = s & 1;
#pragma unroll
                for (int n = 0; n < RN; ++n)
#pragma unroll
                    for (int i = 0; i < 8; ++i) xr[u][n][i] = Bq[(16 * g + i) * XLD + n * SUB + j];
                if (HOISTW) read_w(u, s);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPLIT == 1) {  // bf16 compute mode: the operands rounded to bf16, one product
#pragma unroll
                for (int u = 0; u < NSTEP; ++u) {
                    if (!HOISTW) read_w(u, kid * NSTEP + u);
                    const bf16x8s(&w1)[RM][3] = wf[HOISTW ? u : 0];
#pragma unroll
                    for (int n = 0; n < RN; ++n) {
                        u32x4s q1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) q1[i] = pack_bf16s(xr[u][n][2 * i], xr[u][n][2 * i + 1]);
                        const bf16x8s x1 = __builtin_bit_cast(bf16x8s, q1);
#pragma unroll
                        for (int i = 0; i < RM; ++i)
                            acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1[i][0], acc[i][n], 0, 0, 0);
                    }
                }
            } else {
                // items (k-step u, subtile n): 8 samples split into pieces (VALU), then 6 MFMAs per row tile.  The
                // splitting of item it + 1 is placed in the MFMA gaps of item it: a vector instruction between two MFMAs
                // of the same wave is free, a block of them ahead of the MFMAs costs its full issue time (conv_ws.hip).
                constexpr int NITEM = NSTEP * RN;
                u32x4s qc[3], qn[3];
                auto prep = [&](int it, u32x4s (&q)[3]) {
                    const int u = it / RN, n = it - u * RN;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        unsigned w1_, w2_, w3_;
                        split3s_pair(xr[u][n][2 * i], xr[u][n][2 * i + 1], w1_, w2_, w3_);
                        q[0][i] = w1_;
                        q[1][i] = w2_;
                        q[2][i] = w3_;
                    }
                };
                prep(0, qc);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < NITEM; ++it) {
                    const int u = it / RN, n = it - u * RN;
                    if (!HOISTW && n == 0) read_w(u, kid * NSTEP + u);
                    const bf16x8s(&w1)[RM][3] = wf[HOISTW ? u : 0];
                    if (it + 1 < NITEM) prep(it + 1, qn);
                    const bf16x8s x1 = __builtin_bit_cast(bf16x8s, qc[0]), x2 = __builtin_bit_cast(bf16x8s, qc[1]),
                                  x3 = __builtin_bit_cast(bf16x8s, qc[2]);
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, w1[i][0], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1[i][2], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w1[i][1], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w1[i][0], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1[i][1], acc[i][n], 0, 0, 0);
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1[i][0], acc[i][n], 0, 0, 0);
                    }
                    if (!HOISTW && n == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3 * RM, 0);
#pragma unroll
                    for (int g = 0; g < 6 * RM; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, RM == 1 ? 8 : 4, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 3; ++k) qc[k] = qn[k];
                }
            }
        } else
#pragma unroll
        for (int s4 = kid * (KH / 4 / KS); s4 < (kid + 1) * (KH / 4 / KS); ++s4) {
            f32x4 av[RM];
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = *reinterpret_cast<const f32x4*>(Ap + i * 32 * WLD + 4 * s4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * s4 + u;
                const int cl = s / K;
                const int j = s - cl * K;
                float bv[RN];
#pragma unroll
                for (int n = 0; n < RN; ++n) bv[n] = Bp[cl * XLD + n * SUB + j];
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int n = 0; n < RN; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[n], av[i][u], acc[i][n], 0, 0, 0);
            }
        }
        EEGGAN_STAMP(step, 2);
        if (c_chunk + 1 == nchunks) {
            // epilogue of the finished tile: bias -> activation -> per-(sample, channel) multiplier.
            // The activation is dispatched once (wave-uniform); optional operands are read through
            // an always-valid pointer and neutralised by a select, so the 16*RM*RN stores of a
            // lane are straight-line code.
            const int ty = c_tile / tiles_x, tx = c_tile - ty * tiles_x;
            const int co0 = ty * CT;
            const bool full = co0 + CT <= a.Cout;
            const bool has_bias = a.bias != nullptr, has_drop = a.drop != nullptr;
            if constexpr (RAGC) {
                // The left-over input channels, added to the accumulators of the waves that will run the epilogue:
                // out[t] += sum_j w[j] x[t + j - 1], six inputs per run of four outputs.  All loads come first (from
                // valid addresses, dropped by selects) and nothing is stored in between: one trip to memory.
                if (KS == 1 || kid == 0) {
#pragma unroll
                    for (int n = 0; n < RN; ++n) {
                        const int sub = tx * NT + wn * RN + n;
                        const int b = sub / a.ST;
                        const int t0 = (sub - b * a.ST) * 32 + 4 * h;
                        const bool sok = sub < a.NS;
                        const int bc = sok ? b : 0;
#pragma unroll
                        for (int i = 0; i < RM; ++i) {
                            const int co = co0 + (wm * RM + i) * 32 + l31;
                            const int cc = co < a.Cout ? co : 0;
                            for (int c = nchunks * CK; c < a.Cin; ++c) {
                                const float* wp = a.w + ((long long)cc * a.Cin + c) * 3;
                                const float w0 = wp[0] * a.in_scale, w1 = wp[1] * a.in_scale, w2 = wp[2] * a.in_scale;
                                const long long xrow = ((long long)bc * a.Cin + c) * a.T;
                                const float d = (has_mask && a.mask_drop) ? a.mask_drop[bc * a.Cin + c] : 1.f;
                                f32x4 xm[4], mm[4];
                                float xl[4], xh[4], ml[4], mh[4];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int t = sok ? t0 + 8 * q : 0;
                                    const bool lo = t > 0, hi = t + 4 < a.T;
                                    xm[q] = *reinterpret_cast<const f32x4*>(a.x + xrow + t);
                                    xl[q] = a.x[xrow + (lo ? t - 1 : 0)];
                                    xh[q] = a.x[xrow + (hi ? t + 4 : 0)];
                                    if (has_mask) {
                                        mm[q] = *reinterpret_cast<const f32x4*>(a.mask_y + xrow + t);
                                        ml[q] = a.mask_y[xrow + (lo ? t - 1 : 0)];
                                        mh[q] = a.mask_y[xrow + (hi ? t + 4 : 0)];
                                    }
                                }
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int t = sok ? t0 + 8 * q : 0;
                                    const bool lo = t > 0, hi = t + 4 < a.T;
                                    if (has_mask) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) xm[q][e] *= mm[q][e] > 0.f ? d : d * a.mask_slope;
                                        xl[q] *= ml[q] > 0.f ? d : d * a.mask_slope;
                                        xh[q] *= mh[q] > 0.f ? d : d * a.mask_slope;
                                    }
                                    const float l = lo ? xl[q] : 0.f, r = hi ? xh[q] : 0.f;
                                    acc[i][n][4 * q + 0] += w0 * l + w1 * xm[q][0] + w2 * xm[q][1];
                                    acc[i][n][4 * q + 1] += w0 * xm[q][0] + w1 * xm[q][1] + w2 * xm[q][2];
                                    acc[i][n][4 * q + 2] += w0 * xm[q][1] + w1 * xm[q][2] + w2 * xm[q][3];
                                    acc[i][n][4 * q + 3] += w0 * xm[q][2] + w1 * xm[q][3] + w2 * r;
                                }
                            }
                        }
                    }
                }
            }
            if (KS > 1) {
                // waves kid > 0 hand their partial sums to the kid = 0 wave of the same (wm, wn)
                float* red = smem + 2 * BUF + ((wave % (WM * WN)) * RM * RN) * 16 * 64 + lane;
                if (kid > 0) {
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int n = 0; n < RN; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                red[((i * RN + n) * 16 + r) * 64] = acc[i][n][r];
                                acc[i][n][r] = 0.f;
                            }
                }
                __syncthreads();
                if (kid == 0) {
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int n = 0; n < RN; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][n][r] += red[((i * RN + n) * 16 + r) * 64];
                }
            }
            auto run = [&](auto tag) {
                constexpr int ACT = decltype(tag)::value;
#pragma unroll
                for (int n = 0; n < RN; ++n) {
                    const int sub = tx * NT + wn * RN + n;
                    const int b = sub / a.ST;
                    const int t0 = (sub - b * a.ST) * 32 + 4 * h;
                    const bool sok = sub < a.NS;
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        // accumulator layout: lane = output channel, register r = time (r&3) + 8*(r>>2) + 4*h,
                        // i.e. four runs of 4 consecutive time steps per lane: 16-byte stores
                        const int co = co0 + (wm * RM + i) * 32 + l31;
                        const bool cok = sok && (full || co < a.Cout);
                        const int cc = cok ? co : 0;
                        const int bc = cok ? b : 0;
                        const float bb = has_bias ? pre_b[i] : 0.f;
                        const float dd = has_drop ? pre_d[i][n] : 1.f;
                        const long long row = ((long long)bc * a.Cout + cc) * a.T;
                        // the mask values of the block's four stores, fetched before the first of them: a load
                        // between two stores makes the second wait for the first (one counter for both)
                        f32x4 mk[4];
                        if (ACT == EEGGAN_ACT_MASK && VEC) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int t = t0 + 8 * q;
                                mk[q] = ld4(reinterpret_cast<const AT*>(a.act_aux) + row + ((cok && t < a.T) ? t : 0));
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int t = t0 + 8 * q;
                            const bool ok = cok && t < a.T;
                            const long long o = row + (ok ? t : 0);
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][n][4 * q + e] + bb;
                            if (ACT == EEGGAN_ACT_LRELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.act_param;
                            }
                            if (ACT == EEGGAN_ACT_TANH) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e] * a.act_param);
                            }
                            if (ACT == EEGGAN_ACT_MASK) {
                                if (VEC) {
                                    const f32x4 m = mk[q];
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * a.act_param;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float m = a.act_aux[(ok && t + e < a.T) ? o + e : row];
                                        v[e] = m > 0.f ? v[e] : v[e] * a.act_param;
                                    }
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= dd;
                            {
                                if (VEC) {
                                    if (ok) st4(reinterpret_cast<AT*>(a.y) + o, v);
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if (ok && t + e < a.T) a.y[o + e] = v[e];
                                }
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
                    }
                }
            };
            if (KS == 1 || kid == 0) {
                switch (a.act) {
                    case EEGGAN_ACT_LRELU: run(std::integral_constant<int, EEGGAN_ACT_LRELU>{}); break;
                    case EEGGAN_ACT_TANH: run(std::integral_constant<int, EEGGAN_ACT_TANH>{}); break;
                    case EEGGAN_ACT_MASK: run(std::integral_constant<int, EEGGAN_ACT_MASK>{}); break;
                    default: run(std::integral_constant<int, EEGGAN_ACT_NONE>{}); break;
                }
            }
            c_chunk = 0;
            c_tile += gridDim.x;
        } else {
            ++c_chunk;
        }
        __builtin_amdgcn_sched_barrier(0);  // ... and their consumers below it
        EEGGAN_STAMP(step, 3);
        if (more) commit(s_commit, smem + ((step + 1) & 1) * BUF);
        EEGGAN_STAMP(step, 4);
        __syncthreads();
        EEGGAN_STAMP(step, 5);
    };
    if (DEEP) {
        for (int step = 0; step < nsteps; step += 2) {
            body(step, sa, sb);
            if (step + 1 >= nsteps) break;
            body(step + 1, sb, sa);
        }
    } else {
        for (int step = 0; step < nsteps; ++step) body(step, sa, sa);
    }
}

}  // namespace

// Matrix products of the predicate-free k = 3 layers: 3 = bf16 matrix instructions on three bf16 pieces per fp32
// operand (fp32 products; default), 1 = bf16 matrix instructions on the operands rounded to bf16 (bf16 compute
// mode: fp32 storage and accumulation), 0 = fp32 matrix instructions (comparison runs).  Process-wide; initial
// value from the environment (EEGGAN_FP32_MFMA=1 -> 0, EEGGAN_PRECISION=bf16 -> 1).
static int initial_pieces() {
    if (getenv("EEGGAN_FP32_MFMA") != nullptr) return 0;
    const char* p = getenv("EEGGAN_PRECISION");
    return (p != nullptr && strcmp(p, "bf16") == 0) ? 1 : 3;
}
static std::atomic<int> g_pieces{-1};

int product_pieces() {
    int v = g_pieces.load(std::memory_order_relaxed);
    if (v < 0) {
        v = initial_pieces();
        g_pieces.store(v, std::memory_order_relaxed);
    }
    return v;
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}

int num_cus() {
    static std::atomic<int> cus[MAX_DEVICES];
    const int dev = current_device();
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

namespace {

template <int K, int CT, int NT, int CK, int WM, int WN, bool VEC, bool FULL = false, int KS = 1, int SPLIT = 0, int MASK = -1,
          bool BF = false, bool RAGC = false>
int launch_fwd(const ConvArgs& a, hipStream_t st) {
    constexpr int wld = SPLIT ? 3 * (CK * K * 2 + 16) / 4 : CK * K + 4;  // floats per weight row (the kernel's WLD)
    constexpr size_t lds = ((size_t)2 * (CT * wld + CK * NT * SUB) + (KS > 1 ? 64 * 16 * (CT / 32) * NT : 0)) * sizeof(float);
    auto kernel = &conv_fwd_kernel<K, CT, NT, CK, WM, WN, VEC, FULL, KS, SPLIT, MASK, BF, RAGC>;
    // resident workgroups per CU; also opts in to > 64 KiB of dynamic LDS (once per device)
    static std::atomic<int> per_cu_dev[MAX_DEVICES];
    std::atomic<int>& slot = per_cu_dev[current_device()];
    int per_cu = slot.load(std::memory_order_acquire);
    if (per_cu == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("conv1d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return 1;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), 64 * WM * WN * KS, lds) != hipSuccess || n < 1)
            n = 1;
        per_cu = n > 4 ? 4 : n;
        slot.store(per_cu, std::memory_order_release);
    }
    const long long ntiles = (long long)((a.NS + NT - 1) / NT) * ((a.Cout + CT - 1) / CT);
    long long grid = (long long)num_cus() * per_cu;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * WM * WN * KS), lds, st, a);
    return check_launch("conv1d");
}

// Tile choice: four subtiles per workgroup (one workgroup per CU) while that still gives every CU a workgroup.  The short
// wide layers of the path (T = 32 .. 256 at 128-256 channels, 64 .. 512 subtiles) do not: they take a
// one-subtile tile whose four waves split the reduction, small enough in LDS for two workgroups per
// CU.  (Measured in the captured training step against a two-subtile tile without the split:
// +3.4 % steps/s; the T = 32 layers alone run twice as fast.)
// (comparison switch: EEGGAN_CONV_RAGGED=0 sends layers with 32 n + r input channels, or a partly filled last tile of
// output channels, to the generic predicated kernel)
static bool ragged_enabled() {
    static const bool on = [] {
        const char* e = getenv("EEGGAN_CONV_RAGGED");
        return !(e && e[0] == '0');
    }();
    return on;
}

template <int K>
int dispatch_fwd(const ConvArgs& a, hipStream_t st) {
    const long long ns = a.NS;
    auto tiles = [&](int ct, int nt) { return ((ns + nt - 1) / nt) * ((a.Cout + ct - 1) / ct); };
    {
        int rc = launch_conv_thin(a, K, st);
        if (rc >= 0) return rc;
        rc = launch_conv_ws(a, K, st);
        if (rc >= 0) return rc;
    }
    if ((a.x_bf16 || a.y_bf16) && (K != 3 || !a.vec || a.Cout <= 32)) {
        set_error("conv1d: bfloat16 activations are supported for the predicate-free k = 3 layers and for 1x1 layers "
                  "with at most 8 channels on one side (K=%d Cin=%d Cout=%d T=%d)", K, a.Cin, a.Cout, a.T);
        return 1;
    }
    if (!a.vec) return launch_fwd<K, 64, 2, 32, 2, 2, false>(a, st);
    if (a.Cout <= 32) return launch_fwd<K, 32, 8, 32, 1, 4, true>(a, st);
    const bool full = a.Cin % 32 == 0 && a.Cout % 64 == 0 && a.T % 32 == 0;
    // Rounds of the chip each tile needs (a round of four-subtile tiles is twice as long as a round of pairs of
    // one-subtile tiles): batches that are not a multiple of the chip width lose up to half a round on the big tile.
    const long long cus = num_cus();
    const long long cost4 = (tiles(64, 4) + cus - 1) / cus * 4 * 20, cost1 = (tiles(64, 1) + 2 * cus - 1) / (2 * cus) * 2 * 21;
    // (comparison switch: EEGGAN_CONV_TILE=narrow / big overrides the choice)
    static const int forced = [] {
        const char* e = getenv("EEGGAN_CONV_TILE");
        return !e ? 0 : (e[0] == 'n' ? 1 : (e[0] == 'b' ? 2 : 0));
    }();
    const bool big = forced ? forced == 2 : (tiles(64, 4) >= cus && cost4 <= cost1);
    if constexpr (K == 3) {
        // predicate-free layers: products from bf16 pieces.  The narrow tile is 32 channels x 2 subtiles there (as many
        // tiles as 64 x 1): with the weight chunk as three piece planes that is what still fits twice on a CU.
        const int pieces = product_pieces();
        if (a.x_bf16 || a.y_bf16) {
            // bf16 storage: the predicate-free layers with both sides bfloat16 and bf16 products
            if (!(full && pieces == 1 && a.x_bf16 && a.y_bf16)) {
                set_error("conv1d: bfloat16 activations need Cin %% 32 == 0, Cout %% 64 == 0, T %% 32 == 0, both sides "
                          "bfloat16 and the bf16 matrix precision (Cin=%d Cout=%d T=%d)", a.Cin, a.Cout, a.T);
                return 1;
            }
            if (a.mask_y)
                return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 1, true>(a, st)
                           : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 1, true>(a, st);
            return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 0, true>(a, st)
                       : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 0, true>(a, st);
        }
        // (a last channel tile that is only partly filled is fine for these kernels when it wastes little)
        const bool full_cin = a.Cin % 32 == 0 && a.T % 32 == 0 && a.Cout >= 128;
        if ((full || (full_cin && ragged_enabled())) && pieces == 3) {
            if (a.mask_y)
                return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 3, 1>(a, st)
                           : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 3, 1>(a, st);
            return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 3, 0>(a, st)
                       : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 3, 0>(a, st);
        }
        // a few input channels beyond a multiple of 32 (256 + 1 behind MinibatchStddev): added in the epilogue
        if (pieces == 3 && ragged_enabled() && a.Cin > 64 && a.Cin % 32 != 0 && a.Cin % 32 <= 4 && a.Cout % 64 == 0 && a.T % 32 == 0) {
            if (a.mask_y)
                return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 3, 1, false, true>(a, st)
                           : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 3, 1, false, true>(a, st);
            return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 3, 0, false, true>(a, st)
                       : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 3, 0, false, true>(a, st);
        }
        if ((full || (full_cin && ragged_enabled())) && pieces == 1) {
            if (a.mask_y)
                return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 1>(a, st)
                           : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 1>(a, st);
            return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 0>(a, st)
                       : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 0>(a, st);
        }
        // (the same with bf16 products: the left-over channels' terms stay plain fp32)
        if (pieces == 1 && ragged_enabled() && a.Cin > 64 && a.Cin % 32 != 0 && a.Cin % 32 <= 4 && a.Cout % 64 == 0 && a.T % 32 == 0) {
            if (a.mask_y)
                return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 1, false, true>(a, st)
                           : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 1, false, true>(a, st);
            return big ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2, 1, 0, false, true>(a, st)
                       : launch_fwd<K, 32, 2, 32, 1, 2, true, true, 2, 1, 0, false, true>(a, st);
        }
    }
    if (big)  // (predicate-free form: eight waves, two per SIMD, splitting the reduction: +1.3 % steps/s)
        return full ? launch_fwd<K, 64, 4, 32, 1, 4, true, true, 2>(a, st) : launch_fwd<K, 64, 4, 32, 1, 4, true>(a, st);
    return full ? launch_fwd<K, 64, 1, 32, 2, 1, true, true, 2>(a, st) : launch_fwd<K, 64, 1, 32, 2, 1, true, false, 2>(a, st);
}

__global__ __launch_bounds__(256) void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                           int Cin, int K) {
    const long long n = (long long)Cout * Cin * K;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // i indexes wt[ci][co][j]
    const int j = (int)(i % K);
    const long long r = i / K;
    const int co = (int)(r % Cout);
    const int ci = (int)(r / Cout);
    wt[i] = w[((long long)co * Cin + ci) * K + (K - 1 - j)];
}

// Same for up to 64 weight tensors in one launch (descriptor table by value in the kernel arguments):
// a network's flipped views all go stale together, at its optimizer step.
constexpr int FLIP_MAXT = 64;
struct FlipPack {
    eeggan_flip_desc t[FLIP_MAXT];
    int n;
};

__global__ __launch_bounds__(256) void flip_weights_multi_kernel(FlipPack pk) {
    long long blk = blockIdx.x;
    for (int k = 0; k < pk.n; ++k) {
        const eeggan_flip_desc d = pk.t[k];
        const long long n = (long long)d.Cout * d.Cin * d.K;
        const long long nb = (n + 255) / 256;
        if (blk < nb) {
            const long long i = blk * 256 + threadIdx.x;
            if (i < n) {
                const int j = (int)(i % d.K);
                const long long r = i / d.K;
                const int co = (int)(r % d.Cout);
                const int ci = (int)(r / d.Cout);
                d.wt[i] = d.w[((long long)co * d.Cin + ci) * d.K + (d.K - 1 - j)];
            }
            return;
        }
        blk -= nb;
    }
}

// ---------------------------------------------------------------------------------------
// Weight gradient.  A workgroup owns a 64(co) x 64(ci) x K region of gw and a contiguous
// range of subtile pairs; wave w accumulates the 32x32xK tile (co half = w>>1, ci half = w&1).
struct WgradArgs {
    const float* x;
    const float* gy;
    const float* mask_y;
    const float* mask_drop;
    float mask_slope;
    float* ws;    // [nsplit][Cout*Cin*K]
    float* ws_b;  // [nsplit][Cout] or null
    int B, Cin, Cout, T, ST, NS;
    int nci;      // ci regions
    int nsplit;
    int iters;    // subtile pairs per split
    int vec;
    int x_bf16, gy_bf16;  // bf16 storage mode: x / gy and mask_y hold bfloat16 (2-byte elements behind the pointers)
#ifdef EEGGAN_TRACE
    unsigned long long* trace;  // as ConvArgs::trace
#endif
};
#ifdef EEGGAN_TRACE
unsigned long long* g_wgrad_trace = nullptr;  // (trace builds: where the next weight-gradient launch writes its stamps)
#endif

// FULL: Cin % 64 == 0, Cout % 64 == 0, T % 32 == 0 and T % 4 == 0: staging without predicates on
// channels or time, with one wave-uniform (sample, time) decode per iteration, two LDS buffers and a
// single barrier per iteration.  Otherwise the generic XStage path.
// SPLIT (FULL, K = 3): the products run on the bf16 matrix pipe, each fp32 operand as the sum of three bf16
// pieces with the six leading piece products kept (fp32-exact to 2^-24; see conv_ws.hip): 18 MFMAs of 16
// time steps replace 24 fp32 MFMAs of 2.
// XBF / GBF: x / gy (and mask_y) are bfloat16 in memory.
// W4 (split fp32 form): four waves per workgroup, each taking BOTH subtiles of a pair, and ONE LDS buffer (43 KB), so
// that two workgroups share a CU.  Per SIMD that is still two waves, but of two workgroups that do not wait at each
// other's barriers: while one waits for its loads, the other multiplies.
template <int K, bool FULL, int SPLIT = 0, bool XBF = false, bool GBF = false, bool W4 = false, bool RAGX = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int split, const int region) {
    static_assert(!W4 || (SPLIT != 0 && !XBF && !GBF), "four-wave form: products on the bf16 pipe, float32 tensors");
    // RAGX: Cin is not a multiple of 64 (256 + 1 behind MinibatchStddev).  The last region's missing x rows are read
    // from the last channel again (valid addresses); what they accumulate is never stored (the slab stores are guarded
    // by ci < Cin).  A form of its own: the other layers' loop keeps its immediate row offsets.
    static_assert(!RAGX || (FULL && W4), "ragged input channels: a variant of the four-wave predicate-free form");
    static_assert(!SPLIT || (FULL && K == 3), "split products: predicate-free k = 3 form only");
    static_assert(!(FULL && (XBF || GBF)) || (SPLIT == 1 && XBF && GBF), "bf16 storage, predicate-free form: bf16 products, both sides");
    typedef typename ActT<XBF>::T XT;
    typedef typename ActT<GBF>::T GT;
    const XT* const X0 = reinterpret_cast<const XT*>(a.x);
    const GT* const G0 = reinterpret_cast<const GT*>(a.gy);
    const GT* const M0 = reinterpret_cast<const GT*>(a.mask_y);
    constexpr int PAD = (K - 1) / 2;
    constexpr int NTW = 2;
    constexpr int LD = NTW * SUB + 4;  // 84: (LD/4) odd -> conflict-free b128 row-indexed reads
    constexpr int BUF = 2 * 64 * LD;   // floats per LDS buffer (gy tile + x tile)
    // BFN (both tensors bfloat16, predicate-free form): the LDS tiles hold the stored bfloat16 values themselves, in
    // rows of LDH halfwords (176 bytes: (176 / 16) odd, conflict-free 16-byte row-indexed reads); a subtile takes 40
    // halfwords of a row from halfword 8 on (left halo at 7, right halo at 40).  A lane's eight time steps of a row
    // are then ONE 16-byte LDS read that already is the matrix operand: no conversion in the loop, half the LDS
    // traffic, and 45 KB of tiles instead of 86, so that two workgroups share a CU and twice the loads are in flight
    // (the kernel is bound by memory latency: one workgroup per CU keeps 24 KB per CU in flight).
    constexpr bool BFN = FULL && SPLIT == 1 && XBF && GBF;
    constexpr int LDH = WGRAD_LDH;
    constexpr int BUFH = 2 * 64 * LDH;  // halfwords per LDS buffer in that form
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    // SPLIT runs eight waves, two per SIMD: wave w and wave w + 4 share an output tile and take one subtile of every
    // pair each (their sums meet through LDS at the end).  On the bf16 pipe the matrix work of an iteration is
    // shorter than the splitting that feeds it, and a wave alone on its SIMD runs the two one after the other.
    constexpr int NTHR = (SPLIT && !W4) ? 512 : 256;
    const int wave = tid >> 6;
    const int part = (SPLIT && !W4) ? wave >> 2 : 0;
    const int cot = (wave >> 1) & 1, cit = wave & 1;
    const int co0 = (region / a.nci) * 64;
    const int ci0 = (region % a.nci) * 64;
    const bool do_bias = a.ws_b != nullptr && ci0 == 0 && cit == 0;

    f32x16 acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float bsum = 0.f;

    const int it0 = split * a.iters;
    const int total_pairs = (a.NS + NTW - 1) / NTW;
    int it1 = it0 + a.iters;
    if (it1 > total_pairs) it1 = total_pairs;

    auto mfma_block = [&](const float* Gs, const float* Xs) {
        if constexpr (BFN) {
            const unsigned short* Gr = reinterpret_cast<const unsigned short*>(Gs) + (cot * 32 + l31) * LDH + 8 + 8 * h + part * 40;
            const unsigned short* Xr = reinterpret_cast<const unsigned short*>(Xs) + (cit * 32 + l31) * LDH + 8 + 8 * h + part * 40;
            u32x4s gw[2], xw[2];
            unsigned xl[2], xh[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                gw[hf] = *reinterpret_cast<const u32x4s*>(Gr + 16 * hf);
                xw[hf] = *reinterpret_cast<const u32x4s*>(Xr + 16 * hf);
                xl[hf] = Xr[16 * hf - 1];
                xh[hf] = Xr[16 * hf + 8];
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                // the stored words are samples (1,2) (3,4) (5,6) (7,8) of the ten around this lane's eight time steps:
                // the middle tap's operand as they are, the outer taps re-cut across the word borders
                u32x4s t0, t2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t0[i] = __builtin_amdgcn_alignbit(xw[hf][i], i ? xw[hf][i ? i - 1 : 0] : xl[hf] << 16, 16);
                    t2[i] = __builtin_amdgcn_alignbit(i < 3 ? xw[hf][i < 3 ? i + 1 : 3] : xh[hf], xw[hf][i], 16);
                }
                const bf16x8s G1 = __builtin_bit_cast(bf16x8s, gw[hf]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, __builtin_bit_cast(bf16x8s, t0), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, __builtin_bit_cast(bf16x8s, xw[hf]), acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, __builtin_bit_cast(bf16x8s, t2), acc[2], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        bsum += __builtin_bit_cast(float, gw[hf][i] << 16) + __builtin_bit_cast(float, gw[hf][i] & 0xffff0000u);
                }
            }
            return;
        }
        if constexpr (SPLIT) {
          // (four-wave form: both subtiles of the pair, one after the other)
#pragma unroll
          for (int sub = W4 ? 0 : part; sub < (W4 ? 2 : part + 1); ++sub) {
            const float* Gr = Gs + (cot * 32 + l31) * LD + 4 + 8 * h + sub * SUB;  // (this wave's subtile of the pair)
            const float* Xr = Xs + (cit * 32 + l31) * LD + 4 + 8 * h + sub * SUB;
            // Two halves of 16 time steps; per half this lane's 8 time steps of the gradient row and the 10 input
            // samples around them.  All LDS reads first; the operands of the second half are prepared (VALU) in the
            // MFMA gaps of the first half -- a vector instruction between two MFMAs of the same wave is free, a block
            // of them ahead of the MFMAs is not (see conv_ws.hip).
            f32x4 g0[2], g1[2], x0[2], x1[2];
            float xl[2], xh[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int tb = 16 * hf;
                g0[hf] = *reinterpret_cast<const f32x4*>(Gr + tb);
                g1[hf] = *reinterpret_cast<const f32x4*>(Gr + tb + 4);
                x0[hf] = *reinterpret_cast<const f32x4*>(Xr + tb);
                x1[hf] = *reinterpret_cast<const f32x4*>(Xr + tb + 4);
                xl[hf] = Xr[tb - 1];
                xh[hf] = Xr[tb + 8];
            }
            // operand words of one half: G pieces [piece] (8 values), X pieces [tap][piece] (8 values each)
            struct Ops {
                u32x4s g[3], x[K][3];
            };
            auto prepare = [&](int hf, Ops& o) {
                const float gs[8] = {g0[hf][0], g0[hf][1], g0[hf][2], g0[hf][3], g1[hf][0], g1[hf][1], g1[hf][2], g1[hf][3]};
                const float xs[10] = {xl[hf], x0[hf][0], x0[hf][1], x0[hf][2], x0[hf][3],
                                      x1[hf][0], x1[hf][1], x1[hf][2], x1[hf][3], xh[hf]};
                if (do_bias) bsum += ((gs[0] + gs[1]) + (gs[2] + gs[3])) + ((gs[4] + gs[5]) + (gs[6] + gs[7]));
                if constexpr (SPLIT == 1) {  // bf16 compute mode: the operands rounded to bf16, one product
                    unsigned qw[5];
#pragma unroll
                    for (int i = 0; i < 4; ++i) o.g[0][i] = pack_bf16s(gs[2 * i], gs[2 * i + 1]);
#pragma unroll
                    for (int i = 0; i < 5; ++i) qw[i] = pack_bf16s(xs[2 * i], xs[2 * i + 1]);
#pragma unroll
                    for (int j = 0; j < K; ++j)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            o.x[j][0][i] = (j & 1) ? __builtin_amdgcn_alignbit(qw[i + (j >> 1) + 1], qw[i + (j >> 1)], 16)
                                                   : qw[i + (j >> 1)];
                    return;
                }
                // pieces two values per word (the packed conversion's result is the operand word)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned w1_, w2_, w3_;
                    split3s_pair(gs[2 * i], gs[2 * i + 1], w1_, w2_, w3_);
                    o.g[0][i] = w1_;
                    o.g[1][i] = w2_;
                    o.g[2][i] = w3_;
                }
                unsigned qw[3][5];  // [piece][word]: samples 2 w, 2 w + 1 of the ten
#pragma unroll
                for (int i = 0; i < 5; ++i) split3s_pair(xs[2 * i], xs[2 * i + 1], qw[0][i], qw[1][i], qw[2][i]);
                // tap j reads samples j .. j + 7: whole words for even j, words re-cut across their borders for odd j
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            o.x[j][k][i] = (j & 1) ? __builtin_amdgcn_alignbit(qw[k][i + (j >> 1) + 1], qw[k][i + (j >> 1)], 16)
                                                   : qw[k][i + (j >> 1)];
            };
            auto multiply = [&](const Ops& o) {
                const bf16x8s G1 = __builtin_bit_cast(bf16x8s, o.g[0]);
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const bf16x8s X1 = __builtin_bit_cast(bf16x8s, o.x[j][0]);
                    if constexpr (SPLIT == 1) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, X1, acc[j], 0, 0, 0);
                    } else {
                        const bf16x8s G2 = __builtin_bit_cast(bf16x8s, o.g[1]), G3 = __builtin_bit_cast(bf16x8s, o.g[2]);
                        const bf16x8s X2 = __builtin_bit_cast(bf16x8s, o.x[j][1]), X3 = __builtin_bit_cast(bf16x8s, o.x[j][2]);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G3, X1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, X3, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G2, X2, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G2, X1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, X2, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G1, X1, acc[j], 0, 0, 0);
                    }
                }
            };
            Ops oa, ob;
            prepare(0, oa);
            __builtin_amdgcn_sched_barrier(0);
            prepare(1, ob);
            multiply(oa);
            if constexpr (SPLIT == 3) {
#pragma unroll
                for (int g = 0; g < 6 * K; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            multiply(ob);
          }
          return;
        }
        const float* Gp = Gs + (cot * 32 + l31) * LD + 4 + 4 * h;
        const float* Xp = Xs + (cit * 32 + l31) * LD + 4 + 4 * h;
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(Gp + n * SUB + 8 * g);
                const f32x4 xv = *reinterpret_cast<const f32x4*>(Xp + n * SUB + 8 * g);
                float e[6];
                e[1] = xv[0];
                e[2] = xv[1];
                e[3] = xv[2];
                e[4] = xv[3];
                if (K > 1) {
                    e[0] = Xp[n * SUB + 8 * g - 1];
                    e[5] = Xp[n * SUB + 8 * g + 4];
                }
                if (do_bias) bsum += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], e[q + j + 1 - PAD], acc[j], 0, 0, 0);
            }
        }
    };

    if (FULL) {
        // this thread's items: 16-byte item (row = r0 + 16 i, subtile n, floats 4q..4q+3) of both tiles,
        // and one halo sample (row hr, subtile hn, left/right) of the x tile
        constexpr int RSTEP = NTHR / 16;   // rows covered by one pass of the workgroup's threads
        constexpr int NI = 64 / RSTEP;     // 16-byte items per thread and tile
        const int q = tid & 7, n = (tid >> 3) & 1, r0 = tid >> 4;
        const int hs = tid & 1, hn = (tid >> 1) & 1, hr = tid >> 2;  // (halo items: the first 256 threads)
        typename ActT<GBF>::raw4 gv[NI], mv[NI];
        typename ActT<XBF>::raw4 xv[NI];
        typename ActT<XBF>::raw1 hv = 0;
        float md[NI];
        bool ok = false, hok = false;

        auto issue = [&](int it) {  // loads only; consumers live in commit()
            // wave-uniform decode of the two subtiles of this iteration
            int bb[2], tt[2];
            bool vv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int sub = it * NTW + m;
                vv[m] = sub < a.NS;
                bb[m] = vv[m] ? sub / a.ST : 0;
                tt[m] = vv[m] ? (sub - bb[m] * a.ST) * 32 : 0;
            }
            const int b = n ? bb[1] : bb[0], t0 = n ? tt[1] : tt[0];
            ok = n ? vv[1] : vv[0];
            const GT* gp = G0 + ((long long)b * a.Cout + co0 + r0) * a.T + t0 + 4 * q;
            const XT* xp = X0 + ((long long)b * a.Cin + ci0 + r0) * a.T + t0 + 4 * q;
            const GT* mp = a.mask_y ? M0 + ((long long)b * a.Cout + co0 + r0) * a.T + t0 + 4 * q : G0;
            const float* dp = a.mask_drop ? a.mask_drop + b * a.Cout + co0 + r0 : a.ws;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const long long o = (long long)(RSTEP * i) * a.T;
                gv[i] = ld_raw4(gp + o);
                if constexpr (RAGX) {
                    const int row = ci0 + r0 + RSTEP * i < a.Cin ? ci0 + r0 + RSTEP * i : a.Cin - 1;
                    xv[i] = ld_raw4(X0 + ((long long)b * a.Cin + row) * a.T + t0 + 4 * q);
                } else {
                    xv[i] = ld_raw4(xp + o);
                }
                if (a.mask_y) {  // wave-uniform
                    mv[i] = ld_raw4(mp + o);
                    md[i] = dp[a.mask_drop ? RSTEP * i : 0];
                }
            }
            if (K > 1) {
                const int hb = hn ? bb[1] : bb[0], ht0 = hn ? tt[1] : tt[0];
                const int t = ht0 + (hs ? 32 : -1);
                hok = (hn ? vv[1] : vv[0]) && t >= 0 && t < a.T && hr < 64;
                const int hrow = (RAGX && ci0 + hr >= a.Cin) ? a.Cin - 1 : ci0 + hr;
                hv = ld_raw1(X0 + (hok ? ((long long)hb * a.Cin + hrow) * a.T + t : 0));
            }
        };
        auto commit = [&](float* Gs, float* Xs) {
            if constexpr (BFN) {
                unsigned short* Gh = reinterpret_cast<unsigned short*>(Gs);
                unsigned short* Xh = reinterpret_cast<unsigned short*>(Xs);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    u32x2 gq = gv[i], xq = xv[i];
                    if (a.mask_y) {  // the masked gradient, rounded to bfloat16 once (it is a matrix operand)
                        f32x4 g = widen(gv[i]);
                        const float d = a.mask_drop ? md[i] : 1.f;
                        const f32x4 my = widen(mv[i]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] *= my[e] > 0.f ? d : d * a.mask_slope;
                        gq[0] = pack_bf16s(g[0], g[1]);
                        gq[1] = pack_bf16s(g[2], g[3]);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        gq[e] = ok ? gq[e] : 0u;
                        xq[e] = ok ? xq[e] : 0u;
                    }
                    const int o = (r0 + RSTEP * i) * LDH + n * 40 + 8 + 4 * q;
                    *reinterpret_cast<u32x2*>(Gh + o) = gq;
                    *reinterpret_cast<u32x2*>(Xh + o) = xq;
                }
                if (hr < 64) Xh[hr * LDH + hn * 40 + (hs ? 40 : 7)] = hok ? hv : (unsigned short)0;
                return;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x4 g = widen(gv[i]), x = widen(xv[i]);
                if (a.mask_y) {
                    const float d = a.mask_drop ? md[i] : 1.f;
                    const f32x4 my = widen(mv[i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] *= my[e] > 0.f ? d : d * a.mask_slope;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    g[e] = ok ? g[e] : 0.f;
                    x[e] = ok ? x[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(Gs + (r0 + RSTEP * i) * LD + n * SUB + 4 + 4 * q) = g;
                *reinterpret_cast<f32x4*>(Xs + (r0 + RSTEP * i) * LD + n * SUB + 4 + 4 * q) = x;
            }
            if (K > 1 && hr < 64) Xs[hr * LD + hn * SUB + (hs ? 36 : 3)] = hok ? widen(hv) : 0.f;
        };

        EEGGAN_STAMP(0, 6);
        // (floats per buffer and per tile; the bfloat16-native form counts its halfwords in floats)
        constexpr int BUFF = BFN ? BUFH / 2 : BUF, TILEF = BFN ? 64 * LDH / 2 : 64 * LD;
        if (it0 < it1) {
            issue(it0);
            commit(smem, smem + TILEF);
        }
        __syncthreads();
        EEGGAN_STAMP(0, 7);
        for (int it = it0; it < it1; ++it) {
            const bool more = it + 1 < it1;
            float* cur = smem + (W4 ? 0 : ((it - it0) & 1) * BUFF);
            float* nxt = smem + (W4 ? 0 : ((it - it0 + 1) & 1) * BUFF);
            EEGGAN_STAMP(it - it0, 0);
            if (more) issue(it + 1);
            __builtin_amdgcn_sched_barrier(0);
            EEGGAN_STAMP(it - it0, 1);
            mfma_block(cur, cur + TILEF);
            EEGGAN_STAMP(it - it0, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (W4) __syncthreads();  // (one buffer: every wave has read its operands before the tiles are replaced)
            if (more) commit(nxt, nxt + TILEF);
            EEGGAN_STAMP(it - it0, 3);
            __syncthreads();
            EEGGAN_STAMP(it - it0, 4);
        }
    } else {
        float* Gs = smem;            // [64][LD]
        float* Xs = smem + 64 * LD;  // [64][LD]
        XStage<64, NTW, LD, false> gs;
        XStage<64, NTW, LD, (K > 1)> xs;
        const XT* const XN = nullptr;
        if (it0 < it1) {
            gs.load(G0, M0, a.mask_drop, a.mask_slope, a.Cout, a.T, a.ST, a.NS, it0 * NTW, co0, a.vec);
            xs.load(X0, XN, nullptr, 0.f, a.Cin, a.T, a.ST, a.NS, it0 * NTW, ci0, a.vec);
        }
        for (int it = it0; it < it1; ++it) {
            gs.store(Gs);
            xs.store(Xs);
            __syncthreads();
            if (it + 1 < it1) {
                gs.load(G0, M0, a.mask_drop, a.mask_slope, a.Cout, a.T, a.ST, a.NS, (it + 1) * NTW, co0, a.vec);
                xs.load(X0, XN, nullptr, 0.f, a.Cin, a.T, a.ST, a.NS, (it + 1) * NTW, ci0, a.vec);
            }
            mfma_block(Gs, Xs);
            __syncthreads();
        }
    }

    if (SPLIT && !W4) {
        // waves 4..7 hand their sums to waves 0..3 (the staging buffers are free: every path above ends in a barrier
        // or never used them)
        float* red = smem + ((wave & 3) * (K * 16 + 1)) * 64 + lane;
        if (part == 1) {
#pragma unroll
            for (int j = 0; j < K; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 64] = acc[j][r];
            red[K * 16 * 64] = bsum;
        }
        __syncthreads();
        if (part == 1) return;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] += red[(j * 16 + r) * 64];
        bsum += red[K * 16 * 64];
    }
    const long long E = (long long)a.Cout * a.Cin * K;
    float* slab = a.ws + (long long)split * E;
    const int ci = ci0 + cit * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < a.Cout && ci < a.Cin) {
#pragma unroll
            for (int j = 0; j < K; ++j) slab[((long long)co * a.Cin + ci) * K + j] = acc[j][r];
        }
    }
    if (do_bias) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + cot * 32 + l31;
        if (h == 0 && co < a.Cout) a.ws_b[(long long)split * a.Cout + co] = bsum;
    }
}

// Weight gradient of a 1x1 layer with a thin side (at most 8 channels: from_rgb, to_rgb) at large batches, on the
// vector ALUs.  The matrix-core kernel pads the thin side to 64 channels and runs its generic staging; here the work
// is what it is -- one pass over the wide tensor, a handful of multiply-adds per element -- and the kernel streams:
// a workgroup owns a run of tiles (one sample, 128 time steps, 64 wide channels), stages the wide tile (with the
// fused backward mask) and the thin tile in LDS with full-line loads, the next tile's loads in flight meanwhile, and
// thread (channel c, quarter q) keeps the W sums of its wide row over its 32 time steps of every tile in registers.
//   THIN_X: x is the thin tensor (from_rgb: gw[co][w] = sum gy[co] x[w]); else gy is (to_rgb: gw[w][ci] = sum gy[w] x[ci]).
// Partial sums per split land in the job's slabs like the matrix-core kernels' (same reduction afterwards).
constexpr int WT_TILE = 128;              // time steps per tile
constexpr int WT_LD = WT_TILE + 4;        // LDS row stride of the wide tile in floats ((WT_LD / 4) odd)
template <bool THIN_X, bool XBF, bool GBF>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(WgradArgs a) {
    typedef typename ActT<XBF>::T XT;
    typedef typename ActT<GBF>::T GT;
    typedef typename std::conditional<THIN_X, GT, XT>::type WT;  // storage type of the wide tensor
    typedef typename std::conditional<THIN_X, XT, GT>::type NT_;  // ... of the thin one
    constexpr bool WBF = THIN_X ? GBF : XBF;
    constexpr int EPV = WBF ? 8 : 4;                    // elements per 16-byte vector of the wide tensor
    constexpr int WVEC = 64 * WT_TILE / EPV / 256;      // 16-byte vectors of the wide tile per thread (4 | 8)
    constexpr int VROW = WT_TILE / EPV;                 // vectors per wide row
    __shared__ __attribute__((aligned(16))) float wide_s[64 * WT_LD];
    __shared__ __attribute__((aligned(16))) float thin_s[8 * WT_TILE];
    const int tid = threadIdx.x;
    const int Cw = THIN_X ? a.Cout : a.Cin, Cn = THIN_X ? a.Cin : a.Cout;  // wide / thin channel counts
    const WT* const W0 = reinterpret_cast<const WT*>(THIN_X ? a.gy : a.x);
    const NT_* const N0 = reinterpret_cast<const NT_*>(THIN_X ? a.x : a.gy);
    const GT* const M0 = reinterpret_cast<const GT*>(a.mask_y);
    const bool masked = THIN_X && a.mask_y != nullptr;  // (the mask belongs to gy; a thin gy carries none here)
    const int c0 = blockIdx.y * 64;
    const int tiles_per_sample = a.T / WT_TILE;
    const int ntiles = a.B * tiles_per_sample;
    const int split = blockIdx.x;
    const int t_begin = split * a.iters, t_end = min(t_begin + a.iters, ntiles);

    typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
    u32x4w wv[WVEC], mv[WVEC];
    float md[WVEC];
    float nv[4];  // the thin tile: 8 rows x 128 steps = 1024 values, four per thread
    auto issue = [&](int tile) {
        const int b = tile / tiles_per_sample, t0 = (tile - b * tiles_per_sample) * WT_TILE;
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
            const int v = tid + 256 * i, row = v / VROW, q = v - row * VROW;
            const bool ok = c0 + row < Cw;
            const long long off = ((long long)b * Cw + (ok ? c0 + row : 0)) * a.T + t0 + q * EPV;
            wv[i] = *reinterpret_cast<const u32x4w*>(W0 + off);
            if (masked) {
                mv[i] = *reinterpret_cast<const u32x4w*>(M0 + off);
                md[i] = a.mask_drop ? a.mask_drop[b * Cw + (ok ? c0 + row : 0)] : 1.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e / WT_TILE, t = e - row * WT_TILE;
            nv[i] = row < Cn ? ld1(N0 + ((long long)b * Cn + row) * a.T + t0 + t) : 0.f;
        }
    };
    // one 16-byte vector of the wide tensor as EPV floats (bfloat16: widened)
    auto widen_vec = [](const u32x4w& u, float (&out)[EPV]) {
        if constexpr (WBF) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out[2 * k] = __builtin_bit_cast(float, u[k] << 16);
                out[2 * k + 1] = __builtin_bit_cast(float, u[k] & 0xffff0000u);
            }
        } else {
            const f32x4 f = __builtin_bit_cast(f32x4, u);
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = f[k];
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < WVEC; ++i) {
            const int v = tid + 256 * i, row = v / VROW, q = v - row * VROW;
            const bool ok = c0 + row < Cw;
            float vals[EPV], ms[EPV];
            widen_vec(wv[i], vals);
            if (masked) widen_vec(mv[i], ms);
#pragma unroll
            for (int e = 0; e < EPV; ++e) {
                float g = vals[e];
                if (masked) g *= ms[e] > 0.f ? md[i] : md[i] * a.mask_slope;
                vals[e] = ok ? g : 0.f;
            }
#pragma unroll
            for (int k = 0; k < EPV / 4; ++k) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = vals[4 * k + e];
                *reinterpret_cast<f32x4*>(wide_s + row * WT_LD + q * EPV + 4 * k) = o;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) thin_s[tid + 256 * i] = nv[i];
    };

    const int c = tid & 63, qt = tid >> 6;  // wide row and quarter of the tile's time steps
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    if (t_begin < t_end) issue(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // the previous tile has been read
        commit();
        __syncthreads();
        if (tile + 1 < t_end) issue(tile + 1);
        const float* wr = wide_s + c * WT_LD + 32 * qt;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(wr + 4 * k);
            if (THIN_X) bsum += (g[0] + g[1]) + (g[2] + g[3]);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                if (w < Cn) {  // (uniform)
                    const f32x4 n = *reinterpret_cast<const f32x4*>(thin_s + w * WT_TILE + 32 * qt + 4 * k);
                    acc[w] += (g[0] * n[0] + g[1] * n[1]) + (g[2] * n[2] + g[3] * n[3]);
                }
            }
        }
        if (!THIN_X && c < Cn) {  // bias gradient of a thin gy: row c of the thin tile, this quarter
            const float* nr = thin_s + c * WT_TILE + 32 * qt;
#pragma unroll
            for (int k = 0; k < 32; ++k) bsum += nr[k];
        }
    }
    // the four quarters' sums meet in LDS (in a fixed order), then one thread per wide row writes the slab
    __syncthreads();
    float* red = wide_s;  // [4][64][9]
#pragma unroll
    for (int w = 0; w < 8; ++w) red[(qt * 64 + c) * 9 + w] = acc[w];
    red[(qt * 64 + c) * 9 + 8] = bsum;
    __syncthreads();
    if (qt != 0) return;
    float tot[9];
#pragma unroll
    for (int w = 0; w < 9; ++w) tot[w] = (red[c * 9 + w] + red[(64 + c) * 9 + w]) + (red[(128 + c) * 9 + w] + red[(192 + c) * 9 + w]);
    const long long E = (long long)a.Cout * a.Cin;
    float* slab = a.ws + (long long)split * E;
    if (c0 + c < Cw) {
        for (int w = 0; w < Cn; ++w) {
            if (THIN_X) slab[(long long)(c0 + c) * a.Cin + w] = tot[w];   // [co = wide][ci = thin]
            else slab[(long long)w * a.Cin + c0 + c] = tot[w];            // [co = thin][ci = wide]
        }
        if (THIN_X && a.ws_b) a.ws_b[(long long)split * a.Cout + c0 + c] = tot[8];
    }
    if (!THIN_X && a.ws_b && blockIdx.y == 0 && c < Cn) a.ws_b[(long long)split * a.Cout + c] = tot[8];
}

template <int K, bool FULL, int SPLIT = 0, bool XBF = false, bool GBF = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void conv_wgrad_kernel(WgradArgs a) {
    wgrad_body<K, FULL, SPLIT, XBF, GBF>(a, blockIdx.x, blockIdx.y);
}

// Several independent weight gradients in one launch: the workgroups of the grid are dealt out to the
// jobs in proportion to their work, so that a group of layers shares ONE chip-wide round of
// workgroups.  Each job then has fewer, longer splits: a fraction of the partial slabs to write and
// to sum, and one launch latency for the group.
constexpr int WG_MAXJ = 16;
struct WgradPack {
    WgradArgs j[WG_MAXJ];
    int first[WG_MAXJ + 1];  // first workgroup of job k; first[n] = grid size
    int n;
};

// (the bfloat16-native form asks for registers that let two workgroups of eight waves share a CU: four waves per SIMD)
template <int K, bool FULL, int SPLIT = 0, bool XBF = false, bool GBF = false, bool W4 = false, bool RAGX = false>
__global__ __launch_bounds__((SPLIT && !W4) ? 512 : 256, (FULL && SPLIT == 1 && XBF && GBF) ? 4 : (W4 ? WGRAD_W4_PER_CU : 1))
void conv_wgrad_multi_kernel(WgradPack pk) {
    const int blk = blockIdx.x;
    int k = 0;
    while (k + 1 < pk.n && blk >= pk.first[k + 1]) ++k;
    const WgradArgs a = pk.j[k];
    const int local = blk - pk.first[k];
    const int region = local / a.nsplit;
    wgrad_body<K, FULL, SPLIT, XBF, GBF, W4, RAGX>(a, local - region * a.nsplit, region);
}

// Sum of the partial slabs: 64 consecutive elements x 4 split groups per workgroup, eight
// independent loads in flight per thread, double accumulation (the split count then does not
// enter the rounding error), fixed order => run-to-run deterministic.  One launch covers the
// weight slabs (blocks [0, nb_w)) and, behind them, the bias slabs.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int nsplit, long long E,
                                                           float scale, float* __restrict__ out, int nb_w,
                                                           const float* __restrict__ ws_b, long long Eb,
                                                           float* __restrict__ out_b, float beta, float beta_b) {
    __shared__ double part[4][64];
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    int blk = blockIdx.x;
    if (blk >= nb_w) {  // bias part (uniform per workgroup)
        blk -= nb_w;
        ws = ws_b;
        E = Eb;
        out = out_b;
        scale = 1.f;
        beta = beta_b;
    }
    const long long i = (long long)blk * 64 + el;
    double acc = 0.0;
    if (i < E) {
        int k = sg;
        for (; k + 28 < nsplit; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(long long)(k + 4 * u) * E + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; k < nsplit; k += 4) acc += (double)ws[(long long)k * E + i];
    }
    part[sg][el] = acc;
    __syncthreads();
    if (sg == 0 && i < E) {
        const float v = (float)(((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) * (double)scale);
        out[i] = beta != 0.f ? beta * out[i] + v : v;  // beta == 0: out may hold anything (never read)
    }
}

struct WgradPlan {
    int nco, nci, regions, nsplit, iters;
    size_t slab_elems, bytes;
};

// ``per_cu``: workgroups the kernel variant places on one CU (2 for the bfloat16-native form, WGRAD_W4_PER_CU for the
// four-wave form).  The workspace is always sized for the largest, so that one size serves every variant of a layer.
// (comparison switch: EEGGAN_WGRAD_THIN=0 keeps the 1x1 layers with a thin side on the matrix-core kernel)
inline bool wgrad_thin_enabled() {
    static const bool on = [] {
        const char* e = getenv("EEGGAN_WGRAD_THIN");
        return !(e && e[0] == '0');
    }();
    return on;
}

// (comparison switch: EEGGAN_WGRAD_W4=0 keeps the eight-wave form of the split weight gradient)
inline bool wgrad_four_wave() {
    static const bool on = [] {
        const char* e = getenv("EEGGAN_WGRAD_W4");
        return !(e && e[0] == '0');
    }();
    return on;
}

WgradPlan plan_wgrad(int B, int Cin, int Cout, int T, int K, int per_cu = 1) {
    WgradPlan p;
    p.nco = (Cout + 63) / 64;
    p.nci = (Cin + 63) / 64;
    p.regions = p.nco * p.nci;
    const int ST = (T + 31) / 32;
    const int pairs = (B * ST + 1) / 2;
    auto splits = [&](int wgs, int* iters) {
        int want = (wgs + p.regions - 1) / p.regions;  // ~wgs workgroups over the whole grid
        if (want > pairs) want = pairs;
        if (want < 1) want = 1;
        *iters = (pairs + want - 1) / want;
        return (pairs + *iters - 1) / *iters;
    };
    p.nsplit = splits(256 * per_cu, &p.iters);
    int unused;
    const int most = splits(256 * WGRAD_W4_PER_CU, &unused);
    p.slab_elems = (size_t)Cout * Cin * K;
    p.bytes = ((size_t)(most > p.nsplit ? most : p.nsplit) * (p.slab_elems + Cout) * sizeof(float) + 255) & ~(size_t)255;
    return p;
}

int check_common(int B, int Cin, int Cout, int T, int K) {
    EEGGAN_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && T > 0, "conv1d: non-positive dimension B=%d Cin=%d Cout=%d T=%d", B,
                   Cin, Cout, T);
    EEGGAN_REQUIRE(K == 1 || K == 3, "conv1d: kernel size %d not supported (1 or 3)", K);
    EEGGAN_REQUIRE((long long)B * ((T + 31) / 32) < (1ll << 30), "conv1d: B*T too large");
    EEGGAN_REQUIRE((long long)B * (Cin > Cout ? Cin : Cout) * T < (1ll << 40), "conv1d: tensor too large");
    return 0;
}

}  // namespace
}  // namespace eeggan

using namespace eeggan;

extern "C" int eeggan_set_matrix_precision(int pieces) {
    EEGGAN_REQUIRE(pieces == 0 || pieces == 1 || pieces == 3, "set_matrix_precision: %d is not one of 0, 1, 3", pieces);
    eeggan::g_pieces.store(pieces, std::memory_order_relaxed);
    return 0;
}
extern "C" int eeggan_get_matrix_precision(void) { return eeggan::product_pieces(); }

extern "C" int eeggan_conv1d(const float* x, const float* w, const float* bias, const float* drop, float* y, int B,
                             int Cin, int Cout, int T, int K, float in_scale, int act, float act_param,
                             const float* act_aux, const float* mask_y, const float* mask_drop, float mask_slope,
                             void* stream) {
    return eeggan_conv1d_t(x, w, bias, drop, y, B, Cin, Cout, T, K, in_scale, act, act_param, act_aux, mask_y, mask_drop,
                           mask_slope, 0, 0, stream);
}

extern "C" int eeggan_conv1d_t(const void* xv, const float* w, const float* bias, const float* drop, void* yv, int B,
                               int Cin, int Cout, int T, int K, float in_scale, int act, float act_param,
                               const void* act_auxv, const void* mask_yv, const float* mask_drop, float mask_slope,
                               int x_bf16, int y_bf16, void* stream) {
    const float* x = static_cast<const float*>(xv);
    float* y = static_cast<float*>(yv);
    const float* act_aux = static_cast<const float*>(act_auxv);
    const float* mask_y = static_cast<const float*>(mask_yv);
    if (check_common(B, Cin, Cout, T, K)) return 1;
    EEGGAN_REQUIRE(!(x_bf16 || y_bf16) || T % 4 == 0, "conv1d: bfloat16 activations need T %% 4 == 0");
    EEGGAN_REQUIRE(x && w && y, "conv1d: null x/w/y pointer");
    EEGGAN_REQUIRE(act >= 0 && act <= 3, "conv1d: unknown activation code %d", act);
    EEGGAN_REQUIRE(act != EEGGAN_ACT_MASK || act_aux, "conv1d: EEGGAN_ACT_MASK needs act_aux");
    EEGGAN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                       (!mask_y || (reinterpret_cast<uintptr_t>(mask_y) & 15) == 0),
                   "conv1d: x, w and mask_y must be 16-byte aligned");
    // with T % 4 == 0 the output and the mask of the output are moved in 16-byte pieces as well
    EEGGAN_REQUIRE(T % 4 != 0 || ((reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
                                  (!act_aux || (reinterpret_cast<uintptr_t>(act_aux) & 15) == 0)),
                   "conv1d: y and act_aux must be 16-byte aligned when T is a multiple of 4");
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.drop = drop; a.y = y;
    a.mask_y = mask_y; a.mask_drop = mask_y ? mask_drop : nullptr; a.mask_slope = mask_slope;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T;
    a.ST = (T + 31) / 32;
    a.NS = B * a.ST;
    a.in_scale = in_scale; a.act = act; a.act_param = act_param; a.act_aux = act_aux;
    a.vec = (T % 4 == 0) ? 1 : 0;
    a.wpb = 0;
    a.pn_d = nullptr;
    a.pn_eps = 0.f;
    a.x_bf16 = x_bf16 ? 1 : 0;
    a.y_bf16 = y_bf16 ? 1 : 0;
    hipStream_t st = as_stream(stream);
    return K == 3 ? dispatch_fwd<3>(a, st) : dispatch_fwd<1>(a, st);
}

extern "C" int eeggan_conv1d_pixelnorm_supported(int B, int Cin, int Cout, int T, int K) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0) return 0;
    ConvArgs a{};
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T;
    a.ST = (T + 31) / 32;
    a.NS = B * a.ST;
    return conv_ws_covers(a, K) ? 1 : 0;
}

extern "C" int eeggan_conv1d_pixelnorm(const float* x, const float* w, const float* bias, const float* drop, float* y,
                                       float* d, int B, int Cin, int Cout, int T, int K, float in_scale, float slope,
                                       float eps, void* stream) {
    return eeggan_conv1d_pixelnorm_t(x, w, bias, drop, y, d, B, Cin, Cout, T, K, in_scale, slope, eps, 0, stream);
}

extern "C" int eeggan_conv1d_pixelnorm_t(const void* xv, const float* w, const float* bias, const float* drop, void* yv,
                                         float* d, int B, int Cin, int Cout, int T, int K, float in_scale, float slope,
                                         float eps, int bf16, void* stream) {
    const float* x = static_cast<const float*>(xv);
    float* y = static_cast<float*>(yv);
    if (check_common(B, Cin, Cout, T, K)) return 1;
    EEGGAN_REQUIRE(x && w && y && d, "conv1d_pixelnorm: null x/w/y/d pointer");
    EEGGAN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0,
                   "conv1d_pixelnorm: x, w, y and d must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.drop = drop; a.y = y;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T;
    a.ST = (T + 31) / 32;
    a.NS = B * a.ST;
    a.in_scale = in_scale; a.act = ACT_LRELU_PN; a.act_param = slope;
    a.vec = (T % 4 == 0) ? 1 : 0;
    a.pn_d = d; a.pn_eps = eps;
    a.x_bf16 = a.y_bf16 = bf16 ? 1 : 0;
    EEGGAN_REQUIRE(!bf16 || (Cin == 64 && Cout == 64 && T % 32 == 0 && product_pieces() == 1),
                   "conv1d_pixelnorm: bfloat16 activations need a 64 -> 64 layer with T %% 32 == 0 and the bf16 matrix precision");
    EEGGAN_REQUIRE(conv_ws_covers(a, K), "conv1d_pixelnorm: shape B=%d Cin=%d Cout=%d T=%d K=%d is not covered "
                   "(see eeggan_conv1d_pixelnorm_supported)", B, Cin, Cout, T, K);
    return launch_conv_ws(a, K, as_stream(stream));
}

extern "C" int eeggan_conv1d_flip_weights(const float* w, float* wt, int Cout, int Cin, int K, void* stream) {
    EEGGAN_REQUIRE(w && wt && Cout > 0 && Cin > 0 && K > 0, "conv1d_flip_weights: bad argument");
    const long long n = (long long)Cout * Cin * K;
    hipLaunchKernelGGL(flip_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), w, wt,
                       Cout, Cin, K);
    return check_launch("conv1d_flip_weights");
}

extern "C" int eeggan_conv1d_flip_weights_multi(const eeggan_flip_desc* table, int ntensors, void* stream) {
    EEGGAN_REQUIRE(table && ntensors > 0, "conv1d_flip_weights_multi: bad table");
    for (int first = 0; first < ntensors; first += FLIP_MAXT) {
        FlipPack pk;
        pk.n = ntensors - first < FLIP_MAXT ? ntensors - first : FLIP_MAXT;
        long long blocks = 0;
        for (int i = 0; i < pk.n; ++i) {
            pk.t[i] = table[first + i];
            EEGGAN_REQUIRE(pk.t[i].w && pk.t[i].wt && pk.t[i].Cout > 0 && pk.t[i].Cin > 0 && pk.t[i].K > 0,
                           "conv1d_flip_weights_multi: bad descriptor %d", first + i);
            blocks += ((long long)pk.t[i].Cout * pk.t[i].Cin * pk.t[i].K + 255) / 256;
        }
        EEGGAN_REQUIRE(blocks < (1ll << 31), "conv1d_flip_weights_multi: too many elements");
        hipLaunchKernelGGL(flip_weights_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pk);
        if (check_launch("conv1d_flip_weights_multi")) return 1;
    }
    return 0;
}

extern "C" size_t eeggan_conv1d_wgrad_workspace(int B, int Cin, int Cout, int T, int K) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || (K != 1 && K != 3)) return 0;
    return plan_wgrad(B, Cin, Cout, T, K).bytes;
}

namespace eeggan {
namespace {

// Launches the partial-slab kernel; the slabs (and bias slabs when want_bias) land in `workspace`.
int launch_wgrad_partial(const char* what, const float* x, const float* gy, bool want_bias, int B, int Cin, int Cout,
                         int T, int K, const float* mask_y, const float* mask_drop, float mask_slope,
                         void* workspace, size_t workspace_bytes, hipStream_t st, WgradPlan* plan, float** ws_b) {
    if (check_common(B, Cin, Cout, T, K)) return 1;
    EEGGAN_REQUIRE(x && gy, "%s: null x/gy pointer", what);
    const WgradPlan p = plan_wgrad(B, Cin, Cout, T, K);
    EEGGAN_REQUIRE(workspace && workspace_bytes >= p.bytes, "%s: workspace too small (%zu < %zu bytes)", what,
                   workspace_bytes, p.bytes);
    WgradArgs a;
    a.x = x; a.gy = gy; a.mask_y = mask_y; a.mask_drop = mask_y ? mask_drop : nullptr; a.mask_slope = mask_slope;
    a.ws = static_cast<float*>(workspace);
    a.ws_b = want_bias ? a.ws + (size_t)p.nsplit * p.slab_elems : nullptr;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.T = T;
    a.ST = (T + 31) / 32;
    a.NS = B * a.ST;
    a.nci = p.nci; a.nsplit = p.nsplit; a.iters = p.iters;
    a.vec = (T % 4 == 0) ? 1 : 0;
    a.x_bf16 = a.gy_bf16 = 0;
#ifdef EEGGAN_TRACE
    a.trace = g_wgrad_trace;
#endif
    constexpr size_t lds1 = (size_t)2 * 64 * (2 * SUB + 4) * sizeof(float);  // one buffer: gy tile + x tile
    const bool full = Cin % 64 == 0 && Cout % 64 == 0 && T % 32 == 0;
    dim3 grid(p.nsplit, p.regions);
    if (K == 3 && full) {
        // two buffers = 86 KB: opt in to > 64 KiB of dynamic LDS (once per device)
        static std::atomic<bool> ready_dev[MAX_DEVICES];
        std::atomic<bool>& ready = ready_dev[current_device()];
        if (!ready.load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<3, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<3, true, 3>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<3, true, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
            EEGGAN_REQUIRE(e == hipSuccess, "%s: cannot reserve LDS: %s", what, hipGetErrorString(e));
            ready.store(true, std::memory_order_release);
        }
        const int pieces = product_pieces();
        if (pieces == 3)
            hipLaunchKernelGGL((conv_wgrad_kernel<3, true, 3>), grid, dim3(512), 2 * lds1, st, a);
        else if (pieces == 1)
            hipLaunchKernelGGL((conv_wgrad_kernel<3, true, 1>), grid, dim3(512), 2 * lds1, st, a);
        else
            hipLaunchKernelGGL((conv_wgrad_kernel<3, true>), grid, dim3(256), 2 * lds1, st, a);
    } else if (K == 3) {
        hipLaunchKernelGGL((conv_wgrad_kernel<3, false>), grid, dim3(256), lds1, st, a);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<1, false>), grid, dim3(256), lds1, st, a);
    }
    if (check_launch(what)) return 1;
    *plan = p;
    *ws_b = a.ws_b;
    return 0;
}

// Many slab reductions in one launch (descriptor table by value); a workgroup finds its job by
// walking the table.  Jobs of one launch must write distinct outputs.
constexpr int RED_MAXJ = 48;
struct ReducePack {
    eeggan_reduce_job j[RED_MAXJ];
    int n;
};

// A job whose slabs are whole 16-byte items takes 256 elements per workgroup, four per thread: a wave's load is then
// 1 KB of one slab instead of 256 bytes (the slabs of a job lie E floats apart, so short pieces are scattered
// reads).  Same summation order per element either way.
__host__ __device__ inline bool reduce_vec(const eeggan_reduce_job& j) {
    return j.elems % 4 == 0 && ((uintptr_t)j.slabs & 15) == 0 && ((uintptr_t)j.out & 15) == 0;
}
__host__ __device__ inline long long reduce_blocks(const eeggan_reduce_job& j) {
    return reduce_vec(j) ? (j.elems + 255) / 256 : (j.elems + 63) / 64;
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReducePack pk) {
    __shared__ double part[4][256];
    long long blk = blockIdx.x;
    int k = 0;
    for (; k < pk.n; ++k) {
        const long long nb = reduce_blocks(pk.j[k]);
        if (blk < nb) break;
        blk -= nb;
    }
    if (k == pk.n) return;
    const eeggan_reduce_job job = pk.j[k];
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int nsplit = job.nsplit;
    const float* ws = job.slabs;
    if (reduce_vec(job)) {  // workgroup-uniform
        const long long i = blk * 256 + 4 * el, E = job.elems;
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        if (i < E) {
            int s = sg;
            for (; s + 28 < nsplit; s += 32) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(ws + (long long)(s + 4 * u) * E + i);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) a4[e] += (double)v[u][e];
            }
            if (s < nsplit) {
                // the last (in a grouped launch: the only) up to seven slabs of the wave, all in flight at once: one
                // load per iteration made this a chain of dependent round trips (the grouped weight gradient leaves
                // ~16 slabs per layer, i.e. four per wave, and none for the loop above).  Same order of additions.
                f32x4 v[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) {
                    const int su = s + 4 * u < nsplit ? s + 4 * u : s;
                    v[u] = *reinterpret_cast<const f32x4*>(ws + (long long)su * E + i);
                }
#pragma unroll
                for (int u = 0; u < 7; ++u)
                    if (s + 4 * u < nsplit) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) a4[e] += (double)v[u][e];
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[sg][4 * el + e] = a4[e];
        __syncthreads();
        if (sg == 0 && i < E) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * el + e;
                o[e] = (float)(((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) * (double)job.scale);
            }
            if (job.beta != 0.f) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(job.out + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = job.beta * old[e] + o[e];
            }
            *reinterpret_cast<f32x4*>(job.out + i) = o;
        }
        return;
    }
    const long long i = blk * 64 + el, E = job.elems;
    double acc = 0.0;
    if (i < E) {
        int s = sg;
        for (; s + 28 < nsplit; s += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(long long)(s + 4 * u) * E + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)v[u];
        }
        for (; s < nsplit; s += 4) acc += (double)ws[(long long)s * E + i];
    }
    part[sg][el] = acc;
    __syncthreads();
    if (sg == 0 && i < E) {
        const float v = (float)(((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) * (double)job.scale);
        job.out[i] = job.beta != 0.f ? job.beta * job.out[i] + v : v;
    }
}

}  // namespace
}  // namespace eeggan

extern "C" int eeggan_conv1d_wgrad(const float* x, const float* gy, float* gw, float* gb, int B, int Cin, int Cout,
                                   int T, int K, float in_scale, const float* mask_y, const float* mask_drop,
                                   float mask_slope, float beta_w, float beta_b, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    EEGGAN_REQUIRE(gw, "conv1d_wgrad: null gw pointer");
    hipStream_t st = as_stream(stream);
    WgradPlan p;
    float* ws_b;
    if (launch_wgrad_partial("conv1d_wgrad", x, gy, gb != nullptr, B, Cin, Cout, T, K, mask_y, mask_drop, mask_slope,
                             workspace, workspace_bytes, st, &p, &ws_b))
        return 1;
    const long long E = (long long)p.slab_elems;
    const int nb_w = (int)((E + 63) / 64), nb_b = gb ? (Cout + 63) / 64 : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(nb_w + nb_b)), dim3(256), 0, st,
                       static_cast<const float*>(workspace), p.nsplit, E, in_scale, gw, nb_w, ws_b, (long long)Cout, gb,
                       beta_w, beta_b);
    if (check_launch("conv1d_wgrad reduce")) return 1;
    return 0;
}

extern "C" int eeggan_conv1d_wgrad_partial(const float* x, const float* gy, int want_bias, int B, int Cin, int Cout,
                                           int T, int K, float in_scale, const float* mask_y, const float* mask_drop,
                                           float mask_slope, void* workspace, size_t workspace_bytes,
                                           eeggan_reduce_job* job_w, eeggan_reduce_job* job_b, void* stream) {
    EEGGAN_REQUIRE(job_w && (!want_bias || job_b), "conv1d_wgrad_partial: null job pointer");
    WgradPlan p;
    float* ws_b;
    if (launch_wgrad_partial("conv1d_wgrad_partial", x, gy, want_bias != 0, B, Cin, Cout, T, K, mask_y, mask_drop,
                             mask_slope, workspace, workspace_bytes, as_stream(stream), &p, &ws_b))
        return 1;
    job_w->slabs = static_cast<const float*>(workspace);
    job_w->out = nullptr;
    job_w->elems = (int64_t)p.slab_elems;
    job_w->nsplit = p.nsplit;
    job_w->scale = in_scale;
    job_w->beta = 0.f;
    if (want_bias) {
        job_b->slabs = ws_b;
        job_b->out = nullptr;
        job_b->elems = Cout;
        job_b->nsplit = p.nsplit;
        job_b->scale = 1.f;
        job_b->beta = 0.f;
    }
    return 0;
}

extern "C" int eeggan_conv1d_wgrad_partial_multi(const eeggan_wgrad_job* jobs, int njobs, eeggan_reduce_job* jobs_w,
                                                 eeggan_reduce_job* jobs_b, void* stream) {
    EEGGAN_REQUIRE(jobs && jobs_w && jobs_b && njobs > 0, "conv1d_wgrad_partial_multi: bad table");
    hipStream_t st = as_stream(stream);
    constexpr size_t lds1 = (size_t)2 * 64 * (2 * SUB + 4) * sizeof(float);
    // kernel variant per job: (0 = K 3 without predicates, 1 = K 3 generic, 2 = K 1) * 4 + storage types of x and gy
    // 1x1 layers with a thin side at large batches take the vector-ALU kernel, one launch per layer (below that
    // size the shared matrix-core launch wins: its cost is the launches it saves)
    auto thin = [](const eeggan_wgrad_job& j) {
        const int lo = j.Cin < j.Cout ? j.Cin : j.Cout;
        const bool thin_x = j.Cin <= j.Cout;
        const void* wide = thin_x ? j.gy : j.x;
        return j.K == 1 && lo <= 8 && j.T % WT_TILE == 0 && (long long)j.B * j.T >= (1ll << 18) && wgrad_thin_enabled() &&
               (reinterpret_cast<uintptr_t>(wide) & 15) == 0 && (!j.mask_y || (reinterpret_cast<uintptr_t>(j.mask_y) & 15) == 0) &&
               !(j.mask_y && !thin_x);
    };
    auto variant = [&](const eeggan_wgrad_job& j) {
        if (thin(j)) return -1;
        // (12: the four-wave split form with ragged input channels -- float32 tensors, fp32 or bf16 products)
        if (j.K == 3 && j.Cin % 64 != 0 && j.Cin > 64 && j.Cout % 64 == 0 && j.T % 32 == 0 && !j.x_bf16 && !j.gy_bf16 &&
            product_pieces() != 0 && wgrad_four_wave() && ragged_enabled())
            return 12;
        const bool full = j.Cin % 64 == 0 && j.Cout % 64 == 0 && j.T % 32 == 0;
        return (j.K == 3 ? (full ? 0 : 1) : 2) * 4 + (j.x_bf16 ? 2 : 0) + (j.gy_bf16 ? 1 : 0);
    };
    for (int i = 0; i < njobs; ++i) {
        const eeggan_wgrad_job& j = jobs[i];
        if (check_common(j.B, j.Cin, j.Cout, j.T, j.K)) return 1;
        EEGGAN_REQUIRE(j.x && j.gy, "conv1d_wgrad_partial_multi: job %d: null x/gy pointer", i);
        const WgradPlan p = plan_wgrad(j.B, j.Cin, j.Cout, j.T, j.K);
        EEGGAN_REQUIRE(j.workspace && j.workspace_bytes >= p.bytes,
                       "conv1d_wgrad_partial_multi: job %d: workspace too small (%zu < %zu bytes)", i,
                       j.workspace_bytes, p.bytes);
        const int v = variant(j);
        if (v < 0) continue;
        // bfloat16 activations: the predicate-free k = 3 layers with both tensors bfloat16 and bf16 products, and
        // the 1x1 layers in any combination
        EEGGAN_REQUIRE((v & 3) == 0 || v / 4 == 2 || (v == 3 && product_pieces() == 1),
                       "conv1d_wgrad_partial_multi: job %d: bfloat16 activations need K = 1, or K = 3 with Cin %% 64 == 0, "
                       "Cout %% 64 == 0, T %% 32 == 0, both tensors bfloat16 and the bf16 matrix precision", i);
    }
    for (int i = 0; i < njobs; ++i) {
        const eeggan_wgrad_job& j = jobs[i];
        if (!thin(j)) continue;
        const bool thin_x = j.Cin <= j.Cout;
        const int Cw = thin_x ? j.Cout : j.Cin;
        const int groups = (Cw + 63) / 64;
        const long long tiles = (long long)j.B * (j.T / WT_TILE);
        const size_t E = (size_t)j.Cout * j.Cin;
        long long nsplit = 4ll * num_cus() / groups;  // (38 KB of LDS and four waves per workgroup: four per CU)
        const long long room = (long long)(j.workspace_bytes / ((E + j.Cout) * sizeof(float)));
        if (nsplit > room) nsplit = room;
        if (nsplit > tiles) nsplit = tiles;
        EEGGAN_REQUIRE(nsplit >= 1, "conv1d_wgrad_partial_multi: job %d: workspace too small", i);
        const long long iters = (tiles + nsplit - 1) / nsplit;
        nsplit = (tiles + iters - 1) / iters;
        WgradArgs a{};
        a.x = j.x; a.gy = j.gy; a.mask_y = j.mask_y; a.mask_drop = j.mask_y ? j.mask_drop : nullptr;
        a.mask_slope = j.mask_slope;
        a.ws = static_cast<float*>(j.workspace);
        a.ws_b = j.want_bias ? a.ws + (size_t)nsplit * E : nullptr;
        a.B = j.B; a.Cin = j.Cin; a.Cout = j.Cout; a.T = j.T;
        a.nsplit = (int)nsplit; a.iters = (int)iters;
        a.x_bf16 = j.x_bf16; a.gy_bf16 = j.gy_bf16;
        const dim3 grid((unsigned)nsplit, (unsigned)groups);
        const int tv = (thin_x ? 4 : 0) + (j.x_bf16 ? 2 : 0) + (j.gy_bf16 ? 1 : 0);
        switch (tv) {
            case 0: hipLaunchKernelGGL((wgrad_thin_kernel<false, false, false>), grid, dim3(256), 0, st, a); break;
            case 1: hipLaunchKernelGGL((wgrad_thin_kernel<false, false, true>), grid, dim3(256), 0, st, a); break;
            case 2: hipLaunchKernelGGL((wgrad_thin_kernel<false, true, false>), grid, dim3(256), 0, st, a); break;
            case 3: hipLaunchKernelGGL((wgrad_thin_kernel<false, true, true>), grid, dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL((wgrad_thin_kernel<true, false, false>), grid, dim3(256), 0, st, a); break;
            case 5: hipLaunchKernelGGL((wgrad_thin_kernel<true, false, true>), grid, dim3(256), 0, st, a); break;
            case 6: hipLaunchKernelGGL((wgrad_thin_kernel<true, true, false>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((wgrad_thin_kernel<true, true, true>), grid, dim3(256), 0, st, a); break;
        }
        if (check_launch("conv1d_wgrad_partial_multi (thin 1x1 layer)")) return 1;
        eeggan_reduce_job& rw = jobs_w[i];
        rw.slabs = a.ws; rw.out = nullptr; rw.elems = (int64_t)E; rw.nsplit = (int)nsplit;
        rw.scale = j.in_scale; rw.beta = 0.f; rw.reserved = 0;
        eeggan_reduce_job& rb = jobs_b[i];
        rb.slabs = a.ws_b; rb.out = nullptr; rb.elems = j.want_bias ? j.Cout : 0; rb.nsplit = (int)nsplit;
        rb.scale = 1.f; rb.beta = 0.f; rb.reserved = 0;
    }
    for (int v = 0; v < 13; ++v) {
        int idx[WG_MAXJ];
        int n = 0;
        // (two workgroups per CU: the bfloat16-native form, and the four-wave form of the split fp32 products)
        const bool four_wave = (v == 0 && product_pieces() != 0 && wgrad_four_wave()) || v == 12;
        // (the generic and the 1x1 forms: four waves and one 43 KB buffer as well)
        const int per_cu = four_wave ? WGRAD_W4_PER_CU : ((v == 3 || v / 4 >= 1) ? 2 : 1);
        const int cus = per_cu * num_cus();
        auto flush = [&]() -> int {
            if (n == 0) return 0;
            WgradPlan plan[WG_MAXJ];
            long long pairs[WG_MAXJ], units = 0, longest = 1;
            for (int k = 0; k < n; ++k) {
                const eeggan_wgrad_job& j = jobs[idx[k]];
                plan[k] = plan_wgrad(j.B, j.Cin, j.Cout, j.T, j.K, per_cu);
                pairs[k] = ((long long)j.B * ((j.T + 31) / 32) + 1) / 2;
                units += pairs[k] * plan[k].regions;
                if (pairs[k] > longest) longest = pairs[k];
            }
            // the smallest common split length that keeps the whole group within one workgroup per CU
            // (never shorter than a job's own plan: its workspace is sized for that many slabs)
            long long iters = (units + cus - 1) / cus;
            if (iters < 1) iters = 1;
            for (;; ++iters) {
                long long wgs = 0;
                for (int k = 0; k < n; ++k) {
                    const long long it = iters > plan[k].iters ? iters : plan[k].iters;
                    wgs += ((pairs[k] + it - 1) / it) * plan[k].regions;
                }
                if (wgs <= cus || n == 1 || iters >= longest) break;
            }
            WgradPack pk;
            pk.n = n;
            int first = 0;
            for (int k = 0; k < n; ++k) {
                const eeggan_wgrad_job& j = jobs[idx[k]];
                const long long it = iters > plan[k].iters ? iters : plan[k].iters;
                const int nsplit = (int)((pairs[k] + it - 1) / it);
                WgradArgs& a = pk.j[k];
#ifdef EEGGAN_TRACE
                a.trace = nullptr;
#endif
                a.x = j.x; a.gy = j.gy; a.mask_y = j.mask_y; a.mask_drop = j.mask_y ? j.mask_drop : nullptr;
                a.mask_slope = j.mask_slope;
                a.ws = static_cast<float*>(j.workspace);
                a.ws_b = j.want_bias ? a.ws + (size_t)nsplit * plan[k].slab_elems : nullptr;
                a.B = j.B; a.Cin = j.Cin; a.Cout = j.Cout; a.T = j.T;
                a.ST = (j.T + 31) / 32;
                a.NS = j.B * a.ST;
                a.nci = plan[k].nci; a.nsplit = nsplit; a.iters = (int)it;
                a.vec = (j.T % 4 == 0) ? 1 : 0;
                a.x_bf16 = j.x_bf16; a.gy_bf16 = j.gy_bf16;
                pk.first[k] = first;
                first += nsplit * plan[k].regions;
                eeggan_reduce_job& rw = jobs_w[idx[k]];
                rw.slabs = a.ws; rw.out = nullptr; rw.elems = (int64_t)plan[k].slab_elems; rw.nsplit = nsplit;
                rw.scale = j.in_scale; rw.beta = 0.f; rw.reserved = 0;
                eeggan_reduce_job& rb = jobs_b[idx[k]];
                rb.slabs = a.ws_b; rb.out = nullptr; rb.elems = j.want_bias ? j.Cout : 0; rb.nsplit = nsplit;
                rb.scale = 1.f; rb.beta = 0.f; rb.reserved = 0;
            }
            for (int k = n; k <= WG_MAXJ; ++k) pk.first[k] = first;
            if (v == 12) {
                if (product_pieces() == 3)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 3, false, false, true, true>), dim3(first), dim3(256),
                                       lds1, st, pk);
                else
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 1, false, false, true, true>), dim3(first), dim3(256),
                                       lds1, st, pk);
            } else if (v / 4 == 0) {
                // two LDS buffers = 86 KB: opt in to > 64 KiB of dynamic LDS (once per device)
                static std::atomic<bool> ready_dev[MAX_DEVICES];
                std::atomic<bool>& ready = ready_dev[current_device()];
                if (!ready.load(std::memory_order_acquire)) {
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_multi_kernel<3, true>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
                    if (e == hipSuccess)
                        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_multi_kernel<3, true, 3>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
                    if (e == hipSuccess)
                        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_multi_kernel<3, true, 1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
                    if (e == hipSuccess)
                        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_multi_kernel<3, true, 1, true, true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds1));
                    EEGGAN_REQUIRE(e == hipSuccess, "conv1d_wgrad_partial_multi: cannot reserve LDS: %s",
                                   hipGetErrorString(e));
                    ready.store(true, std::memory_order_release);
                }
                const int pieces = product_pieces();
                if (v == 3)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 1, true, true>), dim3(first), dim3(512), WGRAD_BFN_LDS, st, pk);
                else if (four_wave && pieces == 3)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 3, false, false, true>), dim3(first), dim3(256), lds1, st, pk);
                else if (four_wave)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 1, false, false, true>), dim3(first), dim3(256), lds1, st, pk);
                else if (pieces == 3)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 3>), dim3(first), dim3(512), 2 * lds1, st, pk);
                else if (pieces == 1)
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true, 1>), dim3(first), dim3(512), 2 * lds1, st, pk);
                else
                    hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, true>), dim3(first), dim3(256), 2 * lds1, st, pk);
            } else if (v / 4 == 1) {
                hipLaunchKernelGGL((conv_wgrad_multi_kernel<3, false>), dim3(first), dim3(256), lds1, st, pk);
            } else if (v == 8) {
                hipLaunchKernelGGL((conv_wgrad_multi_kernel<1, false>), dim3(first), dim3(256), lds1, st, pk);
            } else if (v == 9) {
                hipLaunchKernelGGL((conv_wgrad_multi_kernel<1, false, 0, false, true>), dim3(first), dim3(256), lds1, st, pk);
            } else if (v == 10) {
                hipLaunchKernelGGL((conv_wgrad_multi_kernel<1, false, 0, true, false>), dim3(first), dim3(256), lds1, st, pk);
            } else {
                hipLaunchKernelGGL((conv_wgrad_multi_kernel<1, false, 0, true, true>), dim3(first), dim3(256), lds1, st, pk);
            }
            n = 0;
            return check_launch("conv1d_wgrad_partial_multi");
        };
        for (int i = 0; i < njobs; ++i) {
            if (variant(jobs[i]) != v) continue;
            idx[n++] = i;
            if (n == WG_MAXJ && flush()) return 1;
        }
        if (flush()) return 1;
    }
    return 0;
}

extern "C" int eeggan_wgrad_reduce_multi(const eeggan_reduce_job* jobs, int njobs, void* stream) {
    EEGGAN_REQUIRE(jobs && njobs > 0, "wgrad_reduce_multi: bad table");
    for (int first = 0; first < njobs; first += RED_MAXJ) {
        ReducePack pk;
        pk.n = njobs - first < RED_MAXJ ? njobs - first : RED_MAXJ;
        long long blocks = 0;
        for (int i = 0; i < pk.n; ++i) {
            pk.j[i] = jobs[first + i];
            EEGGAN_REQUIRE(pk.j[i].slabs && pk.j[i].out && pk.j[i].elems > 0 && pk.j[i].nsplit > 0,
                           "wgrad_reduce_multi: bad job %d", first + i);
            for (int k = 0; k < i; ++k)
                EEGGAN_REQUIRE(pk.j[k].out != pk.j[i].out, "wgrad_reduce_multi: jobs %d and %d write the same output",
                               first + k, first + i);
            blocks += reduce_blocks(pk.j[i]);
        }
        EEGGAN_REQUIRE(blocks < (1ll << 31), "wgrad_reduce_multi: too many elements");
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pk);
        if (check_launch("wgrad_reduce_multi")) return 1;
    }
    return 0;
}
