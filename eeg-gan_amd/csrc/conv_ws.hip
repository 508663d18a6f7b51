// Weight-stationary MFMA convolution for the 64-channel layers of the path (k = 3, same padding, at
// most 64 input and 64 output channels): forward and, on flipped weights, input gradient.  fp32 in,
// fp32 out, fp32 accumulation; the 64 -> 64 layers form their products from three bf16 pieces per
// operand on the bf16 matrix pipe (SPLIT below: fp32-exact to 2^-24, 2.7x fewer matrix cycles than
// v_mfma_f32_32x32x2_f32), the ragged ones use the fp32 matrix instructions.
//
// These layers carry most of the step's flops but only ~10 us of matrix work per launch, so the
// kernel is organised to keep the matrix pipes fed rather than around a workgroup-wide pipeline:
//   * the whole (scaled) weight matrix [64][64*K] is staged into LDS once per workgroup;
//   * a workgroup is 8 independent waves (2 per SIMD).  A wave owns one work unit at a time -- all 64
//     output channels of one 32-sample subtile -- with a private LDS region for its input subtile;
//     after the initial weight staging there is no workgroup barrier, so the waves drift apart and
//     one wave's loads, LDS traffic and stores overlap the other wave's MFMAs on the same SIMD;
//   * the next unit's input is loaded global -> registers before the MFMA block and written to LDS
//     after the epilogue;
//   * the epilogue transposes the accumulators through the wave's LDS region so that every global
//     store instruction writes eight full 128-byte lines (bias, activation and the per-(sample,
//     channel) multiplier are applied on the way).
#include "common.h"
#include "conv_args.h"
#include <cstdlib>
#include <type_traits>

namespace eeggan {
namespace {

constexpr int WS_WAVES = 8;
constexpr int XL = 40;                   // LDS floats per input row: [3 unused][halo][32 samples][halo][3 unused]
constexpr int XHALF = 32 * XL + 32;      // rows 32..63 start here; the extra 32 floats put the two lane
                                         // halves of an operand read on disjoint LDS banks
constexpr int XREG = XHALF + 32 * XL;    // floats per wave region (2592)
constexpr int EL = 36;                   // row stride of the transposed output tile, which reuses the region
static_assert(64 * EL <= XREG, "output tile must fit the wave's region");

__device__ __forceinline__ int xrow(int row) { return row * XL + (row >= 32 ? 32 : 0); }

// SPLIT form: every fp32 operand is the sum of three bf16 pieces, v = p1 + p2 + p3 exactly up to 2^-24 |v|
// (p1 = bf16(v), p2 = bf16(v - p1), p3 = bf16(v - p1 - p2); the differences are exact in fp32), and a product
// of two such sums keeps the six terms down to 2^-16 of the leading one: p1q1 + p1q2 + p2q1 + p2q2 + p1q3 +
// p3q1; the dropped terms are below 2^-24 of it, the rounding of an fp32 product.  Six bf16 MFMAs of 16
// reduction indices (32 cycles each) replace eight fp32 MFMAs of 2 (64 cycles each).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int WROWB = 400;           // bytes per output channel and piece: 8 groups x 3 taps x 8 channels x 2 B, + 16
constexpr int WPLANE = 64 * WROWB;   // bytes per piece

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Two values at once: the packed conversion (v_cvt_pk_bf16_f32) writes both pieces into one register, which is
// already the operand word; the pieces' float values for the remainders come out of it with a shift and a mask.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& w1, unsigned& w2, unsigned& w3) {
    w1 = pack_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, w1 << 16), rb = b - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = pack_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w2 << 16), sb = rb - __builtin_bit_cast(float, w2 & 0xffff0000u);
    w3 = pack_bf16(sa, sb);
}
__device__ __forceinline__ void split3(float v, __bf16& p1, __bf16& p2, __bf16& p3) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    p3 = (__bf16)r2;
}

// K taps; MASKED: the input is multiplied by a saved LeakyReLU/GDrop mask while it is staged;
// FULL: Cin == Cout == 64 and T % 32 == 0, so no load or store needs a predicate;
// RM: 32-channel output tiles per unit.  RM = 2: a unit is a whole subtile (all 64 output channels).
// RM = 1: a unit is one half of the output channels of a subtile (units 2s and 2s+1 share subtile s
// and are taken by neighbouring waves): twice the input reads, but twice as many, half as long units,
// which balances the four SIMDs of a CU when a layer has only a few subtiles per CU.
// SPLIT: 0 = fp32 matrix instructions; 3 = bf16 matrix instructions on three pieces per operand (fp32 products);
// 1 = bf16 matrix instructions on the operands rounded to bf16 (the bf16 compute mode: one product instead of six,
// no splitting; fp32 accumulation, fp32 in and out).
// BF: activations (x, mask_y, y, act_aux) are bfloat16 in memory (bf16 storage mode; with SPLIT = 1 only: products
// of fp32-exact pieces of values that carry 8 bits would be wasted work).
template <int K, bool MASKED, bool FULL, int RM, int SPLIT = 0, bool BF = false>
__global__ __launch_bounds__(512) void conv_ws_kernel(ConvArgs a) {
    static_assert(!SPLIT || (FULL && K == 3), "the split form covers the 64 -> 64, k = 3 layers");
    static_assert(!BF || SPLIT == 1, "bf16 storage runs the bf16 products");
    typedef typename ActT<BF>::T AT;
    typedef typename ActT<BF>::raw4 R4;
    typedef typename ActT<BF>::raw1 R1;
    const AT* const X0 = reinterpret_cast<const AT*>(a.x);
    const AT* const M0 = reinterpret_cast<const AT*>(a.mask_y);
    static_assert(SPLIT == 0 || SPLIT == 1 || SPLIT == 3, "pieces per operand");
    constexpr int PAD = (K - 1) / 2;
    constexpr int KK = 64 * K;   // reduction length with the input channels padded to 64
    constexpr int WLD = KK + 4;  // (WLD / 4) odd: conflict-free 16-byte row reads
    constexpr int KH = KK / 2;   // k-steps: an MFMA consumes two reduction indices, one per lane half

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WFLOATS = SPLIT ? 3 * WPLANE / 4 : 64 * WLD;  // weight region of the LDS image
    float* Xs = smem + WFLOATS + wave * XREG;

    // staging coordinates of this lane: 16-byte items (row r0 + 8 i, floats 4q..4q+3) and halo items
    const int q = lane & 7, r0 = lane >> 3;
    const int hside = lane & 1, hr0 = lane >> 1;
    const int lo = r0 * a.T + 4 * q;
    const int hlo = hr0 * a.T + (hside ? 32 : -1);

    R4 xv[8], mv[MASKED ? 8 : 1];
    R1 hv[2], mh[MASKED ? 2 : 1];
    float md[MASKED ? 8 : 1], mdh[MASKED ? 2 : 1];

    // The waves of a workgroup take CONSECUTIVE subtiles: their 64 input rows then share pages (and
    // cache lines for the halos), which matters because every row of a unit lies in a different page.
    // A small layer uses fewer waves per workgroup (a.wpb) so that it still touches every CU.
    const int ustride = a.wpb * (int)gridDim.x;
    int u = (int)blockIdx.x * a.wpb + wave;

    // Staging state of a unit: its base pointers and the validity of this lane's items.
    struct Unit {
        const AT* xb;
        const AT* mb;
        int b, t0, half;
        bool t_ok, h_ok;
    };
    auto locate = [&](int unit) {
        Unit r;
        r.half = RM == 1 ? (unit & 1) : 0;
        if (RM == 1) unit >>= 1;
        r.b = unit / a.ST;
        r.t0 = (unit - r.b * a.ST) * 32;
        r.xb = X0 + ((long long)r.b * a.Cin) * a.T + r.t0;
        r.mb = MASKED ? M0 + ((long long)r.b * a.Cin) * a.T + r.t0 : nullptr;
        r.t_ok = FULL || r.t0 + 4 * q < a.T;
        r.h_ok = hside ? (r.t0 + 32 < a.T) : (r.t0 > 0);
        return r;
    };

    // The 64 input rows are staged in four chunks; chunk c holds rows [8c, 8c+8) and [32+8c, 32+8c+8),
    // exactly the rows the k-steps [8cK, 8(c+1)K) read (lane half 0 / half 1), so the MFMAs of chunk c
    // can start as soon as items c and c+4 of this lane have arrived.  Chunk 0 also carries the halos.
    // issue_chunk only issues loads; their consumers live in commit_chunk.
    auto issue_chunk = [&](const Unit& un, int c) {
        if (c == 0 && K > 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = hr0 + 32 * i;
                const bool ok = un.h_ok && (FULL || row < a.Cin);
                const int off = hlo + 32 * i * a.T;
                hv[i] = ld_raw1(ok ? un.xb + off : X0);
                if (MASKED) {
                    mh[MASKED ? i : 0] = ld_raw1(ok ? un.mb + off : M0);
                    mdh[MASKED ? i : 0] = a.mask_drop ? a.mask_drop[ok ? un.b * a.Cin + row : 0] : 1.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = c + 4 * k;
            const int row = r0 + 8 * i;
            const bool ok = FULL || (un.t_ok && row < a.Cin);
            const int off = lo + 8 * i * a.T;
            xv[i] = ld_raw4(ok ? un.xb + off : X0);
            if (MASKED) {
                mv[MASKED ? i : 0] = ld_raw4(ok ? un.mb + off : M0);
                md[MASKED ? i : 0] = a.mask_drop ? a.mask_drop[ok ? un.b * a.Cin + row : 0] : 1.f;
            }
        }
    };

    // registers -> this wave's LDS region
    auto commit_chunk = [&](const Unit& un, int c) {
        if (c == 0 && K > 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = hr0 + 32 * i;
                float v = widen(hv[i]);
                if (MASKED) v *= widen(mh[MASKED ? i : 0]) > 0.f ? mdh[MASKED ? i : 0] : mdh[MASKED ? i : 0] * a.mask_slope;
                const bool ok = un.h_ok && (FULL || row < a.Cin);
                Xs[xrow(row) + (hside ? 36 : 3)] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = c + 4 * k;
            const int row = r0 + 8 * i;
            f32x4 v = widen(xv[i]);
            if (MASKED) {
                const f32x4 my = widen(mv[MASKED ? i : 0]);
                const float d = md[MASKED ? i : 0];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= my[e] > 0.f ? d : d * a.mask_slope;
            }
            if (!FULL) {
                const bool ok = un.t_ok && row < a.Cin;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(Xs + xrow(row) + 4 + 4 * q) = v;
        }
    };

    EEGGAN_STAMP(0, 6);
    // weights -> LDS, scaled, zero-padded to 64 x 64*K (once per workgroup).  Their loads are issued
    // BEFORE the first unit's input loads: a wave's loads return in order, and the weights (served by
    // L2 after the first workgroup) must not queue behind 8 KiB of input coming from HBM.
    constexpr int Q4 = KK / 4;
    constexpr int WV = 64 * Q4 / 512;
    f32x4 wv[FULL ? WV : 1];
    if (SPLIT) {
        // thread = (output channel tid >> 3, group of 8 input channels tid & 7): 24 consecutive floats, which
        // become one 16-byte fragment per tap and piece
        static_assert(!SPLIT || WV == 6, "one (row, 8-channel group) per thread");
#pragma unroll
        for (int i = 0; i < WV; ++i) wv[FULL ? i : 0] = *reinterpret_cast<const f32x4*>(a.w + (long long)tid * 24 + 4 * i);
    } else if (FULL) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int e4 = tid + 512 * i;
            const int co = e4 / Q4, q4 = e4 - co * Q4;
            wv[FULL ? i : 0] = *reinterpret_cast<const f32x4*>(a.w + (long long)co * KK + 4 * q4);
        }
    }
    // (a wave without a unit loads unit 0 and drops it: a load inside a branch would make every later
    // wait count conservative, i.e. the weight staging below would wait for the input as well)
    EEGGAN_STAMP(15, 0);
    const int nunits = a.NS * (3 - RM);
    const bool active = wave < a.wpb && u < nunits;
    Unit cur = locate(active ? u : 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) issue_chunk(cur, c);
    __builtin_amdgcn_sched_barrier(0);
    EEGGAN_STAMP(15, 1);
    if (SPLIT) {
        // piece planes, [co][lane half h][chunk c][tap][8 channels]: the 8 channels 32h + 8c + i of one tap are one
        // 16-byte operand fragment; this thread's group is (h, c) = (group >> 2, group & 3), i.e. slot `group`
        char* Wb = reinterpret_cast<char*>(smem) + (tid >> 3) * WROWB + (tid & 7) * 3 * 16;
        u32x4 p[3][3];  // [tap][piece]: four words of two channels each
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                // this thread's float k is tap k % 3 of channel k / 3 of its group: channels 2 cp and 2 cp + 1
                const int k0 = 6 * cp + tap, k1 = k0 + 3;
                unsigned w1, w2, w3;
                split3_pair(wv[FULL ? k0 / 4 : 0][k0 % 4] * a.in_scale, wv[FULL ? k1 / 4 : 0][k1 % 4] * a.in_scale, w1, w2, w3);
                p[tap][0][cp] = w1;
                p[tap][1][cp] = w2;
                p[tap][2][cp] = w3;
            }
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(Wb + k * WPLANE + tap * 16) = p[tap][k];
    } else if (FULL) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int e4 = tid + 512 * i;
            const int co = e4 / Q4, q4 = e4 - co * Q4;
            f32x4 v = wv[FULL ? i : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= a.in_scale;
            *reinterpret_cast<f32x4*>(Ws + co * WLD + 4 * q4) = v;
        }
    } else {
        for (int idx = tid; idx < 64 * KK; idx += 512) {
            const int co = idx / KK, r = idx - co * KK;
            const int ci = r / K, j = r - ci * K;
            const bool ok = co < a.Cout && ci < a.Cin;
            const float v = a.w[ok ? ((long long)co * a.Cin + ci) * K + j : 0];
            Ws[co * WLD + r] = ok ? v * a.in_scale : 0.f;
        }
    }
    EEGGAN_STAMP(15, 2);
    __syncthreads();  // the only workgroup barrier
    EEGGAN_STAMP(15, 3);
    if (!active) return;

    // epilogue coordinates: this lane handles rows r0 + 8 i (output channels), floats 4q..4q+3
    // Optional operands are read through an always-valid pointer and neutralised by a select at
    // their use: a conditional load ends in a register copy that waits for ALL outstanding loads.
    const bool has_bias = a.bias != nullptr, has_drop = a.drop != nullptr;
    const float* bias_p = has_bias ? a.bias : a.w;
    const float* drop_p = has_drop ? a.drop : a.w;
    const int ylo = r0 * a.T + 4 * q;

    const float* Ap = Ws + l31 * WLD + h * KH;
    const float* Bp = Xs + (h ? XHALF : 0) + (4 - PAD) + l31;
    EEGGAN_STAMP(0, 7);
    int step = 0;
    (void)step;

    // One unit: stage its chunks, multiply, store.  MORE: another unit follows, whose chunks are
    // loaded into the registers as they become free (the last unit is a separate instantiation rather
    // than a branch around the loads, again to keep the wait counts exact).
    auto unit_body = [&](auto more_tag, const Unit& nxt) {
        constexpr bool MORE = decltype(more_tag)::value;
        EEGGAN_STAMP(step, 0);
        const bool st_ok = cur.t_ok;
        constexpr int NE = 4 * RM;       // epilogue items of this lane: output rows co_base + r0 + 8 i
        const int co_base = cur.half * 32;
        float bb[NE], dd[NE];  // (loaded per unit: hoisted out of the loop, their selects would wait for the input)
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int row = (FULL || co_base + r0 + 8 * i < a.Cout) ? co_base + r0 + 8 * i : 0;
            bb[i] = bias_p[row];
            dd[i] = drop_p[has_drop ? cur.b * a.Cout + row : 0];
        }
        EEGGAN_STAMP(step, 1);

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        const float* Au = Ap + co_base * WLD;
        if constexpr (SPLIT == 3) {
            // Twelve steps (chunk c, tap j) of two half steps (one 32-channel output tile: 6 MFMAs) each, software-
            // pipelined so that every MFMA gap carries its share of the other work.  A SIMD issues one vector
            // instruction at a time, and an MFMA holds that issue port for 8 of its 32 cycles: the ~50 VALU
            // instructions that split a step's 8 input samples are free while they sit BETWEEN the MFMAs of the same
            // wave (4-5 per gap), and cost their full issue time in a block of their own -- the partner wave on the
            // SIMD cannot fill such a block with its MFMAs, the older wave's VALU stream takes every issue slot.
            //   during half step (s, tile):  LDS reads of the weight fragments of the next half step (and, on tile 0,
            //   of the input samples of step s + 2), the splitting of half of step s + 1's samples, 6 MFMAs of step s.
            // Chunk c + 1 is committed to LDS at the top of step 3c + 1, before the first reads of its samples (a
            // wave's LDS operations execute in order: its reads see its own writes).
            const char* Wl = reinterpret_cast<const char*>(smem) + (co_base + l31) * WROWB + (h * 4) * 3 * 16;
            float xa[8], xb[8];        // raw samples of steps s + 1 / s + 2 (alternating)
            bf16x8 wc[3], wn[3];       // weight fragments (three piece planes) of this / the next half step
            u32x4 pc[3], pn[3];        // piece vectors of this / the next step's samples (8 bf16 each)
            auto read_x = [&](int st, float (&xr)[8]) {
                const int c = st / 3, j = st - 3 * c;
#pragma unroll
                for (int i = 0; i < 8; ++i) xr[i] = Bp[(8 * c + i) * XL + j];
            };
            auto read_w = [&](int st, int tile, bf16x8 (&wr)[3]) {
                const int c = st / 3, j = st - 3 * c;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    wr[k] = *reinterpret_cast<const bf16x8*>(Wl + c * 3 * 16 + tile * 32 * WROWB + j * 16 + k * WPLANE);
            };
            auto split_half = [&](const float (&xr)[8], int half, u32x4 (&p)[3]) {
#pragma unroll
                for (int i = 2 * half; i < 2 * half + 2; ++i) {
                    unsigned w1, w2, w3;
                    split3_pair(xr[2 * i], xr[2 * i + 1], w1, w2, w3);
                    p[0][i] = w1;
                    p[1][i] = w2;
                    p[2][i] = w3;
                }
            };
            commit_chunk(cur, 0);
            if (MORE) issue_chunk(nxt, 0);
            read_x(0, xb);
            read_x(1, xa);
            read_w(0, 0, wc);
            split_half(xb, 0, pc);
            split_half(xb, 1, pc);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                if (st % 3 == 1 && st < 10) {
                    commit_chunk(cur, st / 3 + 1);
                    if (MORE) issue_chunk(nxt, st / 3 + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                float(&xnext)[8] = (st & 1) ? xb : xa;   // samples of step st + 1 (read during step st - 1)
                float(&xfar)[8] = (st & 1) ? xa : xb;    // buffer the samples of step st + 2 go to
#pragma unroll
                for (int tile = 0; tile < RM; ++tile) {
                    int nreads = 0;
                    if (tile + 1 < RM) {
                        read_w(st, tile + 1, wn);
                        nreads += 3;
                    } else if (st < 11) {
                        read_w(st + 1, 0, wn);
                        nreads += 3;
                    }
                    if (st < 11) split_half(xnext, RM == 2 ? tile : 0, pn);
                    if (RM == 1 && st < 11) split_half(xnext, 1, pn);
                    if (tile == RM - 1 && st < 10) {
                        // (after the splitting in program order: the buffer that held step st + 1's samples on the
                        // previous step is xfar only on the NEXT step, this one is free since step st - 1)
                        read_x(st + 2, xfar);
                        nreads += 4;
                    }
                    f32x16& acc = tile == 0 ? acc0 : acc1;
                    const bf16x8 x1 = __builtin_bit_cast(bf16x8, pc[0]), x2 = __builtin_bit_cast(bf16x8, pc[1]),
                                 x3 = __builtin_bit_cast(bf16x8, pc[2]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, wc[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, wc[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, wc[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, wc[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, wc[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, wc[0], acc, 0, 0, 0);
                    // order inside the half step: the reads, then MFMA / VALU alternating
                    __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, RM == 2 ? 4 : 8, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    (void)nreads;
#pragma unroll
                    for (int k = 0; k < 3; ++k) wc[k] = wn[k];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) pc[k] = pn[k];
            }
        } else
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            commit_chunk(cur, c);
            if (MORE) issue_chunk(nxt, c);
            __builtin_amdgcn_sched_barrier(0);
            if (SPLIT) {
                const char* Wb = reinterpret_cast<const char*>(smem) + (co_base + l31) * WROWB + (h * 4 + c) * 3 * 16;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (SPLIT == 1) {
                        bf16x8 x1;
#pragma unroll
                        for (int i = 0; i < 8; ++i) x1[i] = (__bf16)Bp[(8 * c + i) * XL + j];
#pragma unroll
                        for (int tile = 0; tile < RM; ++tile) {
                            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(Wb + tile * 32 * WROWB + j * 16);
                            f32x16& acc = tile == 0 ? acc0 : acc1;
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1, acc, 0, 0, 0);
                        }
                        continue;
                    }
                    bf16x8 x1, x2, x3;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __bf16 p1, p2, p3;
                        split3(Bp[(8 * c + i) * XL + j], p1, p2, p3);
                        x1[i] = p1;
                        x2[i] = p2;
                        x3[i] = p3;
                    }
#pragma unroll
                    for (int tile = 0; tile < RM; ++tile) {
                        const char* wp = Wb + tile * 32 * WROWB + j * 16;
                        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp);
                        const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + WPLANE);
                        const bf16x8 w3 = *reinterpret_cast<const bf16x8*>(wp + 2 * WPLANE);
                        f32x16& acc = tile == 0 ? acc0 : acc1;
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3, w1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w3, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w2, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x2, w1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w2, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, w1, acc, 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int s4 = c * (2 * K); s4 < (c + 1) * (2 * K); ++s4) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(Au + 4 * s4);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(Au + (RM == 2 ? 32 * WLD : 0) + 4 * s4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int s = 4 * s4 + e;
                    const int cl = s / K, j = s - cl * K;
                    const float bv = Bp[cl * XL + j];
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a0[e], acc0, 0, 0, 0);
                    if (RM == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, a1[e], acc1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        EEGGAN_STAMP(step, 2);

        // accumulators (lane = output channel, register r = time (r&3) + 8 (r>>2) + 4h) -> LDS tile
        // E[channel][time]; LDS executes a wave's operations in order, so no wait is needed between
        // the operand reads above, these writes and the reads below.
        float* E = Xs;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            f32x4 v0, v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v0[e] = acc0[4 * qq + e];
                v1[e] = acc1[4 * qq + e];
            }
            *reinterpret_cast<f32x4*>(E + l31 * EL + 8 * qq + 4 * h) = v0;
            if (RM == 2) *reinterpret_cast<f32x4*>(E + (32 + l31) * EL + 8 * qq + 4 * h) = v1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        AT* yb = reinterpret_cast<AT*>(a.y) + ((long long)cur.b * a.Cout + co_base) * a.T + cur.t0;
        const AT* A0 = reinterpret_cast<const AT*>(a.act_aux);
        const AT* ab = a.act_aux ? A0 + ((long long)cur.b * a.Cout + co_base) * a.T + cur.t0 : X0;
        auto run = [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
            f32x4 m[ACT == EEGGAN_ACT_MASK ? NE : 1];
            if (ACT == EEGGAN_ACT_MASK) {
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const bool ok = st_ok && (FULL || co_base + r0 + 8 * i < a.Cout);
                    m[ACT == EEGGAN_ACT_MASK ? i : 0] = ld4(ok ? ab + ylo + 8 * i * a.T : A0);
                }
            }
            if (ACT == ACT_LRELU_PN) {
                // PixelNorm over the channel axis: this lane holds rows r0 + 8 i of four time steps; the
                // other rows of the same time steps sit in the lanes that differ in lane bits 3..5
                f32x4 vals[NE];
                float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int row = r0 + 8 * i;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(E + row * EL + 4 * q);
                    const float bias_v = has_bias ? bb[i] : 0.f;
                    const bool rok = FULL || co_base + row < a.Cout;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = v[e] + bias_v;
                        t = t > 0.f ? t : t * a.act_param;
                        t = rok ? t : 0.f;
                        vals[i][e] = t;
                        ss[e] += t * t;
                    }
                }
                f32x4 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = ss[e];
                    s += __shfl_xor(s, 8, 64);
                    s += __shfl_xor(s, 16, 64);
                    s += __shfl_xor(s, 32, 64);
                    dv[e] = rsqrtf(s / (float)a.Cout + a.pn_eps);
                }
                if (r0 == 0 && st_ok) *reinterpret_cast<f32x4*>(a.pn_d + (long long)cur.b * a.T + cur.t0 + 4 * q) = dv;
#pragma unroll
                for (int i = 0; i < NE; ++i) {
                    const int row = r0 + 8 * i;
                    const float drop_v = has_drop ? dd[i] : 1.f;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = vals[i][e] * dv[e] * drop_v;
                    if (FULL || (st_ok && co_base + row < a.Cout)) st4(yb + ylo + 8 * i * a.T, v);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < NE; ++i) {
                const int row = r0 + 8 * i;  // row of the unit's tile; output channel co_base + row
                f32x4 v = *reinterpret_cast<const f32x4*>(E + row * EL + 4 * q);
                const float bias_v = has_bias ? bb[i] : 0.f;
                const float drop_v = has_drop ? dd[i] : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[e] + bias_v;
                    if (ACT == EEGGAN_ACT_LRELU) t = t > 0.f ? t : t * a.act_param;
                    if (ACT == EEGGAN_ACT_TANH) t = tanhf(t * a.act_param);
                    if (ACT == EEGGAN_ACT_MASK) t = m[ACT == EEGGAN_ACT_MASK ? i : 0][e] > 0.f ? t : t * a.act_param;
                    v[e] = t * drop_v;
                }
                if (FULL || (st_ok && co_base + row < a.Cout)) st4(yb + ylo + 8 * i * a.T, v);
            }
        };
        switch (a.act) {
            case EEGGAN_ACT_LRELU: run(std::integral_constant<int, EEGGAN_ACT_LRELU>{}); break;
            case EEGGAN_ACT_TANH: run(std::integral_constant<int, EEGGAN_ACT_TANH>{}); break;
            case EEGGAN_ACT_MASK: run(std::integral_constant<int, EEGGAN_ACT_MASK>{}); break;
            case ACT_LRELU_PN:
                if (RM == 2) run(std::integral_constant<int, (RM == 2 ? ACT_LRELU_PN : EEGGAN_ACT_NONE)>{});
                break;
            default: run(std::integral_constant<int, EEGGAN_ACT_NONE>{}); break;
        }
        EEGGAN_STAMP(step, 3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (int nu = u + ustride; nu < nunits; nu += ustride, ++step) {
        const Unit nxt = locate(nu);
        unit_body(std::true_type{}, nxt);
        cur = nxt;
    }
    unit_body(std::false_type{}, cur);
}

template <int K, bool MASKED, bool FULL, int RM, int SPLIT = 0, bool BF = false>
int launch(const ConvArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)((SPLIT ? 3 * WPLANE / 4 : 64 * (64 * K + 4)) + WS_WAVES * XREG) * sizeof(float);
    auto kernel = &conv_ws_kernel<K, MASKED, FULL, RM, SPLIT, BF>;
    // opt in to > 64 KiB of dynamic LDS (once per device)
    static std::atomic<bool> ready_dev[MAX_DEVICES];
    std::atomic<bool>& ready = ready_dev[current_device()];
    if (!ready.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("conv1d: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
            return 1;
        }
        ready.store(true, std::memory_order_release);
    }
    ConvArgs b = a;
    const long long units = (long long)a.NS * (3 - RM);
    b.wpb = (int)((units + num_cus() - 1) / num_cus());
    if (b.wpb > WS_WAVES) b.wpb = WS_WAVES;
    if (RM == 1 && (b.wpb & 1)) ++b.wpb;  // both halves of a subtile in the same workgroup
    long long grid = (units + b.wpb - 1) / b.wpb;
    if (grid > num_cus()) grid = num_cus();
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, st, b);
    return check_launch("conv1d");
}

template <int K, bool MASKED, bool FULL>
int launch_rm(const ConvArgs& a, hipStream_t st) {
    // whole-subtile units once every SIMD has at least two of them to interleave, half units below
    // ... or when halves fill the chip's rounds of waves better (e.g. 2560 subtiles on 2048 waves: 2 rounds of
    // whole units, 3 rounds of half units = 1.5)
    const long long waves = 8ll * num_cus();
    const long long whole = (a.NS + waves - 1) / waves * 2, half = (2ll * a.NS + waves - 1) / waves;
    const bool halves = a.act != ACT_LRELU_PN && (a.NS < waves || half < whole);  // (the norm needs all channels in one unit)
    if constexpr (FULL && K == 3) {
        // (product_pieces() == 0 keeps the fp32 matrix instructions: comparison runs)
        const int pieces = product_pieces();
        if (pieces) {
            // With the split products a unit's cost is mostly the splitting of its input, which a whole unit does
            // once for both halves of the output channels: a whole unit counts 1.3 half units, not 2 (i.e. half
            // units only while they fit one round of the chip's waves; measured +1.2 % steps/s, 1.6 the same).
            const bool halves_s = a.act != ACT_LRELU_PN && half * 10 < (whole / 2) * 13;
            if (a.x_bf16)  // (bf16 storage: checked by the caller to come with bf16 products and a bf16 output)
                return halves_s ? launch<K, MASKED, FULL, 1, 1, true>(a, st) : launch<K, MASKED, FULL, 2, 1, true>(a, st);
            if (pieces == 1)
                return halves_s ? launch<K, MASKED, FULL, 1, 1>(a, st) : launch<K, MASKED, FULL, 2, 1>(a, st);
            return halves_s ? launch<K, MASKED, FULL, 1, 3>(a, st) : launch<K, MASKED, FULL, 2, 3>(a, st);
        }
    }
    return halves ? launch<K, MASKED, FULL, 1>(a, st) : launch<K, MASKED, FULL, 2>(a, st);
}

}  // namespace

bool conv_ws_covers(const ConvArgs& a, int K) {
    if (K != 3 || a.Cin > 64 || a.Cout > 64 || a.Cout <= 32 || a.T % 4 != 0 || a.NS < 256) return false;
    return (long long)a.Cin * a.T * 2 < (1ll << 31) && (long long)a.Cout * a.T * 2 < (1ll << 31);
}

int launch_conv_ws(const ConvArgs& a, int K, hipStream_t st) {
    if (!conv_ws_covers(a, K)) return -1;
    const bool full = a.Cin == 64 && a.Cout == 64 && a.T % 32 == 0;
    if (a.x_bf16 || a.y_bf16) {
        // bf16 storage: the 64 -> 64 layers, both sides bfloat16, bf16 products
        if (!(a.x_bf16 && a.y_bf16 && full && product_pieces() == 1)) return -1;
    }
    if (a.mask_y) return full ? launch_rm<3, true, true>(a, st) : launch_rm<3, true, false>(a, st);
    return full ? launch_rm<3, false, true>(a, st) : launch_rm<3, false, false>(a, st);
}

}  // namespace eeggan
