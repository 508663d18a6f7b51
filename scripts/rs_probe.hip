// Timing aid: shapes of the quad-form rational resampler ([B*64 rows, Tin] -> [., Tout]) beside a copy of the same bytes.
// Coefficients are arbitrary (timing does not depend on them); base[q] follows the ratio like the real tables.
// build: hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops scripts/rs_probe.hip -o scripts/rs_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// PERIOD > 0: the coefficient blocks repeat with that period over the quads (a rational ratio's interior does), so the
// table a launch touches is a few hundred bytes and stays in the vector cache
// PERIOD < 0: the table laid out so that the 64 lanes of a wave read 1 KB in one piece: [q / 64][e][v][q % 64][4]
template <int NV, int ROWS, bool NTS, int PERIOD = 0, bool UNAL = false>
__global__ __launch_bounds__(256) void quad_kernel(const float* __restrict__ x, const int* __restrict__ base,
                                                   const float* __restrict__ coef, float* __restrict__ out, long long rows,
                                                   int Tin, int Tout, int chunks) {
    const int NQ = Tout / 4;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rg = id / NQ;
    const int q = (int)(id - rg * NQ);
    long long r0 = rg * ROWS * chunks;
    if (r0 >= rows) return;
    const int b0 = base[q];
    f32x4 cf[4][NV];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int v = 0; v < NV; ++v)
            cf[e][v] = PERIOD < 0
                           ? *reinterpret_cast<const f32x4*>(coef + ((((long long)(q >> 6) * 4 + e) * NV + v) * 64 + (q & 63)) * 4)
                           : *reinterpret_cast<const f32x4*>(coef + ((long long)(PERIOD > 0 ? q % PERIOD : q) * 4 + e) * (4 * NV) + 4 * v);
    for (int ch = 0; ch < chunks && r0 < rows; ++ch, r0 += ROWS) {
        f32x4 xv[ROWS][NV];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const long long row = r0 + r < rows ? r0 + r : rows - 1;
#pragma unroll
            for (int v = 0; v < NV; ++v)
                xv[r][v] = UNAL ? (f32x4)*reinterpret_cast<const f32x4u*>(x + row * Tin + b0 + 4 * v)
                                : *reinterpret_cast<const f32x4*>(x + row * Tin + b0 + 4 * v);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
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
            if (r0 + r < rows) {
                f32x4* p = reinterpret_cast<f32x4*>(out + (r0 + r) * Tout + 4 * q);
                if (NTS)
                    __builtin_nontemporal_store(o, p);
                else
                    *p = o;
            }
        }
    }
}

// The same walk with the next pass's loads issued before the current pass is combined and stored.
template <int NV, int ROWS, bool NTS>
__global__ __launch_bounds__(256) void quad_pipe_kernel(const float* __restrict__ x, const int* __restrict__ base,
                                                        const float* __restrict__ coef, float* __restrict__ out,
                                                        long long rows, int Tin, int Tout, int chunks) {
    const int NQ = Tout / 4;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long rg = id / NQ;
    const int q = (int)(id - rg * NQ);
    long long r0 = rg * ROWS * chunks;
    if (r0 >= rows) return;
    const int b0 = base[q];
    f32x4 cf[4][NV];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int v = 0; v < NV; ++v) cf[e][v] = *reinterpret_cast<const f32x4*>(coef + ((long long)q * 4 + e) * (4 * NV) + 4 * v);
    f32x4 cur[ROWS][NV], nxt[ROWS][NV];
    const float* xq = x + b0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long long row = r0 + r < rows ? r0 + r : rows - 1;
#pragma unroll
        for (int v = 0; v < NV; ++v) cur[r][v] = *reinterpret_cast<const f32x4*>(xq + row * Tin + 4 * v);
    }
    for (int ch = 0; ch < chunks && r0 < rows; ++ch, r0 += ROWS) {
        if (ch + 1 < chunks) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const long long row = r0 + ROWS + r < rows ? r0 + ROWS + r : rows - 1;
#pragma unroll
                for (int v = 0; v < NV; ++v) nxt[r][v] = *reinterpret_cast<const f32x4*>(xq + row * Tin + 4 * v);
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s += cf[e][v][i] * cur[r][v][i];
                o[e] = s;
            }
            if (r0 + r < rows) {
                f32x4* p = reinterpret_cast<f32x4*>(out + (r0 + r) * Tout + 4 * q);
                if (NTS)
                    __builtin_nontemporal_store(o, p);
                else
                    *p = o;
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int v = 0; v < NV; ++v) cur[r][v] = nxt[r][v];
    }
}

// A wave loads every input vector of its span ONCE (lanes along the row, 1 KB per instruction, no overlap between lanes)
// into LDS and the lanes then pick their NV vectors from there.  A wave owns 64 consecutive quads of one row group.
template <int NV, int ROWS, int NLOAD, bool NT>
__global__ __launch_bounds__(256) void quad_wave_kernel(const float* __restrict__ x, const int* __restrict__ base,
                                                        const float* __restrict__ coef, float* __restrict__ out,
                                                        long long rows, int Tin, int Tout) {
    __shared__ f32x4 stage[4][ROWS][NLOAD * 64];
    const int NQ = Tout / 4, WPR = (NQ + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long w = (long long)blockIdx.x * 4 + wave;
    const long long rg = w / WPR;
    const int wq = (int)(w - rg * WPR);
    const int q = wq * 64 + lane;
    const bool active = q < NQ;
    const long long r0 = rg * ROWS;
    const bool live = r0 < rows;
    const int bf = base[wq * 64];
    const int b0 = active ? base[q] : bf;
    f32x4 cf[4][NV];
    if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                cf[e][v] = *reinterpret_cast<const f32x4*>(coef + ((((long long)(q >> 6) * 4 + e) * NV + v) * 64 + (q & 63)) * 4);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const long long row = r0 + r < rows ? r0 + r : rows - 1;
#pragma unroll
            for (int l = 0; l < NLOAD; ++l) {
                const int at = bf + 4 * (l * 64 + lane);
                if (at + 4 <= Tin) {
                    const f32x4* p = reinterpret_cast<const f32x4*>(x + row * Tin + at);
                    stage[wave][r][l * 64 + lane] = NT ? __builtin_nontemporal_load(p) : *p;
                }
            }
        }
    }
    __syncthreads();
    if (!live || !active) return;
    const int off = (b0 - bf) >> 2;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        f32x4 xv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) xv[v] = stage[wave][r][off + v];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) s += cf[e][v][i] * xv[v][i];
            o[e] = s;
        }
        if (r0 + r < rows) {
            f32x4* p = reinterpret_cast<f32x4*>(out + (r0 + r) * Tout + 4 * q);
            if (NT)
                __builtin_nontemporal_store(o, p);
            else
                *p = o;
        }
    }
}

// The input rows of a workgroup staged through LDS: every input byte is fetched from memory once, by 16-byte loads
// that run along the rows; the taps are then read from LDS.  A workgroup owns RW rows x all quads of those rows in
// pieces of QW quads; thread = (row, quad).
template <int NV, int RW, bool NT>
__global__ __launch_bounds__(256) void lds_kernel(const float* __restrict__ x, const int* __restrict__ base,
                                                  const float* __restrict__ coef, float* __restrict__ out, long long rows,
                                                  int Tin, int Tout) {
    extern __shared__ float tile[];  // [RW][Tin]
    const long long r0 = (long long)blockIdx.x * RW;
    const int nvec = RW * Tin / 4;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        const int r = i / (Tin / 4), c = i - r * (Tin / 4);
        const long long row = r0 + r < rows ? r0 + r : rows - 1;
        const f32x4* p = reinterpret_cast<const f32x4*>(x + row * Tin) + c;
        reinterpret_cast<f32x4*>(tile)[i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
    __syncthreads();
    const int NQ = Tout / 4;
    for (int q = threadIdx.x; q < NQ; q += 256) {
        const int b0 = base[q];
        f32x4 cf[4][NV];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int v = 0; v < NV; ++v)
                cf[e][v] = *reinterpret_cast<const f32x4*>(coef + ((long long)q * 4 + e) * (4 * NV) + 4 * v);
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            f32x4 xv[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) xv[v] = *reinterpret_cast<const f32x4*>(tile + r * Tin + b0 + 4 * v);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s += cf[e][v][i] * xv[v][i];
                o[e] = s;
            }
            if (r0 + r < rows) {
                f32x4* p = reinterpret_cast<f32x4*>(out + (r0 + r) * Tout + 4 * q);
                if (NT)
                    __builtin_nontemporal_store(o, p);
                else
                    *p = o;
            }
        }
    }
}

template <typename F>
static float time_us(F launch, int n) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / n;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    const int Tin = argc > 2 ? atoi(argv[2]) : 1280, Tout = argc > 3 ? atoi(argv[3]) : 1920;
    const int NBUF = 4;
    const long long rows = (long long)B * 64;
    float *xs[NBUF], *os[NBUF];
    for (int i = 0; i < NBUF; ++i) {
        CHECK(hipMalloc(&xs[i], rows * Tin * 4));
        CHECK(hipMalloc(&os[i], rows * Tout * 4));
        CHECK(hipMemset(xs[i], 0, rows * Tin * 4));
    }
    const int NQ = Tout / 4;
    const double bytes = 4.0 * rows * (Tin + Tout);
    const int reps = B >= 256 ? 12 : 48;
    int turn = 0;
    auto report = [&](const char* name, float us) {
        printf("%-44s %8.1f us  %6.0f GB/s  %.2f of 8 TB/s\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
        fflush(stdout);
    };
    printf("--- %d samples, %d -> %d\n", B, Tin, Tout);
    const bool unal = argc > 5;  // bases at any sample (4-byte aligned 16-byte loads), as an unaligned table would have them
    const bool exact = argc > 4;  // taps start exactly at the quad's first input: with Tin == Tout and NV = 1 a row-strided copy
    for (int NV = exact ? 1 : 2; NV <= 3; ++NV) {
        std::vector<int> hb(NQ);
        for (int q = 0; q < NQ; ++q) {
            long long first = (long long)4 * q * Tin / Tout - (exact ? 0 : (NV == 3 ? 2 : 1));
            int b0 = (int)(first < 0 ? 0 : first);
            if (!unal) b0 &= ~3;
            if (b0 > Tin - 4 * NV) b0 = Tin - 4 * NV;
            hb[q] = b0;
        }
        int span = 0;  // widest span of a wave's 64 quads, in 16-byte vectors
        for (int q0 = 0; q0 < NQ; q0 += 64) {
            const int last = q0 + 63 < NQ ? q0 + 63 : NQ - 1;
            const int sv = (hb[last] + 4 * NV - hb[q0]) / 4;
            span = sv > span ? sv : span;
        }
        printf("NV=%d: a wave's span is %d vectors\n", NV, span);
        std::vector<float> hc((size_t)((NQ + 63) / 64 * 64) * 16 * NV, 0.125f);
        int* base;
        float* coef;
        CHECK(hipMalloc(&base, NQ * 4));
        CHECK(hipMalloc(&coef, hc.size() * 4));
        CHECK(hipMemcpy(base, hb.data(), NQ * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
#define QUAD(NV_, ROWS_, NTS_, PER)                                                                                   \
    if (NV == NV_) {                                                                                                  \
        const int chunks = PER / ROWS_;                                                                               \
        const long long threads = (rows + PER - 1) / PER * NQ;                                                        \
        report("quad NV=" #NV_ " rows/pass=" #ROWS_ " nts=" #NTS_ " rows/thread=" #PER, time_us([&] {                \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((quad_kernel<NV_, ROWS_, NTS_>), dim3((unsigned)((threads + 255) / 256)),       \
                                      dim3(256), 0, 0, x, base, coef, o, rows, Tin, Tout, chunks);                    \
               }, reps));                                                                                             \
    }
#define QUADP(NV_, ROWS_, NTS_, PER, PERIOD_)                                                                         \
    if (NV == NV_) {                                                                                                  \
        const int chunks = PER / ROWS_;                                                                               \
        const long long threads = (rows + PER - 1) / PER * NQ;                                                        \
        report("quad NV=" #NV_ " rows/pass=" #ROWS_ " nts=" #NTS_ " rows/thread=" #PER " period=" #PERIOD_, time_us([&] { \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((quad_kernel<NV_, ROWS_, NTS_, PERIOD_>), dim3((unsigned)((threads + 255) / 256)), \
                                      dim3(256), 0, 0, x, base, coef, o, rows, Tin, Tout, chunks);                    \
               }, reps));                                                                                             \
    }
#define PIPE(NV_, ROWS_, NTS_, PER)                                                                                   \
    if (NV == NV_) {                                                                                                  \
        const int chunks = PER / ROWS_;                                                                               \
        const long long threads = (rows + PER - 1) / PER * NQ;                                                        \
        report("pipelined NV=" #NV_ " rows/pass=" #ROWS_ " nts=" #NTS_ " rows/thread=" #PER, time_us([&] {           \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((quad_pipe_kernel<NV_, ROWS_, NTS_>), dim3((unsigned)((threads + 255) / 256)),  \
                                      dim3(256), 0, 0, x, base, coef, o, rows, Tin, Tout, chunks);                    \
               }, reps));                                                                                             \
    }
#define WAVEK(NV_, ROWS_, NLOAD_, NT_)                                                                                \
    if (NV == NV_ && (span + 63) / 64 <= NLOAD_) {                                                                    \
        const long long waves = (rows + ROWS_ - 1) / ROWS_ * ((NQ + 63) / 64);                                        \
        report("wave-staged NV=" #NV_ " rows=" #ROWS_ " loads/row=" #NLOAD_ " nt=" #NT_, time_us([&] {               \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((quad_wave_kernel<NV_, ROWS_, NLOAD_, NT_>), dim3((unsigned)((waves + 3) / 4)), \
                                      dim3(256), 0, 0, x, base, coef, o, rows, Tin, Tout);                            \
               }, reps));                                                                                             \
    }
#define QUADU(NV_, ROWS_, NTS_, PER)                                                                                  \
    if (NV == NV_ && unal) {                                                                                          \
        const int chunks = PER / ROWS_;                                                                               \
        const long long threads = (rows + PER - 1) / PER * NQ;                                                        \
        report("quad UNALIGNED NV=" #NV_ " rows/pass=" #ROWS_ " nts=" #NTS_ " rows/thread=" #PER, time_us([&] {       \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((quad_kernel<NV_, ROWS_, NTS_, -1, true>), dim3((unsigned)((threads + 255) / 256)), \
                                      dim3(256), 0, 0, x, base, coef, o, rows, Tin, Tout, chunks);                    \
               }, reps));                                                                                             \
    }
#define QUADS(NV_)                 \
    QUADU(NV_, 8, true, 8)         \
    QUADU(NV_, 8, false, 8)        \
    WAVEK(NV_, 8, 1, false)        \
    WAVEK(NV_, 8, 1, true)         \
    WAVEK(NV_, 4, 1, true)         \
    WAVEK(NV_, 8, 2, false)        \
    WAVEK(NV_, 8, 2, true)         \
    WAVEK(NV_, 4, 2, true)         \
    QUADP(NV_, 8, false, 8, -1)    \
    QUADP(NV_, 8, true, 8, -1)     \
    QUADP(NV_, 4, true, 4, -1)     \
    QUADP(NV_, 8, true, 16, -1)    \
    QUADP(NV_, 4, true, 8, -1)     \
    QUADP(NV_, 8, true, 32, -1)    \
    PIPE(NV_, 4, false, 32)        \
    PIPE(NV_, 4, true, 32)         \
    PIPE(NV_, 4, true, 64)         \
    PIPE(NV_, 2, true, 32)         \
    PIPE(NV_, 8, true, 32)         \
    PIPE(NV_, 8, true, 64)         \
    QUAD(NV_, 8, false, 16)        \
    QUAD(NV_, 8, true, 16)         \
    QUADP(NV_, 8, false, 8, 3)     \
    QUADP(NV_, 8, true, 8, 3)      \
    QUADP(NV_, 4, false, 4, 3)     \
    QUADP(NV_, 4, true, 4, 3)      \
    QUADP(NV_, 8, false, 16, 3)    \
    QUADP(NV_, 8, true, 16, 3)     \
    QUADP(NV_, 8, true, 32, 3)     \
    QUAD(NV_, 8, false, 32)        \
    QUAD(NV_, 8, true, 32)         \
    QUAD(NV_, 4, false, 32)        \
    QUAD(NV_, 4, true, 32)         \
    QUAD(NV_, 2, false, 32)        \
    QUAD(NV_, 4, false, 16)        \
    QUAD(NV_, 4, false, 8)         \
    QUAD(NV_, 8, false, 8)         \
    QUAD(NV_, 4, false, 64)
        QUADS(1)
        QUADS(2)
        QUADS(3)
#define LDSK(NV_, RW_, NT_)                                                                                           \
    if (NV == NV_) {                                                                                                  \
        report("lds NV=" #NV_ " rows/wg=" #RW_ " nt=" #NT_, time_us([&] {                                             \
                   const float* x = xs[turn % NBUF]; float* o = os[turn % NBUF]; ++turn;                              \
                   hipLaunchKernelGGL((lds_kernel<NV_, RW_, NT_>), dim3((unsigned)((rows + RW_ - 1) / RW_)),          \
                                      dim3(256), (size_t)RW_ * Tin * 4, 0, x, base, coef, o, rows, Tin, Tout);        \
               }, reps));                                                                                             \
    }
        LDSK(2, 4, false) LDSK(2, 4, true) LDSK(2, 8, false) LDSK(2, 8, true)
        LDSK(3, 4, false) LDSK(3, 4, true) LDSK(3, 8, false) LDSK(3, 8, true)
        CHECK(hipFree(base));
        CHECK(hipFree(coef));
    }
    return 0;
}
