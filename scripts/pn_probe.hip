// Timing aid: tile shapes for the per-column channel norm at [B, 64, 1920] beside a plain copy of the same bytes.
// build: hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops scripts/pn_probe.hip -o scripts/pn_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ a, f32x4* __restrict__ o, size_t items) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < items) {
        const f32x4 v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        if (NTS)
            __builtin_nontemporal_store(v, o + i);
        else
            o[i] = v;
    }
}

// workgroup = CG channel groups x UL column units of 4 time steps; CPT channels per thread in registers
template <int CG, int CPT, int UL, bool NT, bool NTS = NT>
__global__ __launch_bounds__(CG* UL) void pn_tile_kernel(const float* __restrict__ h, const float* __restrict__ drop,
                                                         float* __restrict__ y, float* __restrict__ d, int B, int C, int T,
                                                         float eps) {
    __shared__ float red[CG][UL][4];
    const int cg = threadIdx.x / UL, ul = threadIdx.x % UL;
    const int upr = T / 4;
    const long long unit = (long long)blockIdx.x * UL + ul;
    const bool ok = unit < (long long)B * upr;
    const int b = ok ? (int)(unit / upr) : 0;
    const int t = ok ? (int)(unit - (long long)b * upr) * 4 : 0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 keep[CPT];
    if (ok) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const f32x4* p = reinterpret_cast<const f32x4*>(h + ((long long)b * C + cg + i * CG) * T + t);
            keep[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += keep[i][k] * keep[i][k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[cg][ul][k] = s[k];
    __syncthreads();
    if (!ok) return;
    float dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float tot = 0.f;
#pragma unroll
        for (int g = 0; g < CG; ++g) tot += red[g][ul][k];
        dv[k] = rsqrtf(tot / (float)C + eps);
    }
    if (cg == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[(long long)b * T + t + k] = dv[k];
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg + i * CG;
        const float r = drop[b * C + c];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = keep[i][k] * dv[k] * r;
        f32x4* q = reinterpret_cast<f32x4*>(y + ((long long)b * C + c) * T + t);
        if (NTS)
            __builtin_nontemporal_store(o, q);
        else
            *q = o;
    }
}

// a wave on its own: 16 lanes x 4 time steps = 64 columns, 4 channel rows per load instruction, 16 instructions for
// the 64 channels; the four row groups meet through lane exchanges, no LDS and no workgroup barrier
template <bool NT>
__global__ __launch_bounds__(256) void pn_wave_kernel(const float* __restrict__ h, const float* __restrict__ drop,
                                                      float* __restrict__ y, float* __restrict__ d, int B, int T,
                                                      float eps) {
    constexpr int C = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane >> 4, ul = lane & 15;
    const int upr = T / 4;
    const long long unit = ((long long)blockIdx.x * 4 + wave) * 16 + ul;
    const bool ok = unit < (long long)B * upr;
    const int b = ok ? (int)(unit / upr) : 0;
    const int t = ok ? (int)(unit - (long long)b * upr) * 4 : 0;
    f32x4 keep[16];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x4* p = reinterpret_cast<const f32x4*>(h + ((long long)b * C + rg + i * 4) * T + t);
        keep[i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += keep[i][k] * keep[i][k];
    float dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = s[k];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        dv[k] = rsqrtf(v / (float)C + eps);
    }
    if (!ok) return;
    if (rg == 0) *reinterpret_cast<f32x4*>(d + (long long)b * T + t) = f32x4{dv[0], dv[1], dv[2], dv[3]};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = rg + i * 4;
        const float r = drop[b * C + c];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = keep[i][k] * dv[k] * r;
        f32x4* q = reinterpret_cast<f32x4*>(y + ((long long)b * C + c) * T + t);
        if (NT)
            __builtin_nontemporal_store(o, q);
        else
            *q = o;
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
    const int B = argc > 1 ? atoi(argv[1]) : 512, C = 64, T = 1920;
    const size_t n = (size_t)B * C * T;
    // NBUF distinct inputs and outputs taken in turn: no launch finds its input in a cache because the launch before
    // read the same bytes (argv[2] = 1: one pair, what a loop over one tensor measures)
    const int NBUF = argc > 2 ? atoi(argv[2]) : 4;
    float *hs[8], *ys[8], *d, *drop;
    for (int i = 0; i < NBUF; ++i) {
        CHECK(hipMalloc(&hs[i], n * 4));
        CHECK(hipMalloc(&ys[i], n * 4));
    }
    int turn = 0;
    float *h = hs[0], *y = ys[0];
    CHECK(hipMalloc(&d, (size_t)B * T * 4));
    CHECK(hipMalloc(&drop, (size_t)B * C * 4));
    std::vector<float> host(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        host[i] = (float)(s >> 8) * (1.f / 8388608.f) - 1.f;
    }
    for (int i = 0; i < NBUF; ++i) CHECK(hipMemcpy(hs[i], host.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> ones((size_t)B * C, 1.25f);
    CHECK(hipMemcpy(drop, ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
    const double bytes = 2.0 * 4 * n;
    const int reps = B >= 256 ? 12 : 48;
    const long long units = (long long)B * T / 4;
    auto report = [&](const char* name, float us) {
        printf("%-34s %8.1f us  %6.0f GB/s  %.2f of 8 TB/s\n", name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
        fflush(stdout);
    };
#define COPY(NTL, NTS)                                                                                                \
    report("copy, 16 B per thread, nt load=" #NTL " store=" #NTS, time_us([&] {                                       \
               h = hs[turn % NBUF]; y = ys[turn % NBUF]; ++turn;                                                      \
               hipLaunchKernelGGL((copy_kernel<NTL, NTS>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0,    \
                                  reinterpret_cast<const f32x4*>(h), reinterpret_cast<f32x4*>(y), n / 4);             \
           }, reps))
    COPY(false, false);
    COPY(true, false);
    COPY(false, true);
    COPY(true, true);
#define TILE2(CG, CPT, UL, NT, NTS)                                                                                   \
    report("tile CG=" #CG " CPT=" #CPT " UL=" #UL " ntl=" #NT " nts=" #NTS, time_us([&] {                             \
               h = hs[turn % NBUF]; y = ys[turn % NBUF]; ++turn;                                                      \
               hipLaunchKernelGGL((pn_tile_kernel<CG, CPT, UL, NT, NTS>), dim3((unsigned)((units + UL - 1) / UL)),    \
                                  dim3(CG * UL), 0, 0, h, drop, y, d, B, C, T, 1e-8f);                                \
           }, reps))
    TILE2(4, 16, 64, true, false);
    TILE2(4, 16, 64, false, true);
    TILE2(8, 8, 32, true, true);
    TILE2(16, 4, 16, true, true);
#define TILE(CG, CPT, UL, NT)                                                                                         \
    report("tile CG=" #CG " CPT=" #CPT " UL=" #UL " nt=" #NT, time_us([&] {                                           \
               h = hs[turn % NBUF]; y = ys[turn % NBUF]; ++turn;                                                      \
               hipLaunchKernelGGL((pn_tile_kernel<CG, CPT, UL, NT>), dim3((unsigned)((units + UL - 1) / UL)),         \
                                  dim3(CG * UL), 0, 0, h, drop, y, d, B, C, T, 1e-8f);                                \
           }, reps))
    TILE(4, 16, 64, false);
    TILE(4, 16, 64, true);
    TILE(8, 8, 32, false);
    TILE(8, 8, 64, false);
    TILE(16, 4, 64, false);
    TILE(16, 4, 16, false);
    TILE(2, 32, 128, false);
    TILE(4, 16, 32, false);
    TILE(4, 16, 16, false);
    TILE(8, 8, 16, false);
    report("wave 64 ch x 64 columns", time_us([&] {
               h = hs[turn % NBUF]; y = ys[turn % NBUF]; ++turn;
               hipLaunchKernelGGL((pn_wave_kernel<false>), dim3((unsigned)((units + 63) / 64)), dim3(256), 0, 0, h, drop, y,
                                  d, B, T, 1e-8f);
           }, reps));
    report("wave 64 ch x 64 columns nt", time_us([&] {
               h = hs[turn % NBUF]; y = ys[turn % NBUF]; ++turn;
               hipLaunchKernelGGL((pn_wave_kernel<true>), dim3((unsigned)((units + 63) / 64)), dim3(256), 0, 0, h, drop, y,
                                  d, B, T, 1e-8f);
           }, reps));
    return 0;
}
