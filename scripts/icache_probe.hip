// What straight-line code costs at the start of a kernel: a wave runs N dependent multiply-adds once, either as
// N unrolled instructions (code fetched cold, used once) or as a loop over 64 of them (the same work from a few
// cache lines).  256 workgroups of 512 threads like the conv kernels, launched back to back in a captured graph,
// alternating with a second large kernel so that no launch finds its own code in the instruction cache.
// (measurement aid, not part of the library)
//   hipcc -O3 --offload-arch=gfx950 scripts/icache_probe.hip -o scripts/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, bool STRAIGHT, int SALT>
__global__ void __launch_bounds__(512) fma_chain(float* p, float a, float b) {
    float v = (float)threadIdx.x + (float)SALT;
    if (STRAIGHT) {
#pragma unroll
        for (int i = 0; i < N; ++i) v = __builtin_fmaf(v, a, b + (float)(i & 7));
    } else {
#pragma unroll 1
        for (int j = 0; j < N / 64; ++j) {
#pragma unroll
            for (int i = 0; i < 64; ++i) v = __builtin_fmaf(v, a, b + (float)(i & 7));
        }
    }
    if (v == 12345.678f) p[0] = v;
}

template <typename F, typename G>
static float time_pair(hipStream_t st, F launch_a, G launch_b, int pairs) {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < pairs; ++i) {
        launch_a();
        launch_b();
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    return ms * 1000.f / pairs;
}

template <int N>
static void probe(hipStream_t st, float* d) {
    auto s1 = [&] { hipLaunchKernelGGL((fma_chain<N, true, 1>), dim3(256), dim3(512), 0, st, d, 1.0001f, 0.5f); };
    auto s2 = [&] { hipLaunchKernelGGL((fma_chain<N, true, 2>), dim3(256), dim3(512), 0, st, d, 1.0001f, 0.5f); };
    auto l1 = [&] { hipLaunchKernelGGL((fma_chain<N, false, 1>), dim3(256), dim3(512), 0, st, d, 1.0001f, 0.5f); };
    auto l2 = [&] { hipLaunchKernelGGL((fma_chain<N, false, 2>), dim3(256), dim3(512), 0, st, d, 1.0001f, 0.5f); };
    const float same = time_pair(st, s1, s1, 100), alt = time_pair(st, s1, s2, 100), loop = time_pair(st, l1, l2, 100);
    printf("N %5d: unrolled, same kernel twice %.2f us per pair; two different unrolled kernels %.2f; looped %.2f\n", N,
           same, alt, loop);
}

int main() {
    float* d;
    if (hipMalloc(&d, 1024) != hipSuccess) return 1;
    hipStream_t st;
    hipStreamCreate(&st);
    probe<512>(st, d);
    probe<1024>(st, d);
    probe<2048>(st, d);
    probe<4096>(st, d);
    probe<8192>(st, d);
    return 0;
}
