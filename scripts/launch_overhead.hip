// Fixed cost of a kernel launch on this chip as a function of the launch shape: an empty kernel launched back to
// back on one stream (each launch waits for the one before it, like the kernels of a layer chain), timed with one
// pair of events.  (measurement aid, not part of the library)
//   hipcc -O3 --offload-arch=gfx950 scripts/launch_overhead.hip -o scripts/launch_overhead
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ float smem[];
__global__ void empty_kernel(float* p, int touch) {
    if (touch && threadIdx.x == 0) smem[0] = 1.f;
    if (p && blockIdx.x == 0 && threadIdx.x == 0) p[0] = touch ? smem[0] : 0.f;
}

int main() {
    float* d;
    if (hipMalloc(&d, 1024) != hipSuccess) return 1;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int shapes[][3] = {{256, 512, 160 * 1024}, {256, 512, 80 * 1024}, {256, 512, 0}, {512, 256, 80 * 1024}, {256, 256, 0},
                             {256, 64, 0}, {1, 64, 0}, {2048, 256, 0}, {1024, 512, 0}};
    for (auto& s : shapes) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(s[0]), dim3(s[1]), s[2], 0, d, 1);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %5d x %3d threads, %6d B LDS: %.2f us per launch\n", s[0], s[1], s[2], ms * 1000.f / 200);
        }
    }
    // the same through a captured graph of 200 dependent launches
    hipStream_t st;
    hipStreamCreate(&st);
    for (auto& s : shapes) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(s[0]), dim3(s[1]), s[2], st, d, 1);
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
        printf("graph: grid %5d x %3d threads, %6d B LDS: %.2f us per launch\n", s[0], s[1], s[2], ms * 1000.f / 200);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
    }
    return 0;
}
