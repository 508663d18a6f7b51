// In-kernel phase timing of the weight-gradient kernel (debugging aid, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -DEEGGAN_TRACE scripts/wgrad_trace.hip eeg-gan_amd/csrc/lib.hip \
//         eeg-gan_amd/csrc/conv_ws.hip eeg-gan_amd/csrc/conv_thin.hip -o scripts/wgrad_trace
//   scripts/wgrad_trace [B [C [T [masked]]]]
#include "../eeg-gan_amd/csrc/conv_mfma.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512, C = argc > 2 ? atoi(argv[2]) : 64, T = argc > 3 ? atoi(argv[3]) : 1920;
    const bool masked = argc > 4 && atoi(argv[4]) != 0;
    const int K = 3;
    const size_t n = (size_t)B * C * T;
    float *x, *gy, *my;
    unsigned long long* tr;
    hipMalloc(&x, n * 4); hipMalloc(&gy, n * 4); hipMalloc(&my, n * 4);
    const size_t ntr = 4 * 8 * TRACE_STEPS * 8;
    hipMalloc(&tr, ntr * 8);
    std::vector<float> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(gy, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(my, hx.data(), n * 4, hipMemcpyHostToDevice);
    const size_t wsb = eeggan_conv1d_wgrad_workspace(B, C, C, T, K);
    void* ws;
    hipMalloc(&ws, wsb);
    eeggan_reduce_job jw, jb;
    for (int i = 0; i < 3; ++i)
        eeggan_conv1d_wgrad_partial(x, gy, 1, B, C, C, T, K, 0.1f, masked ? my : nullptr, nullptr, 0.2f, ws, wsb, &jw, &jb, nullptr);
    hipDeviceSynchronize();
    hipMemset(tr, 0, ntr * 8);
    g_wgrad_trace = tr;
    eeggan_conv1d_wgrad_partial(x, gy, 1, B, C, C, T, K, 0.1f, masked ? my : nullptr, nullptr, 0.2f, ws, wsb, &jw, &jb, nullptr);
    hipDeviceSynchronize();
    printf("B=%d C=%d T=%d splits=%d\n", B, C, T, jw.nsplit);
    std::vector<unsigned long long> h(ntr);
    hipMemcpy(h.data(), tr, ntr * 8, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 4; blk += 3)
        for (int wv = 0; wv < 4; ++wv) {
            const unsigned long long* p = &h[((blk * 8 + wv) * TRACE_STEPS) * 8];
            printf("block %d wave %d: start %lld prologue %llu\n", blk * 64, wv, (long long)(p[6] - h[6]), p[7] - p[6]);
            for (int s = 0; s < TRACE_STEPS && p[s * 8 + 4]; ++s) {
                const unsigned long long* q = p + s * 8;
                printf("  it %2d: issue %5llu mfma %5llu commit %5llu barrier %5llu | end at %llu\n", s, q[1] - q[0], q[2] - q[1],
                       q[3] - q[2], q[4] - q[3], q[4] - h[6]);
            }
        }
    return 0;
}
