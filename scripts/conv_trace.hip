// In-kernel phase timing of the forward conv kernel (debugging aid, not part of the library).
//   hipcc -O3 --offload-arch=gfx950 -DEEGGAN_TRACE scripts/conv_trace.hip eeg-gan_amd/csrc/lib.hip eeg-gan_amd/csrc/conv_thin.hip -o scripts/conv_trace
#include "../eeg-gan_amd/csrc/conv_mfma.hip"
#include "../eeg-gan_amd/csrc/conv_ws.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
    const int B = 32, C = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 1920, K = 3;
    const size_t n = (size_t)B * C * T;
    float *x, *w, *bias, *drop, *y;
    unsigned long long* tr;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, (size_t)C * C * K * 4);
    hipMalloc(&bias, C * 4); hipMalloc(&drop, (size_t)B * C * 4);
    const size_t ntr = 4 * 8 * TRACE_STEPS * 8;
    hipMalloc(&tr, ntr * 8);
    std::vector<float> hx(n);
    for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hx.data(), (size_t)C * C * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hx.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(drop, hx.data(), (size_t)B * C * 4, hipMemcpyHostToDevice);
    ConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.drop = drop; a.y = y;
    a.B = B; a.Cin = C; a.Cout = C; a.T = T; a.ST = (T + 31) / 32; a.NS = B * a.ST;
    a.in_scale = 0.1f; a.act = EEGGAN_ACT_LRELU; a.act_param = 0.2f; a.vec = 1;
    a.trace = nullptr;
    for (int i = 0; i < 5; ++i) dispatch_fwd<3>(a, 0);
    hipDeviceSynchronize();
    {   // un-traced launches back to back: the kernel's duration as the bench line reports it
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) dispatch_fwd<3>(a, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        printf("back to back: %.2f us per launch\n", ms * 1000.f / 50);
    }
    hipMemset(tr, 0, ntr * 8);
    a.trace = tr;
    dispatch_fwd<3>(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(ntr);
    hipMemcpy(h.data(), tr, ntr * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"loads", "mfma", "epilogue", "commit"};
    for (int blk = 0; blk < 4; blk += 3)
        for (int wv = 0; wv < 8; ++wv) {
            const unsigned long long* p = &h[((blk * 8 + wv) * TRACE_STEPS) * 8];
            printf("block %d wave %d: start %llu prologue %llu (w issued %llu, x issued %llu, w staged %llu, barrier %llu)\n", blk * 64, wv, p[6] - h[6], p[7] - p[6],
                   p[15 * 8 + 0] - p[6], p[15 * 8 + 1] - p[6], p[15 * 8 + 2] - p[6], p[15 * 8 + 3] - p[6]);
            for (int s = 0; s < TRACE_STEPS - 1 && p[s * 8 + 3]; ++s) {
                const unsigned long long* q = p + s * 8;
                printf("  unit %d:", s);
                for (int k = 0; k < 4; ++k)
                    if (q[k + 1]) printf(" %s %llu", names[k], q[k + 1] - q[k]);
                printf("  | end at %llu\n", q[3] - h[6]);
            }
        }
    return 0;
}
