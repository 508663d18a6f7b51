// Argument block shared by the forward/input-gradient convolution kernels.
#pragma once
#include "common.h"

#include <atomic>

namespace eeggan {

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* drop;
    float* y;
    const float* mask_y;
    const float* mask_drop;
    float mask_slope;
    int B, Cin, Cout, T;
    int ST;  // 32-wide subtiles per sample = ceil(T/32)
    int NS;  // total subtiles = B*ST
    float in_scale;
    int act;
    float act_param;
    const float* act_aux;
    int vec;  // T % 4 == 0: 16-byte global loads are aligned
    float* pn_d;   // ACT_LRELU_PN: per-(sample, time) normaliser output [B, T]
    float pn_eps;
    int wpb;    // weight-stationary kernel: waves per workgroup that take work units
    // bf16 storage mode: x and mask_y (x_bf16), y and act_aux (y_bf16) hold bfloat16 instead of float32; the
    // pointers above then address 2-byte elements
    int x_bf16, y_bf16;
#ifdef EEGGAN_TRACE
    unsigned long long* trace;  // [4 traced blocks][8 waves][TRACE_STEPS][8] shader-clock stamps
#endif
};

#ifdef EEGGAN_TRACE
constexpr int TRACE_STEPS = 16;
#define EEGGAN_STAMP(step, k)                                                                              \
    do {                                                                                                   \
        if (a.trace && (blockIdx.x & 63) == 0 && (blockIdx.x >> 6) < 4 && (threadIdx.x & 63) == 0 &&       \
            (step) < TRACE_STEPS)                                                                          \
            a.trace[((((blockIdx.x >> 6) * 8 + (threadIdx.x >> 6)) * TRACE_STEPS) + (step)) * 8 + (k)] =   \
                __builtin_amdgcn_s_memtime();                                                              \
    } while (0)
#else
#define EEGGAN_STAMP(step, k) \
    do {                      \
    } while (0)
#endif

// Internal epilogue code of the weight-stationary kernel: LeakyReLU, then PixelNorm over the output
// channels (all of them live in one work unit), then the multiplier.
constexpr int ACT_LRELU_PN = 4;

// Weight-stationary variant for layers with at most 64 input and 64 output channels (conv_ws.hip).
// Returns -1 when the shape is not covered, otherwise the launch status.
int launch_conv_ws(const ConvArgs& a, int K, hipStream_t st);
bool conv_ws_covers(const ConvArgs& a, int K);
int num_cus();  // compute units of the CURRENT device
int product_pieces();  // pieces per operand of the bf16 matrix products: 3 (fp32 products), 1 (bf16 mode), 0 (fp32 MFMA)
// One-time launch set-up (LDS opt-in, occupancy) is kept per device in arrays of this many atomics: the
// entry points are called from autograd's backward thread as well, and a process may drive several GPUs.
// The set-up calls are idempotent, so two threads racing through one only repeat it.
constexpr int MAX_DEVICES = 64;
int current_device();  // hipGetDevice, clamped to [0, MAX_DEVICES)
// Bandwidth-bound form for 1x1 convs with at most 8 input or output channels (conv_thin.hip); -1 when
// the shape is not covered.
int launch_conv_thin(const ConvArgs& a, int K, hipStream_t st);

}  // namespace eeggan
