/*
 * eeggan_hip.h -- C ABI of libeeggan_hip.so: the MI355X (gfx950) kernels behind the
 * EEG-GAN progressive 1-D WGAN-GP training step.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous float32 unless stated otherwise;
 *     activations are [B, C, T] with T fastest, conv weights [Cout, Cin, K];
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every entry
 *     point only enqueues work on it: no allocation, no synchronisation, no host copies,
 *     so a caller may capture any of them into a hipGraph;
 *   - return value 0 = success, non-zero = error; eeggan_last_error() returns the message
 *     of the last failure on the calling thread.  Shape/argument errors are detected on
 *     the host BEFORE anything is launched.
 *
 * Each declaration cites the reference operation it replaces (path:line in the
 * reference repository).
 */
#ifndef EEGGAN_HIP_H
#define EEGGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ---------------------------------------------------------------------- */
const char* eeggan_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int eeggan_abi_version(void);

/* activation codes for the conv epilogue */
#define EEGGAN_ACT_NONE  0
#define EEGGAN_ACT_LRELU 1   /* act_param = negative slope (nn.LeakyReLU, layers.py:196-200) */
#define EEGGAN_ACT_TANH  2   /* act_param = input scale   (ScaledTanh, layers.py:17-23)      */
#define EEGGAN_ACT_MASK  3   /* v *= (act_aux[b,co,t] > 0 ? 1 : act_param): the LeakyReLU mask of a
                                saved layer output applied to this conv's OUTPUT (second-order pass) */

/* ---- equalized-lr Conv1d, stride 1, "same" zero padding pad=(K-1)/2, K in {1,3} ------
 * Replaces EqualizedConv1d.forward (layers.py:173-174) + the epilogue chain of
 * GeneralConv.forward (layers.py:207-213):
 *   y[b,co,t] = drop[b,co] * act( bias[co] + sum_{ci,j} w[co,ci,j] * in_scale * x[b,ci,t+j-pad] )
 * bias and drop may be NULL.  x, w (and mask_y) must be 16-byte aligned.
 * If mask_y != NULL the input is first multiplied elementwise by
 *   mask_drop[b,ci] * (mask_y[b,ci,t] > 0 ? 1 : mask_slope)
 * (mask_drop may be NULL), i.e. the backward of a fused LeakyReLU+GDrop epilogue
 * (layers.py:26-38,196-204) applied on the fly.
 * The INPUT GRADIENT of this convolution (autograd of nn.Conv1d, layers.py:154-155) is the same
 * entry point called with gy as x and the weight view wt[ci,co,j] = w[co,ci,K-1-j] produced by
 * eeggan_conv1d_flip_weights (valid until w changes):
 *   gx[b,ci,t] = in_scale * sum_{co,j} w[co,ci,j] * gy[b,co,t-j+pad]. */
int eeggan_conv1d(const float* x, const float* w, const float* bias, const float* drop, float* y,
                  int B, int Cin, int Cout, int T, int K, float in_scale,
                  int act, float act_param, const float* act_aux,
                  const float* mask_y, const float* mask_drop, float mask_slope,
                  void* stream);
int eeggan_conv1d_flip_weights(const float* w, float* wt, int Cout, int Cin, int K, void* stream);
/* The generator's layer in one kernel (GeneralConv.forward with act_norm='pixel', layers.py:196-213):
 *   h = lrelu(bias + conv(in_scale*x, w));  d[b,t] = rsqrt(mean_co h^2 + eps);  y = drop * h * d
 * y [B,Cout,T] and the normaliser d [B,T] (what eeggan_pixelnorm_drop_bwd needs) are written; h is not.
 * Only for shapes whose output channels all fit one work unit of the weight-stationary kernel
 * (K = 3, Cin <= 64, 32 < Cout <= 64, T % 4 == 0, B*ceil(T/32) >= 256): ask _supported first. */
int eeggan_conv1d_pixelnorm_supported(int B, int Cin, int Cout, int T, int K);
int eeggan_conv1d_pixelnorm(const float* x, const float* w, const float* bias, const float* drop,
                            float* y, float* d, int B, int Cin, int Cout, int T, int K,
                            float in_scale, float slope, float eps, void* stream);
/* The same for a table of weight tensors (host array, copied into the kernel arguments) in one
 * launch: all views of a network go stale together when its optimizer steps (trainer.py:90,94). */
typedef struct {
    const float* w; float* wt; int Cout, Cin, K, reserved;
} eeggan_flip_desc;
int eeggan_conv1d_flip_weights_multi(const eeggan_flip_desc* table, int ntensors, void* stream);
/* Arithmetic of the matrix products of the predicate-free k = 3 conv layers (forward, input gradient and weight
 * gradient; they carry nearly all of the step's flops), process-wide, for the launches that follow:
 *   3  bf16 matrix instructions on three bf16 pieces per fp32 operand: fp32 products (default; the reference is
 *      fp32 throughout, layers.py:150-174)
 *   1  bf16 compute mode: operands rounded to bf16, one product; fp32 accumulation, fp32 tensors in memory
 *   0  fp32 matrix instructions (comparison runs)
 * Returns non-zero for any other value.  The ragged layers (5 or 257 channels), the k = 1 and the full-length
 * layers use fp32 matrix instructions in every mode. */
int eeggan_set_matrix_precision(int pieces);
int eeggan_get_matrix_precision(void);

/* Weight- and bias-gradient of the same convolution (autograd of nn.Conv1d, layers.py:154-155):
 *   gw[co,ci,j] = in_scale * sum_{b,t} gy[b,co,t] * x[b,ci,t+j-pad],  gb[co] = sum_{b,t} gy[b,co,t]
 * gb may be NULL.  If mask_y != NULL, gy is first multiplied by
 * mask_drop[b,co] * (mask_y[b,co,t] > 0 ? 1 : mask_slope) as above.
 * The results are ACCUMULATED when the factors are non-zero: gw = beta_w * gw + (the sum above),
 * gb = beta_b * gb + (...); with a factor of 0 the old contents are never read (torch's autograd
 * does this accumulation with an extra add kernel per contribution, reference trainer.py:88-94).
 * `workspace` must hold eeggan_conv1d_wgrad_workspace(...) bytes; the sum over the
 * partial slabs is done in a fixed order, so results are run-to-run deterministic. */
size_t eeggan_conv1d_wgrad_workspace(int B, int Cin, int Cout, int T, int K);
int eeggan_conv1d_wgrad(const float* x, const float* gy, float* gw, float* gb,
                        int B, int Cin, int Cout, int T, int K, float in_scale,
                        const float* mask_y, const float* mask_drop, float mask_slope,
                        float beta_w, float beta_b,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The two halves of eeggan_conv1d_wgrad for callers that collect many weight gradients and sum their
 * partial slabs later in few launches (one backward pass of the critic produces ~50 of them):
 * eeggan_conv1d_wgrad_partial runs only the slab kernel and fills job_w (and job_b when want_bias)
 * with slabs/elems/nsplit/scale; the caller sets `out` and `beta` and hands tables of jobs to
 * eeggan_wgrad_reduce_multi:  out = beta * out + scale * (sum of the nsplit slabs), fixed order.
 * Jobs of one call must write distinct outputs; the workspace must stay untouched until then. */
typedef struct {
    const float* slabs; float* out; int64_t elems; int nsplit; float scale; float beta; int reserved;
} eeggan_reduce_job;
int eeggan_conv1d_wgrad_partial(const float* x, const float* gy, int want_bias,
                                int B, int Cin, int Cout, int T, int K, float in_scale,
                                const float* mask_y, const float* mask_drop, float mask_slope,
                                void* workspace, size_t workspace_bytes,
                                eeggan_reduce_job* job_w, eeggan_reduce_job* job_b, void* stream);
int eeggan_wgrad_reduce_multi(const eeggan_reduce_job* jobs, int njobs, void* stream);
/* eeggan_conv1d_wgrad_partial for a group of independent layers in as few launches as their kernel
 * variants allow (one for layers with channels % 64 == 0, T % 32 == 0, K = 3): the chip's workgroups
 * are dealt out to the jobs in proportion to their work, so each job is cut into fewer splits than on
 * its own -- fewer slabs to write and to sum, one launch latency per group.  Arguments per job as in
 * eeggan_conv1d_wgrad_partial (workspace sized by eeggan_conv1d_wgrad_workspace); jobs_w[i] / jobs_b[i]
 * receive job i's reduce descriptors (jobs_b[i].elems = 0 without want_bias).  x, gy and the masks of
 * every job must stay untouched until the call; a backward pass therefore queues its weight gradients
 * and submits them in groups (the reference computes them one by one inside autograd,
 * layers.py:173-174 under loss.backward(), trainer.py:89). */
typedef struct {
    const float* x; const float* gy; const float* mask_y; const float* mask_drop;
    void* workspace; size_t workspace_bytes;
    int B, Cin, Cout, T, K, want_bias;
    float in_scale, mask_slope;
    int x_bf16, gy_bf16;   /* bf16 storage mode: x / gy (and mask_y) point to bfloat16 instead of float32 */
} eeggan_wgrad_job;
int eeggan_conv1d_wgrad_partial_multi(const eeggan_wgrad_job* jobs, int njobs,
                                      eeggan_reduce_job* jobs_w, eeggan_reduce_job* jobs_b, void* stream);

/* ---- depthwise Conv1d (groups = channels): the per-channel half of EqualizedSeparableConv1d (layers.py:137-147;
 * its 1x1 half is eeggan_conv1d with K = 1) ----------------------------------------------------------------
 *   y[b,c,t] = bias[c] + in_scale * sum_j w[c, flip ? K-1-j : j] * x[b,c,t+j-pad],  t in [0, Tin + 2 pad - K + 1)
 * w [C, 1, K]; any K <= 64 and 0 <= pad < K (the full-length layers are K = 32 with pad = 31 or 0).  The input
 * gradient is the same call on gy with flip toggled and pad' = K-1-pad (bias NULL).
 *   wgrad: gw[c, flip ? K-1-j : j] = in_scale * sum_{b,t} gy[b,c,t] * x[b,c,t+j-pad];  gb[c] = sum_{b,t} gy[b,c,t]
 * (gb may be NULL); sums in a fixed order. */
int eeggan_dwconv1d(const float* x, const float* w, const float* bias, float* y, int B, int C, int Tin, int K,
                    int pad, float in_scale, int flip, void* stream);
int eeggan_dwconv1d_wgrad(const float* x, const float* gy, float* gw, float* gb, int B, int C, int Tin, int K,
                          int pad, float in_scale, int flip, void* stream);

/* ---- G head: Conv1d(k=KT, pad=KT-1) on a length-1 signal (network.py:158-159,302-303) --
 *   y[b,co,t] = drop[b,co]*act(bias[co] + in_scale * sum_ci w[co,ci,KT-1-t] * z[b,ci]),  t in [0,KT)
 * wgrad: gw[co,ci,KT-1-t] = in_scale * sum_b gy[b,co,t] * z[b,ci];  gb[co] = sum_{b,t} gy[b,co,t]
 * dgrad: gz[b,ci] = in_scale * sum_{co,t} w[co,ci,KT-1-t] * gy[b,co,t] */
int eeggan_head_fwd(const float* z, const float* w, const float* bias, const float* drop, float* y,
                    int B, int Cin, int Cout, int KT, float in_scale, int act, float act_param,
                    void* stream);
int eeggan_head_wgrad(const float* z, const float* gy, float* gw, float* gb,
                      int B, int Cin, int Cout, int KT, float in_scale, void* stream);
int eeggan_head_dgrad(const float* gy, const float* w, float* gz,
                      int B, int Cin, int Cout, int KT, float in_scale, void* stream);

/* ---- D tail: Conv1d(k=KT, pad=0) on a length-KT signal (network.py:416-418) -----------
 *   y[b,co] = drop[b,co]*act(bias[co] + in_scale * sum_{ci,t} w[co,ci,t] * x[b,ci,t])
 * dgrad: gx[b,ci,t] = in_scale * sum_co w[co,ci,t] * gy[b,co]
 * wgrad: gw[co,ci,t] = in_scale * sum_b gy[b,co] * x[b,ci,t];  gb[co] = sum_b gy[b,co]
 * (B <= 1024 for the weight gradient.) */
int eeggan_tail_fwd(const float* x, const float* w, const float* bias, const float* drop, float* y,
                    int B, int Cin, int Cout, int KT, float in_scale, int act, float act_param,
                    void* stream);
/* The same on the matrix cores with the reduction split over workgroups (Cin*KT a multiple of 128, Cout of 32, x and w
 * 16-byte aligned; other shapes run eeggan_tail_fwd): partial sums go through ``workspace`` of at least
 * eeggan_tail_fwd_workspace(...) bytes (0: the shape takes no workspace) and are summed in a fixed order. */
size_t eeggan_tail_fwd_workspace(int B, int Cin, int Cout, int KT);
int eeggan_tail_fwd_ws(const float* x, const float* w, const float* bias, const float* drop, float* y, int B, int Cin,
                       int Cout, int KT, float in_scale, int act, float act_param, void* workspace, size_t workspace_bytes,
                       void* stream);
int eeggan_tail_dgrad(const float* gy, const float* w, float* gx,
                      int B, int Cin, int Cout, int KT, float in_scale, void* stream);
int eeggan_tail_wgrad(const float* x, const float* gy, float* gw, float* gb,
                      int B, int Cin, int Cout, int KT, float in_scale, void* stream);

/* ---- elementwise / normalisation ----------------------------------------------------- */
/* out = g * drop[b,c] * (y > 0 ? 1 : slope): backward of LeakyReLU (layers.py:200) followed
 * by GDrop (layers.py:32-38); drop may be NULL.  n = B*C*T, ct = T. */
int eeggan_lrelu_drop_bwd(const float* g, const float* y, const float* drop, float* out,
                          int B, int C, int T, float slope, void* stream);
/* PixelNorm (utils.py:133-136) followed by GDrop:  d[b,t] = rsqrt(mean_c h^2 + eps),
 * y = h * d * drop[b,c].  drop may be NULL; d (size B*T) is saved for the backward. */
int eeggan_pixelnorm_drop_fwd(const float* h, const float* drop, float* y, float* d,
                              int B, int C, int T, float eps, void* stream);
/* Backward of LeakyReLU -> PixelNorm -> GDrop w.r.t. the conv output, from the layer
 * OUTPUT y: with u = gy*drop, p = y/drop:  gh = d*(u - p*mean_c(u*p)); gc = gh*(y>0?1:slope).
 * slope = 1 gives the backward of PixelNorm->GDrop alone. */
int eeggan_pixelnorm_drop_bwd(const float* gy, const float* y, const float* drop, const float* d,
                              float* gc, int B, int C, int T, float slope, void* stream);
/* y = tanh(scale * x): ScaledTanh.forward on its own (layers.py:17-23); on the training path it is the
 * epilogue of the to_rgb conv (EEGGAN_ACT_TANH). */
int eeggan_scaled_tanh(const float* x, float* y, size_t n, float scale, void* stream);
/* gc = gy * scale * (1 - y^2): backward of y = tanh(scale*c) (layers.py:17-23). */
int eeggan_tanh_bwd(const float* gy, const float* y, float* gc, size_t n, float scale, void* stream);
/* out = ca*a + cb*b (b may be NULL: out = ca*a).  Fade-in blends network.py:393-396,533-540. */
int eeggan_axpby(const float* a, float ca, const float* b, float cb, float* out, size_t n, void* stream);
/* The same blends with the fade-in factor read from DEVICE memory (alpha[0]), so that a captured
 * launch follows the factor the depth manager updates at every iteration (plugins.py:115-117):
 *   mode 0: out = alpha*a + (1-alpha)*b;  mode 1: out = alpha*a;  mode 2: out = (1-alpha)*a  (b unused). */
int eeggan_blend(const float* a, const float* b, const float* alpha, int mode, float* out, size_t n,
                 void* stream);
/* Channel windows: over the whole of dst [B, Cd, T],
 *   dst[b, cd, t] = (base ? base[b, cd, t] : 0) + (d0 <= cd < d0 + n ? src[b, s0 + cd - d0, t] : 0)
 * with src [B, Cs, Ts], Ts = T or 1 (one sample repeated over time); base NULL, a [B, Cd, T] tensor, or dst itself.  The skips of the
 * reference's `deep` blocks are made of it: y[:, :C] + x (PassChannelResidual, layers.py:216-223) and
 * h + cat([x, net(x)], dim=1) (ConcatResidual, layers.py:226-239), and so are their gradients (a slice is a window
 * into a smaller dst, its transpose a window into a zero-filled larger one). */
int eeggan_channel_window(const float* src, const float* base, float* dst, int B, int Cs, int Cd, int T, int Ts,
                          int s0, int d0, int n, void* stream);
/* out[b,:] = mix[b]*real[b,:] + (1-mix[b])*fake[b,:]  (losses.py:107-108); per = C*T. */
int eeggan_mix_batch(const float* real, const float* fake, const float* mix, float* out,
                     int B, size_t per, void* stream);
/* out = base ** n elementwise (GDrop multipliers (1+strength)**N(0,1), layers.py:36). */
int eeggan_pow_base(const float* n, float* out, size_t count, float base, void* stream);
/* The same multipliers drawn on the device in one launch: out[i] = base ** z_i with z_i ~ N(0, 1) from a counter-based
 * generator (Philox-4x32-10, Box-Muller) keyed by (seed, salt).  state: EEGGAN_LOGNORMAL_STATE_WORDS 32-bit words of device memory, zero before
 * the first call and owned by one call site (launches that may run side by side need states of their own): the kernel
 * advances it by itself, so a captured launch draws a new stream at every replay (GDropLayer, layers.py:26-38). */
#define EEGGAN_LOGNORMAL_STATE_WORDS 256
int eeggan_lognormal_fill(float* out, size_t count, float base, unsigned long long seed, unsigned int salt,
                          unsigned int* state, void* stream);

/* ---- rational resampler (utils.py:204-251) --------------------------------------------
 * out[r, m] = sum_{k<taps} coef[m*taps+k] * x[r, idx[m*taps+k]],  r in [0,rows), m in [0,Tout).
 * The host builds (idx, coef) for "linear-interpolate x p then average-pool / q" and for its
 * transpose; both directions use this one gather kernel. */
int eeggan_resample(const float* x, const int32_t* idx, const float* coef, float* out,
                    size_t rows, int Tin, int Tout, int taps, void* stream);
/* The same operator in quad form, for tables whose taps are local (every resampler of this path): a thread
 * produces 4 consecutive outputs from the 4*nv input samples starting at base[q] (a multiple of 4, with
 * base[q] + 4*nv <= Tin):  out[r, 4q+e] = sum_{j<4nv} C(q, e, j) * x[r, base[q] + j]  -- 16-byte
 * loads and stores only, no address that depends on a table value but base.  Tin, Tout multiples of 4,
 * nv in [1,4]; x, out, coef 16-byte aligned.  coef is lane-interleaved: quads in groups of 64 (the last group
 * padded: ceil(Tout/4 / 64) * 64 * 16 * nv floats in all), and
 *   C(q, e, 4v + i) = coef[((((q / 64) * 4 + e) * nv + v) * 64 + q % 64) * 4 + i],
 * so that 64 consecutive quads (the lanes of a wavefront) read each of their vectors from 1 KB in one piece.
 * With the dense coefficients = the sparse ones laid out by input position it is the same sum in the same order
 * as eeggan_resample (equal up to how the compiler fuses the multiply-adds: the last bit or two). */
int eeggan_resample_quads(const float* x, const int32_t* base, const float* coef, float* out,
                          size_t rows, int Tin, int Tout, int nv, void* stream);

/* ---- MinibatchStddev (layers.py:67-90), one temporal window ------------------------------
 * fwd: out[B, C+1, T]: channels [0,C) = x, channel C = mean_{c,t} sqrt(mean_g (x - mean_g x)^2)
 *      over the G = min(B, group) interleaved members {m, m+B/G, ...} of sample b's group.
 * bwd: gx = gout[:, :C] + d(out[:,C])/dx^T gout[:, C]
 * bwd_bwd: given q = cotangent of gx:  ggout[B, C+1, T] and gx2[B, C, T] (the derivative of
 *      bwd w.r.t. gout and w.r.t. x). */
/* fwd and bwd_bwd need eeggan_mbstd_workspace(...) bytes of device scratch (partial sums). */
size_t eeggan_mbstd_workspace(int B, int C, int T, int group);
int eeggan_mbstd_fwd(const float* x, float* out, void* workspace, int B, int C, int T, int group,
                     void* stream);
/* eeggan_mbstd_fwd for callers that cannot allocate device memory (code under hipGraph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The partial
 * sums live in a 256 KiB buffer built into the library (one copy per GPU, held until the process ends), so
 * there is no workspace argument and the call makes no allocation of any kind; B/min(B,group) *
 * ceil(C*T/256) must not exceed 65536 floats, otherwise the call fails with a message and launches nothing.
 * All calls on one GPU share that buffer: calls that overlap on one GPU (two streams) race.  The first call
 * on a device resolves the buffer's address and must not be made while a stream is capturing. */
int eeggan_mbstd_fwd_static(const float* x, float* out, int B, int C, int T, int group, void* stream);
int eeggan_mbstd_bwd(const float* x, const float* gout, float* gx, int B, int C, int T, int group,
                     void* stream);
int eeggan_mbstd_bwd_bwd(const float* x, const float* gout, const float* q, float* ggout, float* gx2,
                         void* workspace, int B, int C, int T, int group, void* stream);
/* The same three passes on ONE temporal window, time steps [t0, t0 + Tw) of T: the reference's layer with
 * temporal_groups_per_window > 1 takes its statistics per window of kernel_size / temporal_groups_per_window steps
 * (layers.py:72, 81-89).  A call reads and writes only its window's columns of every tensor (all still [B, C(+1), T]),
 * so a signal of several windows takes one call per window; with t0 = 0, Tw = T they are the calls above.  The
 * workspace of eeggan_mbstd_workspace(B, C, T, group) is large enough for any window. */
int eeggan_mbstd_fwd_window(const float* x, float* out, void* workspace, int B, int C, int T, int t0, int Tw,
                            int group, void* stream);
int eeggan_mbstd_bwd_window(const float* x, const float* gout, float* gx, int B, int C, int T, int t0, int Tw,
                            int group, void* stream);
int eeggan_mbstd_bwd_bwd_window(const float* x, const float* gout, const float* q, float* ggout, float* gx2,
                                void* workspace, int B, int C, int T, int t0, int Tw, int group, void* stream);

/* ---- WGAN-GP loss (losses.py:94-100,114-119 and 47-48) -----------------------------------
 * loss = mean(d_fake) - mean(d_real) + drift*mean(d_real^2) + lambda/target^2 * mean((|g_b|_2 - target)^2)
 * g is [B, per]; norms[B] is written for the backward; loss is one float. */
int eeggan_wgan_gp_loss_fwd(const float* d_real, const float* d_fake, const float* g, float* norms,
                            float* loss, int B, size_t per, float drift, float lambda, float target,
                            void* stream);
/* gl = upstream scalar gradient (device pointer to one float). */
int eeggan_wgan_gp_loss_bwd(const float* gl, const float* d_real, const float* g, const float* norms,
                            float* g_d_real, float* g_d_fake, float* g_g,
                            int B, size_t per, float drift, float lambda, float target, void* stream);
/* The reference's other loss types (losses.py:47-60 generator, 82-101 critic), on two score vectors a, b of
 * length B: loss[0] = value, ga/gb = its gradient w.r.t. a and b for an upstream gradient of one.
 *   kind 1 hinge        mean relu(1 - a) + mean relu(1 + b)
 *        2 rsgan        mean softplus(-(a - b))
 *        3 rasgan       (mean softplus(-(a - mean b)) + mean softplus(b - mean a)) / 2
 *        4 rahinge      (mean relu(1 - (a - mean b)) + mean relu(1 + (b - mean a))) / 2
 *        5 wgan_theirs  mean b - mean a + eps (mean a + mean b)^2      (score part)
 *        6              relu(mean a - mean b)                          (wgan_theirs' gain on its penalty)
 * Critic: a = D(real), b = D(fake).  The generator's relativistic losses are the same functions with
 * a = D(fake), b = D(real).  eps is used by kind 5 only. */
int eeggan_score_loss(int kind, const float* a, const float* b, int B, float eps, float* loss, float* ga,
                      float* gb, void* stream);
/* out[i] = s[0] * v[i]: a stored loss gradient times the upstream scalar gradient (device scalar). */
int eeggan_scale_by_scalar(const float* s, const float* v, float* out, size_t n, void* stream);
/* The gradient penalty alone (losses.py:113-118): loss[0] = lambda/target^2 * mean(p_b^2) with
 * p_b = |g_b|_2 - target, or relu(|g_b|_2 - target) when one_sided (wgan_theirs).  g is [B, per];
 * norms[B] is written for the backward, which returns g_g = d loss / d g times gl[0]. */
int eeggan_grad_penalty_fwd(const float* g, float* norms, float* loss, int B, size_t per, float lambda,
                            float target, int one_sided, void* stream);
int eeggan_grad_penalty_bwd(const float* gl, const float* g, const float* norms, float* g_g, int B,
                            size_t per, float lambda, float target, int one_sided, void* stream);
/* Feature matching term of the generator loss (losses.py:62-69) on the critic's features of reals and fakes,
 * real / fake [B, F]:  diff[f] = mean_b real[b,f] - mean_b fake[b,f] (written for the backward),
 * loss[0] = mean_f diff[f]^2;  bwd: g_fake[b,f] = gl[0] * d loss / d fake[b,f] = -2 gl[0] diff[f] / (B F). */
int eeggan_feature_match_fwd(const float* real, const float* fake, float* diff, float* loss, int B, int F,
                             void* stream);
int eeggan_feature_match_bwd(const float* gl, const float* diff, float* g_fake, int B, int F, void* stream);
/* ---- batch normalisation inside a generator layer (act_norm = 'batch': ConditionalBatchNorm, layers.py:93-134,
 * placed between conv and activation by GeneralConv, layers.py:190-213).  c, y [B, C, T] float32:
 *   n = (c - mean_ch) rstd_ch,  statistics over (batch, time), biased variance, eps inside the root;
 *   o = gamma_ch n + beta_ch when gamma / beta are given ('BN' mode), else n;
 *   o = o + gmod[b,ch] o + bmod[b,ch] when gmod / bmod [B, C] are given (self modulation, layers.py:129-134);
 *   y = drop[b,ch] lrelu_slope(o)     (drop NULL: 1; slope 1: no activation).
 * train != 0: mean / rstd [C] are computed and written, and run_mean / run_var (NULL: none) are updated like
 * nn.BatchNorm1d (momentum, unbiased variance); train == 0: mean / rstd are taken from run_mean / run_var. */
int eeggan_bn_lrelu_drop_fwd(const float* c, const float* gamma, const float* beta, const float* gmod, const float* bmod,
                             const float* drop, float* run_mean, float* run_var, float* mean, float* rstd, float* y, int B,
                             int C, int T, float eps, float momentum, float slope, int train, void* stream);
/* Backward of the above given gy = d loss / d y: gc = d loss / d c; g_gamma / g_beta [C] (always written: the channel
 * sums the input gradient needs; the affine parameters' gradients in 'BN' mode); g_gmod / g_bmod [B, C] with the
 * modulation.  train == 0: the statistics are constants. */
int eeggan_bn_lrelu_drop_bwd(const float* gy, const float* y, const float* c, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, const float* gmod, const float* drop, float* gc,
                             float* g_gamma, float* g_beta, float* g_gmod, float* g_bmod, int B, int C, int T, float slope,
                             int train, void* stream);

/* The pieces of a batch norm that is differentiated twice (the critic's layers under the gradient penalty, layers.py:93-134
 * with act_norm = 'batch' in the Discriminator, network.py:496-497): each is linear in each argument, so every
 * derivative of the norm is made of the same three. */
/* out[ch] = scale * sum over (batch, time) of x[b, ch, t], in double, fixed order. */
int eeggan_channel_sum(const float* x, float* out, int B, int C, int T, float scale, void* stream);
/* y[b, ch, t] = x[b, ch, t] * s[ch] + t[ch]   (x NULL: y = t[ch]; s NULL: 1; t NULL: 0). */
int eeggan_channel_affine(const float* x, const float* s, const float* t, float* y, int B, int C, int T, void* stream);
/* out = a * b elementwise. */
int eeggan_product(const float* a, const float* b, float* out, size_t count, void* stream);

/* The same for a pixel norm that is differentiated twice (act_norm = 'pixel' in the Discriminator; PixelNorm,
 * layers.py:10-14): sums over the channels per (sample, time step) and a per-position scale. */
/* out[b, t] = scale * sum over ch of x[b, ch, t]. */
int eeggan_position_sum(const float* x, float* out, int B, int C, int T, float scale, void* stream);
/* y[b, ch, t] = x[b, ch, t] * s[b, t]   (x NULL: y = s[b, t]). */
int eeggan_position_scale(const float* x, const float* s, float* y, int B, int C, int T, void* stream);

/* ---- spectral normalisation of a weight (torch.nn.utils.spectral_norm as layers.py:170-171 and network.py:520-521
 * apply it): the weight as a matrix W [rows = output channels, cols = the other dimensions]. -------------------- */
/* power_iteration != 0 (training): v <- normalize(W^T u), u <- normalize(W v) in place (normalize(x) = x / max(|x|, eps));
 * then sigma = u^T W v and w_sn = w_orig / sigma.  u [rows] and v [cols] are the module's vectors; u_used / v_used
 * receive the vectors sigma was computed with (for the backward); sigma: one float; scratch: rows floats. */
int eeggan_spectral_norm_fwd(const float* w_orig, float* u, float* v, float* w_sn, float* u_used, float* v_used,
                             float* sigma, float* scratch, int rows, int cols, float eps, int power_iteration, void* stream);
/* g_orig = (g_sn - <g_sn, w_sn> u v^T) / sigma: u and v are constants of the forward.  scratch: one float. */
int eeggan_spectral_norm_bwd(const float* g_sn, const float* w_sn, const float* u_used, const float* v_used,
                             const float* sigma, float* scratch, float* g_orig, int rows, int cols, void* stream);

/* ---- self attention (SelfAttention, layers.py:41-64): out = gamma * Wf (value P) + x with
 * P = softmax over dim 1 of key^T query, key / value max-pooled by 4 along time.  The 1x1 convs are eeggan_conv1d;
 * the rest is made of the entry points below (float32 tensors). ------------------------------------------------ */
/* Batched product c[b] (M x N) = alpha * op(a[b]) op(b[b]) of dense row-major matrices: a[b] is stored M x K, or
 * K x M when trans_a; b[b] K x N, or N x K when trans_b (torch.bmm at layers.py:60,62 and every gradient of the two). */
int eeggan_bmm(const float* a, const float* b, float* c, int batch, int M, int N, int K, int trans_a, int trans_b,
               float alpha, void* stream);
/* p = softmax(s, dim=1) for s [B, M, N] (F.softmax(..., dim=1), layers.py:60). */
int eeggan_softmax_dim1_fwd(const float* s, float* p, int B, int M, int N, void* stream);
/* gx = y * (gy - sum_m gy y): backward of that softmax given its output y; also the derivative of this expression
 * with respect to gy, applied to a cotangent passed as gy. */
int eeggan_softmax_dim1_bwd(const float* gy, const float* y, float* gx, int B, int M, int N, void* stream);
/* The derivative of gx = y (gy - sum_m gy y) with respect to y applied to the cotangent q:
 * out = q (gy - sum_m gy y) - gy sum_m(q y)   (second-order pass of the gradient penalty). */
int eeggan_softmax_dim1_bwd_dy(const float* q, const float* gy, const float* y, float* out, int B, int M, int N,
                               void* stream);
/* nn.MaxPool1d(pool) along the last axis of x [rows, Tin] (layers.py:48,58-59): y [rows, Tin / pool] and the position
 * of each maximum inside its window (first one on ties), idx [rows, Tin / pool] bytes. */
int eeggan_maxpool_fwd(const float* x, float* y, unsigned char* idx, size_t rows, int Tin, int pool, void* stream);
/* With the positions fixed the pooling is linear: scatter = 0: dst [rows, Tin / pool] = src[r, pool t + idx[r, t]]
 * (the pooling applied to another tensor: backward of the scatter); scatter = 1: dst [rows, Tin] = src [rows, Tin / pool]
 * placed at the maxima, zero elsewhere (backward of the pooling). */
int eeggan_maxpool_select(const float* src, const unsigned char* idx, float* dst, size_t rows, int Tin, int pool,
                          int scatter, void* stream);
/* out[0] = sum_i a[i] b[i], summed in double in a fixed order (gradient of the scalar gamma, layers.py:47,64).
 * workspace: eeggan_dot_workspace() bytes of device memory. */
size_t eeggan_dot_workspace(void);
int eeggan_dot(const float* a, const float* b, float* out, size_t n, void* workspace, size_t workspace_bytes, void* stream);

/* out[0] = scale * mean(v[0..n))  (generator loss -mean(d_fake): scale = -1). */
int eeggan_mean_scale(const float* v, float* out, int n, float scale, void* stream);
/* out[i] = gl[0] * scale / n  for i in [0,n): backward of eeggan_mean_scale. */
int eeggan_mean_scale_bwd(const float* gl, float* out, int n, float scale, void* stream);

/* ---- multi-tensor Adam (torch.optim.Adam built at train.py:131-136, stepped at trainer.py:90,94)
 * table: HOST array of `ntensors` descriptors (copied into the kernel arguments, so the call is
 * graph-capturable and needs no device table); one launch per 64 tensors updates all of them:
 *   step += 1;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * `step` points to a DEVICE float holding that tensor's step count (advanced by the call, so a
 * captured launch counts correctly on every replay).  grad_scale multiplies every gradient first
 * (1/world_size after a summed all-reduce). */
typedef struct {
    float* p; const float* g; float* m; float* v; float* step; int64_t n;
} eeggan_adam_tensor;
int eeggan_adam_step(const eeggan_adam_tensor* table, int ntensors,
                     float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
/* p[k] = g[k] for every tensor of the table (m, v, step unused): packs gradients into the slices
 * of one flat bucket for a single all-reduce. */
int eeggan_multi_copy(const eeggan_adam_tensor* table, int ntensors, void* stream);
/* p[k] = alpha * p[k] + beta * g[k] for every tensor of the table (m, v, step unused): the running average of
 * the generator's weights the reference keeps for its sample outputs (plugins.py:305-307: alpha = old_weight,
 * beta = 1 - old_weight). */
int eeggan_multi_axpby(const eeggan_adam_tensor* table, int ntensors, float alpha, float beta, void* stream);

/* ---- 1-D sliced Wasserstein distance (SlicedWDistance, plugins.py:383-395, 426-451) -----------------
 * The metric of the reference's evaluation plugin, on the device.  Per pyramid level:
 *   eeggan_swd_patches: out[(j*ppi + k), ch, p] = batch[j, ch, (starts[j*ppi + k] + p) * step]
 *       (plugins.py:439-444; `step` is the decimation of the level, batchs[i][:, :, ::scale] applied
 *       `level` times, plugins.py:449, without materialising the decimated batch; T = full length);
 *   eeggan_swd_normalize: the patches read as rows of c numbers, desc[rows, c] (the reference's
 *       reshape((-1, c)), plugins.py:445):  desc = (desc - mean_col) / (std_col + eps), std unbiased
 *       (plugins.py:446-447); sums in double precision, fixed order;
 *   eeggan_swd_project_keys: keys[d*rows + m] = d << 32 | ord(sum_k desc[m,k] * dirs[k,d]) for the D unit
 *       directions dirs[c, D] (plugins.py:386-390); ord() maps a float to an unsigned integer of the same order;
 *   eeggan_swd_sort: one device-wide radix sort of the keys over 32 + ceil(log2 D) bits = every direction's
 *       projections sorted and contiguous (torch.sort(dim=0), plugins.py:391-392);
 *   eeggan_swd_distance: out[0] = mean |a - b| over the two sorted key arrays (plugins.py:393-394).
 * Workspaces are caller-allocated device memory of the sizes the _workspace functions return. */
int eeggan_swd_patches(const float* batch, const int32_t* starts, float* out, int items, int c, int T,
                       int step, int patches_per_item, int patch_size, void* stream);
size_t eeggan_swd_normalize_workspace(void);
int eeggan_swd_normalize(float* desc, size_t rows, int c, float eps, void* workspace, size_t workspace_bytes,
                         void* stream);
int eeggan_swd_project_keys(const float* desc, const float* dirs, uint64_t* keys, size_t rows, int c, int D,
                            void* stream);
size_t eeggan_swd_sort_workspace(size_t nkeys, int D);
int eeggan_swd_sort(const uint64_t* keys_in, uint64_t* keys_out, size_t nkeys, int D, void* workspace,
                    size_t workspace_bytes, void* stream);
size_t eeggan_swd_distance_workspace(void);
int eeggan_swd_distance(const uint64_t* a, const uint64_t* b, size_t nkeys, float* out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- CPC feature network, inference (cpc/cpc_network.py) and Frechet-distance statistics (plugins.py:503-572)
 * eeggan_cpc_encoder: ConvEncoder in eval mode (cpc_network.py:137-158 with its defaults: 32/64/128 channels,
 *   strides 5/4/3, kernels 9/7/5, padding (k-1)/2; ReLU, dropout off, BatchNorm1d after the first two layers as
 *   the per-channel scale/shift of its running statistics; s1/t1, s2/t2 NULL = no normalisation):
 *     h1 = s1 * relu(conv(x, w1) + b1) + t1;  h2 = s2 * relu(conv(h1, w2) + b2) + t2;  z = relu(conv(h2, w3) + b3)
 *   x [B, Cin, T], T a multiple of 60; weights TRANSPOSED: wNt[(ci*K + j)*Cout + co] = wN[co, ci, j];
 *   z [B, 128, T/60]; zt (may be NULL) the same codes time-major, [B, T/60, 128].
 * eeggan_pnorm_pool: out[b, c] = sum_t |x[b,c,t]|, out[b, C + c] = max_t |x[b,c,t]|  (PNormPooling.forward with
 *   p = [1, inf], cpc_network.py:185-188, before its MLP).
 * eeggan_dense: y[n, m] = shift[m] + scale[m] * act(bias[m] + alpha * sum_k x[n*ldx + k] * W(m, k)) + eye * (n == m)
 *   with W(m, k) = w[m*K + k] (nn.Linear, cpc_network.py:178) or w[k*M + m] when w_is_kn (a plain product x w:
 *   the bmm calls of sqrt_newton_schulz, plugins.py:551-553); act = ReLU when relu != 0; bias, scale/shift
 *   may be NULL (a following eval-mode BatchNorm1d is scale/shift, cpc_network.py:181-182).
 * eeggan_gru_forward: torch.nn.GRU(128, 128), one layer, h0 = 0 (cpc_network.py:203, 211-216) given the input
 *   projections gi[(b*T + t)*384 + g*128 + j] = (W_ih x_t + b_ih)[g*128 + j]; writes c[b, j, t] = h_t[b, j].
 * eeggan_feature_stats: mean[f] and the unbiased covariance cov[f, g] of the rows of X[N, F]
 *   (FidCalculator.torch_cov with rowvar=False, plugins.py:503-514); sums in a fixed order, partials in double. */
int eeggan_cpc_encoder(const float* x, const float* w1t, const float* b1, const float* s1, const float* t1,
                       const float* w2t, const float* b2, const float* s2, const float* t2, const float* w3t,
                       const float* b3, float* z, float* zt, int B, int Cin, int T, void* stream);
int eeggan_pnorm_pool(const float* x, float* out, int B, int C, int T, void* stream);
int eeggan_dense(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                 float* y, size_t N, int K, int M, size_t ldx, size_t ldy, int w_is_kn, float alpha, float eye,
                 int relu, void* stream);
int eeggan_gru_forward(const float* gi, const float* whh, const float* bhh, float* c, int B, int H, int T,
                       void* stream);
size_t eeggan_feature_stats_workspace(size_t N, int F);
int eeggan_feature_stats(const float* X, size_t N, int F, float* mean, float* cov, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ---- bfloat16 activation storage (the large-batch mode: 512 samples per GPU) ---------------------------
 * The reference is float32 throughout (layers.py:150-174); these forms exist for the large-batch configuration,
 * whose 64-channel layers are bound by HBM traffic once their products run on the bf16 matrix pipe.  The `_t`
 * ("typed") entry points are the entry points of the same name with storage flags: a flag of 1 means that the
 * tensors it covers hold bfloat16 (2 bytes per element, round to nearest even on store) instead of float32;
 * with all flags 0 they ARE the float32 entry points.  Arithmetic, weights, biases, multipliers (drop,
 * mask_drop), the PixelNorm normaliser d and every reduction stay float32.
 *   eeggan_conv1d_t           x_bf16 covers x and mask_y, y_bf16 covers y and act_aux.  bfloat16 tensors are accepted
 *                             by the predicate-free k = 3 layers (Cin % 32 == 0, Cout % 64 == 0, T % 32 == 0; both
 *                             flags set; matrix precision 1, eeggan_set_matrix_precision) and by 1x1 layers with at
 *                             most 8 channels on one side (to_rgb / from_rgb and their input gradients: any
 *                             combination of flags, T % 4 == 0); anything else fails with a message.
 *   eeggan_conv1d_pixelnorm_t bf16 covers x and y (64 -> 64 layers, T % 32 == 0, matrix precision 1).
 *   eeggan_wgrad_job          x_bf16 covers x, gy_bf16 covers gy and mask_y: k = 3 layers with Cin % 64 == 0,
 *                             Cout % 64 == 0, T % 32 == 0 (both flags, matrix precision 1) and every 1x1 layer.
 *   eeggan_resample_t / eeggan_resample_quads_t   in_bf16 / out_bf16: the two sides independently (the resampler
 *                             is where the float32 tensors of the 32-sample stage meet the bfloat16 ones).
 *   eeggan_lrelu_drop_bwd_t, eeggan_pixelnorm_drop_fwd_t, eeggan_pixelnorm_drop_bwd_t, eeggan_axpby_t,
 *   eeggan_blend_t            bf16 covers every activation-shaped tensor of the call (T resp. n a multiple of 4).
 *   eeggan_cast               out[i] = in[i] converted between the two types (equal flags: a copy). */
int eeggan_conv1d_t(const void* x, const float* w, const float* bias, const float* drop, void* y,
                    int B, int Cin, int Cout, int T, int K, float in_scale,
                    int act, float act_param, const void* act_aux,
                    const void* mask_y, const float* mask_drop, float mask_slope,
                    int x_bf16, int y_bf16, void* stream);
int eeggan_conv1d_pixelnorm_t(const void* x, const float* w, const float* bias, const float* drop,
                              void* y, float* d, int B, int Cin, int Cout, int T, int K,
                              float in_scale, float slope, float eps, int bf16, void* stream);
int eeggan_resample_t(const void* x, const int32_t* idx, const float* coef, void* out,
                      size_t rows, int Tin, int Tout, int taps, int in_bf16, int out_bf16, void* stream);
int eeggan_resample_quads_t(const void* x, const int32_t* base, const float* coef, void* out,
                            size_t rows, int Tin, int Tout, int nv, int in_bf16, int out_bf16, void* stream);
/* eeggan_resample_quads_t without the table of quad starts: base[q] = min(max(A q + C, 0) / D / 4 * 4, Tin - 4 nv)
 * (integer division), for tables whose starts follow that form exactly (every ratio of this path does; the host
 * checks it when it builds a table) -- the kernel then issues its loads without waiting for a look-up. */
int eeggan_resample_quads_affine_t(const void* x, int A, int C, int D, const float* coef, void* out,
                                   size_t rows, int Tin, int Tout, int nv, int in_bf16, int out_bf16, void* stream);
int eeggan_lrelu_drop_bwd_t(const void* g, const void* y, const float* drop, void* out,
                            int B, int C, int T, float slope, int bf16, void* stream);
int eeggan_pixelnorm_drop_fwd_t(const void* h, const float* drop, void* y, float* d,
                                int B, int C, int T, float eps, int bf16, void* stream);
int eeggan_pixelnorm_drop_bwd_t(const void* gy, const void* y, const float* drop, const float* d,
                                void* gc, int B, int C, int T, float slope, int bf16, void* stream);
int eeggan_axpby_t(const void* a, float ca, const void* b, float cb, void* out, size_t n, int bf16, void* stream);
int eeggan_blend_t(const void* a, const void* b, const float* alpha, int mode, void* out, size_t n, int bf16,
                   void* stream);
int eeggan_cast(const void* in, void* out, size_t n, int in_bf16, int out_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EEGGAN_HIP_H */
