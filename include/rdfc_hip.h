/*
 * rdfc_hip.h -- C ABI of librdfc_hip.so: the MI355X (gfx950) kernels behind the
 * RDFC-GAN generator / discriminator train-step hot path.
 *
 * Boundary rules
 *   - plain C: device pointers, ints, floats, one POD descriptor struct; no torch /
 *     ATen types.  Every entry point enqueues work on `stream` (a hipStream_t passed
 *     as void*) and returns immediately; nothing here allocates or synchronises.
 *   - return value: 0 on success, non-zero on a rejected argument or a failed
 *     launch; rdfc_last_error() returns the message (upstream only printf'ed launch
 *     failures: modulated_deform_im2col_cuda.cuh:348-352 -- here they are errors).
 *   - activations are "pixel-major": element (n, y, x, c) of a feature map lives at
 *     base[((n*H + y)*W + x) * ld + c], ld >= C (ld > C addresses a channel slice of
 *     a wider map, which is how channel concatenation is done without a copy pass).
 *     1-channel maps are therefore bit-identical to upstream's NCHW tensors.
 *   - dtype: RDFC_F32 (exact-f32 MFMA path, parity mode), RDFC_BF16 (bf16 storage,
 *     bf16 MFMA, f32 accumulation) or RDFC_F16 (IEEE half storage, f16 MFMA, f32
 *     accumulation: BASELINE config "fp16 activations with fp32 loss accumulate").
 *     Statistics, scales, losses, weight gradients and optimizer state are always f32.
 *
 * Each group below cites the upstream interface it stands in for (paths relative to
 * RDFC-GAN/).  The python side (rdfc_gan_amd/) binds this file with ctypes; a
 * maintainer of upstream would bind it the same way.
 */
#ifndef RDFC_HIP_H
#define RDFC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RDFC_F32 = 0, RDFC_BF16 = 1, RDFC_F16 = 2 };
enum { RDFC_ACT_NONE = 0, RDFC_ACT_RELU = 1, RDFC_ACT_LRELU = 2, RDFC_ACT_TANH = 3, RDFC_ACT_SIGMOID = 4,
       RDFC_ACT_PRELU = 5 /* slope read from device memory */ };

const char *rdfc_last_error(void);
/* name of the kernel instantiation the calling thread's last rdfc_conv2d_* call launched (e.g.
 * "conv3x3_slab_kernel<__bf16, 64, 128, 1, 256>"), matching the demangled symbol in a rocprofv3 kernel trace.
 * Measurement aid only (bench.py's per-kernel roofline); no upstream counterpart. */
const char *rdfc_last_kernel(void);
/* Measurement aid (tools/conv_tall_bench.py, tests): force the pixel-tile height of the 128-channel 3x3 / stride-1 kernel
 * (rows of 16 pixels per wave: 4 | 6 | 8 | 10; 0 = chosen per launch, the default; same as RDFC_GLDS_JR in the environment).
 * Results do not depend on it.  No upstream counterpart. */
void rdfc_tune_glds_rows(int rows_per_wave);
/* Measurement aid (tools/wgrad_ring_bench.py): which weight-gradient launches the ring kernel takes -- 0 none (tap-gather / segment-walk
 * kernels with float atomics), 1 3x3 / stride 1 only, 2 (default) also stride 2 and 4x4, 3 also the launches the default leaves to the
 * older kernels by measurement; the initial value comes from RDFC_WGRAD_RING, read once.  Process-wide, not thread-safe.  No upstream
 * counterpart. */
void rdfc_tune_wgrad_ring(int mode);
/* Measurement / test aid: 1 (default; RDFC_WGRAD_KG) = the three kernel rows of a 64-channel tile of the ring weight gradient run as ONE
 * 768-thread workgroup in lockstep (one fabric read of dY and X instead of up to three), 0 = one row per 256-thread workgroup as in
 * rounds 3-5.  Same bits either way.  No upstream counterpart. */
void rdfc_tune_wgrad_kg(int on);
int rdfc_version(void);
/* sha256 (first 16 hex digits) of the kernel sources this library was built from (rdfc_gan_amd/build.py: source_hash()): measurement
 * artefacts under profiles/ carry it, and bench.py refuses counters taken from another build.  No upstream counterpart. */
const char *rdfc_build_hash(void);

/* ------------------------------------------------------------------------------
 * Convolution family (implicit GEMM on MFMA).
 * Stands in for ATen conv2d / conv_transpose2d fwd+bwd as called by
 *   lib/models/generator/rdf_generator/encoder_decoder/common.py:29-61 (conv_bn_relu,
 *   convt_bn_relu), torchvision BasicBlock, lib/models/generator/resnet_generator.py:19-50,
 *   lib/models/module/conv_norm_act.py:66-75, model_utils.py:39-50 (EqualLinear == 1x1).
 *
 * One descriptor describes the *forward* convolution  y[N,Ho,Wo,Co] = conv(x[N,Hi,Wi,Ci], w[Co,Ci,KH,KW]).
 * ConvTranspose2d is expressed through the same three primitives with the roles of
 * x and y exchanged (its forward is rdfc_conv2d_dgrad, its input-gradient is
 * rdfc_conv2d_forward, its weight-gradient is rdfc_conv2d_wgrad with x<->dy).
 * ------------------------------------------------------------------------------ */
typedef struct rdfc_conv_desc {
    int32_t N, Hi, Wi, Ci;      /* input map                                         */
    int32_t Ho, Wo, Co;         /* output map                                        */
    int32_t KH, KW;             /* kernel                                            */
    int32_t sh, sw, ph, pw, dh, dw;
    int32_t ldx, ldy;           /* pixel strides (elements) of x / y                 */
    int32_t dtype;              /* RDFC_F32 | RDFC_BF16 | RDFC_F16: storage + MFMA type of x,y,w */
    int32_t act;                /* epilogue activation applied after bias (forward)  */
    float   slope;              /* LeakyReLU slope                                   */
    int32_t out_f32;            /* 1: result stored as f32 regardless of dtype       */
} rdfc_conv_desc;

/* OIHW f32 master weights -> MFMA operand layouts in `dtype`, multiplied by `scale`
 * (EqualLR's sqrt(2/fan_in), model_utils.py:11-16; 1.0 elsewhere):
 *   w_fwd [O][KH*KW][I]   rows of the forward GEMM          (may be NULL)
 *   w_bwd [I][KH*KW][O]   rows of the input-gradient GEMM   (may be NULL)            */
int rdfc_pack_weight(const float *w_oihw, void *w_fwd, void *w_bwd, int O, int I, int KH, int KW,
                     int dtype, float scale, void *stream);
/* Same from master weights stored [O][KH][KW][I] (torch.channels_last: the layout rdfc_conv2d_wgrad produces, so that
 * the weight gradient accumulates straight into the parameter's .grad and needs no unpack). */
int rdfc_pack_weight_ohwi(const float *w_ohwi, void *w_fwd, void *w_bwd, int O, int I, int KH, int KW,
                          int dtype, float scale, void *stream);
/* Every weight of a network in ONE launch (after the optimizer step: lib/models/rdfc_gan.py:324-334 runs optimizer.step(), the
 * next forward then needs every operand again).  jobs: DEVICE array of njobs descriptors sorted by block0; job j owns the
 * workgroups [block0_j, block0_{j+1}): one per (tap, 64-row tile of O, 64-column tile of I), nblocks in all.  Source element
 * (o, t, i) sits at w[o * so + t * st + i * si] (OIHW: so = I T, si = T, st = 1; [O][T][I]: so = T I, st = I, si = 1). */
typedef struct rdfc_pack_job {
    const float *w;             /* f32 master weight                                   */
    void *w_fwd, *w_bwd;        /* as in rdfc_pack_weight (either may be NULL)           */
    int32_t O, I, T;            /* T = KH * KW                                          */
    int32_t so, st, si;         /* source strides (elements)                            */
    float   scale;
    int32_t block0;             /* first workgroup of this job                          */
} rdfc_pack_job;
int rdfc_pack_weights_batch(const rdfc_pack_job *jobs, int njobs, int nblocks, int dtype, void *stream);
/* dw_packed f32 [O][KH*KW][I] -> grad_oihw f32 [O][I][KH][KW] = scale * dw_packed (overwrite) */
int rdfc_unpack_wgrad(const float *dw_packed, float *grad_oihw, int O, int I, int KH, int KW,
                      float scale, void *stream);

/* Reflection padding folded into the conv (resnet_generator.py:60-100: ReflectionPad2d(1) + Conv2d(3x3, padding 0) in every
 * ResnetBlock): d describes the conv on the UNPADDED map x with ph = pw = the reflection width; out-of-image taps read the
 * mirrored pixel.  Covered where forward and weight gradient run on the LDS-DMA 3x3 kernels (rdfc_conv2d_reflect_supported
 * returns 1); elsewhere pad with rdfc_reflect_pad and run the pad-0 conv.  The input gradient is the pad-0 conv's
 * (rdfc_conv2d_dgrad on the padded geometry) folded by rdfc_reflect_pad_adjoint[_add]. */
int rdfc_conv2d_reflect_supported(const rdfc_conv_desc *d);
int rdfc_conv2d_forward_reflect(const rdfc_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y, void *stream);
int rdfc_conv2d_wgrad_reflect(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw_packed, void *stream);
/* 1 when rdfc_conv2d_wgrad (reflect = 0) / rdfc_conv2d_wgrad_reflect (reflect = 1) of this layer adds its split-K partial sums in a
 * fixed order -- bitwise reproducible weight gradients (the 3x3 / 4x4, stride 1 / 2 layers with >= 64 channels on both sides:
 * conv_wgrad_ring.hip) -- and 0 when the launch goes to a kernel that accumulates partials with f32 atomics (same values up to
 * the order of the adds).  torch.nn.grad.conv2d_weight in upstream's autograd (torch.nn.Conv2d, common.py:11-17) has no such
 * guarantee either; a caller that needs run-to-run identical training asks here. */
int rdfc_conv2d_wgrad_deterministic(const rdfc_conv_desc *d, int reflect);
/* The deterministic kernels keep their split-K partials in a scratch slab per (device, stream), grown on demand -- except inside a
 * stream capture, where nothing can be allocated: a launch recorded into a HIP graph on a stream without a slab goes to the
 * float-atomics kernels (rdfc_conv2d_wgrad_deterministic does not know about captures).  Call this BEFORE capturing on `stream` to
 * give it its slab (bytes 0: the default 64 MB, enough for every layer of the hot path); a slab a captured graph refers to is never
 * freed or moved afterwards. */
int rdfc_wgrad_ring_reserve(void *stream, size_t bytes);

/* y = act(conv(x, w_fwd) + bias).  bias: f32[Co] or NULL. */
int rdfc_conv2d_forward(const rdfc_conv_desc *d, const void *x, const void *w_fwd, const float *bias,
                        void *y, void *stream);
/* Forward convolution that also produces the per-tile partial statistics {sum, sum of squares}[G][nchunk][2][Co] of the
 * values it stores, for the normalisation layer behind it: G groups of N / G consecutive images each (G = 1 batch norm, G = N
 * instance norm, in between BatchNorm over G stacked batches); the separate rdfc_norm_stats pass over y disappears.
 * nchunk = rdfc_conv_stats_nchunk(d, G) rows per group; 0 means no fused kernel covers the shape (call rdfc_conv2d_forward +
 * rdfc_norm_stats instead).  Finalize with rdfc_norm_finalize_n / rdfc_norm_finalize_seq / rdfc_norm_finalize_apply_seq. */
int rdfc_conv_stats_nchunk(const rdfc_conv_desc *d, int G);
int rdfc_conv2d_forward_stats(const rdfc_conv_desc *d, const void *x, const void *w_fwd, const float *bias, void *y,
                              float *partial, int G, void *stream);
/* dx[N,Hi,Wi,Ci] = conv^T(dy[N,Ho,Wo,Co], w_bwd); ld of dy = d->ldy, ld of dx = d->ldx.
 * Also the forward of ConvTranspose2d (then bias/act apply to the result). */
int rdfc_conv2d_dgrad(const rdfc_conv_desc *d, const void *dy, const void *w_bwd, const float *bias,
                      void *dx, void *stream);
/* dx = rdfc_conv2d_dgrad(dy) + addend (same dtype and layout as dx, pixel stride ld_add): the gradient of an activation that
 * feeds this convolution and a second consumer -- torchvision BasicBlock's `out += identity` (common.py:11-17), the
 * downsample path -- without a separate summation pass. */
int rdfc_conv2d_dgrad_add(const rdfc_conv_desc *d, const void *dy, const void *w_bwd, const void *addend, int ld_add,
                          void *dx, void *stream);
/* the same with two maps added (a passthrough block's alias cotangent and a skip connection's: functional._Chain) */
int rdfc_conv2d_dgrad_add2(const rdfc_conv_desc *d, const void *dy, const void *w_bwd, const void *addend, int ld_add,
                           const void *addend2, int ld_add2, void *dx, void *stream);
/* dw_packed f32 [Co][KH*KW][Ci] += sum_pixels dy (x) x   (caller zeroes dw_packed first). */
int rdfc_conv2d_wgrad(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw_packed, void *stream);
/* Weight AND bias gradient of a conv block WITHOUT a normalisation layer with the backward of its activation folded in (round 6): the
 * stems -- conv_bn_relu(..., bn=False) on the 3- / 1-channel inputs at full resolution (rdf_generator.py:60-66, common.py:29-44), where
 * autograd runs ReLU's backward (a pass over dz and y that writes dy), the bias gradient (a pass over dy) and the weight gradient (another
 * one).  dz: the cotangent BEHIND the activation (pixel stride d->ldy), y_act: the block's activated output (pixel stride ld_y);
 * dy = dz * act'(y_act) is formed while dz is staged, rounded once like rdfc_act_bwd's result; d->act / d->slope name the activation
 * (RDFC_ACT_RELU | RDFC_ACT_LRELU).  dw_packed [Co][KH*KW][Ci] and dbias [Co] (may be NULL) are ACCUMULATED into, like rdfc_conv2d_wgrad.
 * Covered: thin inputs (Ci <= 4), stride 1, 2-byte dtypes -- ask rdfc_conv2d_wgrad_act_supported first (1 = covered). */
int rdfc_conv2d_wgrad_act_supported(const rdfc_conv_desc *d);
int rdfc_conv2d_wgrad_act(const rdfc_conv_desc *d, const void *x, const void *dz, const void *y_act, int ld_y, float *dw_packed,
                          float *dbias, void *stream);

/* ------------------------------------------------------------------------------
 * Layout / data movement.
 * ------------------------------------------------------------------------------ */
/* upstream NCHW f32 tensor -> pixel-major `dtype` (and back, f32 out). */
int rdfc_nchw_to_pm(const float *src, void *dst, int N, int C, int H, int W, int ld, int dtype, void *stream);
int rdfc_pm_to_nchw(const void *src, float *dst, int N, int C, int H, int W, int ld, int dtype, void *stream);
/* dst[n, y, x, 0:C] (map Hd x Wd, ld_dst) = src[n, y, x, 0:C] (map Hs x Ws, ld_src) for the top-left Hw x Ww
 * window: the crop of rdf_generator.py:244-260 (and its adjoint, a zero-extended copy) and the channel-slice
 * copy of torch.cat.  accumulate != 0: dst += src. */
int rdfc_copy_window(const void *src, void *dst, int N, int Hs, int Ws, int Hd, int Wd, int Hw, int Ww, int C,
                     int ld_src, int ld_dst, int dtype, int accumulate, void *stream);
/* nn.ReflectionPad2d(pad) forward, and its adjoint (fold the border back). resnet_generator.py:19,50,68,84 */
int rdfc_reflect_pad(const void *src, void *dst, int N, int H, int W, int C, int pad, int dtype, void *stream);
int rdfc_reflect_pad_adjoint(const void *dpad, void *dsrc, int N, int H, int W, int C, int pad, int dtype, void *stream);
/* same, + addend (the un-padded map's cotangent from a second consumer: ResnetBlock's `x + conv_block(x)`, resnet_generator.py:97-99) */
int rdfc_reflect_pad_adjoint_add(const void *dpad, const void *addend, int ld_add, void *dsrc, int N, int H, int W, int C, int pad,
                                 int dtype, void *stream);

/* ------------------------------------------------------------------------------
 * Normalisation (+ residual + activation).  Stands in for nn.BatchNorm2d (train and
 * eval), nn.InstanceNorm2d and the activation that follows them
 * (common.py:34-39,52-57; torchvision BasicBlock; resnet_generator.py:21-44;
 *  conv_norm_act.py:117-125).
 * groups G = 1 (batch norm: statistics over N*H*W) or N (instance norm).
 * ------------------------------------------------------------------------------ */
/* pass 1: per-(group, channel) sum and sum of squares -> partial[G][nchunk][2][C] (f32),
 * nchunk = rdfc_norm_nchunk(P, C, dtype) where P = pixels per group (scratch sizing for every
 * two-stage reduction below; rdfc_channel_sum: nchunk = rdfc_norm_nchunk(P, C, dtype) with G = 1). */
int rdfc_norm_nchunk(int pixels_per_group, int C, int dtype);
int rdfc_norm_stats(const void *x, float *partial, int G, int P, int C, int ld, int dtype, void *stream);
/* pass 2: mean[G][C], rstd[G][C] (biased variance, eps); optional running-stat update
 * (momentum, unbiased variance) when run_mean != NULL. */
int rdfc_norm_finalize(const float *partial, float *mean, float *rstd, int G, int P, int C, int dtype, float eps,
                       float *run_mean, float *run_var, float momentum, void *stream);
/* rdfc_norm_finalize for partials with an explicit chunk count (rdfc_conv2d_forward_stats). */
int rdfc_norm_finalize_n(const float *partial, int nchunk, float *mean, float *rstd, int G, int P, int C, float eps,
                         float *run_mean, float *run_var, float momentum, void *stream);
/* BatchNorm over G stacked batches (G consecutive calls of one nn.BatchNorm2d, run as ONE launch over the concatenated
 * batch): statistics per group; the running statistics advance once per group, in group order, with
 * group_momentum[g] (HOST array of G floats, G <= 8) -- what G separate forwards leave behind.  nchunk 0: derived from
 * (P, C, dtype) like rdfc_norm_finalize; > 0: explicit (rdfc_conv2d_forward_stats partials). */
int rdfc_norm_finalize_seq(const float *partial, int nchunk, float *mean, float *rstd, int G, int P, int C, int dtype, float eps,
                           float *run_mean, float *run_var, const float *group_momentum, void *stream);
/* z = act(gamma*(x-mean)*rstd + beta + res).  gamma/beta f32[C] or NULL (1/0); res may be NULL.
 * prelu: device pointer to the slope when act == RDFC_ACT_PRELU. */
int rdfc_norm_apply(const void *x, const float *mean, const float *rstd, const float *gamma, const float *beta,
                    const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype,
                    int act, float slope, const float *prelu, void *stream);
/* backward pass 1: u recomputed from (x, stats, res); du = dz*act'(u);
 * partial[G][nchunk][3][C] = sum du, sum du*xhat, sum dz*min(u,0) (PReLU slope grad). */
int rdfc_norm_bwd_reduce(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                         const float *beta, const void *res, float *partial, int G, int P, int C,
                         int lddz, int ldx, int ldr, int dtype, int act, float slope, const float *prelu,
                         void *stream);
/* sums[G][3][C] from the partials (fixed order => deterministic). */
int rdfc_norm_bwd_finalize(const float *partial, float *sums, int G, int P, int C, int dtype, void *stream);
/* Same; additionally dbeta_acc[C] += sum du and dgamma_acc[C] += sum du*xhat (either may be NULL; G == 1 only): the affine
 * gradients of a BatchNorm land in the parameters' persistent .grad without an autograd add each.  With several statistics groups
 * (BatchNorm over stacked batches) use rdfc_norm_backward: its apply pass adds the groups' sums in group order (reproducible). */
int rdfc_norm_bwd_finalize_acc(const float *partial, float *sums, int G, int P, int C, int dtype, float *dbeta_acc,
                               float *dgamma_acc, void *stream);
/* backward pass 2: dx = gamma*rstd*(du - S1/P - xhat*S2/P) (training) or gamma*rstd*du (frozen stats);
 * dres (may be NULL) = du. */
int rdfc_norm_bwd_apply(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                        const float *beta, const void *res, const float *sums, void *dx, void *dres,
                        int G, int P, int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype,
                        int act, float slope, const float *prelu, int frozen_stats, void *stream);
/* same; additionally the affine gradients land in the parameters' persistent .grad: dbeta_acc[c] += sum over groups of sums[g][0][c],
 * dgamma_acc[c] += ... sums[g][1][c] (groups added in group order by one thread per channel: reproducible); either may be NULL */
int rdfc_norm_bwd_apply_acc(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                            const float *beta, const void *res, const float *sums, void *dx, void *dres,
                            int G, int P, int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype,
                            int act, float slope, const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, void *stream);
/* Round 6: the apply pass that ALSO sums the input gradient it stores -- the gradient of a conv bias in FRONT of the norm layer
 * (ResnetGenerator's convs, resnet_generator.py:60-100; analytically zero, in practice the rounding noise that Adam turns into lr-sized
 * steps, upstream's too): dbias_acc[c] += sum over groups and pixels of dx as stored; fixed-order two-stage sum (one partial row per
 * workgroup in dxsum_scratch [rdfc_norm_dxsum_scratch_floats(G, P, C, dtype)], then rdfc_channel_sum's finalize launch): bitwise
 * reproducible, and one pass over dx less than rdfc_channel_sum_acc behind rdfc_norm_bwd_apply.  dbias_acc NULL: rdfc_norm_bwd_apply_acc. */
int rdfc_norm_dxsum_scratch_floats(int G, int P, int C, int dtype);
int rdfc_norm_bwd_apply_dxsum(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                              const float *beta, const void *res, const float *sums, void *dx, void *dres,
                              int G, int P, int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype,
                              int act, float slope, const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc,
                              float *dxsum_scratch, float *dbias_acc, void *stream);
/* One layer's launches behind one call (the same kernels, in the same order, on the same stream as the three-call sequences above:
 * reduction launch -> fixed-order f64 finalize launch -> apply pass; bitwise reproducible).  scratch: f32
 * [rdfc_norm_scratch_floats(G, rdfc_norm_nchunk(P, C, dtype), K, C)], K = 2 (forward) / 3 (backward): the partial rows + the f64
 * per-group statistics a stacked-batch BatchNorm hands from its finalize to its normalise pass.  (Round 4's in-launch finalize by the
 * last-arriving workgroup measured slower than the finalize launch and was removed in round 5: profiles/r04_experiments.txt item 1.) */
int rdfc_norm_scratch_floats(int G, int nchunk, int K, int C);
/* statistics only (W-AdaIN's instance norm of the content map, model_utils.py:86): mean / rstd [G][C] (+ running statistics, G == 1) */
int rdfc_norm_stats_final(const void *x, float *scratch, float *mean, float *rstd, int G, int P, int C, int ld, int dtype, float eps,
                          float *run_mean, float *run_var, float momentum, void *stream);
int rdfc_norm_forward(const void *x, float *scratch, float *mean, float *rstd, const float *gamma, const float *beta,
                      const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype, float eps,
                      float *run_mean, float *run_var, float momentum, int act, float slope, const float *prelu, void *stream);
/* rdfc_norm_forward with rdfc_norm_finalize_seq in the middle (BatchNorm over G stacked batches). */
int rdfc_norm_forward_seq(const void *x, float *scratch, float *mean, float *rstd, const float *gamma, const float *beta,
                          const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype, float eps,
                          float *run_mean, float *run_var, const float *group_momentum, int act, float slope,
                          const float *prelu, void *stream);
/* The part of rdfc_norm_forward_seq behind the partial rows (round 4): statistics per group in parallel (one small workgroup per
 * (group, 32 channels)), then the normalise pass, ONE thread per channel of which also advances the running statistics once per
 * group in group order -- rdfc_norm_finalize_seq walks the groups inside one workgroup instead.  partial: [G][nchunk][2][C] from
 * rdfc_norm_stats (nchunk 0: derived from (P, C, dtype)) or rdfc_conv2d_forward_stats; muvar: f64 [G][2][C] scratch. */
int rdfc_norm_finalize_apply_seq(const float *partial, int nchunk, double *muvar, const void *x, float *mean, float *rstd,
                                 const float *gamma, const float *beta, const void *res, void *z, int G, int P, int C, int ldx,
                                 int ldr, int ldz, int dtype, float eps, float *run_mean, float *run_var,
                                 const float *group_momentum, int act, float slope, const float *prelu, void *stream);
int rdfc_norm_backward(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                       const float *beta, const void *res, float *scratch, float *sums, void *dx, void *dres, int G, int P, int C,
                       int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope, const float *prelu,
                       int frozen_stats, float *dbeta_acc, float *dgamma_acc, void *stream);
/* rdfc_norm_backward + the bias gradient of rdfc_norm_bwd_apply_dxsum */
int rdfc_norm_backward_dxsum(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                       const float *beta, const void *res, float *scratch, float *sums, void *dx, void *dres, int G, int P, int C,
                       int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope, const float *prelu,
                       int frozen_stats, float *dbeta_acc, float *dgamma_acc,
                             float *dxsum_scratch, float *dbias_acc, void *stream);

/* W-AdaIN (model_utils.py:72-90): out = gb[:, 0:C] * IN(x) + gb[:, C:2C], gb = per-pixel style affine. */
int rdfc_adain_apply(const void *x, const float *mean, const float *rstd, const void *gb, void *out,
                     int N, int P, int C, int ldx, int ldgb, int ldo, int dtype, void *stream);
int rdfc_adain_bwd_reduce(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb,
                          float *partial, int N, int P, int C, int lddo, int ldx, int ldgb, int dtype, void *stream);
int rdfc_adain_bwd_apply(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb,
                         const float *sums, void *dx, void *dgb, int N, int P, int C, int lddo, int ldx,
                         int ldgb, int lddx, int lddgb, int dtype, void *stream);

/* sums + apply of the W-AdaIN backward in one call; scratch: f32 [rdfc_norm_scratch_floats(N, rdfc_norm_nchunk(P, C, dtype), 3, C)], sums f32 [N][3][C] */
int rdfc_adain_backward(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb, float *scratch,
                        float *sums, void *dx, void *dgb, int N, int P, int C, int lddo, int ldx, int ldgb, int lddx, int lddgb,
                        int dtype, void *stream);

/* per-channel sum over all pixels (bias gradient): out[C] = sum_p x[p, c].
 * partial: scratch of rdfc_channel_sum_scratch_floats(P, C, ld, dtype) floats. */
int rdfc_channel_sum_scratch_floats(int P, int C, int ld, int dtype);
int rdfc_channel_sum(const void *x, float *partial, float *out, int P, int C, int ld, int dtype, void *stream);
/* same, and acc[c] += out[c] (f32, may be NULL): the bias gradient lands in the parameter's persistent .grad without a pass of
 * autograd's own (torch accumulates a returned gradient with a stock elementwise add) */
int rdfc_channel_sum_acc(const void *x, float *partial, float *out, float *acc, int P, int C, int ld, int dtype, void *stream);

/* elementwise activation backward for conv epilogues without a norm:
 * dx = dy * act'(y) given the *output* y (relu / lrelu / tanh / sigmoid). */
int rdfc_act_bwd(const void *dy, const void *y, void *dx, int P, int C, int lddy, int ldy, int lddx,
                 int dtype, int y_is_f32, int act, float slope, void *stream);

/* ------------------------------------------------------------------------------
 * Generator output head (rdf_generator.py:400-406): clamp, 2-way softmax of the
 * confidences, confidence-weighted sum.  All maps 1-channel f32, n elements.
 * ------------------------------------------------------------------------------ */
int rdfc_fusion_head(const float *d1, const float *c1, const float *d2_in, const float *c2,
                     float *d2_clamped, float *pred, int n, void *stream);
int rdfc_fusion_head_bwd(const float *d1, const float *c1, const float *d2_in, const float *c2,
                         const float *g_pred, const float *g_d2c, float *g_d1, float *g_c1, float *g_d2, float *g_c2,
                         int n, void *stream);

/* ------------------------------------------------------------------------------
 * Per-pixel weighted loss reductions (lib/losses/gan_loss.py: L1_loss :6-22,
 * L2_loss :25-42, mse_loss :104-120 as used by GANLoss :194-206).
 *   loss = sum_i w_i * |p_i - t_i|^k ,  grad_i = scale * w_i * d|.|^k/dp
 * pred f32 [N][C][HW] (upstream NCHW), target: tensor or constant, weight: [N][1][HW]
 * broadcast over C, or uniform 1/(numel + 1e-6).
 * ------------------------------------------------------------------------------ */
int rdfc_loss_nchunk(int numel);
int rdfc_weighted_loss(const float *pred, const float *target, float target_const, const float *weight,
                       float uniform_w, int N, int C, int HW, int kind /*1=L1, 2=L2*/,
                       float *partial, float *loss_out, float *grad /* may be NULL */, float grad_scale,
                       void *stream);

/* ------------------------------------------------------------------------------
 * Modulated deformable convolution: the ABI underneath upstream's pybind module
 * `DCN` (deformconv/src/modulated_deform_conv.h:10-87, vision.cpp:5-12).  NCHW f32/f64
 * upstream; here f32, contiguous.  Kernels: modulated_deform_im2col_cuda.cuh.
 * The column buffer of upstream is never materialised.
 * ------------------------------------------------------------------------------ */
typedef struct rdfc_mdconv_desc {
    int32_t B, C, H, W, Co, KH, KW, sh, sw, ph, pw, dh, dw, group, dgroup;
} rdfc_mdconv_desc;
int rdfc_mdconv_forward(const rdfc_mdconv_desc *d, const float *input, const float *offset, const float *mask,
                        const float *weight, const float *bias, float *output, void *stream);
/* any of the five gradient outputs may be NULL; non-NULL ones are overwritten
 * (grad_input is zeroed inside and accumulated with atomics like upstream). */
int rdfc_mdconv_backward(const rdfc_mdconv_desc *d, const float *input, const float *offset, const float *mask,
                         const float *weight, const float *grad_out, float *grad_input, float *grad_offset,
                         float *grad_mask, float *grad_weight, float *grad_bias, void *stream);
/* the same in double precision (upstream dispatches AT_DISPATCH_FLOATING_TYPES, modulated_deform_conv_cuda.cu:94,172) */
int rdfc_mdconv_forward_f64(const rdfc_mdconv_desc *d, const double *input, const double *offset, const double *mask,
                            const double *weight, const double *bias, double *output, void *stream);
int rdfc_mdconv_backward_f64(const rdfc_mdconv_desc *d, const double *input, const double *offset, const double *mask,
                             const double *weight, const double *grad_out, double *grad_input, double *grad_offset,
                             double *grad_mask, double *grad_weight, double *grad_bias, void *stream);

/* ------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam semantics (rdfc_gan.py:135-140) over one flat f32 buffer.
 * ------------------------------------------------------------------------------ */
int rdfc_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                   float lr, float beta1, float beta2, float eps, int step, float grad_scale, void *stream);
/* torch.optim.AdamW (rdfc_gan.py:141-144, G_normal's optimizer): the same update after the decoupled decay
 * param *= 1 - lr * weight_decay. */
int rdfc_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int step, float grad_scale, void *stream);
/* Same update; lr and the (1-based) step count are read from device memory, hyper = {lr, step} (f32), so the launch
 * can be captured into a HIP graph and replayed while the schedule advances. */
int rdfc_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, const float *hyper,
                       float beta1, float beta2, float eps, float grad_scale, void *stream);

/* Dynamic loss scaling for the f16 mode (the reference trains in f32 and has no counterpart; half precision needs the per-pixel
 * loss gradients, ~1e-6, lifted into its normal range).  All state lives in device memory so that a step never waits for the host:
 *   scaler  f32[4] = {scale, 1 / scale, steps since the last overflow, an optimizer skipped in this step}   (one per model)
 *   ostate  f32[2] = {this step's gradients hold an inf / NaN, steps skipped so far}                          (one per optimizer)
 * rdfc_grad_nonfinite sets ostate[0] from the (all-reduced) flat gradient; rdfc_adam_step_scaled is rdfc_adam_step / _adamw_step /
 * _step_dev (hyper != NULL: {lr, step} from device memory) with the gradient scale grad_scale * scaler[1], skipped as a whole when
 * ostate[0] != 0 (it then counts the skip and raises scaler[3]), bias corrections over the steps really taken; weight_decay = 0:
 * Adam.  rdfc_loss_scale_update (once per step, after every optimizer): overflow -> scale *= backoff, else after `interval` clean
 * steps scale *= growth (torch.cuda.amp.GradScaler's rule), clamped to [min_scale, max_scale]. */
int rdfc_grad_nonfinite(const float *grad, size_t n, float *ostate, void *stream);
int rdfc_adam_step_scaled(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, const float *hyper,
                          int step, float beta1, float beta2, float eps, float weight_decay, float grad_scale, float *scaler,
                          float *ostate, void *stream);
int rdfc_loss_scale_update(float *scaler, float growth, float backoff, int interval, float min_scale, float max_scale, void *stream);

/* Eval path: per-image RMSE, MAE, iRMSE, iMAE, REL, delta^1, delta^2, delta^3 of
 * RDFGANMetric.evaluate_all (lib/metrics/rdf_gan_metric.py:59-151) -> out[N][8] (f32), computed from the NORMALISED
 * maps gt_norm / pred_norm [N][HW] with the evaluator's de-normalisation x * std + mean folded in
 * (lib/evaluator/evaluator.py:28-29); valid pixels: gt > t_valid (and evaluate_mask != 0 when given, u8 [N][HW]).
 * The caller averages over images (rdf_gan_metric.py:139-140). */
int rdfc_depth_metrics(const float *gt_norm, const float *pred_norm, const uint8_t *evaluate_mask, int N, int HW, float std,
                       float mean, float t_valid, float *out, void *stream);

/* ------------------------------------------------------------------------------
 * Semantic-label net: the operators of PSPNet / dilated ResNet-50 that are not convolutions or
 * normalisations (lib/models/generator/label_net/labels_model.py:12-94, backbone_resnet50.py:150-235) and the
 * label loss (rdfc_gan.py:59,238-239).  Pixel-major maps [N][H][W][C] with leading dimension ld (elements), dtype
 * RDFC_F32 | RDFC_BF16; C and ld multiples of the 16-byte vector (4 | 8) except for the bilinear resize and the
 * cross entropy, which take any C (the 14-class logits).  Every backward is a gather: no atomics.
 * ------------------------------------------------------------------------------ */
/* nn.MaxPool2d(k, s, p) (backbone_resnet50.py:172): first maximum in scan order wins, NaN propagates;
 * idx u8 [N][Ho][Wo][C] = kh * k + kw of the winner, consumed by the backward. */
int rdfc_maxpool2d_forward(const void *x, void *y, uint8_t *idx, int N, int Hi, int Wi, int C, int ldx, int ldy, int k, int s,
                           int p, int Ho, int Wo, int dtype, void *stream);
int rdfc_maxpool2d_backward(const void *dy, const uint8_t *idx, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx,
                            int k, int s, int p, int Ho, int Wo, int dtype, void *stream);
/* nn.AdaptiveAvgPool2d((Ho, Wo)) (labels_model.py:28): bin = [floor(o * I / O), ceil((o + 1) * I / O)). */
int rdfc_adaptive_avgpool_forward(const void *x, void *y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho, int Wo,
                                  int dtype, void *stream);
int rdfc_adaptive_avgpool_backward(const void *dy, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx, int Ho, int Wo,
                                   int dtype, void *stream);
/* F.interpolate(size=(Ho, Wo), mode='bilinear', align_corners=True) (labels_model.py:37-38,82,87).  out_f32 / dy_f32:
 * the resized map (its gradient) is f32 although the source is bf16 -- the full-resolution logits feed the loss. */
int rdfc_bilinear_forward(const void *x, void *y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho, int Wo, int dtype,
                          int out_f32, void *stream);
int rdfc_bilinear_backward(const void *dy, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx, int Ho, int Wo,
                           int dtype, int dy_f32, void *stream);
/* nn.Dropout2d's arithmetic (labels_model.py:23,71): y = x * scale[n][c], scale = keep / (1 - p) drawn by the caller. */
int rdfc_scale_nc(const void *x, const float *scale, void *y, int N, int HW, int C, int ldx, int ldy, int dtype, void *stream);
/* torch.nn.CrossEntropyLoss() (rdfc_gan.py:59): logits [P][C] (ld), labels i64 [P]; mean of lse(z) - z[t] over the
 * pixels with t != ignore_index.  partial: f64 [2 * rdfc_cross_entropy_nchunk(P)] scratch; loss_out / count_out: f32
 * scalars on the device.  Backward: dlogits = (softmax(z) - onehot(t)) * grad_scale * (*grad_scale_dev, if given)
 * / count, zero rows for ignored pixels. */
int rdfc_cross_entropy_nchunk(int P);
int rdfc_cross_entropy(const void *logits, const int64_t *labels, int P, int C, int ld, int dtype, int64_t ignore_index,
                       double *partial, float *loss_out, float *count_out, void *stream);
int rdfc_cross_entropy_backward(const void *logits, const int64_t *labels, int P, int C, int ld, int dtype,
                                int64_t ignore_index, const float *count, const float *grad_scale_dev, float grad_scale,
                                void *dlogits, int lddl, void *stream);

/* nor_loss (lib/losses/gan_loss.py:122-132) and manhattan_loss (:50-99): out8 = {nor, floor, wall, ceiling} -- each
 * sum / (count of non-zero terms + 1e-6), unweighted -- followed by the four counts (consumed by the backward).  normal / gt_normal f32 [N][3][HW], norm_mask f32 [N][HW] or NULL,
 * logits f32 pixel-major [N*HW][C] (ld) -- their arg-max gives the wall / floor / ceiling masks.  gt_normal NULL skips
 * nor_loss, logits NULL skips the manhattan terms.  partial: f64 [8 * rdfc_normal_losses_nchunk(HW)].
 * manhattan_loss's [B,H,W] x [B,1,H,W] product broadcasts to [B,B,H,W] upstream (:73-74): every image's term at a pixel is
 * counted once per image of the batch whose pixel has the class; reproduced. */
int rdfc_normal_losses_nchunk(int P);
int rdfc_normal_losses(const float *normal, const float *gt_normal, const float *norm_mask, const float *logits, int N, int HW,
                       int C, int ld, int label_wall, int label_floor, int label_ceiling, double *partial, float *out8,
                       void *stream);
/* grad_normal f32 [N][3][HW] = d(grad4 . {nor, floor, wall, ceiling}) / d normal (grad4: f32 [4] on the device, the
 * cotangents of the four terms).  Masks and counts are constants of the graph, as upstream (dot.detach(), arg-max). */
int rdfc_normal_losses_backward(const float *normal, const float *gt_normal, const float *norm_mask, const float *logits, int N,
                                int HW, int C, int ld, int label_wall, int label_floor, int label_ceiling, const float *out8,
                                const float *grad4, float *grad_normal, void *stream);

/* addPepperNoise (helper.py:384-402) given the host-drawn mask (0 keep, 1 -> +1, 2 -> -1), in place. */
/* rdfc_draw_pepper_mask: the same per-pixel distribution drawn on the device (u8 [N][HW]; P(0) = snr, P(1) = P(2) =
 * (1 - snr) / 2) from a stateless counter-based generator keyed by (seed, counter, pixel): SURVEY 8f rank 4.  A different random
 * stream than upstream's numpy draw, hence opt-in. */
int rdfc_draw_pepper_mask(uint8_t *mask, int N, int HW, uint64_t seed, uint64_t counter, float snr, void *stream);
int rdfc_apply_pepper(float *img, const uint8_t *mask, int N, int C, int HW, void *stream);

/* ------------------------------------------------------------------------------
 * NLSPN refinement (rdf_generator/nlspn/nlspn_model.py), fused around the single-channel 3x3 deformable gather.
 * All maps f32, NCHW planes [B][c][H][W] as upstream's ModulatedDeformConvFunction takes them, except offset_aff.
 * ------------------------------------------------------------------------------ */
/* NLPSN._get_offset_affinity (nlspn_model.py:68-138) behind the guidance conv: offset_aff = conv_offset_aff(guidance),
 * pixel-major [B][H][W][24] (pixel stride ld, dtype f32 | bf16; channels 2k, 2k+1 = (dh, dw) of neighbour k, 16+k = its raw
 * affinity) -> offset [B][18][H][W] (zero reference tap inserted at tap 4, :77-80), aff [B][9][H][W]: tanh / scale
 * (affinity 2 'TC', 3 'TGASS': / (scale + 1e-8), :84-87), x the eight 1x1 deformable confidence gathers with detached
 * offsets (confidence != NULL, :96-118; w_conf / b_conf: device scalars, NULL = 1 / 0), / max-clamped abs-sum (0 'AS',
 * 1 'ASS', 3, :121-128), reference weight 1 - sum (:130-136).  conf_samples [B][8][H][W] receives the gathered
 * confidences (kept for the backward). */
int rdfc_nlspn_affinity_forward(const void *offset_aff, int ld, int dtype, const float *confidence, const float *scale,
                                const float *w_conf, const float *b_conf, int affinity, int B, int H, int W, float *offset,
                                float *aff, float *conf_samples, void *stream);
/* its backward: grad_offset [B][18][H][W] (may be NULL) and grad_aff [B][9][H][W] -> grad_offset_aff (pixel-major, stride
 * ldg, same dtype as offset_aff), grad_confidence [B][H][W] (may be NULL; zero-filled here, then scattered into),
 * grad_scale (device f32, may be NULL; zero-filled here). */
int rdfc_nlspn_affinity_backward(const void *offset_aff, int ld, int dtype, const float *conf_samples, const float *scale,
                                 const float *w_conf, int affinity, int B, int H, int W, const float *grad_offset,
                                 const float *grad_aff, void *grad_offset_aff, int ldg, float *grad_confidence,
                                 float *grad_scale, void *stream);
/* NLPSN.forward's propagation loop (:166-173, preserve_input = False): T steps of
 *   feat <- ModulatedDeformConvFunction(feat, offset, aff, w, b, stride 1, pad 1, dil 1, groups 1, deformable_groups 1)
 * feats [T+1][B][H][W]: slot 0 receives a copy of feat0, slot i the map after step i (the saved inputs of the backward;
 * slot T is the result).  weight f32[9], bias f32[1] or NULL: device pointers (the module's frozen `w`, `b`). */
int rdfc_nlspn_propagate_forward(const float *feat0, const float *offset, const float *aff, const float *weight, const float *bias,
                                 int B, int H, int W, int T, float *feats, void *stream);
/* backward of the loop.  grads [T+1][B][H][W]: slot T holds the cotangent of the result on entry; slots T-1 .. 0 are filled
 * (slot 0 = gradient of feat0).  grad_offset / grad_aff (both or neither): the sums over all T steps, one pass. */
int rdfc_nlspn_propagate_backward(const float *feats, const float *offset, const float *aff, const float *weight, int B, int H, int W,
                                  int T, float *grads, float *grad_offset, float *grad_aff, void *stream);

/* ------------------------------------------------------------------------------
 * Training-batch preparation (lib/datasets/nyuv2/nyuv2_dataset_training.py:75-134 -> get_train_data's sample dict),
 * whole batch, on the device; bit-identical to the PIL / torchvision chain upstream runs per sample on the host:
 * CutOffBlackBorder (preprocessing.py:312-327) -> TF.rotate(angle, NEAREST) -> T.Resize(height) (bilinear; labels nearest)
 * -> T.CenterCrop -> ToTensor -> masks (:119-120,130) -> T.Normalize (:126-133).
 * ------------------------------------------------------------------------------ */
typedef struct rdfc_prep_desc {
    int32_t B, Hraw, Wraw;      /* decoded images: [B][Hraw][Wraw]([3])                                */
    int32_t top, left, Hc, Wc;  /* the window CutOffBlackBorder keeps                                   */
    int32_t Hr, Wr;             /* size after T.Resize                                                  */
    int32_t cy0, cx0, Ho, Wo;   /* T.CenterCrop: origin in the resized image, output size               */
    int32_t ks;                 /* taps per row of the coefficient tables (Resample.c ksize)            */
} rdfc_prep_desc;
/* rgb / normal: uint8 RGB interleaved; raw_mm / gt_mm: uint16 millimetres; labels uint8.  affine: [B][6] int32, the
 * 16.16 fixed-point matrix of Geometry.c affine_fixed for each sample's rotation (host-computed from the angle).
 * h_* / v_*: Resample.c precompute_coeffs tables for the two axes: bounds [n][2] (first tap, count), k8 [n][ks] (22-bit
 * integer coefficients, 8-bit images), kd [n][ks] (double, 'F' images); x_nearest / y_nearest: ImagingScaleAffine's source
 * index per resized coordinate (labels).  mean_std: device f32 {rgb mean[3], rgb std[3], depth mean, depth std}.
 * Outputs: f32 [B][3][Ho][Wo] x2, f32 [B][1][Ho][Wo] x2, u8 (bool) masks [B][1][Ho][Wo] / [B][Ho][Wo], i64 labels. */
int rdfc_nyu_prepare_batch(const rdfc_prep_desc *d, const uint8_t *rgb, const uint8_t *normal, const uint16_t *raw_mm,
                           const uint16_t *gt_mm, const uint8_t *labels, const int32_t *affine, const int32_t *h_bounds,
                           const int32_t *h_k8, const double *h_kd, const int32_t *v_bounds, const int32_t *v_k8,
                           const double *v_kd, const int32_t *x_nearest, const int32_t *y_nearest, const float *mean_std,
                           float *out_rgb, float *out_normal, float *out_raw, float *out_gt, uint8_t *out_depth_mask,
                           uint8_t *out_normal_mask, int64_t *out_labels, void *stream);

/* ------------------------------------------------------------------------------
 * Small elementwise operators that keep stock-framework arithmetic off the path.
 * ------------------------------------------------------------------------------ */
/* hipMemsetAsync(p, 0, bytes) on `stream`. */
int rdfc_zero(void *p, size_t bytes, void *stream);
/* backward of addPepperNoise's in-place assignment (helper.py:384-402): out = mask == 0 ? g : 0; g/out [N][C][HW] f32,
 * mask [N][HW] u8. */
int rdfc_mask_keep(const float *g, const uint8_t *mask, float *out, int N, int C, int HW, void *stream);
/* out[i] = x[i] * (*scalar) * k; scalar: device f32 (NULL = 1): a stored loss gradient times the loss's cotangent
 * (gan_loss.py:6-42 backward). */
int rdfc_scale_dev(const float *x, const float *scalar, float k, float *out, size_t n, void *stream);
/* helper.py:404-408 norm_normalize on channels 0..2 of an NCHW f32 map with Csrc >= 3 channels -> [N][3][HW]. */
int rdfc_unit_normals(const float *src, float *dst, int N, int Csrc, int HW, void *stream);
/* out[i] = weights[i] * *terms[i] (i < n <= 16), out[n] = post * (sequential f32 sum of them, in order): loss_G and its logged
 * terms (rdfc_gan.py:252-288) / loss_D = (real + fake) * 0.5 (:292-300) in one launch.  terms: HOST array of device
 * pointers, weights: HOST array.  _backward: out[i] = *grad_total * post * weights[i] (grad_total NULL = 1). */
int rdfc_combine_scalars(int n, const float *const *terms, const float *weights, float post, float *out, void *stream);
int rdfc_combine_scalars_backward(int n, const float *weights, float post, const float *grad_total, float *out, void *stream);
/* dst[P][C] (pixel stride ld_dst) = sum over n <= 4 pixel-major maps srcs[k] (HOST array of device pointers, pixel strides
 * lds[k], HOST array): the gradient of an activation with several consumers in one pass (stands in for autograd's
 * pairwise adds: encoder features feeding the next stage and a skip connection, rdf_generator.py:286-398). */
int rdfc_sum_maps(int n, const void *const *srcs, const int *lds, void *dst, int P, int C, int ld_dst, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RDFC_HIP_H */
