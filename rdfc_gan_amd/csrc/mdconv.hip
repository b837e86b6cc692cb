// mdconv.hip -- modulated deformable convolution (DCNv2) forward / backward, the op
// underneath upstream's `DCN` module (deformconv/src/cuda/modulated_deform_im2col_cuda.cuh,
// modulated_deform_conv_cuda.cu).  NCHW f32 (and f64 through the generic kernels, like upstream's dispatch).
//
// Upstream materialises a [C*K, B*Ho*Wo] column buffer (im2col), runs a K=9 GEMM on it
// and permutes the result.  Here the gather, the mask multiply and the tap sum are one
// kernel: one thread per output pixel, offsets/masks read coalesced along W, nothing
// written but the output.  HBM-bound: per pixel and tap 2 offsets + 1 mask in, plus the
// input and output maps once.
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int NT = 256;
typedef rdfc_mdconv_desc D;

__device__ inline int odim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

template <typename S> struct CornerT {
    int h0, w0;
    S lh, lw;
    bool v00, v01, v10, v11;
};
template <typename S> __device__ inline CornerT<S> corners(S h, S w, int H, int W) {
    CornerT<S> c;
    c.h0 = (int)floor(h);
    c.w0 = (int)floor(w);
    c.lh = h - c.h0;
    c.lw = w - c.w0;
    const bool hl = c.h0 >= 0, hh = c.h0 + 1 <= H - 1, wl = c.w0 >= 0, wh = c.w0 + 1 <= W - 1;
    c.v00 = hl && wl; c.v01 = hl && wh; c.v10 = hh && wl; c.v11 = hh && wh;
    return c;
}
template <typename S> __device__ inline S bilinear(const S *im, int W, const CornerT<S> &c) {
    const S a = c.v00 ? im[c.h0 * W + c.w0] : S(0), b = c.v01 ? im[c.h0 * W + c.w0 + 1] : S(0);
    const S d = c.v10 ? im[(c.h0 + 1) * W + c.w0] : S(0), e = c.v11 ? im[(c.h0 + 1) * W + c.w0 + 1] : S(0);
    const S hh = S(1) - c.lh, hw = S(1) - c.lw;
    return hh * hw * a + hh * c.lw * b + c.lh * hw * d + c.lh * c.lw * e;
}

typedef CornerT<float> Corner;

template <typename S>
__global__ __launch_bounds__(NT) void fwd_kernel(D s, int Ho, int Wo, const S *__restrict__ in, const S *__restrict__ off,
                                                 const S *__restrict__ msk, const S *__restrict__ wt,
                                                 const S *__restrict__ bias, S *__restrict__ out) {
    const int K = s.KH * s.KW, Cg = s.C / s.group, Cog = s.Co / s.group, Cdg = s.C / s.dgroup;
    const size_t HWo = (size_t)Ho * Wo, total = (size_t)s.B * s.Co * HWo;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const int w = (int)(idx % Wo), h = (int)((idx / Wo) % Ho);
        const int co = (int)((idx / HWo) % s.Co), b = (int)(idx / (HWo * s.Co));
        const int g = co / Cog;
        const size_t p = (size_t)h * Wo + w;
        S acc = bias ? bias[co] : S(0);
        for (int cl = 0; cl < Cg; ++cl) {
            const int c = g * Cg + cl, dgi = c / Cdg;
            const S *im = in + ((size_t)b * s.C + c) * s.H * s.W;
            const S *o = off + ((size_t)b * s.dgroup + dgi) * 2 * K * HWo + p;
            const S *m = msk + ((size_t)b * s.dgroup + dgi) * K * HWo + p;
            const S *wp = wt + ((size_t)co * Cg + cl) * K;
            for (int i = 0; i < s.KH; ++i)
                for (int j = 0; j < s.KW; ++j) {
                    const int k = i * s.KW + j;
                    const S hi = h * s.sh - s.ph + i * s.dh + o[(size_t)(2 * k) * HWo];
                    const S wi = w * s.sw - s.pw + j * s.dw + o[(size_t)(2 * k + 1) * HWo];
                    if (hi > -S(1) && wi > -S(1) && hi < s.H && wi < s.W)
                        acc += wp[k] * m[(size_t)k * HWo] * bilinear(im, s.W, corners(hi, wi, s.H, s.W));
                }
        }
        out[idx] = acc;
    }
}

// grad_offset / grad_mask (gather, deterministic) and grad_input (scatter, atomics):
// one thread per (b, deformable group, tap, h, w).
template <typename S>
__global__ __launch_bounds__(NT) void bwd_kernel(D s, int Ho, int Wo, const S *__restrict__ in, const S *__restrict__ off,
                                                 const S *__restrict__ msk, const S *__restrict__ wt,
                                                 const S *__restrict__ go, S *__restrict__ gin, S *__restrict__ goff,
                                                 S *__restrict__ gmsk) {
    const int K = s.KH * s.KW, Cg = s.C / s.group, Cog = s.Co / s.group, Cdg = s.C / s.dgroup;
    const size_t HWo = (size_t)Ho * Wo, total = (size_t)s.B * s.dgroup * K * HWo;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const int w = (int)(idx % Wo), h = (int)((idx / Wo) % Ho);
        const int k = (int)((idx / HWo) % K), dgi = (int)((idx / (HWo * K)) % s.dgroup), b = (int)(idx / (HWo * K * s.dgroup));
        const int i = k / s.KW, j = k % s.KW;
        const size_t p = (size_t)h * Wo + w;
        const size_t ob = ((size_t)b * s.dgroup + dgi) * 2 * K * HWo + p, mb = ((size_t)b * s.dgroup + dgi) * K * HWo + p;
        const S m = msk[mb + (size_t)k * HWo];
        const S hi = h * s.sh - s.ph + i * s.dh + off[ob + (size_t)(2 * k) * HWo];
        const S wi = w * s.sw - s.pw + j * s.dw + off[ob + (size_t)(2 * k + 1) * HWo];
        const bool inside = !(hi <= -S(1) || wi <= -S(1) || hi >= s.H || wi >= s.W);
        S g_oh = S(0), g_ow = S(0), g_m = S(0);
        if (inside) {
            const CornerT<S> c = corners(hi, wi, s.H, s.W);
            const S hh = S(1) - c.lh, hw = S(1) - c.lw;
            for (int cd = 0; cd < Cdg; ++cd) {
                const int ch = dgi * Cdg + cd, g = ch / Cg, cl = ch % Cg;
                S cg = S(0);   // column gradient = W^T * grad_out
                for (int q = 0; q < Cog; ++q) {
                    const int co = g * Cog + q;
                    cg += wt[((size_t)co * Cg + cl) * K + k] * go[((size_t)b * s.Co + co) * HWo + p];
                }
                const S *im = in + ((size_t)b * s.C + ch) * s.H * s.W;
                const S a = c.v00 ? im[c.h0 * s.W + c.w0] : S(0), bb = c.v01 ? im[c.h0 * s.W + c.w0 + 1] : S(0);
                const S d = c.v10 ? im[(c.h0 + 1) * s.W + c.w0] : S(0), e = c.v11 ? im[(c.h0 + 1) * s.W + c.w0 + 1] : S(0);
                g_m += cg * (hh * hw * a + hh * c.lw * bb + c.lh * hw * d + c.lh * c.lw * e);
                g_oh += cg * m * (-hw * a - c.lw * bb + hw * d + c.lw * e);
                g_ow += cg * m * (-hh * a + hh * bb - c.lh * d + c.lh * e);
                if (gin) {
                    S *gi = gin + ((size_t)b * s.C + ch) * s.H * s.W;
                    const S t = cg * m;
                    if (c.v00) atomicAdd(gi + c.h0 * s.W + c.w0, hh * hw * t);
                    if (c.v01) atomicAdd(gi + c.h0 * s.W + c.w0 + 1, hh * c.lw * t);
                    if (c.v10) atomicAdd(gi + (c.h0 + 1) * s.W + c.w0, c.lh * hw * t);
                    if (c.v11) atomicAdd(gi + (c.h0 + 1) * s.W + c.w0 + 1, c.lh * c.lw * t);
                }
            }
        }
        if (goff) {
            goff[ob + (size_t)(2 * k) * HWo] = g_oh;
            goff[ob + (size_t)(2 * k + 1) * HWo] = g_ow;
        }
        if (gmsk) gmsk[mb + (size_t)k * HWo] = g_m;
    }
}

// Single-channel case (NLSPN: C = Co = 1, stride 1, dilation 1): the generic kernel above spends most of its time in
// the 4 corners x K taps global atomics per pixel (36 for a 3x3: ~20 M atomics per propagation step, executed at the
// memory side).  Samples land within a few pixels of their tap, so a workgroup that owns an 8 x 64 pixel tile keeps the
// input window (tile + tap reach + R pixels of margin) and a gradient window of the same shape in LDS: corners inside the
// window are read from / added to LDS (ds_add_f32), only the rare far sample goes to global memory, and the window is
// flushed with one global atomic per touched cell (~1 150 per tile instead of ~18 000).
constexpr int C1_TH = 8, C1_TW = 64, C1_R = 3;

__global__ __launch_bounds__(NT) void bwd_c1_kernel(D s, int Ho, int Wo, int tiles_x, int tiles_y, const float *__restrict__ in,
                                                    const float *__restrict__ off, const float *__restrict__ msk,
                                                    const float *__restrict__ wt, const float *__restrict__ go,
                                                    float *__restrict__ gin, float *__restrict__ goff, float *__restrict__ gmsk) {
    extern __shared__ float sm[];
    const int K = s.KH * s.KW;
    const int WH = C1_TH + (s.KH - 1) + 2 * C1_R, WW = C1_TW + (s.KW - 1) + 2 * C1_R, WN = WH * WW;
    float *win = sm, *gacc = sm + WN;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, b = t / tiles_y;
    const int y0 = ty * C1_TH, x0 = tx * C1_TW;
    const int wy0 = y0 - s.ph - C1_R, wx0 = x0 - s.pw - C1_R;          // window origin in input coordinates
    const float *im = in + (size_t)b * s.H * s.W;
    for (int e = threadIdx.x; e < WN; e += NT) {
        const int r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
        win[e] = ((unsigned)iy < (unsigned)s.H && (unsigned)ix < (unsigned)s.W) ? im[(size_t)iy * s.W + ix] : 0.f;
        gacc[e] = 0.f;
    }
    __syncthreads();
    const size_t HWo = (size_t)Ho * Wo;
    float *gi = gin ? gin + (size_t)b * s.H * s.W : nullptr;
    for (int q = threadIdx.x; q < C1_TH * C1_TW; q += NT) {
        const int h = y0 + q / C1_TW, w = x0 + q % C1_TW;
        if (h >= Ho || w >= Wo) continue;
        const size_t p = (size_t)h * Wo + w;
        const float g = go[(size_t)b * HWo + p];
        const size_t ob = (size_t)b * 2 * K * HWo + p, mb = (size_t)b * K * HWo + p;
        for (int k = 0; k < K; ++k) {
            const int i = k / s.KW, j = k - i * s.KW;
            const float m = msk[mb + (size_t)k * HWo];
            const float hi = h - s.ph + i + off[ob + (size_t)(2 * k) * HWo];
            const float wi = w - s.pw + j + off[ob + (size_t)(2 * k + 1) * HWo];
            float g_oh = 0.f, g_ow = 0.f, g_m = 0.f;
            if (!(hi <= -1.f || wi <= -1.f || hi >= s.H || wi >= s.W)) {
                const Corner c = corners(hi, wi, s.H, s.W);
                const float hh = 1.f - c.lh, hw = 1.f - c.lw;
                const float cg = wt[k] * g;
                const int r0 = c.h0 - wy0, c0 = c.w0 - wx0;              // window coordinates of the top-left corner
                const bool inwin = r0 >= 0 && c0 >= 0 && r0 + 1 < WH && c0 + 1 < WW;
                float a, bb, d, e;
                if (inwin) {
                    const float *wp = win + r0 * WW + c0;
                    a = c.v00 ? wp[0] : 0.f; bb = c.v01 ? wp[1] : 0.f; d = c.v10 ? wp[WW] : 0.f; e = c.v11 ? wp[WW + 1] : 0.f;
                } else {
                    a = c.v00 ? im[(size_t)c.h0 * s.W + c.w0] : 0.f; bb = c.v01 ? im[(size_t)c.h0 * s.W + c.w0 + 1] : 0.f;
                    d = c.v10 ? im[(size_t)(c.h0 + 1) * s.W + c.w0] : 0.f; e = c.v11 ? im[(size_t)(c.h0 + 1) * s.W + c.w0 + 1] : 0.f;
                }
                g_m = cg * (hh * hw * a + hh * c.lw * bb + c.lh * hw * d + c.lh * c.lw * e);
                g_oh = cg * m * (-hw * a - c.lw * bb + hw * d + c.lw * e);
                g_ow = cg * m * (-hh * a + hh * bb - c.lh * d + c.lh * e);
                if (gi) {
                    const float tt = cg * m;
                    if (inwin) {
                        float *ga = gacc + r0 * WW + c0;
                        if (c.v00) atomicAdd(ga, hh * hw * tt);
                        if (c.v01) atomicAdd(ga + 1, hh * c.lw * tt);
                        if (c.v10) atomicAdd(ga + WW, c.lh * hw * tt);
                        if (c.v11) atomicAdd(ga + WW + 1, c.lh * c.lw * tt);
                    } else {
                        if (c.v00) atomicAdd(gi + (size_t)c.h0 * s.W + c.w0, hh * hw * tt);
                        if (c.v01) atomicAdd(gi + (size_t)c.h0 * s.W + c.w0 + 1, hh * c.lw * tt);
                        if (c.v10) atomicAdd(gi + (size_t)(c.h0 + 1) * s.W + c.w0, c.lh * hw * tt);
                        if (c.v11) atomicAdd(gi + (size_t)(c.h0 + 1) * s.W + c.w0 + 1, c.lh * c.lw * tt);
                    }
                }
            }
            if (goff) {
                goff[ob + (size_t)(2 * k) * HWo] = g_oh;
                goff[ob + (size_t)(2 * k + 1) * HWo] = g_ow;
            }
            if (gmsk) gmsk[mb + (size_t)k * HWo] = g_m;
        }
    }
    if (gi == nullptr) return;
    __syncthreads();
    for (int e = threadIdx.x; e < WN; e += NT) {
        const float v = gacc[e];
        if (v != 0.f) {
            const int r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
            if ((unsigned)iy < (unsigned)s.H && (unsigned)ix < (unsigned)s.W) atomicAdd(gi + (size_t)iy * s.W + ix, v);
        }
    }
}

// grad_weight[co][cl][k] (block per element) and grad_bias[co] (extra blocks), block reduction over pixels
template <typename S>
__global__ __launch_bounds__(NT) void wgrad_kernel(D s, int Ho, int Wo, const S *__restrict__ in, const S *__restrict__ off,
                                                   const S *__restrict__ msk, const S *__restrict__ go,
                                                   S *__restrict__ gw, S *__restrict__ gb) {
    const int K = s.KH * s.KW, Cg = s.C / s.group, Cog = s.Co / s.group, Cdg = s.C / s.dgroup;
    const size_t HWo = (size_t)Ho * Wo, npix = (size_t)s.B * HWo;
    const int nW = s.Co * Cg * K;
    const int e = blockIdx.x;
    S acc = S(0);
    if (e < nW) {
        const int k = e % K, cl = (e / K) % Cg, co = e / (K * Cg);
        const int i = k / s.KW, j = k % s.KW, g = co / Cog, c = g * Cg + cl, dgi = c / Cdg;
        for (size_t q = threadIdx.x; q < npix; q += NT) {
            const int b = (int)(q / HWo);
            const size_t p = q % HWo;
            const int h = (int)(p / Wo), w = (int)(p % Wo);
            const size_t ob = ((size_t)b * s.dgroup + dgi) * 2 * K * HWo + p, mb = ((size_t)b * s.dgroup + dgi) * K * HWo + p;
            const S hi = h * s.sh - s.ph + i * s.dh + off[ob + (size_t)(2 * k) * HWo];
            const S wi = w * s.sw - s.pw + j * s.dw + off[ob + (size_t)(2 * k + 1) * HWo];
            if (hi > -S(1) && wi > -S(1) && hi < s.H && wi < s.W)
                acc += go[((size_t)b * s.Co + co) * HWo + p] * msk[mb + (size_t)k * HWo] *
                       bilinear(in + ((size_t)b * s.C + c) * s.H * s.W, s.W, corners(hi, wi, s.H, s.W));
        }
    } else {
        const int co = e - nW;
        for (size_t q = threadIdx.x; q < npix; q += NT) acc += go[((size_t)(q / HWo) * s.Co + co) * HWo + q % HWo];
    }
    for (int off_ = 32; off_ > 0; off_ >>= 1) acc += __shfl_down(acc, off_, 64);
    __shared__ S red[NT / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const S t = red[0] + red[1] + red[2] + red[3];
        if (e < nW) { if (gw) gw[e] = t; }
        else if (gb) gb[e - nW] = t;
    }
}

int check(const D *d, const char *who) {
    RDFC_REQUIRE(d != nullptr, "%s: null descriptor", who);
    RDFC_REQUIRE(d->B > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->Co > 0 && d->KH > 0 && d->KW > 0, "%s: non-positive size", who);
    RDFC_REQUIRE(d->group > 0 && d->dgroup > 0 && d->C % d->group == 0 && d->Co % d->group == 0,
                 "%s: channels(%d) and channels_out(%d) must divide group(%d)", who, d->C, d->Co, d->group);   // cu:62
    RDFC_REQUIRE(d->C % d->dgroup == 0, "%s: channels(%d) must divide deformable_group(%d)", who, d->C, d->dgroup);
    RDFC_REQUIRE(d->sh > 0 && d->sw > 0 && d->dh > 0 && d->dw > 0, "%s: bad stride / dilation", who);
    return 0;
}
inline int host_odim(int in, int pad, int dil, int k, int stride) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }

}  // namespace

template <typename S>
static int mdconv_forward_t(const rdfc_mdconv_desc *d, const S *input, const S *offset, const S *mask, const S *weight, const S *bias,
                            S *output, void *stream) {
    if (int e = check(d, "rdfc_mdconv_forward")) return e;
    const int Ho = host_odim(d->H, d->ph, d->dh, d->KH, d->sh), Wo = host_odim(d->W, d->pw, d->dw, d->KW, d->sw);
    RDFC_REQUIRE(Ho > 0 && Wo > 0, "rdfc_mdconv_forward: empty output");
    const size_t total = (size_t)d->B * d->Co * Ho * Wo;
    const int nb = (int)min((size_t)8192, (total + NT - 1) / NT);
    hipLaunchKernelGGL(fwd_kernel<S>, dim3(nb), dim3(NT), 0, (hipStream_t)stream, *d, Ho, Wo, input, offset, mask, weight, bias, output);
    RDFC_LAUNCH_CHECK("rdfc_mdconv_forward");
    return 0;
}

template <typename S>
static int mdconv_backward_t(const rdfc_mdconv_desc *d, const S *input, const S *offset, const S *mask, const S *weight, const S *grad_out,
                             S *grad_input, S *grad_offset, S *grad_mask, S *grad_weight, S *grad_bias, void *stream) {
    if (int e = check(d, "rdfc_mdconv_backward")) return e;
    const int Ho = host_odim(d->H, d->ph, d->dh, d->KH, d->sh), Wo = host_odim(d->W, d->pw, d->dw, d->KW, d->sw);
    RDFC_REQUIRE(Ho > 0 && Wo > 0, "rdfc_mdconv_backward: empty output");
    hipStream_t st = (hipStream_t)stream;
    if (grad_input) {
        if (hipMemsetAsync(grad_input, 0, sizeof(S) * d->B * d->C * d->H * d->W, st) != hipSuccess) {
            rdfc_set_error("rdfc_mdconv_backward: memset failed");
            return 2;
        }
    }
    static const bool c1_off = getenv("RDFC_MDCONV_GENERIC") != nullptr;
    const bool c1 = sizeof(S) == 4 && !c1_off && d->C == 1 && d->Co == 1 && d->group == 1 && d->dgroup == 1 && d->sh == 1 && d->sw == 1 &&
                    d->dh == 1 && d->dw == 1 && d->KH <= 3 && d->KW <= 3;
    if ((grad_input || grad_offset || grad_mask) && c1) {
        const int tiles_x = (Wo + C1_TW - 1) / C1_TW, tiles_y = (Ho + C1_TH - 1) / C1_TH;
        const size_t smem = 2 * sizeof(float) * (C1_TH + d->KH - 1 + 2 * C1_R) * (C1_TW + d->KW - 1 + 2 * C1_R);
        if constexpr (sizeof(S) == 4)
            hipLaunchKernelGGL(bwd_c1_kernel, dim3(tiles_x * tiles_y * d->B), dim3(NT), smem, st, *d, Ho, Wo, tiles_x, tiles_y, input,
                               offset, mask, weight, grad_out, grad_input, grad_offset, grad_mask);
    } else if (grad_input || grad_offset || grad_mask) {
        const size_t total = (size_t)d->B * d->dgroup * d->KH * d->KW * Ho * Wo;
        const int nb = (int)min((size_t)8192, (total + NT - 1) / NT);
        hipLaunchKernelGGL(bwd_kernel<S>, dim3(nb), dim3(NT), 0, st, *d, Ho, Wo, input, offset, mask, weight, grad_out, grad_input,
                           grad_offset, grad_mask);
    }
    if (grad_weight || grad_bias) {
        const int nW = d->Co * (d->C / d->group) * d->KH * d->KW;
        hipLaunchKernelGGL(wgrad_kernel<S>, dim3(nW + d->Co), dim3(NT), 0, st, *d, Ho, Wo, input, offset, mask, grad_out, grad_weight, grad_bias);
    }
    RDFC_LAUNCH_CHECK("rdfc_mdconv_backward");
    return 0;
}

extern "C" int rdfc_mdconv_forward(const rdfc_mdconv_desc *d, const float *input, const float *offset, const float *mask,
                                   const float *weight, const float *bias, float *output, void *stream) {
    return mdconv_forward_t<float>(d, input, offset, mask, weight, bias, output, stream);
}
extern "C" int rdfc_mdconv_backward(const rdfc_mdconv_desc *d, const float *input, const float *offset, const float *mask,
                                    const float *weight, const float *grad_out, float *grad_input, float *grad_offset,
                                    float *grad_mask, float *grad_weight, float *grad_bias, void *stream) {
    return mdconv_backward_t<float>(d, input, offset, mask, weight, grad_out, grad_input, grad_offset, grad_mask, grad_weight, grad_bias, stream);
}
// double precision: upstream dispatches AT_DISPATCH_FLOATING_TYPES (modulated_deform_conv_cuda.cu:94) -- the generic kernels in f64
extern "C" int rdfc_mdconv_forward_f64(const rdfc_mdconv_desc *d, const double *input, const double *offset, const double *mask,
                                       const double *weight, const double *bias, double *output, void *stream) {
    return mdconv_forward_t<double>(d, input, offset, mask, weight, bias, output, stream);
}
extern "C" int rdfc_mdconv_backward_f64(const rdfc_mdconv_desc *d, const double *input, const double *offset, const double *mask,
                                        const double *weight, const double *grad_out, double *grad_input, double *grad_offset,
                                        double *grad_mask, double *grad_weight, double *grad_bias, void *stream) {
    return mdconv_backward_t<double>(d, input, offset, mask, weight, grad_out, grad_input, grad_offset, grad_mask, grad_weight, grad_bias, stream);
}
