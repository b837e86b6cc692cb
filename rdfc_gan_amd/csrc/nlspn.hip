// nlspn.hip -- NLSPN refinement (nlspn_model.py:68-175) as four fused kernels around the single-channel modulated
// deformable gather (modulated_deform_im2col_cuda.cuh:24-328 with C = Co = 1, 3x3, stride 1, pad 1, dilation 1).
//
//   affinity_fwd    conv_offset_aff's 24-channel output -> offsets [B,18,H,W] + affinities [B,9,H,W]: offset packing with
//                   the zero reference tap, tanh / TGASS scaling, the eight 1x1 confidence gathers, abs-sum normalisation
//                   and the reference weight -- upstream: ~15 elementwise launches + 8 deformable-conv launches
//   affinity_bwd    its backward (gradient of the 24 channels, of the confidence map -- scattered through a fixed-point LDS window --
//                   and of aff_scale_const)
//   prop_fwd        one propagation step: out[p] = b + sum_t w_t aff_t(p) bilinear(in, p + D_t + off_t(p))
//   prop_bwd_feat   gradient of one step w.r.t. its input map only (the scatter)
//   prop_bwd_coeff  gradients of offsets and affinities SUMMED OVER ALL STEPS in one pass: a thread keeps the sums of one pixel's
//                   three taps (blockIdx.y = tap group) in registers while the T saved input maps stream through LDS
//
// Why this split: offsets and affinities are shared by the T = 18 steps, so their gradients are sums over the steps.
// Step by step (what autograd does with upstream's loop) that is 18 x (27 gradient planes written + 27 added by an
// elementwise kernel); here the chain of steps only carries the 1-channel cotangent (116 B per pixel and step, like the
// forward) and the 27 sums are produced once.  HBM-bound: prop_fwd / prop_bwd_feat read 18 offsets + 9 affinities + 1 map
// value and write 1 per pixel (116 B); sampled corners come from an LDS window of the map (tile + tap reach + 3 px of
// margin), samples beyond it from global memory.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int TH = 8, TW = 64, R = 3;                 // pixel tile of a workgroup; window margin beyond the tap reach
constexpr int WH = TH + 2 + 2 * R, WW = TW + 2 + 2 * R, WN = WH * WW;
constexpr int PPT = TH * TW / NT;                     // pixels per thread (2)
constexpr int NB = 8, NTAP = 9, REF = 4;              // neighbours, taps (3x3), index of the reference tap

struct Geo {
    int B, H, W, tiles_x, tiles_y;
};

struct Smp {                                          // one bilinear sample (modulated_deform_im2col_cuda.cuh:24-54)
    int h0, w0;
    float lh, lw;
    bool v00, v01, v10, v11;
};
__device__ __forceinline__ Smp sample(float h, float w, int H, int W) {
    Smp c;
    c.h0 = (int)floorf(h);
    c.w0 = (int)floorf(w);
    c.lh = h - c.h0;
    c.lw = w - c.w0;
    const bool hl = c.h0 >= 0, hh = c.h0 + 1 <= H - 1, wl = c.w0 >= 0, wh = c.w0 + 1 <= W - 1;
    c.v00 = hl && wl; c.v01 = hl && wh; c.v10 = hh && wl; c.v11 = hh && wh;
    return c;
}
__device__ __forceinline__ bool inside(float h, float w, int H, int W) { return h > -1.f && w > -1.f && h < H && w < W; }

// corner values of a sample: from the LDS window when all four cells are in it, else from global memory
__device__ __forceinline__ void corners(const Smp &c, const float *__restrict__ win, int wy0, int wx0, const float *__restrict__ im,
                                        int W, float &a, float &b, float &d, float &e) {
    const int r0 = c.h0 - wy0, c0 = c.w0 - wx0;
    if (__builtin_expect(r0 >= 0 && c0 >= 0 && r0 + 1 < WH && c0 + 1 < WW, 1)) {
        // (the window holds zeros outside the image, so a corner that is out of the image reads 0 from it: no per-corner test)
        const float *wp = win + r0 * WW + c0;
        a = wp[0]; b = wp[1]; d = wp[WW]; e = wp[WW + 1];
    } else {
        a = c.v00 ? im[(size_t)c.h0 * W + c.w0] : 0.f; b = c.v01 ? im[(size_t)c.h0 * W + c.w0 + 1] : 0.f;
        d = c.v10 ? im[(size_t)(c.h0 + 1) * W + c.w0] : 0.f; e = c.v11 ? im[(size_t)(c.h0 + 1) * W + c.w0 + 1] : 0.f;
    }
}
__device__ __forceinline__ float bilin(const Smp &c, float a, float b, float d, float e) {
    const float hh = 1.f - c.lh, hw = 1.f - c.lw;
    return hh * hw * a + hh * c.lw * b + c.lh * hw * d + c.lh * c.lw * e;
}

__device__ __forceinline__ void tile_of(const Geo &g, int &b, int &y0, int &x0) {
    int t = blockIdx.x;
    const int tx = t % g.tiles_x;
    t /= g.tiles_x;
    const int ty = t % g.tiles_y;
    b = t / g.tiles_y;
    y0 = ty * TH;
    x0 = tx * TW;
}
template <int NTH = NT>
__device__ __forceinline__ void load_window(float *win, const float *__restrict__ im, int wy0, int wx0, int H, int W) {
    for (int e = threadIdx.x; e < WN; e += NTH) {
        const int r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
        win[e] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? im[(size_t)iy * W + ix] : 0.f;
    }
}

// ---- one propagation step ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void prop_fwd_kernel(Geo g, const float *__restrict__ in, const float *__restrict__ off,
                                                      const float *__restrict__ aff, const float *__restrict__ wt,
                                                      const float *__restrict__ bias, float *__restrict__ out) {
    __shared__ float win[WN];
    int b, y0, x0;
    tile_of(g, b, y0, x0);
    const int H = g.H, W = g.W;
    const size_t HW = (size_t)H * W;
    const int wy0 = y0 - 1 - R, wx0 = x0 - 1 - R;
    const float *im = in + (size_t)b * HW;
    // the pixel's 27 coefficients first, the window of the map after them: both fetches are in flight together (with the window
    // load and its barrier in front, a step was two memory latencies back to back)
    float hi[PPT][NTAP], wi[PPT][NTAP], cf[PPT][NTAP];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pix = threadIdx.x + q * NT, h = y0 + pix / TW, w = x0 + pix % TW;
        const bool live = h < H && w < W;
        const size_t p = live ? (size_t)h * W + w : 0;
        const float *o = off + (size_t)b * 2 * NTAP * HW + p, *m = aff + (size_t)b * NTAP * HW + p;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            hi[q][t] = h - 1 + t / 3 + o[(size_t)(2 * t) * HW];
            wi[q][t] = w - 1 + t % 3 + o[(size_t)(2 * t + 1) * HW];
            cf[q][t] = wt[t] * m[(size_t)t * HW];
        }
    }
    load_window(win, im, wy0, wx0, H, W);
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pix = threadIdx.x + q * NT, h = y0 + pix / TW, w = x0 + pix % TW;
        if (h >= H || w >= W) continue;
        float acc = b0;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            if (inside(hi[q][t], wi[q][t], H, W)) {
                const Smp c = sample(hi[q][t], wi[q][t], H, W);
                float a, bb, d, e;
                corners(c, win, wy0, wx0, im, W, a, bb, d, e);
                acc += cf[q][t] * bilin(c, a, bb, d, e);
            }
        }
        out[(size_t)b * HW + (size_t)h * W + w] = acc;
    }
}

// ---- gradient of one step w.r.t. its input map (scatter) ---------------------------------------------------------------
// LDS-privatised: a workgroup adds its 512 pixels x 9 taps x 4 corners into a window in LDS and flushes the window with one
// global atomic per touched cell.  The window accumulates in 64-BIT FIXED POINT: LDS float atomics run at 0.34 lane-adds
// per clock and CU on gfx950, integer ones at 7-10 (tools/micro/lds_atomics.hip) -- with ds_add_f32 this kernel took 247 us
// against 44 us for the forward gather of the same bytes.  The scale is a power of two chosen per workgroup from the largest
// contribution it is going to add (2^46 / that, rounded down to a power of two): one contribution is below 2^46, a window
// cell receives at most 512 * 9 * 4 < 2^15 of them, so the sum cannot overflow, and every add is exact to 2^-46 of the
// largest one -- integer adds are also order-independent, so a window's sums are bit-reproducible.
__global__ __launch_bounds__(NT) void prop_bwd_feat_kernel(Geo g, const float *__restrict__ off, const float *__restrict__ aff,
                                                           const float *__restrict__ wt, const float *__restrict__ gout,
                                                           float *__restrict__ gin /* zero-filled */) {
    __shared__ long long gacc[WN];
    __shared__ float wmax[NT / 64];
    int b, y0, x0;
    tile_of(g, b, y0, x0);
    const int H = g.H, W = g.W;
    const size_t HW = (size_t)H * W;
    const int wy0 = y0 - 1 - R, wx0 = x0 - 1 - R;
    for (int e = threadIdx.x; e < WN; e += NT) gacc[e] = 0;
    float *gi = gin + (size_t)b * HW;
    // this thread's pixels: cotangent x tap weight x affinity per tap, sampling positions
    float tt[PPT][NTAP], hi[PPT][NTAP], wi[PPT][NTAP];
    float big = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pix = threadIdx.x + q * NT, h = y0 + pix / TW, w = x0 + pix % TW;
        const bool live = h < H && w < W;
        const size_t p = live ? (size_t)h * W + w : 0;
        const float go = live ? gout[(size_t)b * HW + p] : 0.f;
        const float *o = off + (size_t)b * 2 * NTAP * HW + p, *m = aff + (size_t)b * NTAP * HW + p;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            hi[q][t] = h - 1 + t / 3 + o[(size_t)(2 * t) * HW];
            wi[q][t] = w - 1 + t % 3 + o[(size_t)(2 * t + 1) * HW];
            tt[q][t] = live ? wt[t] * go * m[(size_t)t * HW] : 0.f;
            big = fabsf(tt[q][t]) <= 3.0e38f ? fmaxf(big, fabsf(tt[q][t])) : __builtin_inff();      // NaN / inf: flagged, not dropped
        }
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) big = fmaxf(big, __shfl_xor(big, o_, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = big;
    __syncthreads();
    big = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    if (big == 0.f) return;                             // nothing to add (uniform over the workgroup)
    int ex = 0;
    if (big <= 3.0e38f) frexpf(big, &ex);               // big < 2^ex
    // fixed point needs a finite cotangent (NaN / inf go through the float atomics below and stay visible) that is not so
    // small that 2^(46 - ex) leaves the float range (|values| < 2^-80: float path as well)
    const bool fixed = big <= 3.0e38f && 46 - ex <= 126;
    const float up = fixed ? ldexpf(1.f, 46 - ex) : 0.f;                   // contributions * up < 2^46 (bilinear weights <= 1)
    const double down = fixed ? 1.0 / (double)up : 0.0;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            const float h_ = hi[q][t], w_ = wi[q][t], v = tt[q][t];
            if (h_ <= -1.f || w_ <= -1.f || h_ >= H || w_ >= W || v == 0.f) continue;
            const Smp c = sample(h_, w_, H, W);
            const float hh = 1.f - c.lh, hw = 1.f - c.lw;
            const int r0 = c.h0 - wy0, c0 = c.w0 - wx0;
            if (__builtin_expect(fixed && r0 >= 0 && c0 >= 0 && r0 + 1 < WH && c0 + 1 < WW, 1)) {
                // four unconditional LDS adds: a corner outside the image lands in a window cell that is never flushed
                long long *ga = gacc + r0 * WW + c0;
                const float s_ = v * up;
#ifdef RDFC_EXP_NOSCATTER                                             // (timing builds only: tools/build_variant.sh)
                if (s_ == 12345.678f)
#endif
                {
                    atomicAdd(reinterpret_cast<unsigned long long *>(ga), (unsigned long long)(long long)(hh * hw * s_));
                    atomicAdd(reinterpret_cast<unsigned long long *>(ga + 1), (unsigned long long)(long long)(hh * c.lw * s_));
                    atomicAdd(reinterpret_cast<unsigned long long *>(ga + WW), (unsigned long long)(long long)(c.lh * hw * s_));
                    atomicAdd(reinterpret_cast<unsigned long long *>(ga + WW + 1), (unsigned long long)(long long)(c.lh * c.lw * s_));
                }
            } else {
                if (c.v00) atomicAdd(gi + (size_t)c.h0 * W + c.w0, hh * hw * v);
                if (c.v01) atomicAdd(gi + (size_t)c.h0 * W + c.w0 + 1, hh * c.lw * v);
                if (c.v10) atomicAdd(gi + (size_t)(c.h0 + 1) * W + c.w0, c.lh * hw * v);
                if (c.v11) atomicAdd(gi + (size_t)(c.h0 + 1) * W + c.w0 + 1, c.lh * c.lw * v);
            }
        }
    }
    __syncthreads();
#ifdef RDFC_EXP_NOFLUSH
    if (down == 12345.678) 
#endif
    for (int e = threadIdx.x; e < WN; e += NT) {
        const long long a = gacc[e];
        if (a != 0) {
            const int r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) atomicAdd(gi + (size_t)iy * W + ix, (float)((double)a * down));
        }
    }
}

// ---- gradients of offsets / affinities, summed over the T steps --------------------------------------------------------
// feats[i] = input map of step i + 1 (i = 0 .. T-1), gouts[i] = cotangent of that step's output; both [T][B][H][W]
// (feats may have a larger stride between steps: the buffer also holds the final map).
// One pixel per thread, THREE TAPS per workgroup (blockIdx.y = tap group): the sampling geometry of a tap is the same for every
// step, so a thread keeps it (position, bilinear weights, window address: ~10 values per tap) beside the tap's three sums; with
// all nine taps in one thread that was 256 registers and spills at two waves per SIMD (606 us in the step for a pass over
// 250 MB).  The T saved maps stream through a double-buffered LDS window: the next step's window is fetched into registers before
// the current step's arithmetic and deposited after it, one barrier per step.
constexpr int CNT = TH * TW, TG = 3, NGRP = NTAP / TG;
constexpr int NS = (WN + CNT - 1) / CNT;              // window cells staged per thread (3)
__global__ __launch_bounds__(CNT) void prop_bwd_coeff_kernel(Geo g, int T, const float *__restrict__ feats, size_t feat_stride,
                                                            const float *__restrict__ gouts, size_t gout_stride,
                                                            const float *__restrict__ off, const float *__restrict__ aff,
                                                            const float *__restrict__ wt, float *__restrict__ goff,
                                                            float *__restrict__ gaff) {
    __shared__ float win[2][WN];
    int b, y0, x0;
    tile_of(g, b, y0, x0);
    const int H = g.H, W = g.W, t0 = (int)blockIdx.y * TG;
    const size_t HW = (size_t)H * W;
    const int wy0 = y0 - 1 - R, wx0 = x0 - 1 - R;
    const int pix = threadIdx.x, h = y0 + pix / TW, w = x0 + pix % TW;
    const bool live = h < H && w < W;
    const size_t pp = live ? (size_t)h * W + w : 0;
    float hi[TG], wi[TG], cw[TG], am[TG], s_h[TG], s_w[TG], s_m[TG];
    {
        const float *o = off + (size_t)b * 2 * NTAP * HW + pp, *m = aff + (size_t)b * NTAP * HW + pp;
#pragma unroll
        for (int k = 0; k < TG; ++k) {
            const int t = t0 + k;
            hi[k] = h - 1 + t / 3 + (live ? o[(size_t)(2 * t) * HW] : 0.f);
            wi[k] = w - 1 + t % 3 + (live ? o[(size_t)(2 * t + 1) * HW] : 0.f);
            am[k] = live ? m[(size_t)t * HW] : 0.f;
            cw[k] = wt[t];
            s_h[k] = s_w[k] = s_m[k] = 0.f;
        }
    }
    // this thread's share of the window: image offsets of its cells (-1: outside the image -> 0), the same for every step
    int cell[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int e = threadIdx.x + k * CNT, r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
        cell[k] = (e < WN && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? iy * W + ix : -1;
    }
    float stage[NS];
    auto fetch = [&](const float *im) {
#pragma unroll
        for (int k = 0; k < NS; ++k) stage[k] = cell[k] >= 0 ? im[cell[k]] : 0.f;
    };
    auto deposit = [&](float *wn) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
            if (threadIdx.x + k * CNT < WN) wn[threadIdx.x + k * CNT] = stage[k];
    };
    fetch(feats + (size_t)b * HW);
    deposit(win[0]);
    for (int i = 0; i < T; ++i) {
        const float *im = feats + (size_t)i * feat_stride + (size_t)b * HW;
        __syncthreads();                                  // window i is complete; every thread is done with window i - 1
        if (i + 1 < T) fetch(im + feat_stride);           // (in flight during the arithmetic below)
        if (live) {
            const float go = gouts[(size_t)i * gout_stride + (size_t)b * HW + pp];
#pragma unroll
            for (int k = 0; k < TG; ++k) {
                const float h_ = hi[k], w_ = wi[k];
                if (h_ <= -1.f || w_ <= -1.f || h_ >= H || w_ >= W) continue;
                const Smp c = sample(h_, w_, H, W);
                float a, bb, d, e;
                corners(c, win[i & 1], wy0, wx0, im, W, a, bb, d, e);
                const float hh = 1.f - c.lh, hw = 1.f - c.lw, cg = cw[k] * go;
                s_m[k] += cg * (hh * hw * a + hh * c.lw * bb + c.lh * hw * d + c.lh * c.lw * e);
                s_h[k] += cg * am[k] * (-hw * a - c.lw * bb + hw * d + c.lw * e);
                s_w[k] += cg * am[k] * (-hh * a + hh * bb - c.lh * d + c.lh * e);
            }
        }
        if (i + 1 < T) deposit(win[(i + 1) & 1]);         // last read in step i - 1: every thread is past this step's barrier
    }
    if (!live) return;
    float *go_ = goff + (size_t)b * 2 * NTAP * HW + pp, *gm_ = gaff + (size_t)b * NTAP * HW + pp;
#pragma unroll
    for (int k = 0; k < TG; ++k) {
        const int t = t0 + k;
        go_[(size_t)(2 * t) * HW] = s_h[k];
        go_[(size_t)(2 * t + 1) * HW] = s_w[k];
        gm_[(size_t)t * HW] = s_m[k];
    }
}

// ---- offsets / affinities from the 24-channel guidance conv output (nlspn_model.py:68-138) ------------------------------
struct AffMode {
    int use_tanh;      // 'TC', 'TGASS': aff = tanh(aff) / scale'
    int norm;          // 'AS', 'ASS', 'TGASS': aff /= (sum |aff| + 1e-4)
    int clamp;         // 'ASS', 'TGASS': the divisor is 1 where it is below 1
    float scale_eps;   // 'TGASS': scale' = scale + 1e-8; 'TC': scale
};

// oa: pixel-major [B][H][W][24] (pixel stride ld): channels 2k, 2k+1 = (dh, dw) of neighbour k (cat(o1, o2).view(B, 8, 2, H, W), :76),
// 16 + k = raw affinity of neighbour k.  Tap of neighbour k: k < 4 ? k : k + 1 (the zero reference offset sits at tap 4).
template <typename T>
__global__ __launch_bounds__(NT) void affinity_fwd_kernel(Geo g, AffMode md, const T *__restrict__ oa, int ld, const float *__restrict__ conf,
                                                          const float *__restrict__ scale, const float *__restrict__ wconf,
                                                          const float *__restrict__ bconf, float *__restrict__ off,
                                                          float *__restrict__ aff, float *__restrict__ conf_s) {
    const int H = g.H, W = g.W;
    const size_t HW = (size_t)H * W, total = (size_t)g.B * HW;
    const float inv_scale = md.use_tanh ? 1.f / (scale[0] + md.scale_eps) : 1.f;
    const float wc = wconf ? wconf[0] : 1.f, bc = bconf ? bconf[0] : 0.f;
    for (size_t idx = (size_t)blockIdx.x * NT + threadIdx.x; idx < total; idx += (size_t)gridDim.x * NT) {
        const int b = (int)(idx / HW);
        const size_t p = idx - (size_t)b * HW;
        const int h = (int)(p / W), w = (int)(p - (size_t)h * W);
        const T *src = oa + idx * ld;
        float v[3 * NB];
#pragma unroll
        for (int c = 0; c < 3 * NB; ++c) v[c] = Elem<T>::ld(src + c);
        float *o = off + (size_t)b * 2 * NTAP * HW + p, *m = aff + (size_t)b * NTAP * HW + p;
        const float *cf = conf ? conf + (size_t)b * HW : nullptr;
        float a[NB], s = 0.f, sum = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int t = k < REF ? k : k + 1;
            const float oh = v[2 * k], ow = v[2 * k + 1];
            o[(size_t)(2 * t) * HW] = oh;
            o[(size_t)(2 * t + 1) * HW] = ow;
            float x = v[2 * NB + k];
            if (md.use_tanh) x = tanhf(x) * inv_scale;
            if (cf) {                                   // 1x1 deformable gather of the confidence: pad 0, mask 1, weight w_conf, bias b
                const float hi = h + oh, wi = w + ow;
                float cs = bc;
                if (inside(hi, wi, H, W)) {
                    const Smp c = sample(hi, wi, H, W);
                    const float aa = c.v00 ? cf[(size_t)c.h0 * W + c.w0] : 0.f, bb = c.v01 ? cf[(size_t)c.h0 * W + c.w0 + 1] : 0.f;
                    const float dd = c.v10 ? cf[(size_t)(c.h0 + 1) * W + c.w0] : 0.f, ee = c.v11 ? cf[(size_t)(c.h0 + 1) * W + c.w0 + 1] : 0.f;
                    cs += wc * bilin(c, aa, bb, dd, ee);
                }
                conf_s[((size_t)b * NB + k) * HW + p] = cs;
                x *= cs;
            }
            a[k] = x;
            s += fabsf(x);
        }
        o[(size_t)(2 * REF) * HW] = 0.f;
        o[(size_t)(2 * REF + 1) * HW] = 0.f;
        s += 1e-4f;
        if (md.clamp && s < 1.f) s = 1.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (md.norm) a[k] = a[k] / s;
            sum += a[k];
            m[(size_t)(k < REF ? k : k + 1) * HW] = a[k];
        }
        m[(size_t)REF * HW] = 1.f - sum;
    }
}

// the 24 channels of one pixel <-> floats: three (16-bit) or six (f32) 16-byte accesses when the rows are 16-byte aligned
template <typename T> __device__ __forceinline__ void load24(const T *__restrict__ src, bool vec, float (&v)[3 * NB]) {
    constexpr int VEC = Elem<T>::VEC;
    if (vec) {
#pragma unroll
        for (int q = 0; q < 3 * NB / VEC; ++q) {
            Vec16<T> t;
            t.load(src + q * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[q * VEC + e] = t.get(e);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3 * NB; ++c) v[c] = Elem<T>::ld(src + c);
    }
}
template <typename T> __device__ __forceinline__ void store24(T *__restrict__ dst, bool vec, const float (&v)[3 * NB]) {
    constexpr int VEC = Elem<T>::VEC;
    if (vec) {
#pragma unroll
        for (int q = 0; q < 3 * NB / VEC; ++q) {
            Vec16<T> t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t.set(e, v[q * VEC + e]);
            t.store(dst + q * VEC);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3 * NB; ++c) Elem<T>::st(dst + c, v[c]);
    }
}

// backward: g_oa (pixel-major, dtype of the guidance conv's output gradient), g_conf (scatter, zero-filled), g_scale (one
// atomic per workgroup into a zero-filled f32).  Pixel tiles like the propagation kernels; the confidence cotangents (eight
// bilinear scatters per pixel) go through the 64-bit fixed-point LDS window of prop_bwd_feat_kernel and leave the workgroup as
// one global atomic per touched cell (they were 32 global float atomics per pixel: 365 us in the step); the pixel's 24 input and
// 24 output channels move as 16-byte vectors.
template <typename T>
__global__ __launch_bounds__(NT) void affinity_bwd_kernel(Geo g, AffMode md, const T *__restrict__ oa, int ld, const float *__restrict__ conf_s,
                                                          const float *__restrict__ scale, const float *__restrict__ wconf,
                                                          const float *__restrict__ goff,
                                                          const float *__restrict__ gaff, T *__restrict__ goa, int ldg,
                                                          float *__restrict__ gconf, float *__restrict__ gscale, int vec_in, int vec_out) {
    __shared__ long long gacc[WN];
    __shared__ float wmax[NT / 64], red[NT / 64];
    int b, y0, x0;
    tile_of(g, b, y0, x0);
    const int H = g.H, W = g.W;
    const size_t HW = (size_t)H * W;
    const int wy0 = y0 - 1 - R, wx0 = x0 - 1 - R;
    const float inv_scale = md.use_tanh ? 1.f / (scale[0] + md.scale_eps) : 1.f;
    const float wc = wconf ? wconf[0] : 1.f;
    const bool scatter = gconf != nullptr && conf_s != nullptr;
    if (scatter)
        for (int e = threadIdx.x; e < WN; e += NT) gacc[e] = 0;
    float gc[PPT][NB], sy[PPT][NB], sx[PPT][NB];          // confidence cotangents of this thread's pixels and where they land
    float gs_local = 0.f, big = 0.f;
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pix = threadIdx.x + q * NT, h = y0 + pix / TW, w = x0 + pix % TW;
#pragma unroll
        for (int k = 0; k < NB; ++k) gc[q][k] = sy[q][k] = sx[q][k] = 0.f;
        if (h >= H || w >= W) continue;
        const size_t p = (size_t)h * W + w, idx = (size_t)b * HW + p;
        float v[3 * NB];
        load24(oa + idx * ld, vec_in != 0, v);
        const float *go = goff ? goff + (size_t)b * 2 * NTAP * HW + p : nullptr, *gm = gaff + (size_t)b * NTAP * HW + p;
        // recompute the forward up to the normalisation
        float th[NB], cs[NB], bk[NB], s = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const float x = v[2 * NB + k];
            th[k] = md.use_tanh ? tanhf(x) : x;
            cs[k] = conf_s ? conf_s[((size_t)b * NB + k) * HW + p] : 1.f;
            bk[k] = (md.use_tanh ? th[k] * inv_scale : th[k]) * cs[k];
            s += fabsf(bk[k]);
        }
        s += 1e-4f;
        const bool clamped = md.clamp && s < 1.f;
        if (clamped) s = 1.f;
        const float g_ref = gm[(size_t)REF * HW];
        float ga[NB], dot = 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            ga[k] = gm[(size_t)(k < REF ? k : k + 1) * HW] - g_ref;          // ref = 1 - sum a_k
            dot += ga[k] * bk[k];
        }
        float out[3 * NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            float gb = ga[k];
            if (md.norm) {
                gb = ga[k] / s;
                if (!clamped) gb -= (bk[k] > 0.f ? 1.f : (bk[k] < 0.f ? -1.f : 0.f)) * dot / (s * s);
            }
            const float t_scaled = md.use_tanh ? th[k] * inv_scale : th[k];
            const float g_c = gb * t_scaled * wc, g_t = gb * cs[k];            // b_k = t_k * c_k, c_k = b + w_conf * sample
            float gx = g_t;
            if (md.use_tanh) {
                gx = g_t * (1.f - th[k] * th[k]) * inv_scale;
                gs_local += -g_t * th[k] * inv_scale * inv_scale;             // d/dscale of tanh / (scale + eps)
            }
            out[2 * NB + k] = gx;
            const int t = k < REF ? k : k + 1;
            out[2 * k] = go ? go[(size_t)(2 * t) * HW] : 0.f;                 // (the confidence gathers use detached offsets)
            out[2 * k + 1] = go ? go[(size_t)(2 * t + 1) * HW] : 0.f;
            if (scatter) {
                gc[q][k] = g_c;
                sy[q][k] = h + v[2 * k];
                sx[q][k] = w + v[2 * k + 1];
                big = fabsf(g_c) <= 3.0e38f ? fmaxf(big, fabsf(g_c)) : __builtin_inff();      // NaN / inf: flagged, not dropped
            }
        }
        store24(goa + idx * ldg, vec_out != 0, out);
    }
    if (scatter) {
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) big = fmaxf(big, __shfl_xor(big, o_, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = big;
        __syncthreads();                                    // (also: the window is zeroed)
        big = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (big != 0.f) {                                   // uniform over the workgroup
            int ex = 0;
            if (big <= 3.0e38f) frexpf(big, &ex);
            const bool fixed = big <= 3.0e38f && 46 - ex <= 126;            // see prop_bwd_feat_kernel
            const float up = fixed ? ldexpf(1.f, 46 - ex) : 0.f;
            const double down = fixed ? 1.0 / (double)up : 0.0;
            float *gi = gconf + (size_t)b * HW;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const float h_ = sy[q][k], w_ = sx[q][k], vv = gc[q][k];
                    if (h_ <= -1.f || w_ <= -1.f || h_ >= H || w_ >= W || vv == 0.f) continue;
                    const Smp c = sample(h_, w_, H, W);
                    const float hh = 1.f - c.lh, hw = 1.f - c.lw;
                    const int r0 = c.h0 - wy0, c0 = c.w0 - wx0;
                    if (__builtin_expect(fixed && r0 >= 0 && c0 >= 0 && r0 + 1 < WH && c0 + 1 < WW, 1)) {
                        long long *ga_ = gacc + r0 * WW + c0;
                        const float s_ = vv * up;
                        atomicAdd(reinterpret_cast<unsigned long long *>(ga_), (unsigned long long)(long long)(hh * hw * s_));
                        atomicAdd(reinterpret_cast<unsigned long long *>(ga_ + 1), (unsigned long long)(long long)(hh * c.lw * s_));
                        atomicAdd(reinterpret_cast<unsigned long long *>(ga_ + WW), (unsigned long long)(long long)(c.lh * hw * s_));
                        atomicAdd(reinterpret_cast<unsigned long long *>(ga_ + WW + 1), (unsigned long long)(long long)(c.lh * c.lw * s_));
                    } else {
                        if (c.v00) atomicAdd(gi + (size_t)c.h0 * W + c.w0, hh * hw * vv);
                        if (c.v01) atomicAdd(gi + (size_t)c.h0 * W + c.w0 + 1, hh * c.lw * vv);
                        if (c.v10) atomicAdd(gi + (size_t)(c.h0 + 1) * W + c.w0, c.lh * hw * vv);
                        if (c.v11) atomicAdd(gi + (size_t)(c.h0 + 1) * W + c.w0 + 1, c.lh * c.lw * vv);
                    }
                }
            }
            __syncthreads();
            for (int e = threadIdx.x; e < WN; e += NT) {
                const long long a = gacc[e];
                if (a != 0) {
                    const int r = e / WW, c = e - r * WW, iy = wy0 + r, ix = wx0 + c;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) atomicAdd(gi + (size_t)iy * W + ix, (float)((double)a * down));
                }
            }
        }
    }
    if (gscale != nullptr && md.use_tanh) {
        gs_local = wave_sum(gs_local);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs_local;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(gscale, red[0] + red[1] + red[2] + red[3]);
    }
}

int check_geo(const char *who, int B, int H, int W) {
    RDFC_REQUIRE(B > 0 && H > 0 && W > 0, "%s: non-positive size", who);
    RDFC_REQUIRE((double)B * H * W * 24 < 2147483647.0, "%s: map too large for 32-bit pixel indices", who);
    return 0;
}
Geo make_geo(int B, int H, int W) { return Geo{B, H, W, cdiv(W, TW), cdiv(H, TH)}; }
int mode_of(const char *who, int affinity, AffMode &md) {
    switch (affinity) {
        case 0: md = AffMode{0, 1, 0, 0.f}; return 0;        // 'AS'
        case 1: md = AffMode{0, 1, 1, 0.f}; return 0;        // 'ASS'
        case 2: md = AffMode{1, 0, 0, 0.f}; return 0;        // 'TC'
        case 3: md = AffMode{1, 1, 1, 1e-8f}; return 0;      // 'TGASS'
    }
    rdfc_set_error("%s: affinity mode %d (0 AS, 1 ASS, 2 TC, 3 TGASS)", who, affinity);
    return 1;
}

}  // namespace

#define DISPATCH(dtype, ...) RDFC_DISPATCH(dtype, __VA_ARGS__)

extern "C" int rdfc_nlspn_affinity_forward(const void *offset_aff, int ld, int dtype, const float *confidence, const float *scale,
                                           const float *w_conf, const float *b_conf, int affinity, int B, int H, int W, float *offset,
                                           float *aff, float *conf_samples, void *stream) {
    if (int e = check_geo("rdfc_nlspn_affinity_forward", B, H, W)) return e;
    AffMode md;
    if (int e = mode_of("rdfc_nlspn_affinity_forward", affinity, md)) return e;
    RDFC_REQUIRE(rdfc_dtype_ok(dtype), "rdfc_nlspn_affinity_forward: bad dtype");
    RDFC_REQUIRE(ld >= 24 && offset_aff && offset && aff, "rdfc_nlspn_affinity_forward: null pointer / ld < 24");
    RDFC_REQUIRE(!md.use_tanh || scale != nullptr, "rdfc_nlspn_affinity_forward: TC / TGASS need aff_scale_const");
    RDFC_REQUIRE(confidence == nullptr || conf_samples != nullptr, "rdfc_nlspn_affinity_forward: conf_prop needs the sample buffer");
    const Geo g = make_geo(B, H, W);
    const size_t total = (size_t)B * H * W;
    const int nb = (int)min((size_t)8192, (total + NT - 1) / NT);
    DISPATCH(dtype, hipLaunchKernelGGL(affinity_fwd_kernel<T>, dim3(nb), dim3(NT), 0, (hipStream_t)stream, g, md, (const T *)offset_aff, ld,
                                       confidence, scale, w_conf, b_conf, offset, aff, conf_samples));
    RDFC_LAUNCH_CHECK("rdfc_nlspn_affinity_forward");
    return 0;
}

extern "C" int rdfc_nlspn_affinity_backward(const void *offset_aff, int ld, int dtype, const float *conf_samples, const float *scale,
                                            const float *w_conf, int affinity, int B, int H, int W, const float *grad_offset, const float *grad_aff,
                                            void *grad_offset_aff, int ldg, float *grad_confidence, float *grad_scale, void *stream) {
    if (int e = check_geo("rdfc_nlspn_affinity_backward", B, H, W)) return e;
    AffMode md;
    if (int e = mode_of("rdfc_nlspn_affinity_backward", affinity, md)) return e;
    RDFC_REQUIRE(rdfc_dtype_ok(dtype), "rdfc_nlspn_affinity_backward: bad dtype");
    RDFC_REQUIRE(ld >= 24 && ldg >= 24 && offset_aff && grad_aff && grad_offset_aff, "rdfc_nlspn_affinity_backward: null pointer / ld < 24");
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * H * W;
    if (grad_confidence && hipMemsetAsync(grad_confidence, 0, total * sizeof(float), st) != hipSuccess) {
        rdfc_set_error("rdfc_nlspn_affinity_backward: memset failed");
        return 2;
    }
    if (grad_scale && hipMemsetAsync(grad_scale, 0, sizeof(float), st) != hipSuccess) {
        rdfc_set_error("rdfc_nlspn_affinity_backward: memset failed");
        return 2;
    }
    const Geo g = make_geo(B, H, W);
    const size_t esz = dtype == RDFC_F32 ? 4 : 2;
    const int vec_in = (ld * esz) % 16 == 0 && (size_t)offset_aff % 16 == 0, vec_out = (ldg * esz) % 16 == 0 && (size_t)grad_offset_aff % 16 == 0;
    DISPATCH(dtype, hipLaunchKernelGGL(affinity_bwd_kernel<T>, dim3(g.tiles_x * g.tiles_y * B), dim3(NT), 0, st, g, md, (const T *)offset_aff, ld,
                                       conf_samples, scale, w_conf, grad_offset, grad_aff, (T *)grad_offset_aff, ldg, grad_confidence, grad_scale,
                                       vec_in, vec_out));
    RDFC_LAUNCH_CHECK("rdfc_nlspn_affinity_backward");
    return 0;
}

extern "C" int rdfc_nlspn_propagate_forward(const float *feat0, const float *offset, const float *aff, const float *weight,
                                            const float *bias, int B, int H, int W, int T, float *feats, void *stream) {
    if (int e = check_geo("rdfc_nlspn_propagate_forward", B, H, W)) return e;
    RDFC_REQUIRE(T >= 1 && feat0 && offset && aff && weight && feats, "rdfc_nlspn_propagate_forward: null pointer / T < 1");
    const Geo g = make_geo(B, H, W);
    const size_t n = (size_t)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(feats, feat0, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        rdfc_set_error("rdfc_nlspn_propagate_forward: copy of the initial map failed");
        return 2;
    }
    for (int i = 0; i < T; ++i)
        hipLaunchKernelGGL(prop_fwd_kernel, dim3(g.tiles_x * g.tiles_y * B), dim3(NT), 0, st, g, feats + (size_t)i * n, offset, aff, weight,
                           bias, feats + (size_t)(i + 1) * n);
    RDFC_LAUNCH_CHECK("rdfc_nlspn_propagate_forward");
    return 0;
}

extern "C" int rdfc_nlspn_propagate_backward(const float *feats, const float *offset, const float *aff, const float *weight, int B, int H,
                                             int W, int T, float *grads, float *grad_offset, float *grad_aff, void *stream) {
    if (int e = check_geo("rdfc_nlspn_propagate_backward", B, H, W)) return e;
    RDFC_REQUIRE(T >= 1 && feats && offset && aff && weight && grads, "rdfc_nlspn_propagate_backward: null pointer / T < 1");
    const Geo g = make_geo(B, H, W);
    const size_t n = (size_t)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    // grads[T] holds the cotangent of the final map; steps T .. 1 fill grads[T-1] .. grads[0]
    if (hipMemsetAsync(grads, 0, (size_t)T * n * sizeof(float), st) != hipSuccess) {
        rdfc_set_error("rdfc_nlspn_propagate_backward: memset failed");
        return 2;
    }
    const dim3 grid(g.tiles_x * g.tiles_y * B);
    for (int i = T; i >= 1; --i)
        hipLaunchKernelGGL(prop_bwd_feat_kernel, grid, dim3(NT), 0, st, g, offset, aff, weight, grads + (size_t)i * n, grads + (size_t)(i - 1) * n);
    if (grad_offset != nullptr || grad_aff != nullptr) {
        RDFC_REQUIRE(grad_offset != nullptr && grad_aff != nullptr, "rdfc_nlspn_propagate_backward: offset and affinity gradients come together");
        hipLaunchKernelGGL(prop_bwd_coeff_kernel, dim3(grid.x, NGRP), dim3(CNT), 0, st, g, T, feats, n, grads + n, n, offset, aff, weight, grad_offset, grad_aff);
    }
    RDFC_LAUNCH_CHECK("rdfc_nlspn_propagate_backward");
    return 0;
}
