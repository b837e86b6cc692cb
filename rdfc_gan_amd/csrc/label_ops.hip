// label_ops.hip -- the non-convolution operators of the semantic-label net (lib/models/generator/label_net/
// labels_model.py:12-94, backbone_resnet50.py:150-235) and its loss (rdfc_gan.py:59,238-239) on pixel-major maps
// [N][H][W][C] (leading dimension ld):
//   MaxPool2d(3, 2, 1), AdaptiveAvgPool2d(b), F.interpolate(bilinear, align_corners=True), Dropout2d's per-(n, c)
//   scale, CrossEntropyLoss (mean over the pixels whose label is not ignore_index).
// All HBM-bound or tiny; every backward is a gather (no atomics): results do not depend on the launch order.
#include "common.h"

namespace {

constexpr int NT = 256;
inline int nblk(size_t total) { return (int)min((size_t)65535 * 16, (total + NT - 1) / NT); }
#define GRID_STRIDE(i, total) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

// ---- MaxPool2d ------------------------------------------------------------------------------------------------------
// first maximum in window scan order wins (strict >, NaN propagates) like at::max_pool2d; idx = kh * k + kw of the winner
template <typename T>
__global__ void maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, uint8_t *__restrict__ idx, int N, int Hi, int Wi,
                                   int C, int ldx, int ldy, int k, int s, int p, int Ho, int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * Ho * Wo * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float best[VEC];
        int bi[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) { best[v] = -INFINITY; bi[v] = 255; }
        for (int kh = 0; kh < k; ++kh) {
            const int h = oh * s - p + kh;
            if ((unsigned)h >= (unsigned)Hi) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int w = ow * s - p + kw;
                if ((unsigned)w >= (unsigned)Wi) continue;
                Vec16<T> xv;
                xv.load(x + ((size_t)(n * Hi + h) * Wi + w) * ldx + cv * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float val = xv.get(v);
                    if (bi[v] == 255 || val > best[v] || val != val) { best[v] = val; bi[v] = kh * k + kw; }
                }
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, best[v]);
        o.store(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + cv * VEC);
        uint8_t *ip = idx + ((size_t)(n * Ho + oh) * Wo + ow) * C + cv * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) ip[v] = (uint8_t)bi[v];
    }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T *__restrict__ dy, const uint8_t *__restrict__ idx, T *__restrict__ dx, int N, int Hi,
                                   int Wi, int C, int lddy, int lddx, int k, int s, int p, int Ho, int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * Hi * Wi * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int w = (int)(r % Wi); r /= Wi;
        const int h = (int)(r % Hi);
        const int n = (int)(r / Hi);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        // windows that contain (h, w): oh * s - p <= h <= oh * s - p + k - 1
        int oh0 = h + p - k + 1;
        oh0 = oh0 <= 0 ? 0 : (oh0 + s - 1) / s;
        int ow0 = w + p - k + 1;
        ow0 = ow0 <= 0 ? 0 : (ow0 + s - 1) / s;
        const int oh1 = min((h + p) / s, Ho - 1), ow1 = min((w + p) / s, Wo - 1);
        for (int oh = oh0; oh <= oh1; ++oh)
            for (int ow = ow0; ow <= ow1; ++ow) {
                const int off = (h - (oh * s - p)) * k + (w - (ow * s - p));
                const size_t o = (size_t)(n * Ho + oh) * Wo + ow;
                Vec16<T> dv;
                dv.load(dy + o * lddy + cv * VEC);
                const uint8_t *ip = idx + o * C + cv * VEC;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (ip[v] == off) acc[v] += dv.get(v);
            }
        Vec16<T> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, acc[v]);
        o.store(dx + ((size_t)(n * Hi + h) * Wi + w) * lddx + cv * VEC);
    }
}

// ---- AdaptiveAvgPool2d ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bin_start(int o, int O, int I) { return (int)floorf((float)(o * I) / O); }
__device__ __forceinline__ int bin_end(int o, int O, int I) { return (int)ceilf((float)((o + 1) * I) / O); }

template <typename T>
__global__ void avgpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho,
                                   int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * Ho * Wo * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int h0 = bin_start(oh, Ho, Hi), h1 = bin_end(oh, Ho, Hi), w0 = bin_start(ow, Wo, Wi), w1 = bin_end(ow, Wo, Wi);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) {
                Vec16<T> xv;
                xv.load(x + ((size_t)(n * Hi + h) * Wi + w) * ldx + cv * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += xv.get(v);
            }
        const float area = (float)((h1 - h0) * (w1 - w0));
        Vec16<T> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, acc[v] / area);
        o.store(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + cv * VEC);
    }
}

// large bins (the 1x1 .. 3x3 pyramid levels over a 29x38 map): one WAVE per (n, bin, channel vector), lanes stride the bin's
// pixels and a butterfly adds the 64 partial sums -- the thread-per-output version walks up to 1102 pixels serially with 2 048
// threads on the whole chip (169 us per call inside the label net, this one ~15 us)
template <typename T>
__global__ __launch_bounds__(NT) void avgpool_fwd_wave_kernel(const T *__restrict__ x, T *__restrict__ y, int N, int Hi, int Wi, int C, int ldx,
                                                              int ldy, int Ho, int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * Ho * Wo * CV;
    const int lane = threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6); i < total; i += (size_t)gridDim.x * (NT / 64)) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int h0 = bin_start(oh, Ho, Hi), h1 = bin_end(oh, Ho, Hi), w0 = bin_start(ow, Wo, Wi), w1 = bin_end(ow, Wo, Wi);
        const int bw = w1 - w0, area = (h1 - h0) * bw;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int q = lane; q < area; q += 64) {
            const int h = h0 + q / bw, w = w0 + q % bw;
            Vec16<T> xv;
            xv.load(x + ((size_t)(n * Hi + h) * Wi + w) * ldx + cv * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += xv.get(v);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            for (int off = 32; off > 0; off >>= 1) acc[v] += __shfl_xor(acc[v], off, 64);
        if (lane == 0) {
            Vec16<T> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.set(v, acc[v] / (float)area);
            o.store(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + cv * VEC);
        }
    }
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T *__restrict__ dy, T *__restrict__ dx, int N, int Hi, int Wi, int C, int lddy, int lddx,
                                   int Ho, int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * Hi * Wi * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int w = (int)(r % Wi); r /= Wi;
        const int h = (int)(r % Hi);
        const int n = (int)(r / Hi);
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int oh = 0; oh < Ho; ++oh) {
            const int h0 = bin_start(oh, Ho, Hi), h1 = bin_end(oh, Ho, Hi);
            if (h < h0 || h >= h1) continue;
            for (int ow = 0; ow < Wo; ++ow) {
                const int w0 = bin_start(ow, Wo, Wi), w1 = bin_end(ow, Wo, Wi);
                if (w < w0 || w >= w1) continue;
                const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
                Vec16<T> dv;
                dv.load(dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + cv * VEC);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += dv.get(v) * inv;
            }
        }
        Vec16<T> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, acc[v]);
        o.store(dx + ((size_t)(n * Hi + h) * Wi + w) * lddx + cv * VEC);
    }
}

// ---- bilinear resize, align_corners = True (at::upsample_bilinear2d) -------------------------------------------------
__device__ __forceinline__ float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_of(int o, float scale, int in) {
    const float src = scale * (float)o;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <typename TI, typename TO>
__global__ void bilinear_fwd_kernel(const TI *__restrict__ x, TO *__restrict__ y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho,
                                    int Wo) {
    const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
    const size_t total = (size_t)N * Ho * Wo * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const Lerp a = lerp_of(oh, sh, Hi), b = lerp_of(ow, sw, Wi);
        const TI *base = x + (size_t)n * Hi * Wi * ldx + c;
        const float v00 = Elem<TI>::ld(base + ((size_t)a.i0 * Wi + b.i0) * ldx), v01 = Elem<TI>::ld(base + ((size_t)a.i0 * Wi + b.i1) * ldx);
        const float v10 = Elem<TI>::ld(base + ((size_t)a.i1 * Wi + b.i0) * ldx), v11 = Elem<TI>::ld(base + ((size_t)a.i1 * Wi + b.i1) * ldx);
        const float o = a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11);
        Elem<TO>::st(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + c, o);
    }
}

// adjoint as a gather: input pixel (h, w) collects every output pixel whose two taps along each axis include it
__device__ __forceinline__ void out_range(int i, float scale, int in, int out, int &lo, int &hi) {
    if (scale == 0.f) { lo = 0; hi = out - 1; return; }
    lo = (int)floorf((float)(i - 1) / scale) - 1;
    hi = (int)ceilf((float)(i + 1) / scale) + 1;
    lo = max(lo, 0);
    hi = min(hi, out - 1);
}
__device__ __forceinline__ float tap_weight(int o, int i, float scale, int in) {
    const Lerp a = lerp_of(o, scale, in);
    return (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
}

template <typename TI, typename TO>   // TI: dtype of dx (the forward's input), TO: dtype of dy
__global__ void bilinear_bwd_kernel(const TO *__restrict__ dy, TI *__restrict__ dx, int N, int Hi, int Wi, int C, int lddy, int lddx,
                                    int Ho, int Wo) {
    const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
    const size_t total = (size_t)N * Hi * Wi * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int w = (int)(r % Wi); r /= Wi;
        const int h = (int)(r % Hi);
        const int n = (int)(r / Hi);
        int oh0, oh1, ow0, ow1;
        out_range(h, sh, Hi, Ho, oh0, oh1);
        out_range(w, sw, Wi, Wo, ow0, ow1);
        float acc = 0.f;
        for (int oh = oh0; oh <= oh1; ++oh) {
            const float wh = tap_weight(oh, h, sh, Hi);
            if (wh == 0.f) continue;
            float row = 0.f;
            for (int ow = ow0; ow <= ow1; ++ow) {
                const float ww = tap_weight(ow, w, sw, Wi);
                if (ww != 0.f) row += ww * Elem<TO>::ld(dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + c);
            }
            acc += wh * row;
        }
        Elem<TI>::st(dx + ((size_t)(n * Hi + h) * Wi + w) * lddx + c, acc);
    }
}

// small sources (the pyramid levels: 1x1 .. 6x6 -> 29x38): one WAVE per (n, source pixel, channel vector); lanes stride the window
// of output pixels that can touch the source pixel, butterfly at the end (thread-per-source-element: 282 us per call, 4 096 threads
// walking up to 1102 outputs each)
template <typename T>
__global__ __launch_bounds__(NT) void bilinear_bwd_wave_kernel(const T *__restrict__ dy, T *__restrict__ dx, int N, int Hi, int Wi, int C, int lddy,
                                                               int lddx, int Ho, int Wo) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const float sh = ac_scale(Hi, Ho), sw = ac_scale(Wi, Wo);
    const size_t total = (size_t)N * Hi * Wi * CV;
    const int lane = threadIdx.x & 63;
    for (size_t i = (size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6); i < total; i += (size_t)gridDim.x * (NT / 64)) {
        const int cv = (int)(i % CV);
        size_t r = i / CV;
        const int w = (int)(r % Wi); r /= Wi;
        const int h = (int)(r % Hi);
        const int n = (int)(r / Hi);
        int oh0, oh1, ow0, ow1;
        out_range(h, sh, Hi, Ho, oh0, oh1);
        out_range(w, sw, Wi, Wo, ow0, ow1);
        const int bw = ow1 - ow0 + 1, area = (oh1 - oh0 + 1) * bw;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int q = lane; q < area; q += 64) {
            const int oh = oh0 + q / bw, ow = ow0 + q % bw;
            const float wt = tap_weight(oh, h, sh, Hi) * tap_weight(ow, w, sw, Wi);
            if (wt == 0.f) continue;
            Vec16<T> dv;
            dv.load(dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + cv * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += wt * dv.get(v);
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            for (int off = 32; off > 0; off >>= 1) acc[v] += __shfl_xor(acc[v], off, 64);
        if (lane == 0) {
            Vec16<T> o;
#pragma unroll
            for (int v = 0; v < VEC; ++v) o.set(v, acc[v]);
            o.store(dx + ((size_t)(n * Hi + h) * Wi + w) * lddx + cv * VEC);
        }
    }
}

// ---- Dropout2d: y = x * scale[n][c] -------------------------------------------------------------------------------
template <typename T>
__global__ void scale_nc_kernel(const T *__restrict__ x, const float *__restrict__ scale, T *__restrict__ y, int N, int HW, int C, int ldx,
                                int ldy) {
    constexpr int VEC = Elem<T>::VEC;
    const int CV = C / VEC;
    const size_t total = (size_t)N * HW * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        const size_t px = i / CV;
        const int n = (int)(px / HW);
        Vec16<T> xv, o;
        xv.load(x + px * ldx + cv * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, xv.get(v) * scale[n * C + cv * VEC + v]);
        o.store(y + px * ldy + cv * VEC);
    }
}

// ---- CrossEntropyLoss over pixel-major logits [P][C] ---------------------------------------------------------------
constexpr int CE_MAXC = 4096;
template <typename T>   // log-sum-exp of one pixel's logits; zt = the target's logit (two passes over the row, no per-thread array)
__device__ __forceinline__ float pixel_lse(const T *__restrict__ row, int C, int t, float &zt) {
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, Elem<T>::ld(row + c));
    float s = 0.f;
    zt = 0.f;
    for (int c = 0; c < C; ++c) {
        const float z = Elem<T>::ld(row + c);
        s += expf(z - m);
        if (c == t) zt = z;
    }
    return m + logf(s);
}

template <typename T>
__global__ __launch_bounds__(NT) void ce_fwd_kernel(const T *__restrict__ logits, const long long *__restrict__ labels, int P, int C, int ld,
                                                    long long ignore_index, double *__restrict__ partial) {
    __shared__ double red[2][NT / 64];
    double s = 0.0, cnt = 0.0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < (size_t)P; i += (size_t)gridDim.x * NT) {
        const long long t = labels[i];
        if (t == ignore_index || t < 0 || t >= C) continue;
        float zt;
        const float lse = pixel_lse(logits + i * ld, C, (int)t, zt);
        s += (double)(lse - zt);
        cnt += 1.0;
    }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); cnt += __shfl_down(cnt, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < NT / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        partial[2 * blockIdx.x] = a;
        partial[2 * blockIdx.x + 1] = b;
    }
}

__global__ void ce_finalize_kernel(const double *__restrict__ partial, int nb, float *__restrict__ loss, float *__restrict__ count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nb; ++k) { a += partial[2 * k]; b += partial[2 * k + 1]; }
        *loss = (float)(a / b);          // 0 / 0 = NaN like torch when every pixel is ignored
        *count = (float)b;
    }
}

template <typename T>
__global__ void ce_bwd_kernel(const T *__restrict__ logits, const long long *__restrict__ labels, int P, int C, int ld, long long ignore_index,
                              const float *__restrict__ count, const float *__restrict__ gscale_dev, float gscale, T *__restrict__ dl,
                              int lddl) {
    const float g = gscale * (gscale_dev ? *gscale_dev : 1.f) / *count;
    GRID_STRIDE(i, (size_t)P) {
        const long long t = labels[i];
        T *o = dl + i * lddl;
        if (t == ignore_index || t < 0 || t >= C) {
            for (int c = 0; c < C; ++c) Elem<T>::st(o + c, 0.f);
            continue;
        }
        float zt;
        const float lse = pixel_lse(logits + i * ld, C, (int)t, zt);
        for (int c = 0; c < C; ++c) Elem<T>::st(o + c, (expf(Elem<T>::ld(logits + i * ld + c) - lse) - (c == t ? 1.f : 0.f)) * g);
    }
}


// ---- nor_loss + manhattan_loss (lib/losses/gan_loss.py:122-132, 50-99): values of the four normal-consistency terms ------
// normal / gt_normal: f32 [N][3][HW] (upstream NCHW), norm_mask f32 [N][HW] or NULL, logits f32 pixel-major [N*HW][C].
// cosine_similarity as ATen computes it: (a / max(|a|, 1e-8)) . (b / max(|b|, 1e-8)).
// manhattan_loss multiplies its per-image cosine map [B,H,W] with the unsqueezed class mask [B,1,H,W] (:73-74,83,93):
// the product broadcasts to [B,B,H,W], i.e. image j's term at pixel p is counted once for every image i whose pixel p
// has the class.  Reproduced as is: a thread owns one pixel POSITION p and walks the batch,
//   sum_k += M_k(p) * sum_j t_kj(p),  count_k += M_k(p) * #{j : t_kj(p) != 0},   M_k(p) = #{i : class_i(p) == k},
// with t_kj = 1 - cos_j inside image j's class-k mask and 1 - 0 outside it (floor / ceiling), 2 cos_j inside and 0
// outside (wall).
// sums (f64): 0 nor sum, 1 nor count, 2 floor sum, 3 floor count, 4 wall sum, 5 wall count, 6 ceiling sum, 7 ceiling count
__global__ __launch_bounds__(NT) void normal_losses_kernel(const float *__restrict__ nrm, const float *__restrict__ gtn,
                                                           const float *__restrict__ nmask, const float *__restrict__ logits, int N, int HW,
                                                           int C, int ld, int wall, int floor_, int ceiling, double *__restrict__ partial) {
// the reference multiplies, subtracts and compares in separately rounded steps ((1 - cos) != 0 decides the count)
#pragma clang fp contract(off)
    __shared__ double red[8][NT / 64];
    double s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.0;
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < (size_t)HW; p += (size_t)gridDim.x * NT) {
        int M[3] = {0, 0, 0};                 // images whose pixel p is floor / wall / ceiling
        float T[3] = {0.f, 0.f, 0.f};         // sum over images of the term
        int Z[3] = {0, 0, 0};                 // images whose term is non-zero
        for (int n = 0; n < N; ++n) {
            const size_t i = (size_t)n * HW + p;
            const float *a = nrm + (size_t)n * 3 * HW + p;
            const float ax = a[0], ay = a[HW], az = a[2 * (size_t)HW];
            const float an = fmaxf(sqrtf(ax * ax + ay * ay + az * az), 1e-8f);
            if (gtn != nullptr) {
                const float *b = gtn + (size_t)n * 3 * HW + p;
                const float bx = b[0], by = b[HW], bz = b[2 * (size_t)HW];
                const float bn = fmaxf(sqrtf(bx * bx + by * by + bz * bz), 1e-8f);
                const float dot = (ax / an) * (bx / bn) + (ay / an) * (by / bn) + (az / an) * (bz / bn);
                const float m = nmask ? nmask[i] : 1.f;
                const bool valid = m * (dot < 0.999f ? 1.f : 0.f) * (dot > -0.999f ? 1.f : 0.f) > 0.f;
                const float v = valid ? (dot * -1.f) + 1.f : 0.f;
                s[0] += (double)v;
                s[1] += v != 0.f ? 1.0 : 0.0;
            }
            if (logits != nullptr) {
                const float *z = logits + i * ld;
                int best = 0;
                float bz = z[0];
                for (int c = 1; c < C; ++c) {
                    const float v = z[c];
                    if (v > bz) { bz = v; best = c; }
                }
                const float cy = ay / an;                              // cos with (0, 1, 0); (0, -1, 0): -cy
                const float tf = ((best == floor_ ? cy : 0.f) * -1.f) + 1.f;
                const float tw = (best == wall ? cy : 0.f) * 2.f;
                const float tc = ((best == ceiling ? -cy : 0.f) * -1.f) + 1.f;
                M[0] += best == floor_; M[1] += best == wall; M[2] += best == ceiling;
                T[0] += tf; T[1] += fabsf(tw); T[2] += tc;
                Z[0] += tf != 0.f; Z[1] += tw != 0.f; Z[2] += tc != 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            s[2 + 2 * k] += (double)M[k] * (double)T[k];
            s[3 + 2 * k] += (double)M[k] * (double)Z[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = s[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        double a = 0.0;
        for (int w = 0; w < NT / 64; ++w) a += red[threadIdx.x][w];
        partial[(size_t)blockIdx.x * 8 + threadIdx.x] = a;
    }
}

__global__ void normal_losses_finalize_kernel(const double *__restrict__ partial, int nb, float *__restrict__ out) {
    if (threadIdx.x < 4 && blockIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nb; ++k) { a += partial[(size_t)k * 8 + 2 * threadIdx.x]; b += partial[(size_t)k * 8 + 2 * threadIdx.x + 1]; }
        out[threadIdx.x] = (float)((float)a / ((float)b + 1e-6f));
        out[4 + threadIdx.x] = (float)b;                         // the count, for the backward
    }
}

// gradient with respect to `normal` of  g[0] * nor + g[1] * floor + g[2] * wall + g[3] * ceiling  (the counts and the masks are
// constants of the graph upstream too: dot.detach() in the validity mask, arg-max class masks, `!= 0` counts).
//   d cos(a, b) / d a = (b^ - cos * a^) / |a|   for |a| > eps
__global__ __launch_bounds__(NT) void normal_losses_bwd_kernel(const float *__restrict__ nrm, const float *__restrict__ gtn,
                                                               const float *__restrict__ nmask, const float *__restrict__ logits, int N,
                                                               int HW, int C, int ld, int wall, int floor_, int ceiling,
                                                               const float *__restrict__ out8, const float *__restrict__ g4,
                                                               float *__restrict__ grad) {
#pragma clang fp contract(off)
    const float gn = g4[0] / (out8[4] + 1e-6f), gf = g4[1] / (out8[5] + 1e-6f), gw = g4[2] / (out8[6] + 1e-6f),
                gc = g4[3] / (out8[7] + 1e-6f);
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < (size_t)HW; p += (size_t)gridDim.x * NT) {
        int M[3] = {0, 0, 0};
        if (logits != nullptr)
            for (int n = 0; n < N; ++n) {
                const float *z = logits + ((size_t)n * HW + p) * ld;
                int best = 0;
                float bz = z[0];
                for (int c = 1; c < C; ++c)
                    if (z[c] > bz) { bz = z[c]; best = c; }
                M[0] += best == floor_; M[1] += best == wall; M[2] += best == ceiling;
            }
        for (int n = 0; n < N; ++n) {
            const size_t i = (size_t)n * HW + p;
            const float *a = nrm + (size_t)n * 3 * HW + p;
            const float ax = a[0], ay = a[HW], az = a[2 * (size_t)HW];
            const float an_raw = sqrtf(ax * ax + ay * ay + az * az), an = fmaxf(an_raw, 1e-8f);
            const float hx = ax / an, hy = ay / an, hz = az / an;
            const float live = an_raw > 1e-8f ? 1.f / an : 0.f;  // clamp_min(eps): no gradient through a clamped norm
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (gtn != nullptr) {
                const float *b = gtn + (size_t)n * 3 * HW + p;
                const float bx = b[0], by = b[HW], bz_ = b[2 * (size_t)HW];
                const float bn = fmaxf(sqrtf(bx * bx + by * by + bz_ * bz_), 1e-8f);
                const float ux = bx / bn, uy = by / bn, uz = bz_ / bn;
                const float dot = hx * ux + hy * uy + hz * uz;
                const float m = nmask ? nmask[i] : 1.f;
                if (m * (dot < 0.999f ? 1.f : 0.f) * (dot > -0.999f ? 1.f : 0.f) > 0.f) {     // term (1 - dot)
                    const float s = -gn * live;
                    dx += s * (ux - dot * hx); dy += s * (uy - dot * hy); dz += s * (uz - dot * hz);
                }
            }
            if (logits != nullptr) {
                const float *z = logits + i * ld;
                int best = 0;
                float bz = z[0];
                for (int c = 1; c < C; ++c)
                    if (z[c] > bz) { bz = z[c]; best = c; }
                // d cy / d a = (e_y - cy * a^) / |a| with cy = a^_y; only inside image n's own class mask (normal * mask)
                float coef = 0.f;
                if (best == floor_) coef += -gf * (float)M[0];                          // 1 - cy
                if (best == wall) coef += gw * (float)M[1] * 2.f * (hy > 0.f ? 1.f : (hy < 0.f ? -1.f : 0.f));   // |2 cy|
                if (best == ceiling) coef += gc * (float)M[2];                           // 1 + cy
                coef *= live;
                dx += coef * (0.f - hy * hx); dy += coef * (1.f - hy * hy); dz += coef * (0.f - hy * hz);
            }
            float *o = grad + (size_t)n * 3 * HW + p;
            o[0] = dx; o[HW] = dy; o[2 * (size_t)HW] = dz;
        }
    }
}

#define DISPATCH(dtype, ...) RDFC_DISPATCH(dtype, __VA_ARGS__)
#define CHECK_DTYPE(who, dtype) RDFC_REQUIRE(rdfc_dtype_ok(dtype), "%s: bad dtype %d", who, dtype)
#define CHECK_VEC(who, dtype, C, ...)                                                                                     \
    {                                                                                                                     \
        const int vec_ = rdfc_vec(dtype);                                                                                  \
        RDFC_REQUIRE((C) > 0 && (C) % vec_ == 0, "%s: C = %d must be a positive multiple of %d", who, C, vec_);            \
        for (int l_ : {__VA_ARGS__}) RDFC_REQUIRE(l_ >= (C) && l_ % vec_ == 0, "%s: bad leading dimension %d", who, l_); \
    }

}  // namespace

extern "C" int rdfc_maxpool2d_forward(const void *x, void *y, uint8_t *idx, int N, int Hi, int Wi, int C, int ldx, int ldy, int k, int s,
                                      int p, int Ho, int Wo, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_maxpool2d_forward", dtype);
    CHECK_VEC("rdfc_maxpool2d_forward", dtype, C, ldx, ldy);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && k > 0 && k * k < 255 && s > 0 && p >= 0 && 2 * p <= k, "rdfc_maxpool2d_forward: bad geometry");
    RDFC_REQUIRE(Ho == (Hi + 2 * p - k) / s + 1 && Wo == (Wi + 2 * p - k) / s + 1, "rdfc_maxpool2d_forward: output %dx%d does not match", Ho, Wo);
    RDFC_REQUIRE(x && y && idx, "rdfc_maxpool2d_forward: null pointer");
    const size_t total = (size_t)N * Ho * Wo * (C / (rdfc_vec(dtype)));
    DISPATCH(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(nblk(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)x, (T *)y, idx, N,
                                       Hi, Wi, C, ldx, ldy, k, s, p, Ho, Wo));
    RDFC_LAUNCH_CHECK("rdfc_maxpool2d_forward");
    return 0;
}

extern "C" int rdfc_maxpool2d_backward(const void *dy, const uint8_t *idx, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx, int k,
                                       int s, int p, int Ho, int Wo, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_maxpool2d_backward", dtype);
    CHECK_VEC("rdfc_maxpool2d_backward", dtype, C, lddy, lddx);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && k > 0 && k * k < 255 && s > 0 && p >= 0, "rdfc_maxpool2d_backward: bad geometry");
    RDFC_REQUIRE(dy && dx && idx, "rdfc_maxpool2d_backward: null pointer");
    const size_t total = (size_t)N * Hi * Wi * (C / (rdfc_vec(dtype)));
    DISPATCH(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(nblk(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)dy, idx, (T *)dx,
                                       N, Hi, Wi, C, lddy, lddx, k, s, p, Ho, Wo));
    RDFC_LAUNCH_CHECK("rdfc_maxpool2d_backward");
    return 0;
}

extern "C" int rdfc_adaptive_avgpool_forward(const void *x, void *y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho, int Wo, int dtype,
                                             void *stream) {
    CHECK_DTYPE("rdfc_adaptive_avgpool_forward", dtype);
    CHECK_VEC("rdfc_adaptive_avgpool_forward", dtype, C, ldx, ldy);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && x && y, "rdfc_adaptive_avgpool_forward: bad arguments");
    const size_t total = (size_t)N * Ho * Wo * (C / (rdfc_vec(dtype)));
    if ((Hi / Ho) * (Wi / Wo) >= 64) {         // wave per output vector
        DISPATCH(dtype, hipLaunchKernelGGL(avgpool_fwd_wave_kernel<T>, dim3(nblk(total * 64)), dim3(NT), 0, (hipStream_t)stream, (const T *)x,
                                           (T *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo));
    } else {
        DISPATCH(dtype, hipLaunchKernelGGL(avgpool_fwd_kernel<T>, dim3(nblk(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)x, (T *)y, N,
                                           Hi, Wi, C, ldx, ldy, Ho, Wo));
    }
    RDFC_LAUNCH_CHECK("rdfc_adaptive_avgpool_forward");
    return 0;
}

extern "C" int rdfc_adaptive_avgpool_backward(const void *dy, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx, int Ho, int Wo,
                                              int dtype, void *stream) {
    CHECK_DTYPE("rdfc_adaptive_avgpool_backward", dtype);
    CHECK_VEC("rdfc_adaptive_avgpool_backward", dtype, C, lddy, lddx);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && dy && dx, "rdfc_adaptive_avgpool_backward: bad arguments");
    const size_t total = (size_t)N * Hi * Wi * (C / (rdfc_vec(dtype)));
    DISPATCH(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<T>, dim3(nblk(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)dy, (T *)dx, N, Hi,
                                       Wi, C, lddy, lddx, Ho, Wo));
    RDFC_LAUNCH_CHECK("rdfc_adaptive_avgpool_backward");
    return 0;
}

extern "C" int rdfc_bilinear_forward(const void *x, void *y, int N, int Hi, int Wi, int C, int ldx, int ldy, int Ho, int Wo, int dtype,
                                     int out_f32, void *stream) {
    CHECK_DTYPE("rdfc_bilinear_forward", dtype);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && ldx >= C && ldy >= C && x && y, "rdfc_bilinear_forward: bad arguments");
    const size_t total = (size_t)N * Ho * Wo * C;
    const dim3 g(nblk(total)), b(NT);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == RDFC_BF16 && out_f32)
        hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t *)x, (float *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo);
    else if (dtype == RDFC_BF16)
        hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t *)x, (bf16_t *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo);
    else if (dtype == RDFC_F16 && out_f32)
        hipLaunchKernelGGL((bilinear_fwd_kernel<f16_t, float>), g, b, 0, st, (const f16_t *)x, (float *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo);
    else if (dtype == RDFC_F16)
        hipLaunchKernelGGL((bilinear_fwd_kernel<f16_t, f16_t>), g, b, 0, st, (const f16_t *)x, (f16_t *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo);
    else
        hipLaunchKernelGGL((bilinear_fwd_kernel<float, float>), g, b, 0, st, (const float *)x, (float *)y, N, Hi, Wi, C, ldx, ldy, Ho, Wo);
    RDFC_LAUNCH_CHECK("rdfc_bilinear_forward");
    return 0;
}

extern "C" int rdfc_bilinear_backward(const void *dy, void *dx, int N, int Hi, int Wi, int C, int lddy, int lddx, int Ho, int Wo, int dtype,
                                      int dy_f32, void *stream) {
    CHECK_DTYPE("rdfc_bilinear_backward", dtype);
    RDFC_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0 && lddx >= C && lddy >= C && dx && dy, "rdfc_bilinear_backward: bad arguments");
    const size_t total = (size_t)N * Hi * Wi * C;
    const dim3 g(nblk(total)), b(NT);
    hipStream_t st = (hipStream_t)stream;
    const int vec = rdfc_vec(dtype);
    if (!dy_f32 && C % vec == 0 && lddy % vec == 0 && lddx % vec == 0 && (Ho / Hi) * (Wo / Wi) >= 16) {      // wave per source vector
        const dim3 gw(nblk(total / vec * 64));
        DISPATCH(dtype, hipLaunchKernelGGL(bilinear_bwd_wave_kernel<T>, gw, b, 0, st, (const T *)dy, (T *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo));
        RDFC_LAUNCH_CHECK("rdfc_bilinear_backward");
        return 0;
    }
    if (dtype == RDFC_BF16 && dy_f32)
        hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, float>), g, b, 0, st, (const float *)dy, (bf16_t *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo);
    else if (dtype == RDFC_BF16)
        hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t *)dy, (bf16_t *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo);
    else if (dtype == RDFC_F16 && dy_f32)
        hipLaunchKernelGGL((bilinear_bwd_kernel<f16_t, float>), g, b, 0, st, (const float *)dy, (f16_t *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo);
    else if (dtype == RDFC_F16)
        hipLaunchKernelGGL((bilinear_bwd_kernel<f16_t, f16_t>), g, b, 0, st, (const f16_t *)dy, (f16_t *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo);
    else
        hipLaunchKernelGGL((bilinear_bwd_kernel<float, float>), g, b, 0, st, (const float *)dy, (float *)dx, N, Hi, Wi, C, lddy, lddx, Ho, Wo);
    RDFC_LAUNCH_CHECK("rdfc_bilinear_backward");
    return 0;
}

extern "C" int rdfc_scale_nc(const void *x, const float *scale, void *y, int N, int HW, int C, int ldx, int ldy, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_scale_nc", dtype);
    CHECK_VEC("rdfc_scale_nc", dtype, C, ldx, ldy);
    RDFC_REQUIRE(N > 0 && HW > 0 && x && y && scale, "rdfc_scale_nc: bad arguments");
    const size_t total = (size_t)N * HW * (C / (rdfc_vec(dtype)));
    DISPATCH(dtype, hipLaunchKernelGGL(scale_nc_kernel<T>, dim3(nblk(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)x, scale, (T *)y, N, HW,
                                       C, ldx, ldy));
    RDFC_LAUNCH_CHECK("rdfc_scale_nc");
    return 0;
}

extern "C" int rdfc_cross_entropy_nchunk(int P) { return (int)min((size_t)1024, ((size_t)P + NT - 1) / NT); }

extern "C" int rdfc_cross_entropy(const void *logits, const int64_t *labels, int P, int C, int ld, int dtype, int64_t ignore_index,
                                  double *partial, float *loss_out, float *count_out, void *stream) {
    CHECK_DTYPE("rdfc_cross_entropy", dtype);
    RDFC_REQUIRE(P > 0 && C > 0 && C <= CE_MAXC && ld >= C, "rdfc_cross_entropy: bad shape (P %d, C %d <= %d, ld %d)", P, C, CE_MAXC, ld);
    RDFC_REQUIRE(logits && labels && partial && loss_out && count_out, "rdfc_cross_entropy: null pointer");
    const int nb = rdfc_cross_entropy_nchunk(P);
    DISPATCH(dtype, hipLaunchKernelGGL(ce_fwd_kernel<T>, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (const T *)logits,
                                       (const long long *)labels, P, C, ld, (long long)ignore_index, partial));
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, loss_out, count_out);
    RDFC_LAUNCH_CHECK("rdfc_cross_entropy");
    return 0;
}

extern "C" int rdfc_cross_entropy_backward(const void *logits, const int64_t *labels, int P, int C, int ld, int dtype, int64_t ignore_index,
                                           const float *count, const float *grad_scale_dev, float grad_scale, void *dlogits, int lddl,
                                           void *stream) {
    CHECK_DTYPE("rdfc_cross_entropy_backward", dtype);
    RDFC_REQUIRE(P > 0 && C > 0 && C <= CE_MAXC && ld >= C && lddl >= C, "rdfc_cross_entropy_backward: bad shape");
    RDFC_REQUIRE(logits && labels && count && dlogits, "rdfc_cross_entropy_backward: null pointer");
    DISPATCH(dtype, hipLaunchKernelGGL(ce_bwd_kernel<T>, dim3(nblk((size_t)P)), dim3(NT), 0, (hipStream_t)stream, (const T *)logits,
                                       (const long long *)labels, P, C, ld, (long long)ignore_index, count, grad_scale_dev, grad_scale,
                                       (T *)dlogits, lddl));
    RDFC_LAUNCH_CHECK("rdfc_cross_entropy_backward");
    return 0;
}

extern "C" int rdfc_normal_losses_nchunk(int P) { return (int)min((size_t)512, ((size_t)P + NT - 1) / NT); }

extern "C" int rdfc_normal_losses(const float *normal, const float *gt_normal, const float *norm_mask, const float *logits, int N, int HW,
                                  int C, int ld, int label_wall, int label_floor, int label_ceiling, double *partial, float *out8,
                                  void *stream) {
    RDFC_REQUIRE(normal && partial && out8 && (gt_normal || logits), "rdfc_normal_losses: null pointer");
    RDFC_REQUIRE(N > 0 && HW > 0 && (logits == nullptr || (C > 0 && ld >= C)), "rdfc_normal_losses: bad shape");
    const int nb = rdfc_normal_losses_nchunk(HW);
    hipLaunchKernelGGL(normal_losses_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, normal, gt_normal, norm_mask, logits, N, HW, C, ld,
                       label_wall, label_floor, label_ceiling, partial);
    hipLaunchKernelGGL(normal_losses_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, out8);
    RDFC_LAUNCH_CHECK("rdfc_normal_losses");
    return 0;
}

extern "C" int rdfc_normal_losses_backward(const float *normal, const float *gt_normal, const float *norm_mask, const float *logits, int N,
                                           int HW, int C, int ld, int label_wall, int label_floor, int label_ceiling, const float *out8,
                                           const float *grad4, float *grad_normal, void *stream) {
    RDFC_REQUIRE(normal && out8 && grad4 && grad_normal && (gt_normal || logits), "rdfc_normal_losses_backward: null pointer");
    RDFC_REQUIRE(N > 0 && HW > 0 && (logits == nullptr || (C > 0 && ld >= C)), "rdfc_normal_losses_backward: bad shape");
    hipLaunchKernelGGL(normal_losses_bwd_kernel, dim3(rdfc_normal_losses_nchunk(HW)), dim3(NT), 0, (hipStream_t)stream, normal, gt_normal,
                       norm_mask, logits, N, HW, C, ld, label_wall, label_floor, label_ceiling, out8, grad4, grad_normal);
    RDFC_LAUNCH_CHECK("rdfc_normal_losses_backward");
    return 0;
}
