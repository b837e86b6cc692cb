// pointwise.hip -- weight packing, layout changes, window copies, reflection pad,
// generator output head, weighted loss reductions, Adam, pepper noise.
// All HBM-bound; one element (or one 16-byte vector) per thread, grid-stride.
#include "common.h"

#include <stdarg.h>

// ---- error plumbing (shared by every translation unit) --------------------------------
static thread_local char g_err[512] = "";
void rdfc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *rdfc_last_error(void) { return g_err; }
static thread_local char g_kernel[128] = "";
void rdfc_note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof g_kernel, fmt, ap);
    va_end(ap);
}
extern "C" const char *rdfc_last_kernel(void) { return g_kernel; }
extern "C" int rdfc_version(void) { return 100; }


namespace {

constexpr int NT = 256;
inline int nblocks(size_t total, int per = NT) { return (int)min((size_t)4096, (total + per - 1) / per); }
#define GRID_STRIDE(i, total) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (size_t)gridDim.x * blockDim.x)

// ---- weights ------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float *__restrict__ w, T *__restrict__ wf, T *__restrict__ wb, int O, int I, int Tn, float scale) {
    const size_t total = (size_t)O * I * Tn;
    GRID_STRIDE(i, total) {   // i indexes the OIHW source: ((o*I + c)*Tn + t)
        const int t = (int)(i % Tn);
        const int c = (int)((i / Tn) % I);
        const int o = (int)(i / ((size_t)Tn * I));
        const float v = w[i] * scale;
        if (wf) Elem<T>::st(wf + ((size_t)o * Tn + t) * I + c, v);
        if (wb) Elem<T>::st(wb + ((size_t)c * Tn + t) * O + o, v);
    }
}

// same, from a source already stored [O][T][I] (torch.channels_last weights): wf is a cast, wb the [I][T][O] transpose
template <typename T>
__global__ void pack_weight_ohwi_kernel(const float *__restrict__ w, T *__restrict__ wf, T *__restrict__ wb, int O, int I, int Tn, float scale) {
    const size_t total = (size_t)O * I * Tn;
    GRID_STRIDE(i, total) {   // i indexes the OTI source: ((o*Tn + t)*I + c)
        const int c = (int)(i % I);
        const int t = (int)((i / I) % Tn);
        const int o = (int)(i / ((size_t)Tn * I));
        const float v = w[i] * scale;
        if (wf) Elem<T>::st(wf + i, v);
        if (wb) Elem<T>::st(wb + ((size_t)c * Tn + t) * O + o, v);
    }
}

// all weights of a network in one launch: workgroup -> (job, tap, 64 x 64 tile of the [O][I] matrix of that tap); the tile goes
// through LDS so that the source rows, the w_fwd rows (along I) and the w_bwd rows (along O) are all read / written as lines
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const rdfc_pack_job *__restrict__ jobs, int njobs) {
    __shared__ float tile[64][65];
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                // last job whose block0 <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const rdfc_pack_job j = jobs[lo];
    const int tiles_i = (j.I + 63) >> 6, tiles_o = (j.O + 63) >> 6;
    int local = b - j.block0;
    const int ti = local % tiles_i;
    local /= tiles_i;
    const int to = local % tiles_o, t = local / tiles_o;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    T *wf = (T *)j.w_fwd, *wb = (T *)j.w_bwd;
    {
        const int i = ti * 64 + tx;
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {
            const int o = to * 64 + r;
            float v = 0.f;
            if (o < j.O && i < j.I) {
                v = j.w[(size_t)o * j.so + (size_t)t * j.st + (size_t)i * j.si] * j.scale;
                if (wf) Elem<T>::st(wf + ((size_t)o * j.T + t) * j.I + i, v);
            }
            tile[r][tx] = v;
        }
    }
    __syncthreads();
    if (wb) {
        const int o = to * 64 + tx;
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {
            const int i = ti * 64 + r;
            if (o < j.O && i < j.I) Elem<T>::st(wb + ((size_t)i * j.T + t) * j.O + o, tile[tx][r]);
        }
    }
}

__global__ void unpack_wgrad_kernel(const float *__restrict__ dw, float *__restrict__ g, int O, int I, int Tn, float scale) {
    const size_t total = (size_t)O * I * Tn;
    GRID_STRIDE(i, total) {   // i indexes the OIHW destination
        const int t = (int)(i % Tn);
        const int c = (int)((i / Tn) % I);
        const int o = (int)(i / ((size_t)Tn * I));
        g[i] = scale * dw[((size_t)o * Tn + t) * I + c];
    }
}

// ---- layout ---------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_pm_kernel(const float *__restrict__ s, T *__restrict__ d, int N, int C, int HW, int ld) {
    const size_t total = (size_t)N * C * HW;
    GRID_STRIDE(i, total) {   // destination-major: i = (n*HW + p)*C + c
        const int c = (int)(i % C);
        const size_t np = i / C;
        const int p = (int)(np % HW);
        const int n = (int)(np / HW);
        Elem<T>::st(d + np * ld + c, s[((size_t)n * C + c) * HW + p]);
    }
}
template <typename T>
__global__ void pm_to_nchw_kernel(const T *__restrict__ s, float *__restrict__ d, int N, int C, int HW, int ld) {
    const size_t total = (size_t)N * C * HW;
    GRID_STRIDE(i, total) {   // destination-major: i = (n*C + c)*HW + p
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int n = (int)(i / ((size_t)HW * C));
        d[i] = Elem<T>::ld(s + ((size_t)n * HW + p) * ld + c);
    }
}

template <typename T, bool VECP>
__global__ void copy_window_kernel(const T *__restrict__ s, T *__restrict__ d, int N, int Hs, int Ws, int Hd, int Wd, int Hw,
                                   int Ww, int C, int lds, int ldd, int accumulate) {
    constexpr int V = VECP ? Elem<T>::VEC : 1;
    const int CV = C / V;
    const size_t total = (size_t)N * Hw * Ww * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        const size_t pw_ = i / CV;
        const int xx = (int)(pw_ % Ww);
        const int yy = (int)((pw_ / Ww) % Hw);
        const int n = (int)(pw_ / ((size_t)Ww * Hw));
        const size_t ps = ((size_t)n * Hs + yy) * Ws + xx;
        const size_t pd = ((size_t)n * Hd + yy) * Wd + xx;
        if constexpr (VECP) {
            Vec16<T> a;
            a.load(s + ps * lds + cv * V);
            if (accumulate) {
                Vec16<T> b;
                b.load(d + pd * ldd + cv * V);
                for (int v = 0; v < V; ++v) a.set(v, a.get(v) + b.get(v));
            }
            a.store(d + pd * ldd + cv * V);
        } else {
            float v = Elem<T>::ld(s + ps * lds + cv);
            if (accumulate) v += Elem<T>::ld(d + pd * ldd + cv);
            Elem<T>::st(d + pd * ldd + cv, v);
        }
    }
}

__device__ inline int reflect(int i, int n) {   // nn.ReflectionPad2d index rule (no edge repeat)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

template <typename T, bool VECP>
__global__ void reflect_pad_kernel(const T *__restrict__ s, T *__restrict__ d, int N, int H, int W, int C, int pad) {
    constexpr int V = VECP ? Elem<T>::VEC : 1;
    const int CV = C / V, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const size_t total = (size_t)N * Hp * Wp * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        const size_t pd = i / CV;
        const int xx = (int)(pd % Wp), yy = (int)((pd / Wp) % Hp), n = (int)(pd / ((size_t)Wp * Hp));
        const size_t ps = ((size_t)n * H + reflect(yy - pad, H)) * W + reflect(xx - pad, W);
        if constexpr (VECP) {
            Vec16<T> a;
            a.load(s + ps * C + cv * V);
            a.store(d + pd * C + cv * V);
        } else {
            d[pd * C + cv] = s[ps * C + cv];
        }
    }
}

// adjoint as a gather: every source pixel sums the padded positions that read it
template <typename T, bool VECP>
__global__ void reflect_pad_adjoint_kernel(const T *__restrict__ dp, T *__restrict__ ds, int N, int H, int W, int C, int pad,
                                           const T *__restrict__ addend, int ld_add) {
    constexpr int V = VECP ? Elem<T>::VEC : 1;
    const int CV = C / V, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const size_t total = (size_t)N * H * W * CV;
    GRID_STRIDE(i, total) {
        const int cv = (int)(i % CV);
        const size_t ps = i / CV;
        const int xx = (int)(ps % W), yy = (int)((ps / W) % H), n = (int)(ps / ((size_t)W * H));
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = yy + pad;
        if (yy >= 1 && yy <= pad) ys[ny++] = pad - yy;
        if (yy <= H - 2 && yy >= H - 1 - pad) ys[ny++] = 2 * (H - 1) - yy + pad;
        xs[nx++] = xx + pad;
        if (xx >= 1 && xx <= pad) xs[nx++] = pad - xx;
        if (xx <= W - 2 && xx >= W - 1 - pad) xs[nx++] = 2 * (W - 1) - xx + pad;
        float acc[V];
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        if (addend != nullptr) {                       // the cotangent of a second consumer of the un-padded map (ResnetBlock's skip)
            if constexpr (VECP) {
                Vec16<T> t;
                t.load(addend + ps * ld_add + cv * V);
                for (int v = 0; v < V; ++v) acc[v] = t.get(v);
            } else {
                acc[0] = Elem<T>::ld(addend + ps * ld_add + cv);
            }
        }
        for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) {
                const size_t pp = ((size_t)n * Hp + ys[a]) * Wp + xs[b];
                if constexpr (VECP) {
                    Vec16<T> t;
                    t.load(dp + pp * C + cv * V);
                    for (int v = 0; v < V; ++v) acc[v] += t.get(v);
                } else {
                    acc[0] += Elem<T>::ld(dp + pp * C + cv);
                }
            }
        if constexpr (VECP) {
            Vec16<T> o;
            for (int v = 0; v < V; ++v) o.set(v, acc[v]);
            o.store(ds + ps * C + cv * V);
        } else {
            Elem<T>::st(ds + ps * C + cv, acc[0]);
        }
    }
}

// ---- generator output head (rdf_generator.py:400-406) -------------------------------------------
__global__ void fusion_head_kernel(const float *d1, const float *c1, const float *d2, const float *c2, float *d2c, float *pred, int n) {
    GRID_STRIDE(i, (size_t)n) {
        const float b = fminf(fmaxf(d2[i], -1.f), 1.f);
        const float s1 = 1.f / (1.f + expf(c2[i] - c1[i]));   // softmax over (c1, c2), first component
        d2c[i] = b;
        pred[i] = d1[i] * s1 + b * (1.f - s1);
    }
}
__global__ void fusion_head_bwd_kernel(const float *d1, const float *c1, const float *d2, const float *c2, const float *gp,
                                       const float *gd2c, float *gd1, float *gc1, float *gd2, float *gc2, int n) {
    GRID_STRIDE(i, (size_t)n) {
        const float raw = d2[i];
        const float b = fminf(fmaxf(raw, -1.f), 1.f);
        const float s1 = 1.f / (1.f + expf(c2[i] - c1[i]));
        const float g = gp ? gp[i] : 0.f;
        const float gb = g * (1.f - s1) + (gd2c ? gd2c[i] : 0.f);
        gd1[i] = g * s1;
        const float gs = g * (d1[i] - b) * s1 * (1.f - s1);   // d pred / d c1 = (d1 - b) s1 (1 - s1)
        gc1[i] = gs;
        gc2[i] = -gs;
        gd2[i] = (raw >= -1.f && raw <= 1.f) ? gb : 0.f;       // torch.clamp passes the gradient on the closed interval
    }
}

// ---- weighted loss reductions -----------------------------------------------------------------------
constexpr int LCHUNK = 4096;
__global__ __launch_bounds__(NT) void loss_kernel(const float *__restrict__ pred, const float *__restrict__ target, float tconst,
                                                  const float *__restrict__ weight, float uw, int C, int HW, size_t total, int kind,
                                                  float *__restrict__ partial, float *__restrict__ grad, float gscale) {
    const size_t b0 = (size_t)blockIdx.x * LCHUNK;
    float s = 0.f;
    for (int k = threadIdx.x; k < LCHUNK; k += NT) {
        const size_t i = b0 + k;
        if (i >= total) break;
        const float t = target ? target[i] : tconst;
        float w = uw;
        if (weight) {
            const size_t n = i / ((size_t)C * HW);
            w = weight[n * HW + i % HW];
        }
        const float d = pred[i] - t;
        s += w * (kind == 1 ? fabsf(d) : d * d);
        if (grad) grad[i] = gscale * w * (kind == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d);
    }
    s = wave_sum(s);
    __shared__ float red[NT / 64];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void loss_finalize_kernel(const float *partial, int n, float *out) {
    // one wave; fixed order
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (threadIdx.x == 0) *out = (float)s;
}

// ---- Adam (torch.optim.Adam, amsgrad=False, weight_decay=0) ------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale, float keep) {
    GRID_STRIDE(i, n) {                 // keep = 1 - lr * weight_decay (torch.optim.AdamW's decoupled decay; 1 for Adam)
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] * keep - (lr / bc1) * (mi / denom);
    }
}

// same update with the learning rate and the step count read from device memory (hyper = {lr, step}), so that the
// launch can sit in a captured HIP graph: nothing that changes between replays is a kernel argument
__global__ void adam_dev_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                size_t n, const float *__restrict__ hyper, float b1, float b2, float eps, float gscale) {
    const float lr = hyper[0];
    const double t = (double)hyper[1];
    const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// ---- dynamic loss scaling (f16 mode): everything on the device, nothing the host has to wait for ---------------------------
// scaler = {scale, 1 / scale, steps since the last overflow, overflow seen in this step}; ostate (per optimizer) = {a gradient of
// this step is inf / NaN, steps skipped so far}.
__global__ void grad_nonfinite_kernel(const float *__restrict__ g, size_t n, float *__restrict__ ostate) {
    bool bad = false;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 t = reinterpret_cast<const f32x4 *>(g)[i];
        // (x - x) is 0 for finite x, NaN for inf / NaN
        const float z = (t[0] - t[0]) + (t[1] - t[1]) + (t[2] - t[2]) + (t[3] - t[3]);
        bad |= !(z == 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float t = g[n4 * 4 + threadIdx.x];
        bad |= !((t - t) == 0.f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) ostate[0] = 1.f;      // (same value from every writer)
}
// Adam / AdamW with the gradient scale 1 / (loss scale * world size) taken from the device, skipped as a whole when this
// optimizer's gradients overflowed; bias corrections count the steps that were really taken (step - skipped)
__global__ void adam_scaled_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                                   size_t n, float lr_host, const float *__restrict__ hyper, int step_host, float b1, float b2, float eps,
                                   float gscale, float weight_decay, const float *__restrict__ scaler, float *__restrict__ ostate,
                                   float *__restrict__ scaler_flag) {
    if (ostate[0] != 0.f) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            ostate[1] += 1.f;
            *scaler_flag = 1.f;
        }
        return;
    }
    const float lr = hyper != nullptr ? hyper[0] : lr_host;
    const double t = (hyper != nullptr ? (double)hyper[1] : (double)step_host) - (double)ostate[1];
    const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
    const float gs = gscale * scaler[1], keep = 1.f - lr * weight_decay;
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] * keep - (lr / bc1) * (mi / denom);
    }
}
__global__ void loss_scale_update_kernel(float *scaler, float growth, float backoff, float interval, float min_scale, float max_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = scaler[0], good = scaler[2];
    if (scaler[3] != 0.f) {
        s = fmaxf(s * backoff, min_scale);
        good = 0.f;
    } else if (++good >= interval) {
        s = fminf(s * growth, max_scale);
        good = 0.f;
    }
    scaler[0] = s;
    scaler[1] = 1.f / s;
    scaler[2] = good;
    scaler[3] = 0.f;
}

// ---- evaluation metrics (eval path): one workgroup per image, one pass over (gt, pred[, mask]) ------------------------------
// per image: RMSE, MAE, iRMSE, iMAE, REL, delta^1..3 over the pixels with gt > t_valid (& evaluate_mask); inputs are the
// normalised maps, de-normalised on the fly (x * std + mean).  f64 accumulation, fixed order -> reproducible.
constexpr int MNT = 1024;
__global__ __launch_bounds__(MNT) void depth_metrics_kernel(const float *__restrict__ gt_n, const float *__restrict__ pd_n,
                                                            const uint8_t *__restrict__ emask, int HW, float stdv, float meanv,
                                                            float t_valid, float *__restrict__ out) {
// separately rounded multiply and add like the reference's x * std + mean: a contracted fma changes a prediction next to zero
// (x ~ -1) by many ulps and 1/pred with it, and one such pixel can carry most of an image's iRMSE
#pragma clang fp contract(off)
    __shared__ double red[9][MNT / 64];
    const int n = blockIdx.x;
    const float *g = gt_n + (size_t)n * HW, *p = pd_n + (size_t)n * HW;
    const uint8_t *m = emask ? emask + (size_t)n * HW : nullptr;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    const float t1 = 1.25f, t2 = (float)(1.25 * 1.25), t3 = (float)(1.25 * 1.25 * 1.25);
    for (int i = threadIdx.x; i < HW; i += MNT) {
        const float gt = g[i] * stdv + meanv, pd = p[i] * stdv + meanv;
        if (!(gt > t_valid) || (m && !m[i])) continue;
        const float pi = pd <= t_valid ? 0.f : 1.f / (pd + 1e-8f), gi = 1.f / (gt + 1e-8f);
        const float d = pd - gt, di = pi - gi;
        const float ratio = fmaxf(gt / (pd + 1e-8f), pd / (gt + 1e-8f));
        acc[0] += 1.0;
        acc[1] += (double)(d * d);
        acc[2] += (double)fabsf(d);
        acc[3] += (double)(di * di);
        acc[4] += (double)fabsf(di);
        acc[5] += (double)(fabsf(d) / (gt + 1e-8f));
        acc[6] += ratio < t1 ? 1.0 : 0.0;
        acc[7] += ratio < t2 ? 1.0 : 0.0;
        acc[8] += ratio < t3 ? 1.0 : 0.0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s[9];
        for (int k = 0; k < 9; ++k) {
            s[k] = 0.0;
            for (int w = 0; w < MNT / 64; ++w) s[k] += red[k][w];
        }
        const double nv = s[0] + 1e-8;
        float *o = out + (size_t)n * 8;
        o[0] = (float)sqrt(s[1] / nv);
        o[1] = (float)(s[2] / nv);
        o[2] = (float)sqrt(s[3] / nv);
        o[3] = (float)(s[4] / nv);
        o[4] = (float)(s[5] / nv);
        o[5] = (float)(s[6] / nv);
        o[6] = (float)(s[7] / nv);
        o[7] = (float)(s[8] / nv);
    }
}

__global__ void pepper_kernel(float *img, const uint8_t *mask, int C, int HW, size_t total) {
    GRID_STRIDE(i, total) {
        const size_t n = i / ((size_t)C * HW);
        const uint8_t mk = mask[n * HW + i % HW];
        if (mk == 1) img[i] = 1.f;
        else if (mk == 2) img[i] = -1.f;
    }
}

// device-side draw of the same mask distribution (P(keep) = snr, P(+1) = P(-1) = (1 - snr) / 2 per pixel, shared by the
// channels): counter-based, one 64-bit mix (splitmix64) of (seed, call counter, pixel) per pixel -- no state, no host copy.
// NOT upstream's numpy stream: off by default (RDFCGAN(device_pepper=True)), for training at speed.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void pepper_draw_kernel(uint8_t *__restrict__ mask, size_t npix, unsigned long long seed, unsigned long long counter, float snr) {
    GRID_STRIDE(i, npix) {
        const unsigned long long r = mix64(mix64(seed ^ (counter * 0xD1B54A32D192ED03ull)) + i);
        const float u = (float)(r >> 40) * (1.f / 16777216.f);                 // 24 random bits -> [0, 1)
        const float half = (1.f - snr) * 0.5f;
        mask[i] = u < snr ? 0 : (u < snr + half ? 1 : 2);
    }
}

// ---- small elementwise kernels that keep stock-PyTorch arithmetic off the path ------------------------------------------
// salt/pepper backward: the in-place assignment upstream cuts the gradient at the replaced pixels (helper.py:384-402)
__global__ void mask_keep_kernel(const float *__restrict__ g, const uint8_t *__restrict__ mask, float *__restrict__ out, int C, int HW, size_t total) {
    GRID_STRIDE(i, total) {
        const size_t n = i / ((size_t)C * HW);
        out[i] = mask[n * HW + i % HW] == 0 ? g[i] : 0.f;
    }
}

// out = x * (*scalar) * k  (the cotangent of a loss is a 0-dim device tensor; k: a host factor)
__global__ void scale_dev_kernel(const float *__restrict__ x, const float *__restrict__ scalar, float k, float *__restrict__ out, size_t n) {
    const float sc = (scalar ? *scalar : 1.f) * k;
    GRID_STRIDE(i, n) out[i] = x[i] * sc;
}

// helper.py:404-408 norm_normalize: v / (sqrt(sum_c v_c^2) + 1e-10) over the first three channels of an NCHW map
__global__ void unit_normals_kernel(const float *__restrict__ src, float *__restrict__ dst, int Csrc, int HW, size_t npix) {
    GRID_STRIDE(i, npix) {
        const size_t n = i / HW, p = i % HW;
        const float *s = src + n * Csrc * HW + p;
        const float a = s[0], b = s[(size_t)HW], c = s[2 * (size_t)HW];
        const float inv = 1.f / (sqrtf(a * a + b * b + c * c) + 1e-10f);
        float *d = dst + n * 3 * HW + p;
        d[0] = a * inv; d[(size_t)HW] = b * inv; d[2 * (size_t)HW] = c * inv;
    }
}

// loss_G = sum_i w_i * l_i (rdfc_gan.py:252-267) and its logged terms in one launch; sequential f32 adds in the given order
// (what the chain of 0-dim tensor ops upstream computes), then * post
constexpr int MAXTERMS = 16;
struct Terms { const float *p[MAXTERMS]; float w[MAXTERMS]; int n; float post; };
__global__ void combine_kernel(Terms t, float *__restrict__ out /* [n + 1] */) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.f;
    for (int i = 0; i < t.n; ++i) {
        const float v = *t.p[i] * t.w[i];
        out[i] = v;
        acc = i == 0 ? v : acc + v;
    }
    out[t.n] = acc * t.post;
}
__global__ void combine_bwd_kernel(Terms t, const float *__restrict__ g, float *__restrict__ out /* [n] */) {
    const int i = threadIdx.x;
    if (i < t.n) out[i] = (g ? *g : 1.f) * t.post * t.w[i];
}

// dst = sum of up to four pixel-major maps (each with its own pixel stride): the gradient of an activation that feeds
// several consumers, in ONE pass (autograd would add them pairwise with stock elementwise kernels, after making
// channel-slice cotangents contiguous)
struct SumSrc { const void *p[4]; int ld[4]; int n; };
template <typename T>
__global__ void sum_maps_kernel(SumSrc s, T *__restrict__ dst, int C, int ld_dst, size_t total /* P * C / VEC */) {
    constexpr int VEC = Elem<T>::VEC;
    const int cv = C / VEC;
    GRID_STRIDE(i, total) {
        const size_t px = i / cv;
        const int c = (int)(i % cv) * VEC;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
        for (int k = 0; k < s.n; ++k) {
            Vec16<T> x;
            x.load(reinterpret_cast<const T *>(s.p[k]) + px * s.ld[k] + c);
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] += x.get(v);
        }
        Vec16<T> o;
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, acc[v]);
        o.store(dst + px * ld_dst + c);
    }
}
template <typename T>
__global__ void sum_maps_scalar_kernel(SumSrc s, T *__restrict__ dst, int C, int ld_dst, size_t total /* P * C */) {
    GRID_STRIDE(i, total) {
        const size_t px = i / C;
        const int c = (int)(i % C);
        float acc = 0.f;
        for (int k = 0; k < s.n; ++k) acc += Elem<T>::ld(reinterpret_cast<const T *>(s.p[k]) + px * s.ld[k] + c);
        Elem<T>::st(dst + px * ld_dst + c, acc);
    }
}

}  // namespace

#define DISPATCH(dtype, ...) RDFC_DISPATCH(dtype, __VA_ARGS__)
#define CHECK_DTYPE(who, dtype) RDFC_REQUIRE(rdfc_dtype_ok(dtype), "%s: bad dtype %d", who, dtype)

extern "C" int rdfc_pack_weight(const float *w, void *wf, void *wb, int O, int I, int KH, int KW, int dtype, float scale, void *stream) {
    CHECK_DTYPE("rdfc_pack_weight", dtype);
    RDFC_REQUIRE(O > 0 && I > 0 && KH > 0 && KW > 0, "rdfc_pack_weight: bad shape");
    const size_t total = (size_t)O * I * KH * KW;
    DISPATCH(dtype, hipLaunchKernelGGL(pack_weight_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, w, (T *)wf,
                                       (T *)wb, O, I, KH * KW, scale));
    RDFC_LAUNCH_CHECK("rdfc_pack_weight");
    return 0;
}

extern "C" int rdfc_pack_weight_ohwi(const float *w, void *wf, void *wb, int O, int I, int KH, int KW, int dtype, float scale, void *stream) {
    CHECK_DTYPE("rdfc_pack_weight_ohwi", dtype);
    RDFC_REQUIRE(O > 0 && I > 0 && KH > 0 && KW > 0, "rdfc_pack_weight_ohwi: bad shape");
    const size_t total = (size_t)O * I * KH * KW;
    DISPATCH(dtype, hipLaunchKernelGGL(pack_weight_ohwi_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, w, (T *)wf,
                                       (T *)wb, O, I, KH * KW, scale));
    RDFC_LAUNCH_CHECK("rdfc_pack_weight_ohwi");
    return 0;
}

extern "C" int rdfc_pack_weights_batch(const rdfc_pack_job *jobs, int njobs, int nblocks, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_pack_weights_batch", dtype);
    RDFC_REQUIRE(jobs != nullptr && njobs > 0 && nblocks >= njobs, "rdfc_pack_weights_batch: bad job table");
    DISPATCH(dtype, hipLaunchKernelGGL(pack_batch_kernel<T>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, jobs, njobs));
    RDFC_LAUNCH_CHECK("rdfc_pack_weights_batch");
    return 0;
}

extern "C" int rdfc_unpack_wgrad(const float *dw, float *g, int O, int I, int KH, int KW, float scale, void *stream) {
    const size_t total = (size_t)O * I * KH * KW;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, dw, g, O, I, KH * KW, scale);
    RDFC_LAUNCH_CHECK("rdfc_unpack_wgrad");
    return 0;
}

extern "C" int rdfc_nchw_to_pm(const float *src, void *dst, int N, int C, int H, int W, int ld, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_nchw_to_pm", dtype);
    RDFC_REQUIRE(ld >= C, "rdfc_nchw_to_pm: ld < C");
    const size_t total = (size_t)N * C * H * W;
    DISPATCH(dtype, hipLaunchKernelGGL(nchw_to_pm_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, src, (T *)dst, N, C, H * W, ld));
    RDFC_LAUNCH_CHECK("rdfc_nchw_to_pm");
    return 0;
}
extern "C" int rdfc_pm_to_nchw(const void *src, float *dst, int N, int C, int H, int W, int ld, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_pm_to_nchw", dtype);
    RDFC_REQUIRE(ld >= C, "rdfc_pm_to_nchw: ld < C");
    const size_t total = (size_t)N * C * H * W;
    DISPATCH(dtype, hipLaunchKernelGGL(pm_to_nchw_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, (const T *)src, dst, N, C, H * W, ld));
    RDFC_LAUNCH_CHECK("rdfc_pm_to_nchw");
    return 0;
}

extern "C" int rdfc_copy_window(const void *src, void *dst, int N, int Hs, int Ws, int Hd, int Wd, int Hw, int Ww, int C,
                                int ld_src, int ld_dst, int dtype, int accumulate, void *stream) {
    CHECK_DTYPE("rdfc_copy_window", dtype);
    RDFC_REQUIRE(Hw <= Hs && Ww <= Ws && Hw <= Hd && Ww <= Wd && ld_src >= C && ld_dst >= C, "rdfc_copy_window: window / ld mismatch");
    const int vec = rdfc_vec(dtype);
    const bool v = C % vec == 0 && ld_src % vec == 0 && ld_dst % vec == 0;
    const size_t total = (size_t)N * Hw * Ww * (v ? C / vec : C);
    DISPATCH(dtype, if (v) hipLaunchKernelGGL((copy_window_kernel<T, true>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                              (const T *)src, (T *)dst, N, Hs, Ws, Hd, Wd, Hw, Ww, C, ld_src, ld_dst, accumulate);
             else hipLaunchKernelGGL((copy_window_kernel<T, false>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                     (const T *)src, (T *)dst, N, Hs, Ws, Hd, Wd, Hw, Ww, C, ld_src, ld_dst, accumulate));
    RDFC_LAUNCH_CHECK("rdfc_copy_window");
    return 0;
}

extern "C" int rdfc_reflect_pad(const void *src, void *dst, int N, int H, int W, int C, int pad, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_reflect_pad", dtype);
    RDFC_REQUIRE(pad >= 0 && pad < H && pad < W, "rdfc_reflect_pad: pad must be smaller than the map");
    const int vec = rdfc_vec(dtype);
    const bool v = C % vec == 0;
    const size_t total = (size_t)N * (H + 2 * pad) * (W + 2 * pad) * (v ? C / vec : C);
    DISPATCH(dtype, if (v) hipLaunchKernelGGL((reflect_pad_kernel<T, true>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                              (const T *)src, (T *)dst, N, H, W, C, pad);
             else hipLaunchKernelGGL((reflect_pad_kernel<T, false>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                     (const T *)src, (T *)dst, N, H, W, C, pad));
    RDFC_LAUNCH_CHECK("rdfc_reflect_pad");
    return 0;
}
extern "C" int rdfc_reflect_pad_adjoint_add(const void *dpad, const void *addend, int ld_add, void *dsrc, int N, int H, int W, int C, int pad,
                                            int dtype, void *stream);
extern "C" int rdfc_reflect_pad_adjoint(const void *dpad, void *dsrc, int N, int H, int W, int C, int pad, int dtype, void *stream) {
    return rdfc_reflect_pad_adjoint_add(dpad, nullptr, 0, dsrc, N, H, W, C, pad, dtype, stream);
}
extern "C" int rdfc_reflect_pad_adjoint_add(const void *dpad, const void *addend, int ld_add, void *dsrc, int N, int H, int W, int C, int pad,
                                            int dtype, void *stream) {
    CHECK_DTYPE("rdfc_reflect_pad_adjoint", dtype);
    RDFC_REQUIRE(pad >= 0 && pad < H && pad < W, "rdfc_reflect_pad_adjoint: pad must be smaller than the map");
    const int vec = rdfc_vec(dtype);
    RDFC_REQUIRE(addend == nullptr || ld_add >= C, "rdfc_reflect_pad_adjoint_add: ld_add < C");
    const bool v = C % vec == 0 && (addend == nullptr || ld_add % vec == 0);
    const size_t total = (size_t)N * H * W * (v ? C / vec : C);
    DISPATCH(dtype, if (v) hipLaunchKernelGGL((reflect_pad_adjoint_kernel<T, true>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                              (const T *)dpad, (T *)dsrc, N, H, W, C, pad, (const T *)addend, ld_add);
             else hipLaunchKernelGGL((reflect_pad_adjoint_kernel<T, false>), dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream,
                                     (const T *)dpad, (T *)dsrc, N, H, W, C, pad, (const T *)addend, ld_add));
    RDFC_LAUNCH_CHECK("rdfc_reflect_pad_adjoint");
    return 0;
}

extern "C" int rdfc_fusion_head(const float *d1, const float *c1, const float *d2, const float *c2, float *d2c, float *pred, int n, void *stream) {
    hipLaunchKernelGGL(fusion_head_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, d1, c1, d2, c2, d2c, pred, n);
    RDFC_LAUNCH_CHECK("rdfc_fusion_head");
    return 0;
}
extern "C" int rdfc_fusion_head_bwd(const float *d1, const float *c1, const float *d2, const float *c2, const float *gp, const float *gd2c,
                                    float *gd1, float *gc1, float *gd2, float *gc2, int n, void *stream) {
    hipLaunchKernelGGL(fusion_head_bwd_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, d1, c1, d2, c2, gp, gd2c, gd1, gc1, gd2, gc2, n);
    RDFC_LAUNCH_CHECK("rdfc_fusion_head_bwd");
    return 0;
}

extern "C" int rdfc_loss_nchunk(int numel) { return cdiv(numel, LCHUNK); }
extern "C" int rdfc_weighted_loss(const float *pred, const float *target, float tconst, const float *weight, float uw, int N, int C,
                                  int HW, int kind, float *partial, float *loss_out, float *grad, float gscale, void *stream) {
    RDFC_REQUIRE(kind == 1 || kind == 2, "rdfc_weighted_loss: kind must be 1 (L1) or 2 (L2)");
    const size_t total = (size_t)N * C * HW;
    const int nb = cdiv(total, LCHUNK);
    hipLaunchKernelGGL(loss_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, pred, target, tconst, weight, uw, C, HW, total, kind, partial, grad, gscale);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, loss_out);
    RDFC_LAUNCH_CHECK("rdfc_weighted_loss");
    return 0;
}

extern "C" int rdfc_adam_step(float *param, const float *grad, float *m, float *v, size_t n, float lr, float b1, float b2, float eps,
                              int step, float gscale, void *stream) {
    RDFC_REQUIRE(step >= 1, "rdfc_adam_step: step counts from 1");
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, m, v, n, lr, b1, b2, eps,
                       (float)bc1, (float)sqrt(bc2), gscale, 1.f);
    RDFC_LAUNCH_CHECK("rdfc_adam_step");
    return 0;
}

extern "C" int rdfc_adamw_step(float *param, const float *grad, float *m, float *v, size_t n, float lr, float b1, float b2, float eps,
                               float weight_decay, int step, float gscale, void *stream) {
    RDFC_REQUIRE(step >= 1, "rdfc_adamw_step: step must be >= 1 (got %d)", step);
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, m, v, n, lr, b1, b2, eps,
                       (float)bc1, (float)sqrt(bc2), gscale, 1.f - lr * weight_decay);
    RDFC_LAUNCH_CHECK("rdfc_adamw_step");
    return 0;
}

extern "C" int rdfc_adam_step_dev(float *param, const float *grad, float *m, float *v, size_t n, const float *hyper, float b1,
                                  float b2, float eps, float gscale, void *stream) {
    RDFC_REQUIRE(hyper != nullptr, "rdfc_adam_step_dev: null hyper-parameter buffer");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, m, v, n, hyper, b1, b2, eps,
                       gscale);
    RDFC_LAUNCH_CHECK("rdfc_adam_step_dev");
    return 0;
}

extern "C" int rdfc_grad_nonfinite(const float *grad, size_t n, float *ostate, void *stream) {
    RDFC_REQUIRE(grad != nullptr && ostate != nullptr, "rdfc_grad_nonfinite: null pointer");
    RDFC_REQUIRE(((uintptr_t)grad & 15) == 0, "rdfc_grad_nonfinite: the gradient buffer must be 16-byte aligned");
    if (hipMemsetAsync(ostate, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) {
        rdfc_set_error("rdfc_grad_nonfinite: hipMemsetAsync failed");
        return 2;
    }
    hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(nblocks(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, grad, n, ostate);
    RDFC_LAUNCH_CHECK("rdfc_grad_nonfinite");
    return 0;
}

extern "C" int rdfc_adam_step_scaled(float *param, const float *grad, float *m, float *v, size_t n, float lr, const float *hyper, int step,
                                     float b1, float b2, float eps, float weight_decay, float gscale, float *scaler, float *ostate,
                                     void *stream) {
    RDFC_REQUIRE(scaler != nullptr && ostate != nullptr, "rdfc_adam_step_scaled: null scaler state");
    RDFC_REQUIRE(hyper != nullptr || step >= 1, "rdfc_adam_step_scaled: step counts from 1");
    hipLaunchKernelGGL(adam_scaled_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, m, v, n, lr, hyper, step, b1, b2,
                       eps, gscale, weight_decay, (const float *)scaler, ostate, scaler + 3);
    RDFC_LAUNCH_CHECK("rdfc_adam_step_scaled");
    return 0;
}

extern "C" int rdfc_loss_scale_update(float *scaler, float growth, float backoff, int interval, float min_scale, float max_scale,
                                      void *stream) {
    RDFC_REQUIRE(scaler != nullptr, "rdfc_loss_scale_update: null scaler state");
    RDFC_REQUIRE(growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1, "rdfc_loss_scale_update: growth >= 1, 0 < backoff <= 1, interval >= 1");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, growth, backoff, (float)interval, min_scale,
                       max_scale);
    RDFC_LAUNCH_CHECK("rdfc_loss_scale_update");
    return 0;
}

extern "C" int rdfc_depth_metrics(const float *gt_norm, const float *pred_norm, const uint8_t *evaluate_mask, int N, int HW, float std,
                                  float mean, float t_valid, float *out, void *stream) {
    RDFC_REQUIRE(gt_norm != nullptr && pred_norm != nullptr && out != nullptr, "rdfc_depth_metrics: null pointer");
    RDFC_REQUIRE(N > 0 && HW > 0, "rdfc_depth_metrics: empty input");
    hipLaunchKernelGGL(depth_metrics_kernel, dim3(N), dim3(MNT), 0, (hipStream_t)stream, gt_norm, pred_norm, evaluate_mask, HW, std,
                       mean, t_valid, out);
    RDFC_LAUNCH_CHECK("rdfc_depth_metrics");
    return 0;
}

extern "C" int rdfc_draw_pepper_mask(uint8_t *mask, int N, int HW, uint64_t seed, uint64_t counter, float snr, void *stream) {
    RDFC_REQUIRE(mask != nullptr && N > 0 && HW > 0 && snr >= 0.f && snr <= 1.f, "rdfc_draw_pepper_mask: bad arguments");
    const size_t npix = (size_t)N * HW;
    hipLaunchKernelGGL(pepper_draw_kernel, dim3(nblocks(npix)), dim3(NT), 0, (hipStream_t)stream, mask, npix, (unsigned long long)seed,
                       (unsigned long long)counter, snr);
    RDFC_LAUNCH_CHECK("rdfc_draw_pepper_mask");
    return 0;
}

extern "C" int rdfc_apply_pepper(float *img, const uint8_t *mask, int N, int C, int HW, void *stream) {
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(pepper_kernel, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, img, mask, C, HW, total);
    RDFC_LAUNCH_CHECK("rdfc_apply_pepper");
    return 0;
}

extern "C" int rdfc_zero(void *p, size_t bytes, void *stream) {
    if (bytes == 0) return 0;
    RDFC_REQUIRE(p != nullptr, "rdfc_zero: null pointer");
    if (hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) != hipSuccess) {
        rdfc_set_error("rdfc_zero: hipMemsetAsync failed");
        return 2;
    }
    return 0;
}

extern "C" int rdfc_mask_keep(const float *g, const uint8_t *mask, float *out, int N, int C, int HW, void *stream) {
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(mask_keep_kernel, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, g, mask, out, C, HW, total);
    RDFC_LAUNCH_CHECK("rdfc_mask_keep");
    return 0;
}

extern "C" int rdfc_scale_dev(const float *x, const float *scalar, float k, float *out, size_t n, void *stream) {
    hipLaunchKernelGGL(scale_dev_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, x, scalar, k, out, n);
    RDFC_LAUNCH_CHECK("rdfc_scale_dev");
    return 0;
}

extern "C" int rdfc_unit_normals(const float *src, float *dst, int N, int Csrc, int HW, void *stream) {
    RDFC_REQUIRE(Csrc >= 3, "rdfc_unit_normals: needs at least three channels");
    const size_t npix = (size_t)N * HW;
    hipLaunchKernelGGL(unit_normals_kernel, dim3(nblocks(npix)), dim3(NT), 0, (hipStream_t)stream, src, dst, Csrc, HW, npix);
    RDFC_LAUNCH_CHECK("rdfc_unit_normals");
    return 0;
}

extern "C" int rdfc_sum_maps(int n, const void *const *srcs, const int *lds, void *dst, int P, int C, int ld_dst, int dtype, void *stream) {
    CHECK_DTYPE("rdfc_sum_maps", dtype);
    RDFC_REQUIRE(n >= 1 && n <= 4 && P > 0 && C > 0 && ld_dst >= C, "rdfc_sum_maps: 1..4 sources, ld_dst >= C");
    SumSrc s{};
    s.n = n;
    const int vec = rdfc_vec(dtype);
    bool v = C % vec == 0 && ld_dst % vec == 0;
    for (int k = 0; k < n; ++k) {
        RDFC_REQUIRE(srcs[k] != nullptr && lds[k] >= C, "rdfc_sum_maps: bad source %d", k);
        s.p[k] = srcs[k];
        s.ld[k] = lds[k];
        v = v && lds[k] % vec == 0 && ((uintptr_t)srcs[k] & 15) == 0;
    }
    v = v && ((uintptr_t)dst & 15) == 0;
    if (v) {
        const size_t total = (size_t)P * (C / vec);
        DISPATCH(dtype, hipLaunchKernelGGL(sum_maps_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, s, (T *)dst, C, ld_dst, total));
    } else {
        const size_t total = (size_t)P * C;
        DISPATCH(dtype, hipLaunchKernelGGL(sum_maps_scalar_kernel<T>, dim3(nblocks(total)), dim3(NT), 0, (hipStream_t)stream, s, (T *)dst, C, ld_dst, total));
    }
    RDFC_LAUNCH_CHECK("rdfc_sum_maps");
    return 0;
}

extern "C" int rdfc_combine_scalars(int n, const float *const *terms, const float *weights, float post, float *out, void *stream) {
    RDFC_REQUIRE(n >= 1 && n <= MAXTERMS && terms != nullptr && weights != nullptr && out != nullptr, "rdfc_combine_scalars: 1..%d terms", MAXTERMS);
    Terms t{};
    t.n = n;
    t.post = post;
    for (int i = 0; i < n; ++i) {
        RDFC_REQUIRE(terms[i] != nullptr, "rdfc_combine_scalars: null term %d", i);
        t.p[i] = terms[i];
        t.w[i] = weights[i];
    }
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, out);
    RDFC_LAUNCH_CHECK("rdfc_combine_scalars");
    return 0;
}

extern "C" int rdfc_combine_scalars_backward(int n, const float *weights, float post, const float *grad_total, float *out, void *stream) {
    RDFC_REQUIRE(n >= 1 && n <= MAXTERMS && weights != nullptr && out != nullptr, "rdfc_combine_scalars_backward: 1..%d terms", MAXTERMS);
    Terms t{};
    t.n = n;
    t.post = post;
    for (int i = 0; i < n; ++i) t.w[i] = weights[i];
    hipLaunchKernelGGL(combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t, grad_total, out);
    RDFC_LAUNCH_CHECK("rdfc_combine_scalars_backward");
    return 0;
}
