// conv_wgrad.hip -- weight gradient of a convolution as a pixel-contraction GEMM on MFMA.
//
//   dW[co][tap][ci] += sum_m  dY[m][co] * X[src(m, tap)][ci]        (m = output pixel)
//
// One workgroup owns a (TCO x TCI) tile of one tap and a slice of the pixel range
// (split-K over pixels; tiles are combined with f32 atomics, 64-byte segments).
// Both operands are pixel-major in memory, i.e. the contraction index is the *slow*
// index of both tiles.  They are staged into LDS as [pixel][channel] rows (full 16-byte
// coalesced loads) and the MFMA fragments -- which need 8 consecutive k per lane -- are
// produced by the gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) or by plain
// 4-byte reads (f32 MFMA takes one k per lane).
#include "common.h"

#include <stdlib.h>

namespace {

struct WgradParams {
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int KH, KW, sh, sw, ph, pw, dh, dw;
    int ldx, ldy;
    int P;          // N*Ho*Wo
    int steps_per_slice;   // k-steps (of 32 pixels) per split-K slice
};

constexpr int BKP = 32;      // pixels per k-step

__device__ inline bf16x8 tr_frag(const bf16_t *img, int ld, int cbase, int lane) {
    // rows = pixels (k), cols = channels. Group g=lane>>4 covers k = 8g..8g+7.
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const bf16_t *a0 = img + (8 * g + q) * ld + cbase + 4 * pp;
    const bf16_t *a1 = a0 + 4 * ld;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <typename T> struct WMma {       // 2-byte elements (bf16 | f16)
    template <int MT, int NT>
    __device__ static void step(const T *Ys_, const T *Xs_, int ldy, int ldx, int arow, int bcol, int lane,
                                f32x4 (&acc)[MT][NT]) {
        const bf16_t *Ys = reinterpret_cast<const bf16_t *>(Ys_), *Xs = reinterpret_cast<const bf16_t *>(Xs_);
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = tr_frag(Ys, ldy, arow + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = tr_frag(Xs, ldx, bcol + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = mma16<DtOf<T>::value>(a[i], b[j], acc[i][j]);
    }
};
template <> struct WMma<float> {
    template <int MT, int NT>
    __device__ static void step(const float *Ys, const float *Xs, int ldy, int ldx, int arow, int bcol, int lane,
                                f32x4 (&acc)[MT][NT]) {
        const int r = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < BKP / 4; ++kk) {
            float a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = Ys[(4 * kk + kq) * ldy + arow + 16 * i + r];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = Xs[(4 * kk + kq) * ldx + bcol + 16 * j + r];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
};

template <typename T, int TCO, int TCI, int WCO, bool VY, bool VX, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, NTHREADS == 512 && sizeof(T) == 2 ? 4 : 1) void wgrad_kernel(WgradParams p, const T *__restrict__ x,
                                                         const T *__restrict__ dy, float *__restrict__ dw) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int WCI = (NTHREADS / 64) / WCO;
    constexpr int WTCO = TCO / WCO, WTCI = TCI / WCI;
    constexpr int MT = WTCO / 16, NT = WTCI / 16;
    constexpr int LDY = TCO + VEC, LDX = TCI + VEC;
    constexpr int CY = TCO / VEC, CX = TCI / VEC;                     // 16-byte chunks per pixel row
    // loads per thread and k-step: vector path = 16-byte chunks, scalar path = single elements
    constexpr int NY = VY ? (BKP * CY + NTHREADS - 1) / NTHREADS : (BKP * TCO + NTHREADS - 1) / NTHREADS;
    constexpr int NX = VX ? (BKP * CX + NTHREADS - 1) / NTHREADS : (BKP * TCI + NTHREADS - 1) / NTHREADS;
    constexpr int EY = VY ? CY : TCO, EX = VX ? CX : TCI;             // entries per pixel row
    constexpr int SY = VY ? VEC : 1, SX = VX ? VEC : 1;               // elements per entry

    __shared__ __attribute__((aligned(16))) T Ys[2][BKP * LDY];
    __shared__ __attribute__((aligned(16))) T Xs[2][BKP * LDX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_ci = (p.Ci + TCI - 1) / TCI;
    const int co0 = (blockIdx.y / tiles_ci) * TCO, ci0 = (blockIdx.y % tiles_ci) * TCI;
    const int tap = blockIdx.z, kh = tap / p.KW, kw = tap % p.KW;
    const int step0 = blockIdx.x * p.steps_per_slice;
    const int nsteps_total = (p.P + BKP - 1) / BKP;
    int nsteps = nsteps_total - step0;
    if (nsteps > p.steps_per_slice) nsteps = p.steps_per_slice;
    if (nsteps <= 0) return;
    const int HWo = p.Ho * p.Wo;

    // Per-entry state, advanced by BKP pixels per k-step with uniform increments only:
    //   dY: linear pixel index m and element offset;  X: (n, oy, ox) and the element offset of the tapped pixel.
    int ym[NY], yoff[NY];
    bool yok[NY];
#pragma unroll
    for (int j = 0; j < NY; ++j) {
        const int e = tid + j * NTHREADS, row = e / EY, c = co0 + (e % EY) * SY;
        ym[j] = step0 * BKP + row;
        yok[j] = row < BKP && c < p.Co;
        yoff[j] = ym[j] * p.ldy + c;                 // < 2^31 elements (checked on the host)
    }
    int xn[NX], xoy[NX], xox[NX], xoff[NX];
    bool xok[NX];
    const int dx_step = BKP * p.sw * p.ldx;                           // +32 output pixels along a row
    const int dx_row = (p.sh * p.Wi - p.Wo * p.sw) * p.ldx;           // wrap to the next output row
    const int dx_img = (p.Hi - p.Ho * p.sh) * p.Wi * p.ldx;           // wrap to the next image
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int e = tid + j * NTHREADS, row = e / EX, c = ci0 + (e % EX) * SX;
        const int m = step0 * BKP + row;
        const int n = m / HWo, r = m - n * HWo;
        xn[j] = n; xoy[j] = r / p.Wo; xox[j] = r - xoy[j] * p.Wo;
        xok[j] = row < BKP && c < p.Ci;
        const int iy = xoy[j] * p.sh - p.ph + kh * p.dh, ix = xox[j] * p.sw - p.pw + kw * p.dw;
        xoff[j] = ((n * p.Hi + iy) * p.Wi + ix) * p.ldx + c;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    Vec16<T> vy[VY ? NY : 1], vx[VX ? NX : 1];
    T sy[VY ? 1 : NY], sx[VX ? 1 : NX];

    auto load_regs = [&]() {
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const bool ok = yok[j] && ym[j] < p.P;
            if constexpr (VY) {
                vy[j].zero();
                if (ok) vy[j].load(dy + yoff[j]);
            } else {
                sy[j] = ok ? dy[yoff[j]] : (T)0.0f;
            }
            ym[j] += BKP;
            yoff[j] += BKP * p.ldy;
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int iy = xoy[j] * p.sh - p.ph + kh * p.dh, ix = xox[j] * p.sw - p.pw + kw * p.dw;
            const bool ok = xok[j] && xn[j] < p.N && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            if constexpr (VX) {
                vx[j].zero();
                if (ok) vx[j].load(x + xoff[j]);
            } else {
                sx[j] = ok ? x[xoff[j]] : (T)0.0f;
            }
            xox[j] += BKP;
            xoff[j] += dx_step;
            while (xox[j] >= p.Wo) { xox[j] -= p.Wo; ++xoy[j]; xoff[j] += dx_row; }
            while (xoy[j] >= p.Ho) { xoy[j] -= p.Ho; ++xn[j]; xoff[j] += dx_img; }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NY; ++j) {
            const int e = tid + j * NTHREADS, row = e / EY, ch = e % EY;
            if (row < BKP) {
                if constexpr (VY) vy[j].store(&Ys[buf][row * LDY + ch * VEC]);
                else Ys[buf][row * LDY + ch] = sy[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int e = tid + j * NTHREADS, row = e / EX, ch = e % EX;
            if (row < BKP) {
                if constexpr (VX) vx[j].store(&Xs[buf][row * LDX + ch * VEC]);
                else Xs[buf][row * LDX + ch] = sx[j];
            }
        }
    };

    const int wr = wave / WCI, wc = wave % WCI;
    load_regs();
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_regs();
        WMma<T>::template step<MT, NT>(Ys[buf], Xs[buf], LDY, LDX, wr * WTCO, wc * WTCI, lane, acc);
        if (s + 1 < nsteps) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- accumulate the tile: dW[co][tap][ci], row co = 4*(lane>>4)+r, col ci = lane&15 --------------
    const int T_ = p.KH * p.KW;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int ci = ci0 + wc * WTCI + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wr * WTCO + 16 * i + 4 * (lane >> 4) + r;
                if (co < p.Co && ci < p.Ci) atomicAdd(&dw[((size_t)co * T_ + tap) * p.Ci + ci], acc[i][j][r]);
            }
        }
}

template <typename T, int TCO, int TCI, int WCO, int NTHREADS = 256>
void launch_cfg(const WgradParams &p0, const void *x, const void *dy, float *dw, hipStream_t st) {
    WgradParams p = p0;
    const int tiles = cdiv(p.Co, TCO) * cdiv(p.Ci, TCI);
    const int taps = p.KH * p.KW;
    const int nsteps = cdiv(p.P, BKP);
    // split-K over pixels.  Two limits: enough workgroups to fill the chip (~6 per CU), and the f32 atomic
    // traffic of the final accumulation -- slices x |dW| bytes at the chip-wide ~1.3 TB/s atomic rate
    // (MI355X_MICROARCH.md); measured: 42 slices on a 256->256 3x3 layer spent ~50 % of the kernel in atomics.
    int slices = cdiv(1536, tiles * taps);
    const double dw_bytes = 4.0 * p.Co * p.Ci * taps;
    const int atomic_cap = (int)(24.0e6 / dw_bytes);              // <= ~24 MB of atomics (~20 us) per launch
    if (slices > atomic_cap) slices = atomic_cap;
    if (slices > cdiv(nsteps, 4)) slices = cdiv(nsteps, 4);
    if (slices < 1) slices = 1;
    p.steps_per_slice = cdiv(nsteps, slices);
    slices = cdiv(nsteps, p.steps_per_slice);
    dim3 grid(slices, tiles, taps);
    constexpr int VEC = Elem<T>::VEC;
    const bool vy = (p.Co % VEC == 0) && (p.ldy % VEC == 0), vx = (p.Ci % VEC == 0) && (p.ldx % VEC == 0);
    const T *xp = (const T *)x, *yp = (const T *)dy;
    rdfc_note_kernel("wgrad_kernel<%s, %d, %d, %d, %s, %s, %d>", tname<T>(), TCO, TCI, WCO, vy ? "true" : "false", vx ? "true" : "false", NTHREADS);
    if (vy && vx) hipLaunchKernelGGL((wgrad_kernel<T, TCO, TCI, WCO, true, true, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, xp, yp, dw);
    else if (vy) hipLaunchKernelGGL((wgrad_kernel<T, TCO, TCI, WCO, true, false, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, xp, yp, dw);
    else if (vx) hipLaunchKernelGGL((wgrad_kernel<T, TCO, TCI, WCO, false, true, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, xp, yp, dw);
    else hipLaunchKernelGGL((wgrad_kernel<T, TCO, TCI, WCO, false, false, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, xp, yp, dw);
}

template <typename T>
void launch_T(const WgradParams &p, const void *x, const void *dy, float *dw, hipStream_t st) {
    if (p.Co <= 16) launch_cfg<T, 16, 64, 1>(p, x, dy, dw, st);
    else if (p.Ci <= 16) launch_cfg<T, 64, 16, 4>(p, x, dy, dw, st);
    else if (p.Co >= 128 && p.Ci >= 128) launch_cfg<T, 128, 128, 2, (sizeof(T) == 2 ? 512 : 256)>(p, x, dy, dw, st);
    else launch_cfg<T, 64, 64, 2>(p, x, dy, dw, st);
}

}  // namespace

int rdfc_try_wgrad9(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st);   // conv_wgrad9.hip
int rdfc_try_wgrad_glds(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st, int reflect = 0);   // conv_wgrad_glds.hip
int rdfc_try_wgrad_gather_glds(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st);   // conv_wgrad_gather_glds.hip
int rdfc_try_wgrad_thin(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st);   // conv_wgrad_thin.hip
int rdfc_try_wgrad_narrow(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st); // conv_narrow.hip

extern "C" int rdfc_conv2d_reflect_supported(const rdfc_conv_desc *d);
// weight gradient of a conv whose input is reflection-padded by ph = pw (x is the UNPADDED map): see rdfc_conv2d_forward_reflect
extern "C" int rdfc_conv2d_wgrad_reflect(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw_packed, void *stream) {
    RDFC_REQUIRE(d != nullptr && rdfc_conv2d_reflect_supported(d), "rdfc_conv2d_wgrad_reflect: shape not covered (ask rdfc_conv2d_reflect_supported)");
    RDFC_REQUIRE(d->ldx >= d->Ci && d->ldy >= d->Co, "rdfc_conv2d_wgrad_reflect: ld smaller than channel count");
    RDFC_REQUIRE((double)d->N * d->Hi * d->Wi * d->ldx < 2147483647.0 && (double)d->N * d->Ho * d->Wo * d->ldy < 2147483647.0,
                 "rdfc_conv2d_wgrad_reflect: maps of 2^31 elements or more need 64-bit offsets (not built)");
    RDFC_REQUIRE(rdfc_try_wgrad_glds(d, x, dy, dw_packed, (hipStream_t)stream, 1), "rdfc_conv2d_wgrad_reflect: launch refused");
    RDFC_LAUNCH_CHECK("rdfc_conv2d_wgrad_reflect");
    return 0;
}

extern "C" int rdfc_conv2d_wgrad(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw_packed, void *stream) {
    RDFC_REQUIRE(d != nullptr, "rdfc_conv2d_wgrad: null descriptor");
    RDFC_REQUIRE(rdfc_dtype_ok(d->dtype), "rdfc_conv2d_wgrad: bad dtype %d", d->dtype);
    RDFC_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ci > 0 && d->Ho > 0 && d->Wo > 0 && d->Co > 0,
                 "rdfc_conv2d_wgrad: non-positive dimension");
    RDFC_REQUIRE(d->ldx >= d->Ci && d->ldy >= d->Co, "rdfc_conv2d_wgrad: ld smaller than channel count");
    RDFC_REQUIRE(d->KH * d->KW <= 65535, "rdfc_conv2d_wgrad: too many taps");
    RDFC_REQUIRE((double)d->N * d->Hi * d->Wi * d->ldx < 2147483647.0 && (double)d->N * d->Ho * d->Wo * d->ldy < 2147483647.0,
                 "rdfc_conv2d_wgrad: maps of 2^31 elements or more need 64-bit offsets (not built)");
    WgradParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->KH, d->KW, d->sh, d->sw, d->ph, d->pw, d->dh, d->dw,
                  d->ldx, d->ldy, d->N * d->Ho * d->Wo, 0};
    hipStream_t st = (hipStream_t)stream;
    constexpr bool w9 = true;
    if (rdfc_try_wgrad_thin(d, x, dy, dw_packed, st)) {
    } else if (rdfc_try_wgrad_narrow(d, x, dy, dw_packed, st)) {
    } else if (rdfc_try_wgrad_glds(d, x, dy, dw_packed, st)) {
    } else if (rdfc_try_wgrad_gather_glds(d, x, dy, dw_packed, st)) {
    } else if (w9 && rdfc_try_wgrad9(d, x, dy, dw_packed, st)) {
    } else if (d->dtype == RDFC_BF16) launch_T<bf16_t>(p, x, dy, dw_packed, st);
    else if (d->dtype == RDFC_F16) launch_T<f16_t>(p, x, dy, dw_packed, st);
    else launch_T<float>(p, x, dy, dw_packed, st);
    RDFC_LAUNCH_CHECK("rdfc_conv2d_wgrad");
    return 0;
}
