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

namespace {

struct WgradParams {
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int KH, KW, sh, sw, ph, pw, dh, dw;
    int ldx, ldy;
    int P;          // N*Ho*Wo
    int steps_per_slice;   // k-steps (of 32 pixels) per split-K slice
};

constexpr int BKP = 32;      // pixels per k-step
constexpr int NTHREADS = 256;

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

template <typename T> struct WMma;
template <> struct WMma<bf16_t> {
    template <int MT, int NT>
    __device__ static void step(const bf16_t *Ys, const bf16_t *Xs, int ldy, int ldx, int arow, int bcol, int lane,
                                f32x4 (&acc)[MT][NT]) {
        bf16x8 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = tr_frag(Ys, ldy, arow + 16 * i, lane);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = tr_frag(Xs, ldx, bcol + 16 * j, lane);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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

template <typename T, int TCO, int TCI, int WCO>
__global__ __launch_bounds__(NTHREADS) void wgrad_kernel(WgradParams p, const T *__restrict__ x,
                                                         const T *__restrict__ dy, float *__restrict__ dw) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int WCI = 4 / WCO;
    constexpr int WTCO = TCO / WCO, WTCI = TCI / WCI;
    constexpr int MT = WTCO / 16, NT = WTCI / 16;
    constexpr int LDY = TCO + VEC, LDX = TCI + VEC;
    constexpr int CY = TCO / VEC, CX = TCI / VEC;                     // 16-byte chunks per pixel row
    constexpr int NYV = (BKP * CY + NTHREADS - 1) / NTHREADS, NXV = (BKP * CX + NTHREADS - 1) / NTHREADS;
    constexpr int NYS = (BKP * TCO + NTHREADS - 1) / NTHREADS, NXS = (BKP * TCI + NTHREADS - 1) / NTHREADS;

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

    const bool vecY = (p.Co % VEC == 0) && (p.ldy % VEC == 0);
    const bool vecX = (p.Ci % VEC == 0) && (p.ldx % VEC == 0);
    const int HWo = p.Ho * p.Wo;

    // incremental pixel coordinates of the X rows this thread loads (vector path: NXV rows; scalar: NXS)
    constexpr int NXC = NXV > NXS ? NXV : NXS;
    int xn[NXC], xoy[NXC], xox[NXC];
    {
        const int cnt = vecX ? NXV : NXS;
        for (int j = 0; j < NXC; ++j) {
            xn[j] = xoy[j] = xox[j] = 0;
            if (j < cnt) {
                const int e = tid + j * NTHREADS;
                const int row = vecX ? e / CX : e / TCI;
                const int m = step0 * BKP + row;
                const int n = m / HWo, r = m - n * HWo;
                xn[j] = n; xoy[j] = r / p.Wo; xox[j] = r - (r / p.Wo) * p.Wo;
            }
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    Vec16<T> vy[NYV], vx[NXV];
    T sy[NYS], sx[NXS];

    auto load_regs = [&](int s) {
        const int mbase = (step0 + s) * BKP;
        if (vecY) {
#pragma unroll
            for (int j = 0; j < NYV; ++j) {
                const int e = tid + j * NTHREADS, row = e / CY, ch = e % CY;
                vy[j].zero();
                const int m = mbase + row, c = co0 + ch * VEC;
                if (row < BKP && m < p.P && c < p.Co) vy[j].load(dy + (size_t)m * p.ldy + c);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NYS; ++j) {
                const int e = tid + j * NTHREADS, row = e / TCO, c = co0 + e % TCO;
                const int m = mbase + row;
                sy[j] = (row < BKP && m < p.P && c < p.Co) ? dy[(size_t)m * p.ldy + c] : (T)0.0f;
            }
        }
        if (vecX) {
#pragma unroll
            for (int j = 0; j < NXV; ++j) {
                const int e = tid + j * NTHREADS, row = e / CX, ch = e % CX;
                vx[j].zero();
                const int c = ci0 + ch * VEC;
                const int iy = xoy[j] * p.sh - p.ph + kh * p.dh, ix = xox[j] * p.sw - p.pw + kw * p.dw;
                if (row < BKP && xn[j] < p.N && c < p.Ci && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                    vx[j].load(x + ((size_t)(xn[j] * p.Hi + iy) * p.Wi + ix) * p.ldx + c);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NXS; ++j) {
                const int e = tid + j * NTHREADS, row = e / TCI, c = ci0 + e % TCI;
                const int iy = xoy[j] * p.sh - p.ph + kh * p.dh, ix = xox[j] * p.sw - p.pw + kw * p.dw;
                sx[j] = (row < BKP && xn[j] < p.N && c < p.Ci && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                            ? x[((size_t)(xn[j] * p.Hi + iy) * p.Wi + ix) * p.ldx + c] : (T)0.0f;
            }
        }
        // advance the tracked pixels by one k-step
#pragma unroll
        for (int j = 0; j < NXC; ++j) {
            xox[j] += BKP;
            while (xox[j] >= p.Wo) { xox[j] -= p.Wo; ++xoy[j]; }
            while (xoy[j] >= p.Ho) { xoy[j] -= p.Ho; ++xn[j]; }
        }
    };
    auto store_lds = [&](int buf) {
        if (vecY) {
#pragma unroll
            for (int j = 0; j < NYV; ++j) {
                const int e = tid + j * NTHREADS, row = e / CY, ch = e % CY;
                if (row < BKP) vy[j].store(&Ys[buf][row * LDY + ch * VEC]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NYS; ++j) {
                const int e = tid + j * NTHREADS, row = e / TCO;
                if (row < BKP) Ys[buf][row * LDY + e % TCO] = sy[j];
            }
        }
        if (vecX) {
#pragma unroll
            for (int j = 0; j < NXV; ++j) {
                const int e = tid + j * NTHREADS, row = e / CX, ch = e % CX;
                if (row < BKP) vx[j].store(&Xs[buf][row * LDX + ch * VEC]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NXS; ++j) {
                const int e = tid + j * NTHREADS, row = e / TCI;
                if (row < BKP) Xs[buf][row * LDX + e % TCI] = sx[j];
            }
        }
    };

    const int wr = wave / WCI, wc = wave % WCI;
    load_regs(0);
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) load_regs(s + 1);
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

template <typename T, int TCO, int TCI, int WCO>
void launch_cfg(const WgradParams &p0, const void *x, const void *dy, float *dw, hipStream_t st) {
    WgradParams p = p0;
    const int tiles = cdiv(p.Co, TCO) * cdiv(p.Ci, TCI);
    const int taps = p.KH * p.KW;
    const int nsteps = cdiv(p.P, BKP);
    int slices = cdiv(1536, tiles * taps);            // aim for ~6 workgroups per CU in flight
    if (slices > cdiv(nsteps, 4)) slices = cdiv(nsteps, 4);
    if (slices < 1) slices = 1;
    p.steps_per_slice = cdiv(nsteps, slices);
    slices = cdiv(nsteps, p.steps_per_slice);
    dim3 grid(slices, tiles, taps);
    hipLaunchKernelGGL((wgrad_kernel<T, TCO, TCI, WCO>), grid, dim3(NTHREADS), 0, st, p, (const T *)x, (const T *)dy, dw);
}

template <typename T>
void launch_T(const WgradParams &p, const void *x, const void *dy, float *dw, hipStream_t st) {
    if (p.Co <= 16) launch_cfg<T, 16, 64, 1>(p, x, dy, dw, st);
    else if (p.Ci <= 16) launch_cfg<T, 64, 16, 4>(p, x, dy, dw, st);
    else if (p.Co >= 128 && p.Ci >= 128) launch_cfg<T, 128, 128, 2>(p, x, dy, dw, st);
    else launch_cfg<T, 64, 64, 2>(p, x, dy, dw, st);
}

}  // namespace

extern "C" int rdfc_conv2d_wgrad(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw_packed, void *stream) {
    RDFC_REQUIRE(d != nullptr, "rdfc_conv2d_wgrad: null descriptor");
    RDFC_REQUIRE(d->dtype == RDFC_F32 || d->dtype == RDFC_BF16, "rdfc_conv2d_wgrad: bad dtype %d", d->dtype);
    RDFC_REQUIRE(d->N > 0 && d->Hi > 0 && d->Wi > 0 && d->Ci > 0 && d->Ho > 0 && d->Wo > 0 && d->Co > 0,
                 "rdfc_conv2d_wgrad: non-positive dimension");
    RDFC_REQUIRE(d->ldx >= d->Ci && d->ldy >= d->Co, "rdfc_conv2d_wgrad: ld smaller than channel count");
    RDFC_REQUIRE(d->KH * d->KW <= 65535, "rdfc_conv2d_wgrad: too many taps");
    WgradParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->KH, d->KW, d->sh, d->sw, d->ph, d->pw, d->dh, d->dw,
                  d->ldx, d->ldy, d->N * d->Ho * d->Wo, 0};
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype == RDFC_BF16) launch_T<bf16_t>(p, x, dy, dw_packed, st);
    else launch_T<float>(p, x, dy, dw_packed, st);
    RDFC_LAUNCH_CHECK("rdfc_conv2d_wgrad");
    return 0;
}
