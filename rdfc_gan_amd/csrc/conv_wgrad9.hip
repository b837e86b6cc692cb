// conv_wgrad9.hip -- weight gradient of 3x3 / stride-1 convolutions, all nine taps per workgroup.
//
//   dW[co][tap][ci] += sum_m dY[m][co] * X[m + tap][ci]
//
// The per-tap kernel (conv_wgrad.hip) launches one workgroup per (tile, tap): every tap re-reads the same
// dY tile and a shifted copy of X, and a wave gets only 4-8 MFMAs between two barriers.  Here a k-step is a
// segment of 32 consecutive output pixels of one image row: the workgroup stages the dY tile [32 px][64 co]
// once and the X halo patch [3 rows x 34 px][64 ci] once, and each wave then runs 9 taps x (2x2) MFMAs = 36
// MFMAs per barrier against 2 + 9*2 fragment reads.  Fragments need 8 consecutive k (= pixels) per lane while
// both tiles are pixel-major: the gfx950 transposing LDS read ds_read_b64_tr_b16 delivers them (bf16); the
// f32 MFMA takes one k per lane and reads plain dwords.  Split-K over segments, f32 atomics at the end
// (slices capped by the chip-wide atomic rate, see conv_wgrad.hip).
#include "common.h"

namespace {

struct W9Params {
    int N, Hi, Wi, Ci, Ho, Wo, Co, pad;
    int ldx, ldy;
    int segs_per_row, total_segs, segs_per_slice;
};

constexpr int SEG = 32;            // output pixels per k-step
constexpr int PW = SEG + 2;        // patch width
constexpr int NTH = 256;
constexpr int TC = 64;             // tile: 64 co x 64 ci

__device__ inline bf16x8 tr_frag(const bf16_t *img, int ld, int row0, int cbase, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const bf16_t *a0 = img + (row0 + 8 * g + q) * ld + cbase + 4 * pp;
    const bf16_t *a1 = a0 + 4 * ld;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <typename T> struct F9 {           // 2-byte elements (bf16 | f16)
    typedef bf16x8 frag;
    static constexpr int KSUB = 1;
    __device__ static frag ld(const T *img, int ld_, int row0, int cbase, int lane, int) {
        return tr_frag(reinterpret_cast<const bf16_t *>(img), ld_, row0, cbase, lane);
    }
    __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return mma16<DtOf<T>::value>(a, b, c); }
};
template <> struct F9<float> {
    typedef float frag;
    static constexpr int KSUB = SEG / 4;
    __device__ static frag ld(const float *img, int ld_, int row0, int cbase, int lane, int kk) {
        return img[(row0 + 4 * kk + (lane >> 4)) * ld_ + cbase + (lane & 15)];
    }
    __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

template <typename T>
__global__ __launch_bounds__(NTH, sizeof(T) == 2 ? 2 : 1) void wgrad9_kernel(W9Params p, const T *__restrict__ x,
                                                                             const T *__restrict__ dy, float *__restrict__ dw) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int LD = TC + VEC;                       // padded tile row (elements)
    constexpr int CPR = TC / VEC;                      // 16-byte chunks per pixel row
    constexpr int NYV = (SEG * CPR + NTH - 1) / NTH;
    constexpr int NXV = (3 * PW * CPR + NTH - 1) / NTH;

    __shared__ __attribute__((aligned(16))) T Ys[2][SEG * LD];
    __shared__ __attribute__((aligned(16))) T Xs[2][3 * PW * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_ci = (p.Ci + TC - 1) / TC;
    const int co0 = (blockIdx.y / tiles_ci) * TC, ci0 = (blockIdx.y % tiles_ci) * TC;
    const int seg0 = blockIdx.x * p.segs_per_slice;
    int nseg = p.total_segs - seg0;
    if (nseg > p.segs_per_slice) nseg = p.segs_per_slice;
    if (nseg <= 0) return;

    // uniform segment cursor (n, oy, sx), advanced one segment per k-step
    int cn = seg0 / (p.Ho * p.segs_per_row);
    int rem = seg0 - cn * p.Ho * p.segs_per_row;
    int coy = rem / p.segs_per_row, csx = rem - coy * p.segs_per_row;

    // thread-constant parts of the addresses
    int yrow[NYV], ycol[NYV], xr[NXV], xc[NXV], xcol[NXV], xrow[NXV];
#pragma unroll
    for (int j = 0; j < NYV; ++j) {
        const int e = tid + j * NTH;
        yrow[j] = e / CPR;
        ycol[j] = (e % CPR) * VEC;
    }
#pragma unroll
    for (int j = 0; j < NXV; ++j) {
        const int e = tid + j * NTH, pr = e / CPR;
        xrow[j] = pr;                                  // patch pixel index (row-major, PW wide); >= 3*PW: unused slot
        xr[j] = pr / PW;
        xc[j] = pr - xr[j] * PW;
        xcol[j] = (e % CPR) * VEC;
    }

    f32x4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};

    Vec16<T> vy[NYV], vx[NXV];
    auto load_regs = [&]() {
        const int ox0 = csx * SEG;
        const bool segok = cn < p.N;
        const int ybase = ((cn * p.Ho + coy) * p.Wo + ox0) * p.ldy + co0;
        const int xbase = ((cn * p.Hi + coy - p.pad) * p.Wi + ox0 - p.pad) * p.ldx + ci0;
#pragma unroll
        for (int j = 0; j < NYV; ++j) {
            vy[j].zero();
            if (segok && yrow[j] < SEG && ox0 + yrow[j] < p.Wo && co0 + ycol[j] < p.Co)
                vy[j].load(dy + ybase + yrow[j] * p.ldy + ycol[j]);
        }
#pragma unroll
        for (int j = 0; j < NXV; ++j) {
            vx[j].zero();
            const int iy = coy - p.pad + xr[j], ix = ox0 - p.pad + xc[j];
            if (segok && xrow[j] < 3 * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && ci0 + xcol[j] < p.Ci)
                vx[j].load(x + xbase + (xr[j] * p.Wi + xc[j]) * p.ldx + xcol[j]);
        }
        if (++csx == p.segs_per_row) {
            csx = 0;
            if (++coy == p.Ho) { coy = 0; ++cn; }
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NYV; ++j)
            if (yrow[j] < SEG) vy[j].store(&Ys[buf][yrow[j] * LD + ycol[j]]);
#pragma unroll
        for (int j = 0; j < NXV; ++j)
            if (xrow[j] < 3 * PW) vx[j].store(&Xs[buf][xrow[j] * LD + xcol[j]]);
    };

    const int wr = wave >> 1, wc = wave & 1;           // 2 x 2 waves, 32 co x 32 ci each
    load_regs();
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nseg; ++s) {
        const int buf = s & 1;
        if (s + 1 < nseg) load_regs();
#pragma unroll
        for (int kk = 0; kk < F9<T>::KSUB; ++kk) {
            typename F9<T>::frag a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = F9<T>::ld(Ys[buf], LD, 0, wr * 32 + 16 * i, lane, kk);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                typename F9<T>::frag b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = F9<T>::ld(Xs[buf], LD, (t / 3) * PW + (t % 3), wc * 32 + 16 * j, lane, kk);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[t][i][j] = F9<T>::mma(a[i], b[j], acc[t][i][j]);
            }
        }
        if (s + 1 < nseg) store_lds(buf ^ 1);
        __syncthreads();
    }

    // accumulate: row co = 4*(lane>>4)+r, col ci = lane&15
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + wc * 32 + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wr * 32 + 16 * i + 4 * (lane >> 4) + r;
                    if (co < p.Co && ci < p.Ci) atomicAdd(&dw[((size_t)co * 9 + t) * p.Ci + ci], acc[t][i][j][r]);
                }
            }
}

}  // namespace

// Called by rdfc_conv2d_wgrad (conv_wgrad.hip) when the shape qualifies; returns 1 if it took the launch.
int rdfc_try_wgrad9(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st) {
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 && d->ph == d->pw && d->ph <= 1)) return 0;
    if (d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return 0;
    const int vec = rdfc_vec(d->dtype);
    if (d->Co % vec || d->Ci % vec || d->ldx % vec || d->ldy % vec || d->Co < 32 || d->Ci < 32) return 0;
    // measured (tools/ab_env2.sh, one device): +8..45 % over the per-tap kernel's 64x64 tile on the full-resolution
    // 64/32-channel layers; the per-tap 128x128 8-wave tile stays faster for >= 128 x >= 128 channels
    if (d->Co >= 128 && d->Ci >= 128) return 0;
    const int spr = cdiv(d->Wo, SEG);
    if ((double)d->Wo / (spr * SEG) < 0.55) return 0;              // too much padding per row segment
    W9Params p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->ph, d->ldx, d->ldy, spr, d->N * d->Ho * spr, 0};
    const int tiles = cdiv(d->Co, TC) * cdiv(d->Ci, TC);
    int slices = cdiv(512, tiles);
    const int atomic_cap = (int)(24.0e6 / (4.0 * d->Co * d->Ci * 9));
    if (slices > atomic_cap) slices = atomic_cap;
    if (slices > cdiv(p.total_segs, 8)) slices = cdiv(p.total_segs, 8);
    if (slices < 1) slices = 1;
    p.segs_per_slice = cdiv(p.total_segs, slices);
    slices = cdiv(p.total_segs, p.segs_per_slice);
    dim3 grid(slices, tiles);
    rdfc_note_kernel("wgrad9_kernel<%s>", d->dtype == RDFC_BF16 ? "__bf16" : (d->dtype == RDFC_F16 ? "_Float16" : "float"));
    if (d->dtype == RDFC_BF16) hipLaunchKernelGGL(wgrad9_kernel<bf16_t>, grid, dim3(NTH), 0, st, p, (const bf16_t *)x, (const bf16_t *)dy, dw);
    else if (d->dtype == RDFC_F16) hipLaunchKernelGGL(wgrad9_kernel<f16_t>, grid, dim3(NTH), 0, st, p, (const f16_t *)x, (const f16_t *)dy, dw);
    else hipLaunchKernelGGL(wgrad9_kernel<float>, grid, dim3(NTH), 0, st, p, (const float *)x, (const float *)dy, dw);
    return 1;
}
