// conv_wgrad_gather_glds.hip -- weight gradient for every stride / kernel size (bf16), staged by LDS-DMA.
//
//   dW[co][ky][kx][ci] += sum over output pixels m of dY[m][co] * X[m * s + (ky, kx) * d - pad][ci]
//
// conv_wgrad_glds.hip's pipeline for the layers it does not take (stride 2, 4x4, 1x1): a workgroup owns one kernel row ky,
// a 128 (64) co x 64 ci tile and a slice of the output pixels; a k-step is 32 consecutive output pixels of one row.
// dY [32 px][TCO] is staged once per k-step and shared by the KW taps of the row; with a stride the taps no longer
// read shifted views of one X row, so each tap gets its own gathered tile [32 px][64 ci] (a per-lane LDS-DMA source,
// recomputed per k-step: one 16-byte piece per thread and tap).  Ring of four k-step slots (three for KW = 4), one raw
// barrier and one counted vmcnt per k-step, ds_read_b64_tr_b16 fragments, source-side pair swizzle, zero page for
// out-of-image pieces: as in conv_wgrad_glds.hip.
#include "common.h"

#include <stdlib.h>

namespace {

struct WGParams {
    int N, Hi, Wi, Ci, Ho, Wo, Co;
    int KH, KW, s, pad, dil;
    int ldx, ldy;
    int segs_per_row, total_segs, segs_per_slice;
    int tiles_ci;
};

constexpr int SEG = 32, NT = 256;

__device__ inline void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// swizzle of the 32-byte channel-pair index by image row (see header)
template <int ROWB> __device__ inline int pair_swz(int row) {
    if constexpr (ROWB == 256) return (row & 3) | (((row >> 3) & 1) << 2);
    else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

// fragment: 16 channels [cbase, cbase+16) x 32 pixels (image rows row0 .. row0+31) -> 8 consecutive pixels per lane
template <int ROWB> __device__ inline bf16x8 tr_frag(const unsigned char *img, int row0, int cbase, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int r0 = row0 + 8 * g + q, r1 = r0 + 4, cp = cbase >> 4;
    const unsigned char *a0 = img + r0 * ROWB + ((cp ^ pair_swz<ROWB>(r0)) << 5) + pp * 8;
    const unsigned char *a1 = img + r1 * ROWB + ((cp ^ pair_swz<ROWB>(r1)) << 5) + pp * 8;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int TCO, int KWT, int DT = RDFC_BF16>   // KWT = taps per kernel row (1, 3 or 4)
__global__ __launch_bounds__(NT, 2) void wgrad_gather_glds_kernel(WGParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                                  float *__restrict__ dw, const bf16_t *__restrict__ zero) {
    constexpr int TCI = 64;
    constexpr int YROW = TCO * 2, XROW = TCI * 2;
    constexpr int YB = SEG * YROW, XT = SEG * XROW;             // dY tile 8 / 4 KB, one X tap tile 4 KB
    constexpr int NY = YB / (NT * 16);                          // dY DMAs per thread and k-step (2 / 1); one X DMA per tap
    constexpr int SLOT = YB + KWT * XT;
    constexpr int RING = KWT == 4 ? 3 : 4, LOOK = RING - 1;
    constexpr int WCO = TCO / 64, WCI = 4 / WCO;
    constexpr int WTCO = 64, WTCI = TCI / WCI;
    constexpr int MT = WTCO / 16, NTF = WTCI / 16;
    constexpr int PER_STEP = NY + KWT;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.y;
    const int ky = tile % p.KH;
    tile /= p.KH;
    const int ci0 = (tile % p.tiles_ci) * TCI, co0 = (tile / p.tiles_ci) * TCO;
    const int seg0 = blockIdx.x * p.segs_per_slice;
    int nseg = p.total_segs - seg0;
    if (nseg > p.segs_per_slice) nseg = p.segs_per_slice;
    if (nseg <= 0) return;

    int ypx[NY], yoff[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        const int e = k * NT + tid, px = e / (TCO / 8), slot = e % (TCO / 8);
        ypx[k] = px;
        yoff[k] = px * p.ldy + co0 + ((slot ^ (2 * pair_swz<YROW>(px))) << 3);
    }
    // X: thread -> (pixel tid / 8, 16-byte slot tid % 8) of every tap tile
    const int xpx = tid >> 3;
    const int xch = ci0 + (((tid & 7) ^ (2 * pair_swz<XROW>(xpx))) << 3);

    int cn = seg0 / (p.Ho * p.segs_per_row);
    int rem = seg0 - cn * p.Ho * p.segs_per_row;
    int coy = rem / p.segs_per_row, csx = rem - coy * p.segs_per_row;
    auto issue = [&](int s) {
        const bool live = s < nseg;
        const int ox0 = csx * SEG, iy = coy * p.s - p.pad + ky * p.dil;
        const bool rowok = live && (unsigned)iy < (unsigned)p.Hi && ox0 + xpx < p.Wo;
        const bf16_t *yb = dy + ((size_t)(cn * p.Ho + coy) * p.Wo + ox0) * p.ldy;
        const bf16_t *xrow = x + ((long)(cn * p.Hi + iy) * p.Wi) * p.ldx + xch;
        const int ixb = (ox0 + xpx) * p.s - p.pad;
        const unsigned slot = lds0 + (s % RING) * SLOT;
#pragma unroll
        for (int k = 0; k < NY; ++k) {
            const bool ok = live && ox0 + ypx[k] < p.Wo;
            glds16(ok ? yb + yoff[k] : zero, slot + (k * NT + wave * 64) * 16);
        }
#pragma unroll
        for (int t = 0; t < KWT; ++t) {
            const int ix = ixb + t * p.dil;
            const bool ok = rowok && (unsigned)ix < (unsigned)p.Wi;
            glds16(ok ? xrow + (long)ix * p.ldx : zero, slot + YB + t * XT + wave * 1024);
        }
        if (live) {
            if (++csx == p.segs_per_row) {
                csx = 0;
                if (++coy == p.Ho) { coy = 0; ++cn; }
            }
        }
    };

    const int wr = wave / WCI, wc = wave % WCI;
    f32x4 acc[KWT][MT][NTF];
#pragma unroll
    for (int t = 0; t < KWT; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};

#pragma unroll
    for (int h = 0; h < LOOK; ++h) issue(h);
    for (int s = 0; s < nseg; ++s) {
        wait_vm_and_reads<(LOOK - 1) * PER_STEP>();
        __builtin_amdgcn_s_barrier();
        issue(s + LOOK);
        const unsigned char *Ys = smem + (s % RING) * SLOT, *Xs = Ys + YB;
        bf16x8 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = tr_frag<YROW>(Ys, 0, wr * WTCO + 16 * i, lane);
#pragma unroll
        for (int t = 0; t < KWT; ++t) {
            bf16x8 b[NTF];
#pragma unroll
            for (int j = 0; j < NTF; ++j) b[j] = tr_frag<XROW>(Xs + t * XT, 0, wc * WTCI + 16 * j, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTF; ++j) acc[t][i][j] = mma16<DT>(a[i], b[j], acc[t][i][j]);
        }
    }
    wait_vm<0>();

    const int T = p.KH * p.KW;
#pragma unroll
    for (int t = 0; t < KWT; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) {
                const int ci = ci0 + wc * WTCI + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wr * WTCO + 16 * i + 4 * (lane >> 4) + r;
                    if (co < p.Co && ci < p.Ci) atomicAdd(&dw[((size_t)co * T + ky * p.KW + t) * p.Ci + ci], acc[t][i][j][r]);
                }
            }
}

const bf16_t *zero_page() {
    static void *pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (pages[dev] == nullptr) {
        void *ptr = nullptr;
        if (hipMalloc(&ptr, 256) != hipSuccess) return nullptr;
        if (hipMemset(ptr, 0, 256) != hipSuccess) return nullptr;
        pages[dev] = ptr;
    }
    return (const bf16_t *)pages[dev];
}

template <int TCO, int KWT, int DT = RDFC_BF16>
int launch(WGParams p, const void *x, const void *dy, float *dw, const bf16_t *zero, hipStream_t st) {
    constexpr int SMEM = (KWT == 4 ? 3 : 4) * (SEG * TCO * 2 + KWT * SEG * 128);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)wgrad_gather_glds_kernel<TCO, KWT, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    p.tiles_ci = cdiv(p.Ci, 64);
    const int tiles = cdiv(p.Co, TCO) * p.tiles_ci * p.KH;
    constexpr double cap_mb = 48.0;
    int slices = 512 / tiles;                                  // one round of workgroups (see conv_wgrad_glds.hip)
    const int atomic_cap = (int)(cap_mb * 1.0e6 / (4.0 * p.Co * p.Ci * p.KH * p.KW));
    if (slices > atomic_cap) slices = atomic_cap;
    if (slices > cdiv(p.total_segs, 8)) slices = cdiv(p.total_segs, 8);
    if (slices < 1) slices = 1;
    p.segs_per_slice = cdiv(p.total_segs, slices);
    slices = cdiv(p.total_segs, p.segs_per_slice);
    rdfc_note_kernel("wgrad_gather_glds_kernel<%d, %d, %d>", TCO, KWT, DT);
    hipLaunchKernelGGL((wgrad_gather_glds_kernel<TCO, KWT, DT>), dim3(slices, tiles), dim3(NT), SMEM, st, p, (const bf16_t *)x,
                       (const bf16_t *)dy, dw, zero);
    return 1;
}

template <int TCO, int DT = RDFC_BF16>
int launch_kw(const WGParams &p, const void *x, const void *dy, float *dw, const bf16_t *zero, hipStream_t st) {
    if (p.KW == 1) return launch<TCO, 1, DT>(p, x, dy, dw, zero, st);
    if (p.KW == 3) return launch<TCO, 3, DT>(p, x, dy, dw, zero, st);
    if (p.KW == 4) return launch<TCO, 4, DT>(p, x, dy, dw, zero, st);
    return 0;
}

}  // namespace

// Called by rdfc_conv2d_wgrad (conv_wgrad.hip) after the stride-1 3x3 kernel; returns 1 if it took the launch.
int rdfc_try_wgrad_gather_glds(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st) {
    if (!rdfc_is16(d->dtype)) return 0;
    if (d->sh != d->sw || d->dh != d->dw || d->ph != d->pw || d->KH != d->KW) return 0;
    if (d->Co % 64 || d->Ci % 64 || d->ldx % 8 || d->ldy % 8) return 0;
    const int spr = cdiv(d->Wo, SEG);
    if ((double)d->Wo / (spr * SEG) < 0.55) return 0;              // too much padding per row segment
    // A/B on one device (tools/wgrad2_bench.sh): wins on wide maps with <= 3 taps per kernel row (3x3/s2 64->128 @ 228x304:
    // 62 vs 82 us, 64->192: 70 vs 108 us, 1x1 192->384 @ 114x152: 57 vs 74 us); on the narrower maps and the 4x4 layers the
    // register-staged kernel's flat pixel tiles win (4x4/s1 256->512 @ 27x37: 124 vs 90 us)
    if (d->Wo < 96 || d->KW > 3) return 0;
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    WGParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->KH, d->KW, d->sh, d->ph, d->dh, d->ldx, d->ldy, spr,
               d->N * d->Ho * spr, 0, 0};
    if (d->Co % 128 == 0 && d->Co >= 256) return d->dtype == RDFC_F16 ? launch_kw<128, RDFC_F16>(p, x, dy, dw, zero, st) : launch_kw<128>(p, x, dy, dw, zero, st);
    return d->dtype == RDFC_F16 ? launch_kw<64, RDFC_F16>(p, x, dy, dw, zero, st) : launch_kw<64>(p, x, dy, dw, zero, st);
}
