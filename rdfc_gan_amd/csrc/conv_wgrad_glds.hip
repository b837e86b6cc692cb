// conv_wgrad_glds.hip -- weight gradient of 3x3 / stride-1 convolutions (bf16), staged by LDS-DMA.
//
//   dW[co][ky][kx][ci] += sum over pixels m of dY[m][co] * X[m + (ky, kx) - pad][ci]
//
// GEMM view: M = co, N = (kx, ci), K = pixels.  A workgroup owns one kernel row ky, a 128 (or 64) co x 64 ci tile
// and a contiguous slice of the pixel range (split-K, f32 atomics at the end).  A k-step is a segment of 32
// consecutive output pixels of one image row: dY [32 px][TCO] and the X row segment [34 px][64 ci] serve all three kx
// (tap kx reads X rows kx .. kx+31), so dY is staged once per three taps instead of once per tap as in
// conv_wgrad.hip.  Pipeline as in conv_glds.hip: global_load_lds_dwordx4 into a ring of four k-step slots filled
// three steps ahead, one raw barrier per k-step, counted s_waitcnt vmcnt(N) -- every wave issues the same number
// of DMAs per step (out-of-image pieces read a zero page).  Both images are pixel-major, the MFMA wants 8
// consecutive k (pixels) per lane: fragments come from ds_read_b64_tr_b16.  LDS-DMA destinations are lane-linear,
// so the bank swizzle is applied to the SOURCE chunk index: 32-byte channel pairs are XORed with row bits such that
// the 8 rows a 32-lane group touches fall on 8 distinct bank octets (brute-forced: conflict-free for every kx).
#include "common.h"

#include <stdlib.h>

namespace {

struct WGParams {
    int N, Hi, Wi, Ci, Ho, Wo, Co, pad;
    int ldx, ldy;
    int segs_per_row, total_segs, segs_per_slice;
    int tiles, xcd_group;
    int tiles_ci;
    int reflect;                 // X pixels outside the image are their mirror images (reflection padding folded into the loader)
    int flat;                    // k-steps walk a flat pixel index over rows padded to Wo + 2 instead of 32-pixel segments of one row:
                                 // a 76-wide row fills 2.44 k-steps instead of 3, a 38-wide one 1.25 instead of 2 (NKY = 1 only)
};

constexpr int SEG = 32, XPX = SEG + 2;

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

// k-step ring (slots of dY + X rows filled WG_RING-1 steps ahead)
#ifndef RDFC_WGRAD_GLDS_RING
#define RDFC_WGRAD_GLDS_RING 4
#endif
constexpr int WG_RING = RDFC_WGRAD_GLDS_RING;

// TCI = 128 (8 waves, 128 co x 128 ci; experiment, see rdfc_try_wgrad_glds): per k-step the workgroup stages 8 KB of dY + 8.7 KB of X
// for 192 MFMAs instead of 8 + 4.4 KB for 96.
template <int TCO, int NKY, int DT = RDFC_BF16, int TCI = 64>   // NKY = 1: one kernel row (3 taps) per workgroup; NKY = 3: all nine taps (64 x 64 tiles only)
__global__ __launch_bounds__(256 * (TCI / 64), 2) void wgrad_glds_kernel(WGParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                           float *__restrict__ dw, const bf16_t *__restrict__ zero) {
    constexpr int NT = 256 * (TCI / 64);
    constexpr int YROW = TCO * 2, XROW = TCI * 2;               // bytes per pixel row of the two images
    constexpr int YB = SEG * YROW;                              // 8 KB / 4 KB
    constexpr int NY = YB / (NT * 16);                          // dY DMAs per thread and k-step (2 / 1)
    constexpr int NX = (NKY * XPX * (TCI / 8) + NT - 1) / NT;   // X DMAs per thread and k-step: 272 (816) pieces -> 2 (4) rounds
    constexpr int XB = NX * NT * 16;                            // 8 KB (16 KB) region, 4352 B (13056 B) used
    constexpr int SLOT = YB + XB;
    constexpr int WCO = NKY == 3 ? 2 : TCO / 64, WCI = (NT / 64) / WCO; // waves along co / ci
    constexpr int WTCO = TCO / WCO, WTCI = TCI / WCI;           // wave tile: 64 co x (32 | 16) ci, or 32 x 32 with nine taps
    constexpr int MT = WTCO / 16, NTF = WTCI / 16, NTAP = 3 * NKY;
    static_assert(NKY == 1 || TCO == 64, "the nine-tap variant keeps 9 x 2 x 2 accumulators per wave");
    static_assert(TCI == 64 || (TCO == 128 && NKY == 1), "the 128-ci tile is the 8-wave 128 x 128 variant");
    constexpr int PER_STEP = NY + NX;                           // DMAs per wave and k-step
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2.  All the tiles of one pixel slice (the
    // kernel rows, ci tiles and co tiles that re-read the same dY / X pixels) are numbered onto ONE XCD, eight slices
    // side by side: id = (slice group * tiles + tile) * 8 + slice % 8.  (p.xcd_group == 0: plain (slice, tile) grid.)
    int tile = blockIdx.y, slice = blockIdx.x;
    if (p.xcd_group) {
        const int k = blockIdx.x >> 3;
        tile = k % p.tiles;
        slice = (k / p.tiles) * 8 + (blockIdx.x & 7);
    }
    const int ky = NKY == 3 ? 0 : tile % 3;                     // first kernel row of this workgroup
    if (NKY == 1) tile /= 3;
    const int ci0 = (tile % p.tiles_ci) * TCI, co0 = (tile / p.tiles_ci) * TCO;
    const int seg0 = slice * p.segs_per_slice;
    int nseg = p.total_segs - seg0;
    if (nseg > p.segs_per_slice) nseg = p.segs_per_slice;
    if (nseg <= 0) return;

    // ---- lane-constant parts of the DMA sources ---------------------------------------------------------------
    // dY piece e = k*NT + tid: pixel e / (TCO/8), 16-byte slot e % (TCO/8); the slot holds source chunk slot ^ 2*swz(pixel)
    int ypx[NY], yoff[NY], ych[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        const int e = k * NT + tid, px = e / (TCO / 8), slot = e % (TCO / 8);
        ypx[k] = px;
        ych[k] = co0 + ((slot ^ (2 * pair_swz<YROW>(px))) << 3);
        yoff[k] = px * p.ldy + ych[k];
    }
    int xpx[NX], xoff[NX], xch[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int e = k * NT + tid, px = e / (TCI / 8), slot = e % (TCI / 8);
        // image row px = (patch row r, column c); pieces past the last row: always the zero page
        const int r = px / XPX, c = px - r * XPX;
        xpx[k] = px < NKY * XPX ? (r << 8 | c) : -1;
        xch[k] = ci0 + ((slot ^ (2 * pair_swz<XROW>(px))) << 3);
        xoff[k] = (r * p.Wi + c) * p.ldx + xch[k];
    }

    // uniform segment cursor of the NEXT k-step to issue
    int cn = seg0 / (p.Ho * p.segs_per_row);
    int rem = seg0 - cn * p.Ho * p.segs_per_row;
    int coy = rem / p.segs_per_row, csx = rem - coy * p.segs_per_row;
    // flat mode: the k-step starts at flat position (image fn, output row fy, column fj of the row padded to L = Wo + 2 columns)
    const int L = p.Wo + 2;
    int fn = 0, fy = 0, fj = 0;
    if (p.flat) {
        const int f0 = seg0 * SEG;
        fn = f0 / (p.Ho * L);
        const int r0 = f0 - fn * p.Ho * L;
        fy = r0 / L;
        fj = r0 - fy * L;
    }
    auto issue = [&](int s) {                                   // DMAs of k-step s (s >= nseg: zero page only) into ring slot s % WG_RING
        const bool live = s < nseg;
        if (p.flat) {
            // every piece derives its own (image, row, column) from the uniform cursor: one wrap at most (L >= 34).
            // (Tried: the two candidate rows' base pointers and validity as uniform values and a per-piece select -- 6-12 % SLOWER.)
            const unsigned slot = lds0 + (s % WG_RING) * SLOT;
#pragma unroll
            for (int k = 0; k < NY; ++k) {
                int j = fj + ypx[k];
                const int wrap = j >= L;
                j -= wrap ? L : 0;
                const int row = fn * p.Ho + fy + wrap;                          // dY rows are contiguous across images
                const bool ok = live && j < p.Wo && row < p.N * p.Ho;
                glds16(ok ? dy + ((size_t)row * p.Wo + j) * p.ldy + ych[k] : zero, slot + (k * NT + wave * 64) * 16);
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                int j = fj + (xpx[k] & 255);
                const int wrap = j >= L;
                j -= wrap ? L : 0;
                int y = fy + wrap, img = fn;
                if (y >= p.Ho) { y -= p.Ho; ++img; }
                int yy = y + ky - p.pad, xx = j - p.pad;
                if (p.reflect) {
                    yy = yy < 0 ? -yy : (yy >= p.Hi ? 2 * p.Hi - 2 - yy : yy);
                    xx = xx < 0 ? -xx : (xx >= p.Wi ? 2 * p.Wi - 2 - xx : xx);
                }
                const bool ok = live && xpx[k] >= 0 && img < p.N && (unsigned)yy < (unsigned)p.Hi && (unsigned)xx < (unsigned)p.Wi;
                glds16(ok ? x + ((long)(img * p.Hi + yy) * p.Wi + xx) * p.ldx + xch[k] : zero, slot + YB + (k * NT + wave * 64) * 16);
            }
            if (live) {
                fj += SEG;
                while (fj >= L) {
                    fj -= L;
                    if (++fy == p.Ho) { fy = 0; ++fn; }
                }
            }
            return;
        }
        const int ox0 = csx * SEG, iy = coy - p.pad + ky, ix0 = ox0 - p.pad;
        const bf16_t *yb = dy + ((size_t)(cn * p.Ho + coy) * p.Wo + ox0) * p.ldy;
        const bf16_t *xb = x + ((long)(cn * p.Hi + iy) * p.Wi + ix0) * p.ldx;
        const unsigned slot = lds0 + (s % WG_RING) * SLOT;
#pragma unroll
        for (int k = 0; k < NY; ++k) {
            const bool ok = live && ox0 + ypx[k] < p.Wo;
            glds16(ok ? yb + yoff[k] : zero, slot + (k * NT + wave * 64) * 16);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int r = xpx[k] >> 8, c = xpx[k] & 255;
            if (p.reflect) {            // mirrored halo: the source is no longer base + a lane constant
                int yy = iy + r, xx = ix0 + c;
                yy = yy < 0 ? -yy : (yy >= p.Hi ? 2 * p.Hi - 2 - yy : yy);
                xx = xx < 0 ? -xx : (xx >= p.Wi ? 2 * p.Wi - 2 - xx : xx);
                const bool ok = live && xpx[k] >= 0 && (unsigned)yy < (unsigned)p.Hi && (unsigned)xx < (unsigned)p.Wi;
                glds16(ok ? x + ((long)(cn * p.Hi + yy) * p.Wi + xx) * p.ldx + xch[k] : zero, slot + YB + (k * NT + wave * 64) * 16);
                continue;
            }
            const bool ok = live && xpx[k] >= 0 && (unsigned)(iy + r) < (unsigned)p.Hi && (unsigned)(ix0 + c) < (unsigned)p.Wi;
            glds16(ok ? xb + xoff[k] : zero, slot + YB + (k * NT + wave * 64) * 16);
        }
        if (live) {
            if (++csx == p.segs_per_row) {
                csx = 0;
                if (++coy == p.Ho) { coy = 0; ++cn; }
            }
        }
    };

    const int wr = wave / WCI, wc = wave % WCI;
    f32x4 acc[NTAP][MT][NTF];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};

#pragma unroll
    for (int h = 0; h < WG_RING - 1; ++h) issue(h);
    for (int s = 0; s < nseg; ++s) {
        wait_vm<(WG_RING - 2) * PER_STEP>();                    // the k-steps after s that are already issued may stay in flight
        __builtin_amdgcn_s_barrier();
        issue(s + WG_RING - 1);                                 // the slot of k-step s-1: every wave is past its reads (barrier)
        const unsigned char *Ys = smem + (s % WG_RING) * SLOT, *Xs = Ys + YB;
        bf16x8 a[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = tr_frag<YROW>(Ys, 0, wr * WTCO + 16 * i, lane);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            bf16x8 b[NTF];
#pragma unroll
            for (int j = 0; j < NTF; ++j) b[j] = tr_frag<XROW>(Xs, (t / 3) * XPX + (t % 3), wc * WTCI + 16 * j, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTF; ++j) acc[t][i][j] = mma16<DT>(a[i], b[j], acc[t][i][j]);
#ifdef RDFC_WGRAD_EXP_DUP   /* timing experiment only (wrong results): every MFMA twice -> how much of a k-step is not MFMA time */
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTF; ++j) acc[t][i][j] = mma16<DT>(a[i], b[j], acc[t][i][j]);
#endif
        }
    }
    wait_vm<0>();

    // accumulate: row co = 4*(lane>>4)+r, col ci = lane&15
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) {
                const int ci = ci0 + wc * WTCI + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wr * WTCO + 16 * i + 4 * (lane >> 4) + r;
                    if (co < p.Co && ci < p.Ci) atomicAdd(&dw[((size_t)co * 9 + ky * 3 + t) * p.Ci + ci], acc[t][i][j][r]);
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

template <int TCO, int NKY, int DT = RDFC_BF16, int TCI = 64>
int launch(WGParams p, const void *x, const void *dy, float *dw, const bf16_t *zero, hipStream_t st) {
    constexpr int NT = 256 * (TCI / 64);
    constexpr int SMEM = WG_RING * (SEG * TCO * 2 + ((NKY * XPX * (TCI / 8) + NT - 1) / NT) * NT * 16);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)wgrad_glds_kernel<TCO, NKY, DT, TCI>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    p.tiles_ci = cdiv(p.Ci, TCI);
    const int tiles = cdiv(p.Co, TCO) * p.tiles_ci * (3 / NKY);
    // Split-K factor: fill the chip's workgroup slots in ONE round -- 528 workgroups measured 1.3x slower
    // than 480 -- within an atomic budget of 48 MB per launch (sweep on one device, 256->256 @ 57x76: 24 MB / 240
    // workgroups 108 us, 48 MB / 480 workgroups 91 us; under-filling the chip costs more than the extra atomics).
    static const double cap_mb = getenv("RDFC_WGRAD_ATOMIC_MB") ? atof(getenv("RDFC_WGRAD_ATOMIC_MB")) : 48.0;
    // workgroups the chip holds at once: the 64-co kernel (48 KB of LDS, 104 registers) fits three per CU, the 128-co one two.
    // Sizing the 64-co launches for 768 instead of 512: 64->64 @ 228x304 87 -> 80 us, 256->128 @ 114x152 154 -> 132, 128->64 @ 228x304 158 -> 141
    static const int slots_env = getenv("RDFC_WGRAD_SLOTS") ? atoi(getenv("RDFC_WGRAD_SLOTS")) : 0;
    const int slots = slots_env ? slots_env : (TCI == 128 ? 256 : (TCO == 64 && NKY == 1 ? 768 : 512));     // 128 x 128: one 8-wave workgroup per CU
    int slices = slots / tiles;
    const int atomic_cap = (int)(cap_mb * 1.0e6 / (4.0 * p.Co * p.Ci * 9));
    if (slices > atomic_cap) slices = atomic_cap;
    if (slices > cdiv(p.total_segs, 8)) slices = cdiv(p.total_segs, 8);
    if (slices < 1) slices = 1;
    p.segs_per_slice = cdiv(p.total_segs, slices);
    slices = cdiv(p.total_segs, p.segs_per_slice);
    rdfc_note_kernel("wgrad_glds_kernel<%d, %d, %d, %d>", TCO, NKY, DT, TCI);
    static const bool xcd_group = !(getenv("RDFC_WGRAD_XCD") && atoi(getenv("RDFC_WGRAD_XCD")) == 0);
    p.tiles = tiles;
    // measured (tools/wgrad_xcd_ab.sh): 64->64 @ 228x304 (3 tiles) 95 -> 89 us; with 12-24 tiles per slice the slice count
    // (512 / tiles) is no multiple of 8 and the XCDs end up unevenly loaded: 256->256 @ 57x76 88 -> 123 us.  FETCH_SIZE
    // shows the re-reads (3x on the 64-channel layers) either way; the Infinity Cache serves them at full rate.
    p.xcd_group = xcd_group && tiles <= 3 && slices >= 8;
    const dim3 grid = p.xcd_group ? dim3(cdiv(slices, 8) * 8 * tiles, 1) : dim3(slices, tiles);
    hipLaunchKernelGGL((wgrad_glds_kernel<TCO, NKY, DT, TCI>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)dy, dw, zero);
    return 1;
}

}  // namespace

// Called by rdfc_conv2d_wgrad (conv_wgrad.hip) before the register-staged kernels; returns 1 if it took the launch.
static bool wgrad_glds_shape_ok(const rdfc_conv_desc *d);
int rdfc_wgrad_glds_covers(const rdfc_conv_desc *d) { return wgrad_glds_shape_ok(d) ? 1 : 0; }

int rdfc_try_wgrad_glds(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st, int reflect = 0) {
    static const int mode = getenv("RDFC_WGRAD_GLDS") ? atoi(getenv("RDFC_WGRAD_GLDS")) : 1;
    if (!wgrad_glds_shape_ok(d)) return 0;
    if (reflect && (d->ph < 1 || d->ph >= d->Hi || d->pw >= d->Wi)) return 0;
    const int spr = cdiv(d->Wo, SEG);
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    WGParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->ph, d->ldx, d->ldy, spr, d->N * d->Ho * spr, 0, 0, 0, 0, reflect, 0};
    // Flat pixel index where the rows leave more than 30 % of their 32-pixel segments empty (29x38: 41 %: 189 -> 154 us).  The
    // per-piece address arithmetic of the flat walk costs ~9 % of a k-step: at 57x76 (21 % empty) it breaks even (153 -> 149, 149 -> 149 us),
    // on wide rows (152, 304: 4 % empty) it loses 8-9 %.  RDFC_WGRAD_FLAT=0|1 forces.
    static const int flat_env = getenv("RDFC_WGRAD_FLAT") ? atoi(getenv("RDFC_WGRAD_FLAT")) : -1;
    const double ragged = 1.0 - (double)(d->Wo + 2) / (spr * SEG);
    const bool flat_ok = d->Wo >= 32 && (long)d->N * d->Ho * (d->Wo + 2) < 2000000000L;
    static const int nine_env = getenv("RDFC_WGRAD_GLDS_NINE") ? atoi(getenv("RDFC_WGRAD_GLDS_NINE")) : 0;
    if (flat_ok && !nine_env && (flat_env == 1 || (flat_env != 0 && ragged > 0.30))) {
        p.flat = 1;
        p.total_segs = cdiv(d->N * d->Ho * (d->Wo + 2), SEG);
    }
    // 128 x 128 tiles (8 waves, one workgroup per CU): A/B switch RDFC_WGRAD_GLDS=5 -- measured SLOWER at the batched step's sizes
    // (N = 16; 128 x 128 against the default): 256->256 @ 57x76 150-154 vs 146-148 us, 128->128 @ 114x152 189 vs 137 us, 512->512 @
    // 29x38 187 vs 181 us: halving the DMA'd bytes per MFMA does not pay for going from two 4-wave workgroups per CU to one
    if (mode == 5 && d->Co % 128 == 0 && d->Ci % 128 == 0)
        return d->dtype == RDFC_F16 ? launch<128, 1, RDFC_F16, 128>(p, x, dy, dw, zero, st) : launch<128, 1, RDFC_BF16, 128>(p, x, dy, dw, zero, st);
    // 128-co tiles from 256 output channels up; below that the 64-co tile's larger grid wins (128->128: 82 vs 92 us)
    if (d->Co % 128 == 0 && (mode == 3 || (mode == 1 && d->Co >= 256))) return d->dtype == RDFC_F16 ? launch<128, 1, RDFC_F16>(p, x, dy, dw, zero, st) : launch<128, 1>(p, x, dy, dw, zero, st);
    // all nine taps per workgroup (dY and X staged once instead of once per kernel row): measured SLOWER on the
    // full-resolution 64-channel layers (120 vs 97 us: 36 MFMAs against 40 transposing reads per k-step, 244 VGPRs), so
    // L2 traffic is not what bounds them; kept behind RDFC_WGRAD_GLDS_NINE=1 for A/B runs
    static const int nine = getenv("RDFC_WGRAD_GLDS_NINE") ? atoi(getenv("RDFC_WGRAD_GLDS_NINE")) : 0;
    if (nine && d->Co * d->Ci <= 128 * 64) return d->dtype == RDFC_F16 ? launch<64, 3, RDFC_F16>(p, x, dy, dw, zero, st) : launch<64, 3>(p, x, dy, dw, zero, st);
    return d->dtype == RDFC_F16 ? launch<64, 1, RDFC_F16>(p, x, dy, dw, zero, st) : launch<64, 1>(p, x, dy, dw, zero, st);
}

static bool wgrad_glds_shape_ok(const rdfc_conv_desc *d) {
    static const int mode = getenv("RDFC_WGRAD_GLDS") ? atoi(getenv("RDFC_WGRAD_GLDS")) : 1;
    if (!mode || !rdfc_is16(d->dtype)) return false;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 && d->ph == d->pw && d->ph <= 1)) return false;
    if (d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return false;
    if (d->Co % 64 || d->Ci % 64 || d->ldx % 8 || d->ldy % 8) return false;
    const int spr = cdiv(d->Wo, SEG);
    return (double)d->Wo / (spr * SEG) >= 0.55;                   // else: too much padding per row segment
}
