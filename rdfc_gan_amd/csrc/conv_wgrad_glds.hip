// conv_wgrad_glds.hip -- weight gradient of 3x3 / stride-1 convolutions (bf16), staged by LDS-DMA.
// Since round 3 the default dispatch runs conv_wgrad_ring.hip (flat pixel walk, slab reduction: deterministic, 15-20 % faster on the
// 76-wide maps, within -6 % elsewhere: profiles/r03_wgrad_ring_ab.txt); this kernel is what RDFC_WGRAD_RING=0 selects for A/B runs and
// what maps narrower than 24 pixels fall back to.  Removed in round 3, measured slower in round 2: the 128 co x 128 ci eight-wave tile,
// the nine-taps-per-workgroup tile, the per-lane flat walk.
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
    // (assembled from 32-bit halves: an element-wise s16x8 initialiser costs four v_mov per read -- 96 per k-step, as much
    //  vector-issue time as the k-step's 24 MFMAs)
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, r);
}

// k-step ring (slots of dY + X rows filled WG_RING-1 steps ahead)
constexpr int WG_RING = 4;

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
    auto issue = [&](int s) {                                   // DMAs of k-step s (s >= nseg: zero page only) into ring slot s % WG_RING
        const bool live = s < nseg;
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
        wait_vm_and_reads<(WG_RING - 2) * PER_STEP>();                    // the k-steps after s that are already issued may stay in flight
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
    constexpr double cap_mb = 48.0;
    // workgroups the chip holds at once: the 64-co kernel (48 KB of LDS, 104 registers) fits three per CU, the 128-co one two.
    // Sizing the 64-co launches for 768 instead of 512: 64->64 @ 228x304 87 -> 80 us, 256->128 @ 114x152 154 -> 132, 128->64 @ 228x304 158 -> 141
    constexpr int slots_env = 0;
    const int slots = slots_env ? slots_env : (TCO == 64 && NKY == 1 ? 768 : 512);
    int slices = slots / tiles;
    const int atomic_cap = (int)(cap_mb * 1.0e6 / (4.0 * p.Co * p.Ci * 9));
    if (slices > atomic_cap) slices = atomic_cap;
    if (slices > cdiv(p.total_segs, 8)) slices = cdiv(p.total_segs, 8);
    if (slices < 1) slices = 1;
    p.segs_per_slice = cdiv(p.total_segs, slices);
    slices = cdiv(p.total_segs, p.segs_per_slice);
    rdfc_note_kernel("wgrad_glds_kernel<%d, %d, %d, %d>", TCO, NKY, DT, TCI);
    constexpr bool xcd_group = true;
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

int rdfc_try_wgrad_ring(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st, int reflect);   // conv_wgrad_ring.hip

int rdfc_try_wgrad_glds(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st, int reflect = 0) {
    static const int mode = getenv("RDFC_WGRAD_GLDS") ? atoi(getenv("RDFC_WGRAD_GLDS")) : 1;
    if (mode == 1 && rdfc_try_wgrad_ring(d, x, dy, dw, st, reflect)) return 1;       // flat walk + slab reduction (deterministic)
    if (!wgrad_glds_shape_ok(d)) return 0;
    if (reflect && (d->ph < 1 || d->ph >= d->Hi || d->pw >= d->Wi)) return 0;
    const int spr = cdiv(d->Wo, SEG);
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    WGParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->ph, d->ldx, d->ldy, spr, d->N * d->Ho * spr, 0, 0, 0, 0, reflect};
    // 128-co tiles from 256 output channels up; below that the 64-co tile's larger grid wins (128->128: 82 vs 92 us)
    if (d->Co % 128 == 0 && (mode == 3 || (mode == 1 && d->Co >= 256))) return d->dtype == RDFC_F16 ? launch<128, 1, RDFC_F16>(p, x, dy, dw, zero, st) : launch<128, 1>(p, x, dy, dw, zero, st);
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
