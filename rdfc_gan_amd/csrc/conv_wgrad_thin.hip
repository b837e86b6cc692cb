// conv_wgrad_thin.hip -- weight gradient when one side of the convolution has at most 4 channels (bf16, stride 1).
//
//   (a) thin input  (Ci <= 4: the 1- / 3-channel input layers, 7x7 and 3x3):  T = x,  W = dy
//   (b) thin output (Co <= 4: the 1-channel heads, ResnetGenerator's 7x7 -> 3): T = dy, W = x
//
//   G[(tap, c_t)][c_w] = sum over W's pixels p of  T[p + r(tap)][c_t] * W[p][c_w]
//
// On the per-tap MFMA kernel these layers run one 16-wide padded tile per tap and re-read the wide map KH*KW
// times (49x for the 7x7 layers): 160-460 us each for work that is one pass over the wide map.  Here the GEMM's
// M index is the whole (tap, thin channel) set (<= 160 rows), N the wide channels, K the pixels.  A k-step is a
// segment of 32 consecutive pixels of one row of W.  Each wave owns its segments (no workgroup barrier in the
// main loop): it stages the W tile [32 px][<=64 ch] and the KH halo rows of T in its private LDS region, reads the
// W fragments with ds_read_b64_tr_b16 and gathers the T fragments (8 consecutive pixels of one tap/channel per lane)
// with 16-bit LDS reads.  HBM bound: W is read once, T KH times from L2.  The four waves of a workgroup combine
// their accumulators through LDS, then one f32 atomic per element and workgroup.
#include "common.h"

namespace {

struct ThinWParams {
    int N, Hw, Ww, ldw;            // wide map W
    int Ht, Wt, ct, ldt;           // thin map T
    int KH, KW, ry_min, rx_min, flip;
    int M;                         // KH*KW*ct real rows
    int segs_per_row, units;
    int trow;                      // staged elements per T row = (32 + KW - 1) * ct
    int thin_is_x, Ci, Ctot;       // dw index: thin_is_x ? (c_w*T + tap)*Ci + c_t : (c_t*T + tap)*Ci + c_w ;  Ctot = wide channels
    int wave_lds;                  // bytes of LDS per wave
    // round 6 (ACT instantiation, thin input only): the wide map is dz, the cotangent BEHIND the block's activation; dy = dz * act'(y) is
    // formed while dz is staged (y_act: the block's activated output, pixel stride ld_ya; ns: the derivative where y <= 0), and the
    // bias gradient comes out of the same product as one more row of ones (dbias, may be null)
    const void *y_act;
    int ld_ya;
    float ns;
    float *dbias;
};

constexpr int SEG = 32;
constexpr int MAXT = 14;           // 16-bit thin loads per lane and unit (7 rows x 114 elements / 64 lanes)

__device__ inline bf16x8 tr_frag(const bf16_t *img, int ld, int cbase, int lane) {
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const bf16_t *a0 = img + (8 * g + q) * ld + cbase + 4 * pp;
    const bf16_t *a1 = a0 + 4 * ld;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int MF, int NF, int DT = RDFC_BF16, bool ACT = false>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(ThinWParams p, const bf16_t *__restrict__ wide,
                                                         const bf16_t *__restrict__ thin, float *__restrict__ dw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CW = 16 * NF, LDW = CW + 8, VPP = 2 * NF;        // wide tile: channels, padded row, 16-B vectors per pixel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16_t *Wl = reinterpret_cast<bf16_t *>(smem + (size_t)wave * p.wave_lds);
    unsigned short *Tl = reinterpret_cast<unsigned short *>(Wl + SEG * LDW);
    const int zero_row = p.KH * p.trow;                               // 32*ct + 8 zeros behind the staged rows
    for (int e = lane; e < SEG * p.ct + 8; e += 64) Tl[zero_row + e] = 0;
    const int cw0 = blockIdx.y * CW;

    // lane-constant pieces: wide vectors, thin elements, A-fragment bases
    int wpx[NF], wv[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int e = lane + 64 * j;
        wpx[j] = e / VPP;
        wv[j] = e - wpx[j] * VPP;
    }
    const int nthin = p.KH * p.trow, nt = (nthin + 63) / 64;
    int trc[MAXT], tch[MAXT];                                         // (row << 16 | column), channel
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
        const int e = lane + 64 * j;
        trc[j] = -1;
        tch[j] = 0;
        if (j < nt && e < nthin) {
            const int r = e / p.trow, ce = e - r * p.trow, col = ce / p.ct, c = ce - col * p.ct;
            trc[j] = (r << 16) | col;
            tch[j] = c;
        }
    }
    const int kg = lane >> 4;
    int abase[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
        const int m = 16 * f + (lane & 15);
        abase[f] = zero_row;
        if (m < p.M) {
            const int tap = m / p.ct, c = m - tap * p.ct, ky = tap / p.KW, kx = tap - ky * p.KW;
            const int r = p.flip ? p.KH - 1 - ky : ky, cx = p.flip ? p.KW - 1 - kx : kx;
            abase[f] = r * p.trow + (cx + 8 * kg) * p.ct + c;
        } else {
            abase[f] += 8 * kg * p.ct;
        }
    }

    f32x4 acc[MF][NF];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[f][n] = f32x4{0, 0, 0, 0};

    u32x4 rw[NF], ry[ACT ? NF : 1];
    unsigned rt[MAXT];                                                 // (one register per element: see load_regs)
    unsigned wok = 0, tok = 0;                                         // validity masks of the loads in flight, applied when they are staged
    const int wave_id = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    // Every load is UNCONDITIONAL from a clamped position and masked when it is staged one unit later; offsets, loads and masks are
    // separate passes with the scheduler fenced between them.  (A load under a per-element condition makes hipcc branch around it and
    // wait for it before it issues the next -- cdna_hip_programming.md section 5, trap (c): up to 14 + NF dependent round trips per
    // 32-pixel unit were this kernel's time, 155-205 us for the 7x7 layers' one pass over a 142 MB map.)
    auto load_regs = [&](int u) {
        const int n = u / (p.Hw * p.segs_per_row), rem = u - n * p.Hw * p.segs_per_row;
        const int y = rem / p.segs_per_row, xs = (rem - y * p.segs_per_row) * SEG;
        const bf16_t *wb = wide + ((size_t)(n * p.Hw + y) * p.Ww) * p.ldw + cw0;
        const bf16_t *yb = ACT ? reinterpret_cast<const bf16_t *>(p.y_act) + ((size_t)(n * p.Hw + y) * p.Ww) * p.ld_ya + cw0 : nullptr;
        int yoff[NF];
        const int ty0 = y + p.ry_min, tx0 = xs + p.rx_min;
        const unsigned short *timg = reinterpret_cast<const unsigned short *>(thin) + (long)n * p.Ht * p.Wt * p.ldt;
        int woff[NF], toffs[MAXT];
        wok = tok = 0;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int px = xs + wpx[j];
            if (px < p.Ww) wok |= 1u << j;
            woff[j] = min(px, p.Ww - 1) * p.ldw + wv[j] * 8;
            yoff[j] = min(px, p.Ww - 1) * p.ld_ya + wv[j] * 8;
        }
#pragma unroll
        for (int j = 0; j < MAXT; ++j) {
            toffs[j] = 0;
            if (j < nt) {
                const int r = trc[j] >> 16, col = trc[j] & 0xffff;
                const int gy = ty0 + r, gx = tx0 + col;
                if (trc[j] >= 0 && (unsigned)gy < (unsigned)p.Ht && (unsigned)gx < (unsigned)p.Wt) tok |= 1u << j;
                toffs[j] = (min(max(gy, 0), p.Ht - 1) * p.Wt + min(max(gx, 0), p.Wt - 1)) * p.ldt + tch[j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NF; ++j) rw[j] = *reinterpret_cast<const u32x4 *>(wb + woff[j]);
        if constexpr (ACT) {
#pragma unroll
            for (int j = 0; j < NF; ++j) ry[j] = *reinterpret_cast<const u32x4 *>(yb + yoff[j]);
        }
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
            if (j < nt) rt[j] = timg[toffs[j]];
        __builtin_amdgcn_sched_barrier(0);
    };

    int u = wave_id;
    if (u < p.units) load_regs(u);
    for (; u < p.units; u += nwaves) {
        // stage (this wave's LDS region only; LDS executes a wave's instructions in order)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            u32x4 v = rw[j];
            if constexpr (ACT) {                                       // dy = dz * act'(y), rounded once like rdfc_act_bwd's result
                const bf16x8 dzv = __builtin_bit_cast(bf16x8, v), yv = __builtin_bit_cast(bf16x8, ry[j]);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = cvt16<DT>(f32of<DT>(dzv[e]) * (f32of<DT>(yv[e]) > 0.f ? 1.f : p.ns));
                v = __builtin_bit_cast(u32x4, o);
            }
            *reinterpret_cast<u32x4 *>(Wl + wpx[j] * LDW + wv[j] * 8) = (wok >> j) & 1 ? v : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < MAXT; ++j)
            if (j < nt && trc[j] >= 0) Tl[lane + 64 * j] = (tok >> j) & 1 ? (unsigned short)rt[j] : (unsigned short)0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (u + nwaves < p.units) load_regs(u + nwaves);

        bf16x8 b[NF];
#pragma unroll
        for (int n = 0; n < NF; ++n) b[n] = tr_frag(Wl, LDW, 16 * n, lane);
#pragma unroll
        for (int f = 0; f < MF; ++f) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            s16x8 av;
#pragma unroll
            for (int j = 0; j < 8; ++j) av[j] = (short)Tl[abase[f] + j * p.ct];
            if constexpr (ACT) {                                       // row M of the product: ones x dy = the bias gradient (pixels beyond the row are zeros in dy)
                if (p.dbias != nullptr && f == (p.M >> 4) && (lane & 15) == (p.M & 15)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = (short)(DT == RDFC_F16 ? 0x3C00 : 0x3F80);
                }
            }
            const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int n = 0; n < NF; ++n) acc[f][n] = mma16<DT>(a, b[n], acc[f][n]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // combine the four waves through LDS: G[m][c], m = 16 f + 4 (lane>>4) + r, c = 16 n + (lane & 15)
    float *G = reinterpret_cast<float *>(smem);
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int n = 0; n < NF; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (16 * f + 4 * (lane >> 4) + r) * CW + 16 * n + (lane & 15);
                        G[idx] = (w == 0 ? 0.f : G[idx]) + acc[f][n][r];
                    }
        }
    }
    __syncthreads();
    const int T = p.KH * p.KW;
    if constexpr (ACT) {
        if (p.dbias != nullptr && p.M < MF * 16)
            for (int c = tid; c < CW; c += 256)
                if (cw0 + c < p.Ctot) atomicAdd(p.dbias + cw0 + c, G[p.M * CW + c]);
    }
    if (p.thin_is_x) {
        // dw[c_w][tap][c_t] = dw[c_w * M + m]: walk m fastest so that a wave's atomics fall on consecutive addresses
        for (int idx = tid; idx < p.M * CW; idx += 256) {
            const int c = idx / p.M, m = idx - c * p.M, cw = cw0 + c;
            if (cw < p.Ctot) atomicAdd(dw + (size_t)cw * p.M + m, G[m * CW + c]);
        }
    } else {
        for (int idx = tid; idx < p.M * CW; idx += 256) {
            const int m = idx / CW, c = idx - m * CW, cw = cw0 + c;
            if (cw >= p.Ctot) continue;
            const int tap = m / p.ct, c_t = m - tap * p.ct;
            atomicAdd(dw + ((size_t)c_t * T + tap) * p.Ci + cw, G[idx]);
        }
    }
}

template <int MF, int NF, bool ACT = false>
void launch(ThinWParams p, int groups, const void *wide, const void *thin, float *dw, hipStream_t st, int dtype) {
    constexpr int CW = 16 * NF, LDW = CW + 8;
    p.wave_lds = (SEG * LDW * 2 + (p.KH * p.trow + SEG * p.ct + 8) * 2 + 15) / 16 * 16;
    size_t smem = (size_t)4 * p.wave_lds;
    const size_t red = (size_t)MF * 16 * CW * 4;
    if (smem < red) smem = red;
    int blocks = cdiv(p.units, 4 * 4);                        // >= 4 segments per wave
    if (blocks > 512) blocks = 512;
    const int cap = (int)(24.0e6 / ((double)p.M * CW * 4.0 * groups));
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (ACT) rdfc_note_kernel("wgrad_thin_kernel<%d, %d, %d, true>", MF, NF, dtype);
    else rdfc_note_kernel("wgrad_thin_kernel<%d, %d, %d>", MF, NF, dtype);
    if (dtype == RDFC_F16)
        hipLaunchKernelGGL((wgrad_thin_kernel<MF, NF, RDFC_F16, ACT>), dim3(blocks, groups), dim3(256), smem, st, p, (const bf16_t *)wide,
                           (const bf16_t *)thin, dw);
    else
        hipLaunchKernelGGL((wgrad_thin_kernel<MF, NF, RDFC_BF16, ACT>), dim3(blocks, groups), dim3(256), smem, st, p, (const bf16_t *)wide,
                           (const bf16_t *)thin, dw);
}

template <int MF, bool ACT = false>
void launch_M(const ThinWParams &p, int nfrag, const void *wide, const void *thin, float *dw, hipStream_t st, int dtype) {
    // wide channels in groups of <= 64 (grid.y); each group re-reads T (tiny) and its own slice of W
    if (nfrag % 4 == 0) launch<MF, 4, ACT>(p, nfrag / 4, wide, thin, dw, st, dtype);
    else if (nfrag % 3 == 0) launch<MF, 3, ACT>(p, nfrag / 3, wide, thin, dw, st, dtype);
    else if (nfrag % 2 == 0) launch<MF, 2, ACT>(p, nfrag / 2, wide, thin, dw, st, dtype);
    else launch<MF, 1, ACT>(p, nfrag, wide, thin, dw, st, dtype);
}

}  // namespace

// fills p for the shape; false: not this kernel's
static bool thin_params(const rdfc_conv_desc *d, ThinWParams &p, bool &thin_x) {
    if (!rdfc_is16(d->dtype)) return false;
    if (d->sh != 1 || d->sw != 1 || d->dh != 1 || d->dw != 1) return false;
    thin_x = d->Ci <= 4 && d->Co % 16 == 0 && d->ldy % 8 == 0 && d->Co <= 128;
    const bool thin_y = !thin_x && d->Co <= 4 && d->Ci % 16 == 0 && d->ldx % 8 == 0 && d->Ci <= 128;
    if (!thin_x && !thin_y) return false;
    p.N = d->N;
    p.KH = d->KH;
    p.KW = d->KW;
    if (thin_x) {
        p.Hw = d->Ho; p.Ww = d->Wo; p.ldw = d->ldy; p.Ctot = d->Co;
        p.Ht = d->Hi; p.Wt = d->Wi; p.ct = d->Ci; p.ldt = d->ldx;
        p.ry_min = -d->ph; p.rx_min = -d->pw; p.flip = 0;
    } else {
        p.Hw = d->Hi; p.Ww = d->Wi; p.ldw = d->ldx; p.Ctot = d->Ci;
        p.Ht = d->Ho; p.Wt = d->Wo; p.ct = d->Co; p.ldt = d->ldy;
        p.ry_min = d->ph - (d->KH - 1); p.rx_min = d->pw - (d->KW - 1); p.flip = 1;
    }
    p.thin_is_x = thin_x;
    p.Ci = d->Ci;
    p.M = d->KH * d->KW * p.ct;
    p.trow = (SEG + d->KW - 1) * p.ct;
    if (p.M > 160 || d->KH * p.trow > MAXT * 64 || d->KH > 255) return false;
    p.segs_per_row = cdiv(p.Ww, SEG);
    p.units = d->N * p.Hw * p.segs_per_row;
    if (p.units < 512) return false;                        // small maps: the tile kernels' grid is fine there
    p.wave_lds = 0;
    p.y_act = nullptr;
    p.ld_ya = 0;
    p.ns = 0.f;
    p.dbias = nullptr;
    return true;
}

// Called by rdfc_conv2d_wgrad (conv_wgrad.hip) before the MFMA tile kernels; returns 1 if it took the launch.
int rdfc_try_wgrad_thin(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st) {
    ThinWParams p;
    bool thin_x = false;
    if (!thin_params(d, p, thin_x)) return 0;
    const int nfrag = p.Ctot / 16;
    if (p.M <= 16) launch_M<1>(p, nfrag, thin_x ? dy : x, thin_x ? x : dy, dw, st, d->dtype);
    else if (p.M <= 32) launch_M<2>(p, nfrag, thin_x ? dy : x, thin_x ? x : dy, dw, st, d->dtype);
    else if (p.M <= 64) launch_M<4>(p, nfrag, thin_x ? dy : x, thin_x ? x : dy, dw, st, d->dtype);
    else launch_M<10>(p, nfrag, thin_x ? dy : x, thin_x ? x : dy, dw, st, d->dtype);
    return 1;
}

// include/rdfc_hip.h: weight AND bias gradient of a conv block without a normalisation layer, the backward of its activation folded in --
// the stems (3 | 1 -> 64 | 48 | 16 at full resolution, rdf_generator.py:60-66; resnet_generator.py:12-15): dz is the cotangent behind the
// activation, y_act the block's activated output.  Thin inputs (Ci <= 4), ReLU / LeakyReLU, stride 1.
extern "C" int rdfc_conv2d_wgrad_act_supported(const rdfc_conv_desc *d) {
    ThinWParams p;
    bool thin_x = false;
    if (d == nullptr || !thin_params(d, p, thin_x) || !thin_x) return 0;
    if (!(d->act == RDFC_ACT_RELU || d->act == RDFC_ACT_LRELU)) return 0;
    return p.M + 1 <= 160 ? 1 : 0;                          // (one more row of the product for the bias gradient)
}
extern "C" int rdfc_conv2d_wgrad_act(const rdfc_conv_desc *d, const void *x, const void *dz, const void *y_act, int ld_y, float *dw_packed,
                                     float *dbias, void *stream) {
    RDFC_REQUIRE(d != nullptr && x != nullptr && dz != nullptr && y_act != nullptr && dw_packed != nullptr, "rdfc_conv2d_wgrad_act: null pointer");
    RDFC_REQUIRE(rdfc_conv2d_wgrad_act_supported(d), "rdfc_conv2d_wgrad_act: shape / activation not covered (ask rdfc_conv2d_wgrad_act_supported)");
    RDFC_REQUIRE(ld_y >= d->Co && ld_y % 8 == 0 && (reinterpret_cast<size_t>(y_act) & 15) == 0 && (reinterpret_cast<size_t>(dz) & 15) == 0,
                 "rdfc_conv2d_wgrad_act: dz / y_act must be 16-byte aligned with pixel strides that are multiples of 8");
    ThinWParams p;
    bool thin_x = false;
    thin_params(d, p, thin_x);
    p.y_act = y_act;
    p.ld_ya = ld_y;
    p.ns = d->act == RDFC_ACT_RELU ? 0.f : d->slope;
    p.dbias = dbias;
    const int nfrag = p.Ctot / 16, rows = p.M + 1;
    hipStream_t st = (hipStream_t)stream;
    if (rows <= 16) launch_M<1, true>(p, nfrag, dz, x, dw_packed, st, d->dtype);
    else if (rows <= 32) launch_M<2, true>(p, nfrag, dz, x, dw_packed, st, d->dtype);
    else if (rows <= 64) launch_M<4, true>(p, nfrag, dz, x, dw_packed, st, d->dtype);
    else launch_M<10, true>(p, nfrag, dz, x, dw_packed, st, d->dtype);
    RDFC_LAUNCH_CHECK("rdfc_conv2d_wgrad_act");
    return 0;
}
