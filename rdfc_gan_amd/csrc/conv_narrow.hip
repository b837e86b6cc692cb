// conv_narrow.hip -- 3x3 / stride-1 / pad-1 convolutions around NARROW maps (8 or 24 channels), 2-byte elements: forward, input
// gradient and weight gradient.  Round 6.
//
// Where they occur: NLSPN's guidance path (rdf_generator.py:95-102, nlspn_model.py:29-33) -- gd_dec0 (128 -> 8, no norm, no activation)
// feeds conv_offset_aff (8 -> 24, f32 result) at full resolution.  Around those two layers a train step runs
//     conv_offset_aff forward  8 -> 24            its input gradient  24 -> 8            its weight gradient (Ci 8, Co 24)
//     gd_dec0's input gradient 8 -> 128
// and until round 6 all four went to the generic register-staged implicit-GEMM kernels, whose 16 x 256 / 128 x 128 tiles with K-steps of 32
// input channels are 2/3 ... 15/16 padding on these shapes: 189 + 343 + 385 + 313 us inside the step at 16 x 228 x 304
// (profiles/r05_nlspn_kernel_stats.csv) for maps whose HBM time is 20 ... 48 us -- 1.2 of the 3.6 ms that BASELINE configs[2] (NLSPN on)
// was behind configs[1].  These layers are HBM-bound by a wide margin (K = 72 or 216); what they need is a K layout without padding:
//
//   * k runs over (tap, channel) FLAT: with 8 channels a 32-deep MFMA step covers FOUR taps (lane group kg = lane / 16 takes tap
//     4 kb + kg: its B fragment is the 16 bytes of the pixel at that tap's shift), three steps for the nine taps instead of nine
//     three-quarters-empty ones; with 24 channels a step is one tap (three 16-byte chunks, the fourth lane group idles).
//   * the weights of the whole layer are A fragments in REGISTERS (6 ... 24 of them), loaded once per workgroup straight from the
//     packed [O][tap][I] / [I][tap][O] operand: no weight staging, no K loop over slabs.
//   * a workgroup owns 8 x 64 pixels: the halo patch (10 x 66 pixels x 16 | 48 bytes) goes to LDS once, each wave then walks its two
//     rows in groups of 16 pixels: NKB fragment reads + NCB x NKB MFMAs + a store.
// MFMA rows = result channels (as in every conv kernel here: an accumulator lane holds 4 consecutive channels of one pixel).
//
// Weight gradient (Ci = 8, Co <= 32): dW[co][tap][ci] = sum over pixels; GEMM with k = pixels, so both operands must come out of LDS
// TRANSPOSED (8 consecutive pixels per lane): ds_read_b64_tr_b16, whose sixteen lanes may supply sixteen unrelated 8-byte pieces --
// the im2col of the narrow map ("channel" block = two taps x 8 channels, each with its own pixel shift) is done by the addresses.
// Workgroups walk runs of tiles with the 2 x 5 accumulators in registers and leave one partial per workgroup; a second launch adds
// the partials in a fixed order (bit-reproducible, no float atomics).
#include "common.h"

#include <mutex>

namespace {

constexpr int NT = 256, TH = 8, TW = 64, PH = TH + 2, PW = TW + 2;

struct NParams {
    int N, H, W;                 // the gathered map and the result have the same size (stride 1, pad 1)
    int Cin, Cout, ld_in, ld_out;
    int transposed;              // input gradient: weight tap index mirrored (operand [I][tap][O])
    int out_f32;
    int tiles_x, tiles_y;
};

template <int CIN, int NCB, int DT>
__global__ __launch_bounds__(NT) void narrow3x3_kernel(NParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                      const float *__restrict__ bias, void *__restrict__ yout) {
    constexpr int CH = CIN / 8;                                   // 16-byte chunks per pixel: 1 | 3
    constexpr int PXB = CH * 16;                                  // LDS bytes per patch pixel
    constexpr int NKB = CH == 1 ? 3 : 9;                          // 32-deep MFMA steps: four taps each | one tap each
    __shared__ __attribute__((aligned(16))) unsigned char patch[PH * PW * PXB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, n = t / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- the halo patch: every pixel's channel vector as 16-byte pieces, zeros outside the image ---------------------------------
    for (int e = tid; e < PH * PW * CH; e += NT) {
        const int px = e / CH, ch = e - px * CH;
        const int r = px / PW, c = px - r * PW, iy = y0 - 1 + r, ix = x0 - 1 + c;
        u32x4 v = u32x4{0, 0, 0, 0};
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
            v = *reinterpret_cast<const u32x4 *>(x + ((size_t)(n * p.H + iy) * p.W + ix) * p.ld_in + ch * 8);
        *reinterpret_cast<u32x4 *>(patch + px * PXB + ch * 16) = v;
    }

    // ---- the layer's weights as A fragments (rows = result channels), loaded while the patch is on its way -------------------------
    const int row = lane & 15, kg = lane >> 4;
    bf16x8 A[NCB][NKB];
    unsigned boff[NKB];                                           // per step: byte offset of this lane's B piece relative to its pixel (tap shift + chunk)
    bool bval[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        const int tap = CH == 1 ? kb * 4 + kg : kb, chunk = CH == 1 ? 0 : kg;
        const bool valid = CH == 1 ? tap < 9 : kg < CH;
        const int tp = valid ? tap : 0;
        bval[kb] = valid;
        boff[kb] = (unsigned)(((tp / 3) * PW + tp % 3) * PXB + (valid ? chunk : 0) * 16);
        const int tw = p.transposed ? 8 - tp : tp;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int co = cb * 16 + row;
            bf16x8 a;
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = (bf16_t)0.0f;
            if (valid && co < p.Cout) a = *reinterpret_cast<const bf16x8 *>(w + ((size_t)co * 9 + tw) * CIN + (CH == 1 ? 0 : chunk * 8));
            A[cb][kb] = a;
        }
    }
    float bv[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = cb * 16 + 4 * kg + r;
            bv[cb][r] = (bias != nullptr && co < p.Cout) ? bias[co] : 0.f;
        }
    __syncthreads();

    // ---- a wave walks rows wave and wave + 4 of the tile in groups of 16 pixels ------------------------------------------------------
#pragma unroll 1
    for (int g = 0; g < 2 * (TW / 16); ++g) {
        const int r = wave + 4 * (g / (TW / 16)), c0 = (g % (TW / 16)) * 16;
        const int oy = y0 + r, ox = x0 + c0 + row;
        if (oy >= p.H || x0 + c0 >= p.W) continue;               // (wave-uniform)
        const unsigned char *pp = patch + (r * PW + c0 + row) * PXB;
        bf16x8 B[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            B[kb] = *reinterpret_cast<const bf16x8 *>(pp + boff[kb]);
            if (!bval[kb]) {                                      // (its A fragment is zero; the product must not be 0 x NaN)
#pragma unroll
                for (int i = 0; i < 8; ++i) B[kb][i] = (bf16_t)0.0f;
            }
        }
        const bool live = ox < p.W;
        const size_t po = ((size_t)(n * p.H + oy) * p.W + ox) * p.ld_out;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x4 acc = f32x4{bv[cb][0], bv[cb][1], bv[cb][2], bv[cb][3]};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = mma16<DT>(A[cb][kb], B[kb], acc);
            const int co = cb * 16 + 4 * kg;
            if (!live || co >= p.Cout) continue;                  // (Cout is a multiple of 4: a lane's four channels exist together)
            if (p.out_f32) *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(yout) + po + co) = acc;
            else {
                bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = cvt16<DT>(acc[q]);
                *reinterpret_cast<bf16x4 *>(reinterpret_cast<bf16_t *>(yout) + po + co) = o;
            }
        }
    }
}

// ---- weight gradient with an 8-channel map on one side -------------------------------------------------------------------------------
//   dW[co][ky][kx][ci] = sum over (y, x) of dY[y][x][co] * X[y + ky - 1][x + kx - 1][ci]
// The NARROW map (8 channels, 16 bytes per pixel) is staged with a one-pixel halo and is the operand the taps shift; the WIDE map goes
// through in slabs of 32 channels (blockIdx.y), 8 x 64 pixels at a time, without a halo:
//   SWAP = false  narrow = X (Ci = 8), wide = dY (Co <= 32: one slab): conv_offset_aff.          column (tap, ci), tap as written
//   SWAP = true   narrow = dY (Co = 8), wide = X (Ci a multiple of 32): gd_dec0 (128 -> 8).  With (u, v) = the X pixel the sum reads
//                 X[u][v][ci] * dY[u - ky + 1][v - kx + 1][co]: the same walk with the MIRRORED tap 8 - (3 ky + kx) as the shift.
// Per 32-pixel k-step a wave reads 2 wide fragments (channel blocks of 16) and 5 narrow ones (two taps x 8 channels each; 9 taps -> 4.5
// blocks, the last half reads zeros) through ds_read_b64_tr_b16 and issues 10 MFMAs.  Workgroups walk runs of tiles with the accumulators
// in registers, the next tile's pixels on their way in registers while the current one is multiplied; the four waves' sums are added
// in wave order and one partial per workgroup is left for narrow_wgrad_reduce_kernel.
struct NWParams {
    int N, H, W;
    int ld_n, ld_w, wch;         // pixel strides of the narrow / wide map, channels of the wide map
    int tiles_x, tiles_y, ntiles;
    int Co, Ci;
};
constexpr int WB = 64;                                             // LDS bytes per wide-map pixel (a slab of 32 channels)
constexpr int NNB = 5;                                             // 16-column blocks of (tap, narrow channel): 72 -> 80
constexpr int WG_OUT = 32 * NNB * 16;                              // one workgroup's partial: [32 wide channels][80 (tap, narrow channel)]
constexpr int NXS = (PH * PW + NT - 1) / NT, NYS = TH * TW * 4 / NT;   // 16-byte pieces per thread and tile: 3 narrow, 8 wide

template <int DT, bool SWAP>
__global__ __launch_bounds__(NT) void narrow_wgrad_kernel(NWParams p, const bf16_t *__restrict__ nar, const bf16_t *__restrict__ wide,
                                                         float *__restrict__ partial) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[PH * PW * 16 + 16];      // + one zero piece (the tenth "tap")
    __shared__ __attribute__((aligned(16))) unsigned char ys[TH * TW * WB];            // (32 KB; the cross-wave reduction reuses it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    constexpr unsigned ZERO_OFF = PH * PW * 16;
    if (tid < 4) reinterpret_cast<unsigned *>(xs + ZERO_OFF)[tid] = 0u;
    const int cbase = (int)blockIdx.y * 32;                       // this workgroup's slab of the wide map

    // per-lane addressing of the transposing reads (ds_read_b64_tr_b16: the 16 lanes of a group supply 4 pixels x 16 "channels" as
    // 8-byte pieces -- ANY sixteen addresses; lane li receives channel li of those 4 pixels).  k-step = 32 consecutive pixels of one
    // tile row; lane group g covers pixels 8 g ... 8 g + 7 (first read: + q, second: + 4 + q).
    //   wide:   channel block cb = channels 16 cb ... of the slab: piece = 4 channels (pp) of pixel 8 g + q.  The 32-byte halves of a
    //           pixel are swapped where bit 3 of its column is set (groups g and g + 1 would otherwise hit the same banks)
    //   narrow: column block nb = taps 2 nb, 2 nb + 1 x 8 channels: pieces pp = 0, 1 -> tap 2 nb (channels 4 pp ...), pp = 2, 3 -> tap 2 nb + 1
    unsigned aoff[2], boffw[NNB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) aoff[cb] = (unsigned)((8 * g + q) * WB + ((cb ^ (g & 1)) << 5) + pp * 8);
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) {
        const int tap = 2 * nb + (pp >> 1);
        boffw[nb] = tap < 9 ? (unsigned)(((tap / 3) * PW + tap % 3 + 8 * g + q) * 16 + (pp & 1) * 8) : 0xffffffffu;
    }
    f32x4 acc[2][NNB];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) acc[cb][nb] = f32x4{0, 0, 0, 0};

    u32x4 rx[NXS], ry[NYS];                                       // the next tile's pieces, fetched a tile ahead
    auto fetch = [&](int t) {
        const int tx = t % p.tiles_x, t2 = t / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
        const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
        for (int k = 0; k < NXS; ++k) {
            const int e = tid + k * NT, r = e / PW, c = e - r * PW, iy = y0 - 1 + r, ix = x0 - 1 + c;
            rx[k] = u32x4{0, 0, 0, 0};
            if (e < PH * PW && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                rx[k] = *reinterpret_cast<const u32x4 *>(nar + ((size_t)(n * p.H + iy) * p.W + ix) * p.ld_n);
        }
#pragma unroll
        for (int k = 0; k < NYS; ++k) {
            const int e = tid + k * NT, px = e >> 2, ch = e & 3, r = px / TW, c = px - r * TW, iy = y0 + r, ix = x0 + c;
            ry[k] = u32x4{0, 0, 0, 0};
            if (iy < p.H && ix < p.W && cbase + ch * 8 < p.wch)
                ry[k] = *reinterpret_cast<const u32x4 *>(wide + ((size_t)(n * p.H + iy) * p.W + ix) * p.ld_w + cbase + ch * 8);
        }
    };
    if ((int)blockIdx.x < p.ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        __syncthreads();                                          // (the previous tile's reads are done)
#pragma unroll
        for (int k = 0; k < NXS; ++k)
            if (tid + k * NT < PH * PW) *reinterpret_cast<u32x4 *>(xs + (tid + k * NT) * 16) = rx[k];
#pragma unroll
        for (int k = 0; k < NYS; ++k) {
            const int e = tid + k * NT, px = e >> 2, ch = e & 3;
            *reinterpret_cast<u32x4 *>(ys + px * WB + (((ch >> 1) ^ ((px >> 3) & 1)) << 5) + (ch & 1) * 16) = ry[k];
        }
        __syncthreads();
        if (t + (int)gridDim.x < p.ntiles) fetch(t + gridDim.x);
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {                              // this wave's four k-steps: rows wave, wave + 4, two halves each
            const int r = wave + 4 * (s >> 1), c0 = (s & 1) * 32;
            const unsigned char *yb = ys + (r * TW + c0) * WB, *xb = xs + (r * PW + c0) * 16;
            bf16x8 a[2], b[NNB];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(yb + aoff[cb]));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(yb + aoff[cb] + 4 * WB));
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                a[cb] = __builtin_bit_cast(bf16x8, u32x4{l2[0], l2[1], h2[0], h2[1]});
            }
#pragma unroll
            for (int nb = 0; nb < NNB; ++nb) {
                const bool z = boffw[nb] == 0xffffffffu;
                const unsigned char *b0 = z ? xs + ZERO_OFF : xb + boffw[nb], *b1 = z ? xs + ZERO_OFF : xb + boffw[nb] + 4 * 16;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)b0);
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)b1);
                const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
                b[nb] = __builtin_bit_cast(bf16x8, u32x4{l2[0], l2[1], h2[0], h2[1]});
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int nb = 0; nb < NNB; ++nb) acc[cb][nb] = mma16<DT>(a[cb], b[nb], acc[cb][nb]);
        }
    }
    // ---- the four waves' sums, in wave order, -> partial[slab][workgroup][wide channel][(tap, narrow channel)] ----------------------
    // acc[cb][nb][r]: row = wide channel 16 cb + 4 (lane >> 4) + r of the slab, column 16 nb + (lane & 15)
    float *red = reinterpret_cast<float *>(ys);                    // [wave][8 rows][80 columns]
    float *out = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * WG_OUT;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int half = 0; half < 2; ++half) {                    // rows 16 cb + 8 half ... + 7: lanes with (lane >> 5) == half
            __syncthreads();
            if ((lane >> 5) == half) {
#pragma unroll
                for (int nb = 0; nb < NNB; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave * 640 + ((4 * ((lane >> 4) & 1) + r) * NNB + nb) * 16 + (lane & 15)] = acc[cb][nb][r];
            }
            __syncthreads();
            for (int e = tid; e < 8 * NNB * 16; e += NT)
                out[(16 * cb + 8 * half) * (NNB * 16) + e] = ((red[e] + red[640 + e]) + red[1280 + e]) + red[1920 + e];
        }
}

// dW[co][tap][ci] += the workgroups' partials, in workgroup order.  A workgroup owns 16 consecutive elements; its 256 threads are 16
// partial groups x 16 elements: group k adds partials k, k + 16, ... in that order, then the 16 group sums are added in group order
// through LDS (one thread per element walking every partial in turn is a chain of several hundred dependent memory round trips).
template <bool SWAP>
__global__ __launch_bounds__(256) void narrow_wgrad_reduce_kernel(const float *__restrict__ partial, int nwg, int Co, int Ci, float *__restrict__ dw) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el, total = Co * 9 * Ci;
    float s = 0.f;
    if (i < total) {
        const int ci = i % Ci, tap = (i / Ci) % 9, co = i / (9 * Ci);
        const int wch = SWAP ? ci : co, nch = SWAP ? co : ci, tp = SWAP ? 8 - tap : tap;
        const float *src = partial + (size_t)(wch >> 5) * nwg * WG_OUT + (wch & 31) * (NNB * 16) + tp * 8 + nch;
        for (int k = grp; k < nwg; k += 16) s += src[(size_t)k * WG_OUT];
    }
    red[grp][el] = s;
    __syncthreads();
    if (grp == 0 && i < total) {
        float t = red[0][el];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][el];
        atomicAdd(dw + i, t);                                      // (other uses of the parameter may add from another stream: conv_wgrad_ring.hip)
    }
}

template <int CIN, int NCB>
int launch_fwd(const NParams &p, int dtype, const void *x, const void *w, const float *bias, void *y, hipStream_t st) {
    const dim3 grid(p.tiles_x * p.tiles_y * p.N);
    rdfc_note_kernel("narrow3x3_kernel<%d, %d, %d>", CIN, NCB, dtype);
    if (dtype == RDFC_F16)
        hipLaunchKernelGGL((narrow3x3_kernel<CIN, NCB, RDFC_F16>), grid, dim3(NT), 0, st, p, (const bf16_t *)x, (const bf16_t *)w, bias, y);
    else
        hipLaunchKernelGGL((narrow3x3_kernel<CIN, NCB, RDFC_BF16>), grid, dim3(NT), 0, st, p, (const bf16_t *)x, (const bf16_t *)w, bias, y);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's forward / input-gradient entry points; returns 1 if it took the launch.
int rdfc_try_conv_narrow(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out, hipStream_t st) {
    if (!rdfc_is16(d->dtype)) return 0;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->ph == 1 && d->pw == 1 && d->dh == 1 && d->dw == 1)) return 0;
    if (d->Ho != d->Hi || d->Wo != d->Wi || d->act != RDFC_ACT_NONE) return 0;
    if (!(Cin == 8 || Cin == 24) || Cout % 4 != 0 || Cout < 8 || Cout > 128 || (Cin == 24 && Cout > 32)) return 0;
    const int ld_in = transposed ? d->ldy : d->ldx, ld_out = transposed ? d->ldx : d->ldy;
    // 16-byte pieces of the gathered map, 16- / 8-byte stores of the result
    if (ld_in % 8 != 0 || (reinterpret_cast<size_t>(in) & 15) != 0 || (reinterpret_cast<size_t>(w) & 15) != 0) return 0;
    if (ld_out % 4 != 0 || (reinterpret_cast<size_t>(out) & 15) != 0) return 0;
    if ((double)d->N * d->Hi * d->Wi >= 2.0e9 / 128) return 0;
    NParams p{d->N, d->Hi, d->Wi, Cin, Cout, ld_in, ld_out, transposed, d->out_f32, cdiv(d->Wi, TW), cdiv(d->Hi, TH)};
    if (Cin == 24) return Cout <= 16 ? launch_fwd<24, 1>(p, d->dtype, in, w, bias, out, st) : launch_fwd<24, 2>(p, d->dtype, in, w, bias, out, st);
    if (Cout <= 16) return launch_fwd<8, 1>(p, d->dtype, in, w, bias, out, st);
    if (Cout <= 32) return launch_fwd<8, 2>(p, d->dtype, in, w, bias, out, st);
    if (Cout <= 64) return launch_fwd<8, 4>(p, d->dtype, in, w, bias, out, st);
    return launch_fwd<8, 8>(p, d->dtype, in, w, bias, out, st);
}

// Scratch for the per-workgroup partials: one buffer per (device, stream), as conv_wgrad_ring.hip's slabs (launches on one stream run
// one after the other); fixed size, allocated at the first use outside a capture.
namespace {
constexpr int NW_GRID = 768;                                       // workgroups per launch, all slabs together (three per CU: 43 KB of LDS each)
struct NWScratch {
    int dev;
    hipStream_t st;
    float *ptr;
};
NWScratch g_nw[64];
int g_nnw = 0;
std::mutex g_nw_mu;                                                // (the autograd engine's thread and the caller's may both launch weight gradients)
float *nw_scratch(hipStream_t st) {
    std::lock_guard<std::mutex> lock(g_nw_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < g_nnw; ++i)
        if (g_nw[i].dev == dev && g_nw[i].st == st) return g_nw[i].ptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone || g_nnw == 64) return nullptr;
    float *ptr = nullptr;
    if (hipMalloc(&ptr, (size_t)NW_GRID * WG_OUT * sizeof(float)) != hipSuccess) return nullptr;
    g_nw[g_nnw++] = NWScratch{dev, st, ptr};
    return ptr;
}
}  // namespace

// the shapes the narrow weight gradient takes (pointer alignment aside: every map of the train step is 16-byte aligned):
// 3x3 / stride 1 / pad 1 with Ci = 8 and Co <= 32 (conv_offset_aff), or Co = 8 and Ci a multiple of 32 up to 256 (gd_dec0)
int rdfc_wgrad_narrow_covers(const rdfc_conv_desc *d) {
    if (!rdfc_is16(d->dtype)) return 0;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->ph == 1 && d->pw == 1 && d->dh == 1 && d->dw == 1)) return 0;
    if (d->Ho != d->Hi || d->Wo != d->Wi) return 0;
    const bool a = d->Ci == 8 && d->Co % 8 == 0 && d->Co >= 8 && d->Co <= 32, b = d->Co == 8 && d->Ci % 32 == 0 && d->Ci >= 32 && d->Ci <= 256;
    if (!(a || b)) return 0;
    if (d->ldx % 8 != 0 || d->ldy % 8 != 0 || (double)d->N * d->Hi * d->Wi >= 2.0e9 / 256) return 0;
    return cdiv(d->Wi, TW) * cdiv(d->Hi, TH) * d->N >= 64;         // (small maps: the generic kernel's one launch is as good)
}

// Called by rdfc_conv2d_wgrad ahead of the generic kernels; returns 1 if it took the launch.
int rdfc_try_wgrad_narrow(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st) {
    if (!rdfc_wgrad_narrow_covers(d) || (reinterpret_cast<size_t>(x) & 15) != 0 || (reinterpret_cast<size_t>(dy) & 15) != 0) return 0;
    const bool swap = d->Ci != 8;                                  // the narrow map is dY
    NWParams p{d->N, d->Hi, d->Wi, swap ? d->ldy : d->ldx, swap ? d->ldx : d->ldy, swap ? d->Ci : d->Co, cdiv(d->Wi, TW), cdiv(d->Hi, TH), 0, d->Co, d->Ci};
    p.ntiles = p.tiles_x * p.tiles_y * d->N;
    float *scratch = nw_scratch(st);
    if (scratch == nullptr) return 0;
    const int nslab = swap ? d->Ci / 32 : 1;
    int nwg = NW_GRID / nslab;
    if (nwg > p.ntiles) nwg = p.ntiles;
    const dim3 grid(nwg, nslab);
    const void *nar = swap ? dy : x, *wide = swap ? x : dy;
    rdfc_note_kernel("narrow_wgrad_kernel<%d, %s>", d->dtype, swap ? "true" : "false");
#define RDFC_NW_LAUNCH(DT_, SW_)                                                                                                                  \
    hipLaunchKernelGGL((narrow_wgrad_kernel<DT_, SW_>), grid, dim3(NT), 0, st, p, (const bf16_t *)nar, (const bf16_t *)wide, scratch);            \
    hipLaunchKernelGGL((narrow_wgrad_reduce_kernel<SW_>), dim3(cdiv(d->Co * 9 * d->Ci, 16)), dim3(256), 0, st, (const float *)scratch, nwg, d->Co, d->Ci, dw);
    if (d->dtype == RDFC_F16) {
        if (swap) { RDFC_NW_LAUNCH(RDFC_F16, true) } else { RDFC_NW_LAUNCH(RDFC_F16, false) }
    } else {
        if (swap) { RDFC_NW_LAUNCH(RDFC_BF16, true) } else { RDFC_NW_LAUNCH(RDFC_BF16, false) }
    }
#undef RDFC_NW_LAUNCH
    return 1;
}
