// conv_thin_in.hip -- convolutions that READ at most 4 channels and produce a wide map (bf16, MFMA).
//
// The 1- / 3-channel input layers (RDFGenerator's first convs, ResnetGenerator's 7x7 stem, PatchGAN's first 4x4
// stride-2 conv) and the input-gradients of the 1- / 3-channel output layers (heads, 7x7 output conv):
//     out[p][co] = sum over k = (tap, c) of  T[p*s + r(tap)][c] * w[co][k],      K = taps * ct <= 160.
// On the generic kernel the K loop gathers every (pixel, k) element with a scalar 2-byte global load.  Here a
// workgroup stages the thin map's halo patch of a 16x16 output tile (a few KB) and the whole weight matrix
// [Cout][K] in LDS; B fragments (8 consecutive k of one pixel per lane) are gathered from the patch with 16-bit
// LDS reads through a per-lane offset table, A fragments are plain ds_read_b128.  Workgroups are persistent
// (grid-stride over tiles) so the weights are staged once.  The layer is bound by writing its wide output.
#include "common.h"

#include <stdlib.h>

namespace {

struct TIParams {
    int N, Ht, Wt, ct, ldt;        // thin source map
    int Ho, Wo, Cout, ld_out;      // produced map
    int KH, KW, s;                 // taps, stride of the source position per output pixel
    int oy, ox, flip;              // patch origin = tile origin * s + (oy, ox); flip: tap (ky,kx) reads patch (KH-1-ky, KW-1-kx)
    int K, KS;                     // real K = KH*KW*ct, k-steps of 32
    int PH, PW;                    // patch extent in pixels
    int tiles_x, tiles_y, ntiles;
    int act, out_f32;
    float slope;
};

template <int MF, int DT = RDFC_BF16>   // output-channel fragments per wave: Cout <= 16 * MF
__global__ __launch_bounds__(256) void thin_in_kernel(TIParams p, const bf16_t *__restrict__ t, const bf16_t *__restrict__ w,
                                                      const float *__restrict__ bias, void *__restrict__ yout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KP = p.KS * 32, WLD = KP + 8;                       // padded weight row (elements)
    bf16_t *Wl = reinterpret_cast<bf16_t *>(smem);                 // [16*MF][WLD]
    unsigned short *Tl = reinterpret_cast<unsigned short *>(Wl + 16 * MF * WLD);   // [PH*PW*ct] (+8 slack)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- weights: w[co][K] -> Wl[co][KP], zero padded ---------------------------------------------------------------
    for (int e = tid; e < 16 * MF * KP; e += 256) {
        const int co = e / KP, k = e - co * KP;
        bf16_t v = (bf16_t)0.f;
        if (co < p.Cout && k < p.K) v = w[(size_t)co * p.K + k];
        Wl[co * WLD + k] = v;
    }
    // ---- per-lane gather offsets: element offset of k inside the patch, for k = 32 ks + 8 (lane>>4) + j --------------
    constexpr int MAXKS = 5;
    int koff[MAXKS][8];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 32 + 8 * (lane >> 4) + j;
            int o = 0;
            if (ks < p.KS && k < p.K) {
                const int tap = k / p.ct, c = k - tap * p.ct, ky = tap / p.KW, kx = tap - ky * p.KW;
                const int pr = p.flip ? p.KH - 1 - ky : ky, pc = p.flip ? p.KW - 1 - kx : kx;
                o = (pr * p.PW + pc) * p.ct + c;
            }
            koff[ks][j] = o;                                       // k >= K: any valid element (its weight column is zero)
        }
    const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
    const int q4 = (lane >> 4) * 4;
    float bv[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * i + q4 + r;
            bv[i][r] = (bias != nullptr && co < p.Cout) ? bias[co] : 0.f;
        }
    const int prow = p.PW * p.ct;                                  // elements per patch row

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int tt = tile;
        const int tx = tt % p.tiles_x;
        tt /= p.tiles_x;
        const int ty = tt % p.tiles_y, n = tt / p.tiles_y;
        const int y0 = ty * 16, x0 = tx * 16;
        __syncthreads();                                           // previous tile's gathers are done (and Wl is written)
        // ---- stage the patch: rows of PW*ct contiguous elements ---------------------------------------------------
        for (int row = wave; row < p.PH; row += 4) {
            const int gy = y0 * p.s + p.oy + row;
            const bool rok = (unsigned)gy < (unsigned)p.Ht;
            const int gx0 = x0 * p.s + p.ox;
            const unsigned short *src = reinterpret_cast<const unsigned short *>(t) + ((size_t)(n * p.Ht + (rok ? gy : 0)) * p.Wt) * p.ldt;
            for (int e = lane; e < prow; e += 64) {
                const int col = e / p.ct, c = e - col * p.ct, gx = gx0 + col;
                unsigned short v = 0;
                if (rok && (unsigned)gx < (unsigned)p.Wt) v = src[(size_t)gx * p.ldt + c];
                Tl[row * prow + e] = v;
            }
        }
        __syncthreads();

        f32x4 acc[MF][4];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        int pbase[4];                                              // patch element of this lane's pixel in row-fragment j
#pragma unroll
        for (int j = 0; j < 4; ++j) pbase[j] = ((4 * wave + j) * p.s * p.PW + (lane & 15) * p.s) * p.ct;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < p.KS) {
                bf16x8 a[MF];
#pragma unroll
                for (int i = 0; i < MF; ++i)
                    a[i] = *reinterpret_cast<const bf16x8 *>(Wl + (16 * i + (lane & 15)) * WLD + ks * 32 + 8 * (lane >> 4));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    s16x8 bvv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bvv[e] = (short)Tl[pbase[j] + koff[ks][e]];
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bvv);
#pragma unroll
                    for (int i = 0; i < MF; ++i) acc[i][j] = mma16<DT>(a[i], b, acc[i][j]);
                }
            }
        }
        // ---- epilogue: lane holds channels 16 i + q4 .. +3 of pixel (row 4*wave + j, col lane & 15) ------------------
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = y0 + 4 * wave + j, ox = x0 + (lane & 15);
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const size_t po = ((size_t)(n * p.Ho + oy) * p.Wo + ox) * p.ld_out;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int co = 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = acc[i][j][r] + bv[i][r];
                    v[r] = u > 0.f ? u : u * ns;
                }
                if (p.out_f32) {
                    float *y = reinterpret_cast<float *>(yout) + po + co;
                    if ((p.ld_out & 3) == 0 && co + 3 < p.Cout) *reinterpret_cast<f32x4 *>(y) = f32x4{v[0], v[1], v[2], v[3]};
                    else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = v[r];
                } else {
                    bf16_t *y = reinterpret_cast<bf16_t *>(yout) + po + co;
                    if ((p.ld_out & 3) == 0 && co + 3 < p.Cout) {
                        bf16x4 o;
                        for (int r = 0; r < 4; ++r) o[r] = cvt16<DT>(v[r]);
                        *reinterpret_cast<bf16x4 *>(y) = o;
                    } else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = cvt16<DT>(v[r]);
                }
            }
        }
    }
}

template <int MF>
int launch(const TIParams &p, const void *t, const void *w, const float *bias, void *y, hipStream_t st, int dtype) {
    const int KP = p.KS * 32, WLD = KP + 8;
    const size_t smem = (size_t)16 * MF * WLD * 2 + ((size_t)p.PH * p.PW * p.ct + 8) * 2;
    if (smem > 64 * 1024) return 0;
    int blocks = p.ntiles < 1024 ? p.ntiles : 1024;
    rdfc_note_kernel("thin_in_kernel<%d, %d>", MF, dtype);
    if (dtype == RDFC_F16) hipLaunchKernelGGL((thin_in_kernel<MF, RDFC_F16>), dim3(blocks), dim3(256), smem, st, p, (const bf16_t *)t, (const bf16_t *)w, bias, y);
    else hipLaunchKernelGGL((thin_in_kernel<MF, RDFC_BF16>), dim3(blocks), dim3(256), smem, st, p, (const bf16_t *)t, (const bf16_t *)w, bias, y);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's entry points. Returns 1 if it took the launch.
int rdfc_try_conv_thin_in(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                          hipStream_t st) {
    static const bool off = getenv("RDFC_NO_THIN_IN") != nullptr;
    if (off || !rdfc_is16(d->dtype)) return 0;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    if (Cin > 4 || Cout < 16 || Cout > 128) return 0;
    if (d->act == RDFC_ACT_TANH || d->act == RDFC_ACT_SIGMOID) return 0;
    if (d->dh != 1 || d->dw != 1 || d->sh != d->sw) return 0;
    if (transposed && d->sh != 1) return 0;
    TIParams p;
    p.N = d->N;
    p.Ht = transposed ? d->Ho : d->Hi;
    p.Wt = transposed ? d->Wo : d->Wi;
    p.ct = Cin;
    p.ldt = transposed ? d->ldy : d->ldx;
    p.Ho = transposed ? d->Hi : d->Ho;
    p.Wo = transposed ? d->Wi : d->Wo;
    p.Cout = Cout;
    p.ld_out = transposed ? d->ldx : d->ldy;
    p.KH = d->KH;
    p.KW = d->KW;
    p.s = transposed ? 1 : d->sh;
    p.oy = transposed ? d->ph - (d->KH - 1) : -d->ph;
    p.ox = transposed ? d->pw - (d->KW - 1) : -d->pw;
    p.flip = transposed;
    p.K = d->KH * d->KW * Cin;
    p.KS = (p.K + 31) / 32;
    if (p.KS > 5) return 0;
    p.PH = 15 * p.s + d->KH;
    p.PW = 15 * p.s + d->KW;
    p.tiles_x = cdiv(p.Wo, 16);
    p.tiles_y = cdiv(p.Ho, 16);
    p.ntiles = p.tiles_x * p.tiles_y * d->N;
    if (p.ntiles < 64) return 0;
    // A/B on one device (tools/thin_in_bench.sh): wins on the 7x7 layers (3<-64 input gradient 138 vs 236 us, 1->64 stem
    // 85 vs 112 us) and on 16-channel outputs (30 vs 35 us); the 3x3 / 4x4 layers with 48-128 output channels are a tie
    // or slightly behind the generic kernel (per-tile latency: stage, barrier, compute, store do not overlap yet)
    static const bool all = getenv("RDFC_THIN_IN_ALL") != nullptr;
    if (!all && d->KH * d->KW <= 16 && Cout > 16) return 0;
    p.act = d->act;
    p.out_f32 = d->out_f32;
    p.slope = d->slope;
    if (Cout <= 16) return launch<1>(p, in, w, bias, out, st, d->dtype);
    if (Cout <= 32) return launch<2>(p, in, w, bias, out, st, d->dtype);
    if (Cout <= 48) return launch<3>(p, in, w, bias, out, st, d->dtype);
    if (Cout <= 64) return launch<4>(p, in, w, bias, out, st, d->dtype);
    if (Cout <= 96) return launch<6>(p, in, w, bias, out, st, d->dtype);
    return launch<8>(p, in, w, bias, out, st, d->dtype);
}
