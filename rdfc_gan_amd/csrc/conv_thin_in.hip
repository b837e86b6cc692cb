// conv_thin_in.hip -- convolutions that READ at most 4 channels and produce a wide map (2-byte elements, MFMA).
//
// The 1- / 3-channel input layers (RDFGenerator's first convs rdf_generator.py:286-292, ResnetGenerator's 7x7 stem
// resnet_generator.py:19-22, PatchGAN's first 4x4 stride-2 conv patch_gan_discriminator.py:22-26) and the input-gradients of the
// 1- / 3-channel output layers (7x7 output conv resnet_generator.py:48-51):
//     out[p][co] = sum over k = (tap, c) of  T[p*s + r(tap)][c] * w[co][k],      K = taps * ct <= 160.
// The arithmetic is nothing (K <= 160 against 2 * Cout bytes written per pixel): the layer is bound by WRITING its wide output, and what
// has to be avoided is latency -- round 3's kernel staged a 16 x 16 tile's patch with a loop of dependent 2-byte loads between two
// workgroup barriers and ran the 3 <- 64 7x7 input gradient at 0.46 TB/s (324 us for 142 MB), the stems at 0.8 TB/s.
//
// Round 4: every WAVE works alone.  A wave owns tasks of 4 rows x 16 columns of output pixels; it keeps the task's halo patch of the
// thin map ((3 s + KH) x (15 s + KW) pixels, 0.6-2 KB) in its own slice of LDS, fetches the NEXT task's patch into registers before it
// computes the current one (so the global latency sits under the gathers / MFMAs / stores), and shares only the weight matrix
// [Cout][K] (staged once per workgroup) with the other waves -- no barrier after the prologue, 16 waves per CU at different points of
// their tasks.  B fragments (8 consecutive k of one pixel per lane) are gathered from the patch with 16-bit LDS reads through a
// per-lane offset table, A fragments are plain ds_read_b128.
#include "common.h"

#include <stdlib.h>

namespace {

struct TIParams {
    int N, Ht, Wt, ct, ldt;        // thin source map
    int Ho, Wo, Cout, ld_out;      // produced map
    int KH, KW, s;                 // taps, stride of the source position per output pixel
    int oy, ox, flip;              // patch origin = tile origin * s + (oy, ox); flip: tap (ky,kx) reads patch (KH-1-ky, KW-1-kx)
    int K, KS;                     // real K = KH*KW*ct, k-steps of 32
    int PH, PW;                    // patch extent in pixels (per wave task)
    int tiles_x, tiles_y, ntasks;
    int act, out_f32;
    float slope;
};

constexpr int TROWS = 4, TCOLS = 16;        // output pixels of a wave task
constexpr int MAXE = 16;                    // patch elements per lane (64 * 16 = 1024 elements = 2 KB per wave)
constexpr int MAXKS = 5;

template <int MF, int DT = RDFC_BF16>   // output-channel fragments per wave: Cout <= 16 * MF
__global__ __launch_bounds__(256, 2) void thin_in_kernel(TIParams p, const bf16_t *__restrict__ t, const bf16_t *__restrict__ w,
                                                      const float *__restrict__ bias, void *__restrict__ yout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int KP = p.KS * 32, WLD = KP + 8;                       // padded weight row (elements)
    bf16_t *Wl = reinterpret_cast<bf16_t *>(smem);                 // [16*MF][WLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (provably wave-uniform: task geometry stays in SGPRs)
    unsigned short *Kt = reinterpret_cast<unsigned short *>(Wl + 16 * MF * WLD);                               // [MAXKS][4][8] gather offsets
    unsigned short *Tl = Kt + MAXKS * 32 + wave * (64 * MAXE + 8);                                            // this wave's patch

    // ---- weights: w[co][K] -> Wl[co][KP], zero padded (once per workgroup; eight independent loads in flight per thread: every load is
    //      UNCONDITIONAL from a clamped address and masked afterwards -- a load under a per-element condition makes hipcc branch around it
    //      and wait for it before the next one, cdna_hip_programming.md section 5 trap (c): one dependent round trip per element) ------------
    for (int e0 = tid; e0 < 16 * MF * KP; e0 += 256 * 8) {
        bf16_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * 256, co = e / KP, k = e - co * KP;
            const bool ok = e < 16 * MF * KP && co < p.Cout && k < p.K;
            const bf16_t x = w[ok ? (size_t)co * p.K + k : 0];
            v[q] = ok ? x : (bf16_t)0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + q * 256, co = e / KP, k = e - co * KP;
            if (e < 16 * MF * KP) Wl[co * WLD + k] = v[q];
        }
    }
    // ---- gather offsets: element offset of k inside the patch, table [ks][k-group of 8][8] (a lane reads its 8 with one ds_read_b128) ---
    const int prow = p.PW * p.ct;                                  // elements per patch row
    for (int k = tid; k < MAXKS * 32; k += 256) {
        int o = 0;
        if (k < p.K) {
            const int tap = k / p.ct, c = k - tap * p.ct, ky = tap / p.KW, kx = tap - ky * p.KW;
            const int pr = p.flip ? p.KH - 1 - ky : ky, pc = p.flip ? p.KW - 1 - kx : kx;
            o = pr * prow + pc * p.ct + c;
        }
        Kt[k] = (unsigned short)o;                                 // k >= K: any valid element (its weight column is zero)
    }
    __syncthreads();                                               // the only workgroup barrier
    // ---- per-lane staging geometry: patch element e = lane + 64 i -> (row, column, channel) packed, task independent ---------------
    const int ne = p.PH * prow;                                    // <= 64 * MAXE (host)
    int sgeo[MAXE];                                                // row << 16 | col << 2 | c; row = 0x7fff: no element
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int e = lane + 64 * i;
        const int row = e / prow, rem = e - row * prow, col = rem / p.ct, c = rem - col * p.ct;
        sgeo[i] = e < ne ? (row << 16 | col << 2 | c) : (0x7fff << 16);
    }
    const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
    const bool fast_store = !p.out_f32 && (p.Cout & 15) == 0 && (p.ld_out & 3) == 0 && ns >= 0.f && ns <= 1.f && (reinterpret_cast<size_t>(yout) & 7) == 0;
    const int q4 = (lane >> 4) * 4;
    float bv[MF][4];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 16 * i + q4 + r;
            bv[i][r] = (bias != nullptr && co < p.Cout) ? bias[co] : 0.f;
        }
    const unsigned short *tsrc = reinterpret_cast<const unsigned short *>(t);

    const int wstride = gridDim.x * 4;
    int task = blockIdx.x * 4 + wave;
    unsigned okm = 0;
    unsigned pre[MAXE];                                            // (one register per element: packing two per register made hipcc wait for every load before the next)
    auto fetch = [&](int tk) {                                     // the patch of task tk -> registers (loads stay in flight)
        int tt = tk;
        const int tx = tt % p.tiles_x;
        tt /= p.tiles_x;
        const int ty = tt % p.tiles_y, n = tt / p.tiles_y;
        const int gy0 = ty * TROWS * p.s + p.oy, gx0 = tx * TCOLS * p.s + p.ox;
        const unsigned short *img = tsrc + (long)n * p.Ht * p.Wt * p.ldt;
        // three separate passes -- offsets, loads, masks -- with the scheduler fenced between them: left alone hipcc re-uses one register
        // for every load and waits for each before it issues the next (16 dependent round trips per task, the whole kernel's time)
        int off[MAXE];
        okm = 0;
#pragma unroll
        for (int i = 0; i < MAXE; ++i) {
            const int row = sgeo[i] >> 16, col = (sgeo[i] >> 2) & 0x3fff, c = sgeo[i] & 3;
            const int gy = gy0 + row, gx = gx0 + col;
            if ((unsigned)gy < (unsigned)p.Ht && (unsigned)gx < (unsigned)p.Wt) okm |= 1u << i;
            const int cy = min(max(gy, 0), p.Ht - 1), cx = min(max(gx, 0), p.Wt - 1);
            off[i] = (cy * p.Wt + cx) * p.ldt + c;                 // (the map has < 2^31 elements: host)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MAXE; ++i) pre[i] = img[off[i]];       // unconditional, from the clamped position
        __builtin_amdgcn_sched_barrier(0);                         // (the out-of-image mask okm is applied when the values are written to LDS, a task later)
    };
    if (task < p.ntasks) fetch(task);
    for (; task < p.ntasks; task += wstride) {
        int tt = task;
        const int tx = tt % p.tiles_x;
        tt /= p.tiles_x;
        const int ty = tt % p.tiles_y, n = tt / p.tiles_y;
        const int y0 = ty * TROWS, x0 = tx * TCOLS;
        // registers -> this wave's LDS patch (the previous task's gathers were issued before these stores: same wave, in order)
#pragma unroll
        for (int i = 0; i < MAXE; ++i)
            if (lane + 64 * i < ne) Tl[lane + 64 * i] = (okm >> i) & 1 ? (unsigned short)pre[i] : (unsigned short)0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (task + wstride < p.ntasks) fetch(task + wstride);      // next patch: in flight under this task's work
#pragma unroll 1
        for (int j = 0; j < TROWS; ++j) {
            f32x4 acc[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) acc[i] = f32x4{bv[i][0], bv[i][1], bv[i][2], bv[i][3]};     // the bias is the accumulators' start value (PMC: 65 vector instructions per MFMA, most of them this epilogue)
            const int pbase = (j * p.s * p.PW + (lane & 15) * p.s) * p.ct;
#pragma unroll 1
            for (int ks = 0; ks < p.KS; ++ks) {
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
                const u16x8 ko = *reinterpret_cast<const u16x8 *>(Kt + ks * 32 + 8 * (lane >> 4));
                s16x8 bvv;
#pragma unroll
                for (int e = 0; e < 8; ++e) bvv[e] = (short)Tl[pbase + ko[e]];
                const bf16x8 b = __builtin_bit_cast(bf16x8, bvv);
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(Wl + (16 * i + (lane & 15)) * WLD + ks * 32 + 8 * (lane >> 4));
                    acc[i] = mma16<DT>(a, b, acc[i]);
                }
            }
            // ---- epilogue: lane holds channels 16 i + q4 .. +3 of pixel (row y0 + j, col x0 + (lane & 15)) ------------------
            const int oy = y0 + j, ox = x0 + (lane & 15);
            if (oy >= p.Ho || ox >= p.Wo) continue;
            const size_t po = ((size_t)(n * p.Ho + oy) * p.Wo + ox) * p.ld_out + q4;
            if (fast_store) {              // 2-byte elements, every fragment's four channels inside Cout, 8-byte aligned rows: one address per row
                bf16_t *y = reinterpret_cast<bf16_t *>(yout) + po;
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    if (16 * i >= p.Cout) continue;
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = cvt16<DT>(fmaxf(acc[i][r], acc[i][r] * ns));          // 0 <= ns <= 1: max(u, ns u) = leaky / plain ReLU / identity
                    *reinterpret_cast<bf16x4 *>(y + 16 * i) = o;
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int co = 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[i][r] > 0.f ? acc[i][r] : acc[i][r] * ns;
                if (p.out_f32) {
                    float *y = reinterpret_cast<float *>(yout) + po - q4 + co;
                    if ((p.ld_out & 3) == 0 && co + 3 < p.Cout) *reinterpret_cast<f32x4 *>(y) = f32x4{v[0], v[1], v[2], v[3]};
                    else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = v[r];
                } else {
                    bf16_t *y = reinterpret_cast<bf16_t *>(yout) + po - q4 + co;
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
        __builtin_amdgcn_wave_barrier();                           // (compiler: the gathers above stay above the next task's patch stores)
    }
}

template <int MF>
int launch(const TIParams &p, const void *t, const void *w, const float *bias, void *y, hipStream_t st, int dtype) {
    const int KP = p.KS * 32, WLD = KP + 8;
    const size_t smem = (size_t)16 * MF * WLD * 2 + (size_t)MAXKS * 32 * 2 + (size_t)4 * (64 * MAXE + 8) * 2;
    if (smem > 64 * 1024) return 0;
    const int wtasks = cdiv(p.ntasks, 4);
    int blocks = wtasks < 768 ? wtasks : 768;                      // persistent: three workgroups per CU (the weights are staged once per workgroup)
    rdfc_note_kernel("thin_in_kernel<%d, %d>", MF, dtype);
    if (dtype == RDFC_F16) hipLaunchKernelGGL((thin_in_kernel<MF, RDFC_F16>), dim3(blocks), dim3(256), smem, st, p, (const bf16_t *)t, (const bf16_t *)w, bias, y);
    else hipLaunchKernelGGL((thin_in_kernel<MF, RDFC_BF16>), dim3(blocks), dim3(256), smem, st, p, (const bf16_t *)t, (const bf16_t *)w, bias, y);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's entry points. Returns 1 if it took the launch.
int rdfc_try_conv_thin_in(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                          hipStream_t st) {
    constexpr bool off = false;
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
    if (p.KS > MAXKS) return 0;
    p.PH = (TROWS - 1) * p.s + d->KH;
    p.PW = (TCOLS - 1) * p.s + d->KW;
    if (p.PH * p.PW * p.ct > 64 * MAXE) return 0;
    if ((double)p.Ht * p.Wt * p.ldt >= 2.0e9) return 0;           // (per-image element offsets are 32-bit)
    p.tiles_x = cdiv(p.Wo, TCOLS);
    p.tiles_y = cdiv(p.Ho, TROWS);
    p.ntasks = p.tiles_x * p.tiles_y * d->N;
    if (p.ntasks < 256) return 0;
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
