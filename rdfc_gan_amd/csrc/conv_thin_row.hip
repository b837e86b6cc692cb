// conv_thin_row.hip -- wide-tap convolutions that produce a few channels from a wide map (bf16 / f16): ResnetGenerator's
// 7x7 output layer 64 -> 3 (resnet_generator.py:50-52) at full resolution.
//
// conv_thin.hip's tile kernel gives every pixel to a lane and walks the 49 taps with packed dot products: 4 704 v_dot2 per
// pixel, weights through scalar loads that share a counter with the LDS reads -- 376 us for 16 x 228 x 304 pixels, 0.4 TB/s.
// The reduction is deep (K = 49 x 64) but the output is 3 channels wide, so a plain MFMA tile would be 3/16 full.  Here the
// conv is split along kx instead:
//     T_ky[y', x', (co, kx)] = sum over c of X[y', x', c] * W[co][ky][kx][c]            (a 1x1 conv to Cout*KW <= 32 channels: MFMA,
//                                                                                      two 16-row tiles, 21/32 of them real)
//     out[y, x, co]         = sum over ky, kx of T_ky[y + ky, x + kx, (co, kx)]          (shift-and-add out of LDS: KW reads per co)
// per workgroup: a 16x16 output tile, the (16+KH-1) x (16+KW-1) input patch of a 32-channel chunk in LDS (as in thin_tile), for
// each ky one pass of 2 x 22 MFMAs (16 x 22 positions of that patch row band) into an LDS table T, then every lane adds its
// pixel's KW x Cout entries.  MFMAs per pixel: 2.4 instead of the 6.1 a tap-by-tap MFMA tiling needs, B fragments 5x fewer.
#include "common.h"

#include <stdlib.h>

namespace {

struct RowParams {
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, oy, ox;                // patch origin = tile origin + (oy, ox)
    int ld_in, ld_out, act, out_f32;
    float slope;
    int tiles_x, tiles_y, PH, PW, row_bytes, NG;     // NG: groups of 16 positions in a 16-row band of the patch (16 * PW / 16)
};

constexpr int CH = 32, PIX = CH * 2 + 16;          // channel chunk, bytes per patch pixel (padded: conflict-free b128 reads)

// K: square kernel extent, CO: output channels (compile-time: every LDS offset below is an immediate -- with run-time extents the
// address arithmetic alone was 1 200 VALU instructions per wave and kernel row, 4 cycles each on a wave64)
template <int K, int CO, int DT>
__global__ __launch_bounds__(256) void thin_row_kernel(RowParams p, const bf16_t *__restrict__ in, const bf16_t *__restrict__ w,
                                                       const float *__restrict__ bias, void *__restrict__ out) {
    constexpr int PW = 16 + K - 1, PH = 16 + K - 1, NR = CO * K, T9 = K * K;
    constexpr int NG = PW;                                 // groups of 16 positions in a 16-row band of the patch
    constexpr int GW = (NG + 3) / 4;                       // groups per wave
    // T[(co, kx)][position]: a lane's shift-add reads walk consecutive positions (consecutive banks).  TP = 4 mod 8: the four
    // k-groups of a store (rows 4 kg + i, 4 TP floats apart) land on the two halves of the banks -- 2 lanes per bank, the minimum
    constexpr int TP = 16 * PW + 4;
    constexpr int ROWB = (PW * PIX + 255) / 256 * 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *T = reinterpret_cast<float *>(smem + PH * ROWB);            // [NR][TP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx0 = (b % p.tiles_x) * 16;
    b /= p.tiles_x;
    const int ty0 = (b % p.tiles_y) * 16, n = b / p.tiles_y;
    const int ty = tid >> 4, tx = tid & 15;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;

    // this lane's A rows: n0 = lane & 15 (tile 0), n0 + 16 (tile 1) -> (co, kx); k group = lane >> 4
    const int kg = lane >> 4, li = lane & 15;
    int wrow[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int nn = m * 16 + li;
        wrow[m] = nn < NR ? (nn / K) * T9 + nn % K : -1;            // + ky * K selects the kernel row
    }
    // this wave's position groups: LDS byte offset of the B fragment (kernel row 0) and the T column, once
    int boff[GW], tcol[GW];
#pragma unroll
    for (int j = 0; j < GW; ++j) {
        const int g = wave + 4 * j, pos = g * 16 + li, r = pos / PW, col = pos - r * PW;
        boff[j] = g < NG ? r * ROWB + col * PIX + kg * 16 : -1;
        tcol[j] = pos + 4 * kg * TP;
    }
    const float *trow = T + ty * PW + tx;

    for (int c0 = 0; c0 < p.Cin; c0 += CH) {
        // A fragments of every kernel row of this chunk: straight from global (the whole table is a few KB, L1 / L2 resident),
        // issued ahead of the patch staging so that they are back when the first pass starts
        bf16x8 a[K][2];
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                u32x4 v{0u, 0u, 0u, 0u};
                if (wrow[m] >= 0) v = *reinterpret_cast<const u32x4 *>(w + (size_t)(wrow[m] + ky * K) * p.Cin + c0 + kg * 8);
                a[ky][m] = __builtin_bit_cast(bf16x8, v);
            }
        __syncthreads();                                              // (the previous chunk's last T reads / patch reads are done)
        // patch of this chunk: every 16-byte piece (pixel, 8-channel vector) is one load; all of a thread's loads are issued before
        // the first LDS store (a load -> store chain per row costs a memory latency per row)
        {
            constexpr int NPIECE = PH * PW * 4, SP = (NPIECE + 255) / 256;
            u32x4 val[SP];
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                const int e = k * 256 + tid;
                val[k] = u32x4{0u, 0u, 0u, 0u};
                if (e < NPIECE) {
                    const int pix = e >> 2, v = e & 3, row = pix / PW, col = pix - row * PW;
                    const int gy = ty0 + p.oy + row, gx = tx0 + p.ox + col;
                    if ((unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win)
                        val[k] = *reinterpret_cast<const u32x4 *>(in + ((size_t)(n * p.Hin + gy) * p.Win + gx) * p.ld_in + c0 + v * 8);
                }
            }
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                const int e = k * 256 + tid;
                if (e < NPIECE) {
                    const int pix = e >> 2, v = e & 3, row = pix / PW, col = pix - row * PW;
                    *reinterpret_cast<u32x4 *>(smem + row * ROWB + col * PIX + v * 16) = val[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            // T[(co, kx)][pos] for the 16 x PW positions of patch rows ky .. ky+15
#pragma unroll
            for (int j = 0; j < GW; ++j) {
                if (boff[j] < 0) continue;
                const u32x4 xv = *reinterpret_cast<const u32x4 *>(smem + boff[j] + ky * ROWB);
                const bf16x8 bfrag = __builtin_bit_cast(bf16x8, xv);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (m * 16 >= NR) continue;
                    const f32x4 d = mma16<DT>(a[ky][m], bfrag, f32x4{0.f, 0.f, 0.f, 0.f});   // rows 4*kg .. 4*kg+3 of tile m, column pos
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (m * 16 + 4 * kg + i < NR) T[tcol[j] + (m * 16 + i) * TP] = d[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                float s = 0.f;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) s += trow[(co * K + kx) * TP + kx];
                acc[co] += s;
            }
            __syncthreads();                                          // T is overwritten by the next kernel row
        }
    }
    const int yo = ty0 + ty, xo = tx0 + tx;
    if (yo < p.Hout && xo < p.Wout) {
        const size_t off = ((size_t)(n * p.Hout + yo) * p.Wout + xo) * p.ld_out;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = acc[co];
            if (bias != nullptr) v += bias[co];
            v = act_fwd(v, p.act, p.slope);
            if (p.out_f32) reinterpret_cast<float *>(out)[off + co] = v;
            else reinterpret_cast<bf16_t *>(out)[off + co] = cvt16<DT>(v);
        }
    }
}

template <int K, int CO>
int launch_row(const RowParams &p, int dtype, const void *in, const void *w, const float *bias, void *out, hipStream_t st) {
    constexpr int PW = 16 + K - 1, ROWB = (PW * PIX + 255) / 256 * 256, TP = 16 * PW + 4;
    constexpr size_t smem = (size_t)PW * ROWB + (size_t)CO * K * TP * 4;
    static_assert(smem <= 80 * 1024, "two workgroups per CU");
    static bool attr[2] = {false, false};
    const int slot = dtype == RDFC_F16;
    const void *fn = dtype == RDFC_F16 ? (const void *)thin_row_kernel<K, CO, RDFC_F16> : (const void *)thin_row_kernel<K, CO, RDFC_BF16>;
    if (!attr[slot]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return 0;
        attr[slot] = true;
    }
    rdfc_note_kernel("thin_row_kernel<%d, %d, %d>", K, CO, dtype);
    const dim3 grid(p.tiles_x * p.tiles_y * p.N);
    if (dtype == RDFC_F16)
        hipLaunchKernelGGL((thin_row_kernel<K, CO, RDFC_F16>), grid, dim3(256), smem, st, p, (const bf16_t *)in, (const bf16_t *)w, bias, out);
    else
        hipLaunchKernelGGL((thin_row_kernel<K, CO, RDFC_BF16>), grid, dim3(256), smem, st, p, (const bf16_t *)in, (const bf16_t *)w, bias, out);
    return 1;
}

}  // namespace

// Called by rdfc_try_conv_thin (conv_thin.hip) ahead of its tile kernel.  Forward form only (stride 1, dilation 1, square kernel).
int rdfc_try_conv_thin_row(const rdfc_conv_desc *d, const void *in, const void *w, const float *bias, void *out, hipStream_t st) {
    constexpr bool off = false;
    if (off || !rdfc_is16(d->dtype)) return 0;
    if (d->KH != d->KW || d->Co > 4 || d->Ci % CH != 0 || d->ldx % 8 != 0) return 0;
    if (d->sh != 1 || d->sw != 1 || d->dh != 1 || d->dw != 1) return 0;
    if ((long)d->N * d->Ho * d->Wo < 65536) return 0;
    RowParams p;
    p.N = d->N; p.Hin = d->Hi; p.Win = d->Wi; p.Cin = d->Ci; p.Hout = d->Ho; p.Wout = d->Wo; p.Cout = d->Co;
    p.KH = d->KH; p.KW = d->KW; p.oy = -d->ph; p.ox = -d->pw;
    p.ld_in = d->ldx; p.ld_out = d->ldy; p.act = d->act; p.out_f32 = d->out_f32; p.slope = d->slope;
    p.tiles_x = cdiv(d->Wo, 16); p.tiles_y = cdiv(d->Ho, 16);
    p.PH = p.PW = p.row_bytes = p.NG = 0;
    // the wide kernels of the model and of the tests (narrow ones, K < 4, stay on the padded MFMA tiles / the tile kernel)
    const int key = d->KW * 10 + d->Co;
    switch (key) {
        case 73: return launch_row<7, 3>(p, d->dtype, in, w, bias, out, st);
        case 74: return launch_row<7, 4>(p, d->dtype, in, w, bias, out, st);
        case 71: return launch_row<7, 1>(p, d->dtype, in, w, bias, out, st);
        case 52: return launch_row<5, 2>(p, d->dtype, in, w, bias, out, st);
        case 41: return launch_row<4, 1>(p, d->dtype, in, w, bias, out, st);
        default: return 0;
    }
}
