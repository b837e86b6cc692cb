// conv_head_dgrad.hip -- input gradient of the C -> 1 output heads (3x3, stride 1: rgb_pred_dec0 / id_dec0 128 -> 1,
// rgb_conf_dec0 / cf_dec0 96 -> 1, rdf_generator.py:373-398) as a streaming kernel.
//
//   dX[n, y, x, c] = sum_{ky, kx} dY[n, y + pad - ky, x + pad - kx] * W[c][ky][kx]
//
// is a rank-1 update per tap: one 2-byte value of dY per pixel against C weights -- ~0 useful MFMA work, and the whole cost is
// writing the C-channel map once (284 MB at 16 x 228 x 304 x 128).  On the padded MFMA tile this ran at 155-160 us (0.9 TB/s
// of output); here a lane owns 8 channels (its 72 weights live in registers as f32), 64 / (C / 8) neighbouring pixels share a
// wave so that every store instruction writes whole contiguous pixel rows of 2 C bytes, and dY comes from an LDS tile with
// halo (nine 2-byte LDS reads per pixel group instead of nine global gathers).  HBM-bound: reads 2 B, writes 2 C B per pixel.
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int NT = 256, TH = 16, TW = 64;            // pixel tile per workgroup
constexpr int PH = TH + 2, PWD = TW + 2;

struct HDParams {
    int N, H, W, C, ld_out, pad;                     // H x W: both maps (stride 1, "same" geometry checked by the caller)
    int Hy, Wy;                                      // dY size
    int tiles_x, tiles_y;
};

template <int DT>
__global__ __launch_bounds__(NT) void head_dgrad_kernel(HDParams p, const bf16_t *__restrict__ dy, const bf16_t *__restrict__ wb,
                                                        bf16_t *__restrict__ dx) {
    __shared__ float tile[PH * PWD];
    int b = blockIdx.x;
    const int tx = b % p.tiles_x;
    b /= p.tiles_x;
    const int ty = b % p.tiles_y, n = b / p.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    // dY patch: rows y0 + pad - 2 .. y0 + pad + TH - 1, zero outside the map
    const bf16_t *src = dy + (size_t)n * p.Hy * p.Wy;
    for (int e = threadIdx.x; e < PH * PWD; e += NT) {
        const int r = e / PWD, c = e - r * PWD, iy = y0 + p.pad - 2 + r, ix = x0 + p.pad - 2 + c;
        tile[e] = ((unsigned)iy < (unsigned)p.Hy && (unsigned)ix < (unsigned)p.Wy) ? f32of<DT>(src[(size_t)iy * p.Wy + ix]) : 0.f;
    }
    const int chunks = p.C >> 3, ppw = 64 / chunks;                   // 8-channel chunks; pixels a wave covers per pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ck = lane % chunks, pl = lane / chunks;
    const bool lane_on = pl < ppw;
    float w[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int v = 0; v < 8; ++v) w[t][v] = lane_on ? f32of<DT>(wb[(size_t)(ck * 8 + v) * 9 + t]) : 0.f;
    __syncthreads();
    // a wave walks its rows of the tile (TH / 4 rows each), ppw pixels at a time
    for (int r = wave; r < TH; r += NT / 64) {
        const int y = y0 + r;
        if (y >= p.H) break;
        for (int xb = 0; xb < TW; xb += ppw) {
            const int xl = xb + pl, x = x0 + xl;
            if (!lane_on || xl >= TW || x >= p.W) continue;
            float acc[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[v] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float g = tile[(r + 2 - ky) * PWD + xl + 2 - kx];          // dY[y + pad - ky][x + pad - kx]
#pragma unroll
                    for (int v = 0; v < 8; ++v) acc[v] += g * w[ky * 3 + kx][v];
                }
            bf16x8 o;
#pragma unroll
            for (int v = 0; v < 8; ++v) o[v] = cvt16<DT>(acc[v]);
            *reinterpret_cast<bf16x8 *>(dx + ((size_t)(n * p.H + y) * p.W + x) * p.ld_out + ck * 8) = o;
        }
    }
}

}  // namespace

// Called by rdfc_conv2d_dgrad (conv_igemm.hip) first; returns 1 if it took the launch.
int rdfc_try_head_dgrad(const rdfc_conv_desc *d, const void *dy, const void *w_bwd, const float *bias, void *dx, hipStream_t st) {
    constexpr bool off = false;
    if (off || !rdfc_is16(d->dtype) || bias != nullptr || d->act != RDFC_ACT_NONE || d->out_f32) return 0;
    if (d->Co != 1 || d->ldy != 1 || d->KH != 3 || d->KW != 3 || d->sh != 1 || d->sw != 1 || d->dh != 1 || d->dw != 1) return 0;
    if (d->ph != d->pw || d->ph > 2) return 0;
    if (d->Ci % 8 != 0 || d->Ci < 16 || d->Ci > 512) return 0;         // 2 .. 64 eight-channel chunks per pixel
    if (d->ldx % 8 != 0 || d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return 0;
    HDParams p{d->N, d->Hi, d->Wi, d->Ci, d->ldx, d->ph, d->Ho, d->Wo, cdiv(d->Wi, TW), cdiv(d->Hi, TH)};
    if ((long)d->N * d->Hi * d->Wi < 65536) return 0;                  // small maps: the MFMA tile kernels' grids are fine
    dim3 grid(p.tiles_x * p.tiles_y * p.N);
    rdfc_note_kernel("head_dgrad_kernel<%d>", d->dtype);
    if (d->dtype == RDFC_F16) hipLaunchKernelGGL(head_dgrad_kernel<RDFC_F16>, grid, dim3(NT), 0, st, p, (const bf16_t *)dy, (const bf16_t *)w_bwd, (bf16_t *)dx);
    else hipLaunchKernelGGL(head_dgrad_kernel<RDFC_BF16>, grid, dim3(NT), 0, st, p, (const bf16_t *)dy, (const bf16_t *)w_bwd, (bf16_t *)dx);
    return 1;
}
