// conv_thin.hip -- convolutions that produce at most 4 channels from a wide map (bf16).
//
// The 1-channel heads (rdf_generator.py:214-241), PatchGAN's last 4x4 layer (patchgan_discriminator.py:39-41),
// ResnetGenerator's 7x7 output layer (resnet_generator.py:50-52) and the input-gradients of the 1- / 3-channel
// input layers have M = Cout <= 4: on the MFMA kernels 3/4 .. 15/16 of every tile is padding and, for the small
// maps, the grid has a few dozen workgroups.  They are reductions over (tap, channel), so here one wave owns one
// output pixel: lanes stride over the (tap, 8-channel vector) items with v_dot2c_f32_bf16, a 64-lane butterfly
// finishes the sum.  Weights and the per-item (offset, tap) table sit in LDS for the whole workgroup;
// workgroups grid-stride over pixels.  HBM / L2 bound: every input vector is fetched KH*KW times from L2, once
// from HBM.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct ThinParams {
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, sh, sw;
    int oy, ox, ty, tx;      // source pixel = (o*s + oy + ky*ty, o*s + ox + kx*tx)
    int ld_in, ld_out, act, out_f32;
    float slope;
    int CV, items, items_pad, P;
};

constexpr int WAVES = 4;

typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
template <int DT> __device__ inline float dot2(unsigned a, unsigned b, float c) {       // c + a.lo*b.lo + a.hi*b.hi (bf16 | f16 pairs)
    if constexpr (DT == RDFC_F16) return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
    else return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

template <int CO, int DT = RDFC_BF16>
__global__ __launch_bounds__(WAVES * 64) void thin_out_kernel(ThinParams p, const bf16_t *__restrict__ in,
                                                              const bf16_t *__restrict__ w, const float *__restrict__ bias,
                                                              void *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int2 *tab = reinterpret_cast<int2 *>(smem);                                  // [items_pad]
    u32x4 *wl = reinterpret_cast<u32x4 *>(smem + (size_t)p.items_pad * 8);       // [CO][items_pad] 16-B vectors
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int it = tid; it < p.items_pad; it += WAVES * 64) {
        int2 e{0, -1};
        if (it < p.items) {
            const int tap = it / p.CV, cv = it - tap * p.CV;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            e.x = (ky * p.ty * p.Win + kx * p.tx) * p.ld_in + cv * 8;
            e.y = (ky << 16) | kx;
        }
        tab[it] = e;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            u32x4 v{0u, 0u, 0u, 0u};
            if (it < p.items && co < p.Cout) v = *reinterpret_cast<const u32x4 *>(w + ((size_t)co * p.items + it) * 8);
            wl[co * p.items_pad + it] = v;
        }
    }
    __syncthreads();

    const int niter = p.items_pad >> 6;
    for (int px = blockIdx.x * WAVES + wave; px < p.P; px += gridDim.x * WAVES) {
        const int pu = __builtin_amdgcn_readfirstlane(px);
        const int n = pu / (p.Hout * p.Wout), r = pu - n * p.Hout * p.Wout;
        const int yo = r / p.Wout, xo = r - yo * p.Wout;
        const int y0 = yo * p.sh + p.oy, x0 = xo * p.sw + p.ox;
        const long base = ((long)(n * p.Hin + y0) * p.Win + x0) * p.ld_in;
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll 2
        for (int i = 0; i < niter; ++i) {
            const int item = i * 64 + lane;
            const int2 e = tab[item];
            const int ky = e.y >> 16, kx = e.y & 0xffff;
            const int iy = y0 + ky * p.ty, ix = x0 + kx * p.tx;
            const bool ok = e.y >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const u32x4 raw = *reinterpret_cast<const u32x4 *>(in + (ok ? base + e.x : 0));
            // (scalars on purpose: __builtin_bit_cast of an ext-vector element reference reads element 0 on ROCm 7.2)
            const unsigned x0 = ok ? raw.x : 0u, x1 = ok ? raw.y : 0u, x2 = ok ? raw.z : 0u, x3 = ok ? raw.w : 0u;
#pragma unroll
            for (int co = 0; co < CO; ++co) {
                const u32x4 wv = wl[co * p.items_pad + item];
                const unsigned w0 = wv.x, w1 = wv.y, w2 = wv.z, w3 = wv.w;
                acc[co] = dot2<DT>(x0, w0, acc[co]);
                acc[co] = dot2<DT>(x1, w1, acc[co]);
                acc[co] = dot2<DT>(x2, w2, acc[co]);
                acc[co] = dot2<DT>(x3, w3, acc[co]);
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[co] += __shfl_xor(acc[co], off, 64);
        }
        if (lane < p.Cout) {
            float v = acc[0];
#pragma unroll
            for (int co = 1; co < CO; ++co) v = lane == co ? acc[co] : v;
            if (bias != nullptr) v += bias[lane];
            v = act_fwd(v, p.act, p.slope);
            const size_t off = (size_t)pu * p.ld_out + lane;
            if (p.out_f32) reinterpret_cast<float *>(out)[off] = v;
            else reinterpret_cast<bf16_t *>(out)[off] = cvt16<DT>(v);
        }
    }
}

// ---- large maps: one lane per output pixel, the input patch of a 16x16 tile in LDS -----------------------------
// (stride 1, dilation 1).  Per 32- (or 16-) channel chunk the (16+KH-1) x (16+KW-1) patch is staged once; each lane
// walks the taps reading its pixel's vectors with ds_read_b128 (pixel stride 64+16 B / 32+16 B and rows padded to
// 256 B keep every lane group conflict-free) while the weights, uniform across the wave, come through scalar loads.
struct TileParams {
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, oy, ox, flip;          // patch origin = tile origin + (oy, ox); flip: tap (ky,kx) reads patch (KH-1-ky, KW-1-kx)
    int ld_in, ld_out, act, out_f32;
    float slope;
    int tiles_x, tiles_y, PH, PW, row_bytes;
};

template <int CO, int CH, int DT = RDFC_BF16>
__global__ __launch_bounds__(256) void thin_tile_kernel(TileParams p, const bf16_t *__restrict__ in, const bf16_t *__restrict__ w,
                                                        const float *__restrict__ bias, void *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int VPC = CH / 8, PIX = CH * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x;
    const int tx0 = (b % p.tiles_x) * 16;
    b /= p.tiles_x;
    const int ty0 = (b % p.tiles_y) * 16, n = b / p.tiles_y;
    const int ty = tid >> 4, tx = tid & 15;
    const int T = p.KH * p.KW;
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;

    for (int c0 = 0; c0 < p.Cin; c0 += CH) {
        __syncthreads();
        for (int row = wave; row < p.PH; row += 4) {
            const int gy = ty0 + p.oy + row;
            const bool rok = (unsigned)gy < (unsigned)p.Hin;
            const bf16_t *src = in + ((size_t)(n * p.Hin + (rok ? gy : 0)) * p.Win) * p.ld_in + c0;
            unsigned char *dst = smem + row * p.row_bytes;
            for (int e = lane; e < p.PW * VPC; e += 64) {
                const int col = e / VPC, v = e % VPC;
                const int gx = tx0 + p.ox + col;
                u32x4 val{0u, 0u, 0u, 0u};
                if (rok && (unsigned)gx < (unsigned)p.Win) val = *reinterpret_cast<const u32x4 *>(src + (size_t)gx * p.ld_in + v * 8);
                *reinterpret_cast<u32x4 *>(dst + col * PIX + v * 16) = val;
            }
        }
        __syncthreads();
        for (int ky = 0; ky < p.KH; ++ky) {
            const int pr = ty + (p.flip ? p.KH - 1 - ky : ky);
            for (int kx = 0; kx < p.KW; ++kx) {
                const int pc = tx + (p.flip ? p.KW - 1 - kx : kx);
                const u32x4 *xp = reinterpret_cast<const u32x4 *>(smem + pr * p.row_bytes + pc * PIX);
                u32x4 xv[VPC];
#pragma unroll
                for (int v = 0; v < VPC; ++v) xv[v] = xp[v];
                const int tap = ky * p.KW + kx;
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    if (co < p.Cout) {
                        const u32x4 *wp = reinterpret_cast<const u32x4 *>(w + ((size_t)co * T + tap) * p.Cin + c0);   // wave-uniform
#pragma unroll
                        for (int v = 0; v < VPC; ++v) {
                            const u32x4 wv = wp[v];
                            acc[co] = dot2<DT>(xv[v].x, wv.x, acc[co]);
                            acc[co] = dot2<DT>(xv[v].y, wv.y, acc[co]);
                            acc[co] = dot2<DT>(xv[v].z, wv.z, acc[co]);
                            acc[co] = dot2<DT>(xv[v].w, wv.w, acc[co]);
                        }
                    }
                }
            }
        }
    }
    const int yo = ty0 + ty, xo = tx0 + tx;
    if (yo < p.Hout && xo < p.Wout) {
        const size_t off = ((size_t)(n * p.Hout + yo) * p.Wout + xo) * p.ld_out;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            if (co < p.Cout) {
                float v = acc[co];
                if (bias != nullptr) v += bias[co];
                v = act_fwd(v, p.act, p.slope);
                if (p.out_f32) reinterpret_cast<float *>(out)[off + co] = v;
                else reinterpret_cast<bf16_t *>(out)[off + co] = cvt16<DT>(v);
            }
        }
    }
}

template <int CO, int CH>
void launch_tile(TileParams p, const void *in, const void *w, const float *bias, void *out, hipStream_t st, int dtype) {
    constexpr int PIX = CH * 2 + 16;
    p.row_bytes = (p.PW * PIX + 255) / 256 * 256;
    const size_t smem = (size_t)p.PH * p.row_bytes;
    rdfc_note_kernel("thin_tile_kernel<%d, %d, %d>", CO, CH, dtype);
    if (dtype == RDFC_F16)
        hipLaunchKernelGGL((thin_tile_kernel<CO, CH, RDFC_F16>), dim3(p.tiles_x * p.tiles_y * p.N), dim3(256), smem, st, p, (const bf16_t *)in,
                           (const bf16_t *)w, bias, out);
    else
        hipLaunchKernelGGL((thin_tile_kernel<CO, CH, RDFC_BF16>), dim3(p.tiles_x * p.tiles_y * p.N), dim3(256), smem, st, p, (const bf16_t *)in,
                           (const bf16_t *)w, bias, out);
}

template <int CO>
void launch(const ThinParams &p, const void *in, const void *w, const float *bias, void *out, hipStream_t st, int dtype) {
    const size_t smem = (size_t)p.items_pad * 8 + (size_t)CO * p.items_pad * 16;
    int blocks = cdiv(p.P, WAVES);
    if (blocks > 256 * 16) blocks = 256 * 16;
    rdfc_note_kernel("thin_out_kernel<%d, %d>", CO, dtype);
    if (dtype == RDFC_F16)
        hipLaunchKernelGGL((thin_out_kernel<CO, RDFC_F16>), dim3(blocks), dim3(WAVES * 64), smem, st, p, (const bf16_t *)in, (const bf16_t *)w, bias, out);
    else
        hipLaunchKernelGGL((thin_out_kernel<CO, RDFC_BF16>), dim3(blocks), dim3(WAVES * 64), smem, st, p, (const bf16_t *)in, (const bf16_t *)w, bias, out);
}

}  // namespace

// Called by conv_igemm.hip's entry points before the MFMA kernels. Returns 1 if it took the launch.
int rdfc_try_conv_thin_row(const rdfc_conv_desc *d, const void *in, const void *w, const float *bias, void *out, hipStream_t st);   // conv_thin_row.hip

int rdfc_try_conv_thin(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                       hipStream_t st) {
    constexpr bool off = false;
    if (off || !rdfc_is16(d->dtype)) return 0;
    if (!transposed && rdfc_try_conv_thin_row(d, in, w, bias, out, st)) return 1;      // wide kernels (7x7 -> 3 channels): row GEMM + shift-add
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    const int ld_in = transposed ? d->ldy : d->ldx, ld_out = transposed ? d->ldx : d->ldy;
    if (Cout > 4 || Cin % 8 != 0 || ld_in % 8 != 0) return 0;
    if (transposed && (d->sh != 1 || d->sw != 1)) return 0;
    ThinParams p;
    p.N = d->N;
    p.Hin = transposed ? d->Ho : d->Hi;
    p.Win = transposed ? d->Wo : d->Wi;
    p.Hout = transposed ? d->Hi : d->Ho;
    p.Wout = transposed ? d->Wi : d->Wo;
    p.Cin = Cin;
    p.Cout = Cout;
    p.KH = d->KH;
    p.KW = d->KW;
    p.sh = transposed ? 1 : d->sh;
    p.sw = transposed ? 1 : d->sw;
    p.oy = transposed ? d->ph : -d->ph;
    p.ox = transposed ? d->pw : -d->pw;
    p.ty = transposed ? -d->dh : d->dh;
    p.tx = transposed ? -d->dw : d->dw;
    p.ld_in = ld_in;
    p.ld_out = ld_out;
    p.act = d->act;
    p.out_f32 = d->out_f32;
    p.slope = d->slope;
    p.CV = Cin / 8;
    p.items = d->KH * d->KW * p.CV;
    p.items_pad = (p.items + 63) / 64 * 64;
    p.P = d->N * p.Hout * p.Wout;
    const int co_t = Cout <= 1 ? 1 : (Cout <= 2 ? 2 : 4);
    constexpr long tile_min = 65536;
    // A/B on one device (tools/thin_bench.sh): the tile kernel wins on the 7x7 layers (64->3 fwd 215 vs 309 us, 1<-64 dgrad
    // 99 vs 315 us) and on narrow inputs (1<-16 dgrad 12 vs 129 us); the 3x3 layers with >= 64 input channels stay on the
    // padded MFMA tiles (128->1: 46 vs 62 us)
    constexpr bool tile_all = false;
    const bool tile_pays = tile_all || d->KH * d->KW > 9 || Cin < 64;
    if (p.P >= tile_min && !tile_pays) return 0;
    if (p.P >= tile_min && d->sh == 1 && d->sw == 1 && d->dh == 1 && d->dw == 1 && Cin % 16 == 0) {
        TileParams t;
        t.N = d->N; t.Hin = p.Hin; t.Win = p.Win; t.Cin = Cin; t.Hout = p.Hout; t.Wout = p.Wout; t.Cout = Cout;
        t.KH = d->KH; t.KW = d->KW; t.flip = transposed;
        t.oy = transposed ? d->ph - (d->KH - 1) : -d->ph;
        t.ox = transposed ? d->pw - (d->KW - 1) : -d->pw;
        t.ld_in = ld_in; t.ld_out = ld_out; t.act = d->act; t.out_f32 = d->out_f32; t.slope = d->slope;
        t.tiles_x = cdiv(p.Wout, 16); t.tiles_y = cdiv(p.Hout, 16);
        t.PH = 16 + d->KH - 1; t.PW = 16 + d->KW - 1; t.row_bytes = 0;
        const bool c32 = Cin % 32 == 0;
        if ((size_t)t.PH * ((t.PW * (c32 ? 80 : 48) + 255) / 256 * 256) <= 64 * 1024) {
            if (co_t == 1) c32 ? launch_tile<1, 32>(t, in, w, bias, out, st, d->dtype) : launch_tile<1, 16>(t, in, w, bias, out, st, d->dtype);
            else if (co_t == 2) c32 ? launch_tile<2, 32>(t, in, w, bias, out, st, d->dtype) : launch_tile<2, 16>(t, in, w, bias, out, st, d->dtype);
            else c32 ? launch_tile<4, 32>(t, in, w, bias, out, st, d->dtype) : launch_tile<4, 16>(t, in, w, bias, out, st, d->dtype);
            return 1;
        }
    }
    if ((size_t)p.items_pad * 8 + (size_t)co_t * p.items_pad * 16 > 64 * 1024) return 0;
    // one wave per pixel only pays on small maps (grid of the MFMA kernels too small); larger ones stay on MFMA
    if (p.P >= tile_min) return 0;
    if (co_t == 1) launch<1>(p, in, w, bias, out, st, d->dtype);
    else if (co_t == 2) launch<2>(p, in, w, bias, out, st, d->dtype);
    else launch<4>(p, in, w, bias, out, st, d->dtype);
    return 1;
}
