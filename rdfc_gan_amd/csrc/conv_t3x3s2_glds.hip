// conv_t3x3s2_glds.hip -- the TRANSPOSED 3x3 (and 4x4) / stride-2 / pad-1 convolution, staged by LDS-DMA with the source patch reused by all
// taps: the input gradient of the encoders' stride-2 convs (encoder_decoder.py:40-61 via common.py:29-44, resnet_generator.py:8-51)
// and the forward of the decoders' ConvTranspose2d(3, 2, 1, output_padding) (common.py:46-61).
//
//     r(y, x) = sum_{kh, kw} s((y + 1 - kh) / 2, (x + 1 - kw) / 2) . w[kh, kw]      over the taps that make both quotients integers
//
// igemm_glds_kernel walks the result per stride-parity class (even/even: one tap, ..., odd/odd: four) and gathers the source pixels of
// every tap again: 16 KB of pixel DMA per 64-MFMA step on top of the weights, four passes over the source map, K loops of 4-16 steps
// behind a 3-step prologue -- 190-200 us (0.09-0.12 of the peak) on the 64-channel full-resolution results.  Here a workgroup owns
// 8 x 16 SOURCE pixels and 64 result channels, i.e. all FOUR classes of the 16 x 32 result pixels above them:
//   * the 9 x 17 source patch of a 32-channel slab is staged once (10 KB) and every tap reads it at a shift of 0 or 1 pixel:
//     kh = 1 -> even result rows from source row q; kh = 0 -> odd rows from q + 1; kh = 2 -> odd rows from q (columns alike);
//   * a wave owns 64 co x 32 source px and keeps the four classes' accumulators (4 x 32 f32x4 = 128 registers); a phase is one
//     kernel row = three taps = 24 MFMAs per wave behind one barrier, weights (12 KB per phase) in a ring of two phases;
//   * per phase 12 KB of weights + a third of the patch = 15.4 KB for 96 MFMAs (160 B per MFMA; igemm_glds<64>: 320), the source is
//     read once (+ a 13 % halo) instead of four times.
// Pipeline rules as in conv_glds.hip (counted vmcnt + lgkmcnt(0) in front of a raw barrier, zero page for out-of-range pieces so
// that every wave issues the same number of DMAs).
#include "common.h"

#include <stdlib.h>

namespace {

struct TParams {
    int N, Hs, Ws, Hr, Wr;       // source map (the smaller one), result map
    int Cin, Cout;               // source channels (K), result channels
    int ld_in, ld_out;
    int act;
    float slope;
    int out_f32;
    int tiles_x, tiles_y;
    const void *addend;          // optional map added to the result before the store (rdfc_conv2d_dgrad_add)
    int ld_add;
    int row_store;               // epilogue through LDS: whole 128-byte channel rows per store instruction (common.h)
    const void *addend2;         // a second one (rdfc_conv2d_dgrad_add2: the alias cotangent AND a skip connection's)
    int ld_add2;
    unsigned x_bytes, w_bytes;   // extents of the source map and of the packed weights (buffer descriptors of the LDS-DMA loads; < 2^31)
};

constexpr int BK = 32, NT = 256, TCO = 64;
constexpr int TC = 16, TRS = 8, JR = 2, RING = 2;                                 // source tile; source rows per wave; weight ring (phases)
// KS = 3: taps kh = 0, 1, 2 feed result rows of parity 1, 0, 1 from source rows q + 1, q, q (halo: one row / column on the high side).
// KS = 4 (PatchGAN's 4x4 / stride-2 layers, patch_gan_discriminator.py:11-37): kh = 0 .. 3 -> parity 1, 0, 1, 0 from q + 1, q, q, q - 1
// (halo on both sides).  A phase is one kernel row = KS taps.
template <int KS> struct Geo {
    static constexpr int ORG = KS == 4 ? 1 : 0;                                   // patch origin = tile origin - ORG
    static constexpr int PW = TC + 1 + ORG, PH = TRS + 1 + ORG, NPATCH = PH * PW;
    static constexpr int NPIECE = NPATCH * 4, PATCH_BYTES = (NPIECE + 63) / 64 * 1024, NR = (NPIECE + NT - 1) / NT;   // 612 | 720 pieces, 10 | 12 KB, 3 rounds
    static constexpr int WPH = KS * TCO * BK * 2, NWP = KS;                       // weights of one phase: 12 | 16 KB, one DMA per thread and tap
    static constexpr int OFF_W = 2 * PATCH_BYTES, OFF_JUNK = OFF_W + RING * WPH, SMEM = OFF_JUNK + 1024;
    static_assert(NR == 3 && NR <= KS, "one patch round per phase of the previous slab");
};

// one LDS-DMA instruction through a buffer descriptor: lane l reads 16 bytes at base + soff + voff (voff per lane, soff wave-uniform)
// and lands at lds_base + 16 l; a lane whose voff is not below num_records arrives as zeros (conv_glds.hip: bdma16)
constexpr unsigned DEAD = 0x80000000u;
__device__ inline void bdma16(unsigned voff, u32x4 rsrc, unsigned soff, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
                 : "memory");
}
__device__ inline u32x4 make_rsrc(const void *base, unsigned bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
template <int DT, int KS>
__global__ __launch_bounds__(NT, 2) void convt3x3s2_glds_kernel(TParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                                const float *__restrict__ bias, void *__restrict__ yout,
                                                                const bf16_t *__restrict__ zero) {
    using G = Geo<KS>;
    constexpr int ORG = G::ORG, PW = G::PW, NPATCH = G::NPATCH, PATCH_BYTES = G::PATCH_BYTES, NR = G::NR, WPH = G::WPH, NWP = G::NWP;
    constexpr int OFF_W = G::OFF_W, OFF_JUNK = G::OFF_JUNK;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = (int)threadIdx.x;
    const int co0 = blockIdx.y * TCO;
    int t = blockIdx.x;
    {   // contiguous runs of tiles per XCD (conv_glds.hip)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7, k = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, n = t / p.tiles_y;
    const int y0 = ty * TRS, x0 = tx * TC;                       // source coordinates of the tile
    const int ncc = p.Cin / BK;

    // ---- per-thread DMA sources (computed once) --------------------------------------------------------------------------------
    // 32-bit byte offsets through buffer descriptors (conv_glds.hip): a piece outside the map or a channel row beyond Cout is DEAD -> zeros
    const u32x4 xsrc = make_rsrc(x, p.x_bytes), wrs = make_rsrc(w, p.w_bytes);
    unsigned poff[NR], pdst[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int e = r * NT + tid, q = e >> 2, slot = e & 3;
        poff[r] = DEAD;
        if (q < NPATCH) {
            const int py = q / PW, px = q - py * PW, iy = y0 - ORG + py, ix = x0 - ORG + px;
            if ((unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws)
                poff[r] = 2u * (unsigned)(((size_t)(n * p.Hs + iy) * p.Ws + ix) * p.ld_in + ((slot ^ ((q >> 1) & 2)) << 3));
        }
        const int wbase = (r * NT + wave * 64) * 16;
        pdst[r] = wbase + 1024 <= PATCH_BYTES ? (unsigned)wbase : 0xffffffffu;
    }
    unsigned woff[NWP];                                          // DMA k of a phase = tap kw = k, row co = tid >> 2
    {
        const int row = tid >> 2, slot = tid & 3;
#pragma unroll
        for (int k = 0; k < NWP; ++k)
            woff[k] = co0 + row < p.Cout ? 2u * (unsigned)((size_t)(co0 + row) * KS * KS * p.Cin + k * p.Cin + ((slot ^ ((row >> 1) & 2)) << 3)) : DEAD;
    }
    auto issue_w = [&](int it) {                                 // weights of global phase it = cc * KS + kh -> ring slot it % RING
        const int cc = it / KS, kh = it - cc * KS;
        u32x4 rs = wrs;
        if (cc >= ncc) rs[2] = 0;                                // (a phase beyond the last slab: zeros)
        const unsigned soff = 2u * (unsigned)(kh * KS * p.Cin + cc * BK);
#pragma unroll
        for (int k = 0; k < NWP; ++k) bdma16(woff[k], rs, soff, lds0 + OFF_W + (it % RING) * WPH + (k * NT + wave * 64) * 16);
    };
    auto issue_p = [&](int r, int cc) {                          // round r of slab cc -> patch buffer cc & 1
        const unsigned dst = pdst[r] == 0xffffffffu ? lds0 + OFF_JUNK : lds0 + (cc & 1) * PATCH_BYTES + pdst[r];
        u32x4 rs = xsrc;
        if (cc >= ncc) rs[2] = 0;
        bdma16(poff[r], rs, 2u * (unsigned)(cc * BK), dst);
    };
    auto issue_dummy = [&]() {                                   // (keeps the DMA count per phase static where a phase has no patch round)
        u32x4 rs = xsrc;
        rs[2] = 0;
        bdma16(DEAD, rs, 0u, lds0 + OFF_JUNK);
    };

    // ---- fragment coordinates -----------------------------------------------------------------------------------------------------
    int bpp[JR];
#pragma unroll
    for (int j = 0; j < JR; ++j) bpp[j] = (JR * wave + j + ORG) * PW + (lane & 15) + ORG;
    const int arow = lane & 15, kg = lane >> 4;
    const unsigned a_off = (unsigned)((arow << 6) | (kg << 4)) ^ (unsigned)((arow & 4) << 3);      // + 1024 per 16 rows (they share bit 2)

    f32x4 acc[4][4][JR];                                          // [class ry * 2 + rx][16-channel block][source row]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < JR; ++j) acc[c][i][j] = f32x4{0, 0, 0, 0};

    // ---- prologue: slab 0's patch, the first phase's weights --------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < NR; ++r) issue_p(r, 0);
    issue_w(0);

    // phase (cc, KH): kernel row KH feeds result rows of parity RY = (KH + 1) & 1 from source rows q + SY, SY = (RY + 1 - KH) / 2; columns alike
#define RDFC_T3S2_PHASE(KH)                                                                                     \
    if (KH < KS) {                                                                                              \
        constexpr int RY = (KH + 1) & 1, SY = (RY + 1 - KH) / 2;                                                 \
        if (KH == 0) wait_vm_and_reads<0>(); else wait_vm_and_reads<1>();   /* phase 0 reads a patch whose last round was the youngest DMA */ \
        __builtin_amdgcn_s_barrier();                                                                           \
        issue_w(it0 + KH + 1);                                                                                  \
        if (KH < NR) issue_p(KH < NR ? KH : 0, cc + 1);                                                         \
        else issue_dummy();                                /* (a fourth phase issues a dummy piece: the vmcnt arithmetic stays the same) */ \
        const unsigned char *Wt = smem + OFF_W + ((it0 + KH) % RING) * WPH + a_off;                              \
        const unsigned char *Pt = smem + (cc & 1) * PATCH_BYTES;                                                \
        bf16x8 b_[3][JR];                                      /* column shifts -1 (KS = 4 only), 0, +1 */      \
        _Pragma("unroll") for (int sx = (KS == 4 ? -1 : 0); sx <= 1; ++sx)                                      \
            _Pragma("unroll") for (int j = 0; j < JR; ++j) {                                                    \
                const int row = bpp[j] + SY * PW + sx;                                                          \
                b_[sx + 1][j] = *reinterpret_cast<const bf16x8 *>(Pt + (((row << 6) | (kg << 4)) ^ ((row & 4) << 3))); \
            }                                                                                                   \
        _Pragma("unroll") for (int kw = 0; kw < KS; ++kw) {                                                     \
            const int rx = (kw + 1) & 1, sx = (rx + 1 - kw) / 2;                                                \
            bf16x8 a_[4];                                                                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) a_[i] = *reinterpret_cast<const bf16x8 *>(Wt + (kw * 4 + i) * 1024); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
                _Pragma("unroll") for (int j = 0; j < JR; ++j) acc[RY * 2 + rx][i][j] = mma16<DT>(a_[i], b_[sx + 1][j], acc[RY * 2 + rx][i][j]); \
        }                                                                                                       \
    }
    for (int cc = 0; cc < ncc; ++cc) {
        const int it0 = cc * KS;
        RDFC_T3S2_PHASE(0)
        RDFC_T3S2_PHASE(1)
        RDFC_T3S2_PHASE(2)
        RDFC_T3S2_PHASE(3)
    }
#undef RDFC_T3S2_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: bias + activation (+ addend) + store, four consecutive result channels per lane --------------------------------
    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    float bv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * i + q4 + r;
            bv[i][r] = (bias != nullptr && co < p.Cout) ? bias[co] : 0.f;
        }
    auto store_all = [&](auto actf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ry = c >> 1, rx = c & 1;
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                const int oy = 2 * (y0 + JR * wave + j) + ry, ox = 2 * (x0 + (lane & 15)) + rx;
                if (oy >= p.Hr || ox >= p.Wr) continue;
                const size_t po = ((size_t)n * p.Hr + oy) * p.Wr + ox;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * i + q4;
                    if (co >= p.Cout) continue;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = actf(acc[c][i][j][r] + bv[i][r]);
                    auto add_map = [&](const void *a, int ld) {
                        const bf16_t *ad = reinterpret_cast<const bf16_t *>(a) + po * ld + co;
                        if ((ld & 3) == 0 && co + 3 < p.Cout && (reinterpret_cast<size_t>(a) & 7) == 0) {
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4 *>(ad);
                            for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t4[r]);
                        } else
                            for (int r = 0; r < 4; ++r)
                                if (co + r < p.Cout) v[r] += f32of<DT>(ad[r]);
                    };
                    if (p.addend != nullptr) add_map(p.addend, p.ld_add);
                    if (p.addend2 != nullptr) add_map(p.addend2, p.ld_add2);
                    const size_t off = po * p.ld_out + co;
                    if (p.out_f32) {
                        float *y = reinterpret_cast<float *>(yout) + off;
                        if (vec_ok && co + 3 < p.Cout) *reinterpret_cast<f32x4 *>(y) = f32x4{v[0], v[1], v[2], v[3]};
                        else
                            for (int r = 0; r < 4; ++r)
                                if (co + r < p.Cout) y[r] = v[r];
                    } else {
                        bf16_t *y = reinterpret_cast<bf16_t *>(yout) + off;
                        if (vec_ok && co + 3 < p.Cout) {
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
    };
    // ---- whole-row stores (common.h: rs_put / rs_flush): per result-row parity ry the wave's two column classes are 2 rows x 32
    //      CONSECUTIVE result pixels x 64 channels = one 64-pixel tile (tile pixel = 32 j + 2 col + rx) -----------------------------------
    if (p.row_store == 2) {
        // launches with addends (rdfc_conv2d_dgrad_add / _add2): f32 staging, one result row of the wave (32 consecutive pixels) per pass,
        // the addends fetched in whole 128-byte rows a pass ahead and added by the flush (common.h: rsf_put / rsf_fetch / rsf_flush); same
        // bits as the path below
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        const bf16_t *add1 = reinterpret_cast<const bf16_t *>(p.addend) + co0;
        const bf16_t *add2 = p.addend2 != nullptr ? reinterpret_cast<const bf16_t *>(p.addend2) + co0 : nullptr;
        auto ppix = [&](int pass, int px) -> long {                // pass = 2 ry + j: result row 2 (y0 + JR wave + j) + ry, pixels 2 x0 ...
            const int oy = 2 * (y0 + JR * wave + (pass & 1)) + (pass >> 1);
            if (oy >= p.Hr || 2 * x0 + px >= p.Wr) return -1;
            return ((long)n * p.Hr + oy) * p.Wr + 2 * x0 + px;
        };
        static_assert(JR == 2, "pass = 2 ry + j");
        RsfAdd c1, c2, n1, n2;
        rsf_fetch(c1, lane, add1, p.ld_add, p.Cout - co0, [&](int px) { return ppix(0, px); });
        if (add2 != nullptr) rsf_fetch(c2, lane, add2, p.ld_add2, p.Cout - co0, [&](int px) { return ppix(0, px); });
        __syncthreads();
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int ry = pass >> 1, j = pass & 1;
#pragma unroll
            for (int rx = 0; rx < 2; ++rx)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = acc[2 * ry + rx][i][j][r] + bv[i][r];
                        v[r] = u > 0.f ? u : u * ns;
                    }
                    rsf_put(Ot, 2 * (lane & 15) + rx, 16 * i + q4, v);
                }
            if (pass < 3) {
                rsf_fetch(n1, lane, add1, p.ld_add, p.Cout - co0, [&](int px) { return ppix(pass < 3 ? pass + 1 : pass, px); });
                if (add2 != nullptr) rsf_fetch(n2, lane, add2, p.ld_add2, p.Cout - co0, [&](int px) { return ppix(pass < 3 ? pass + 1 : pass, px); });
            }
            rsf_flush<DT>(Ot, lane, reinterpret_cast<bf16_t *>(yout) + co0, p.ld_out, c1, add2 != nullptr ? &c2 : nullptr, p.Cout - co0,
                          [&](int px) { return ppix(pass, px); });
            c1 = n1;
            c2 = n2;
        }
        return;
    }
    if (p.row_store) {
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        __syncthreads();                                           // every wave is done with the operand tiles
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
#pragma unroll
            for (int rx = 0; rx < 2; ++rx) {
                const int c = 2 * ry + rx;
#pragma unroll
                for (int j = 0; j < JR; ++j) {
                    const int oy = 2 * (y0 + JR * wave + j) + ry, ox = 2 * (x0 + (lane & 15)) + rx;
                    const bool live = oy < p.Hr && ox < p.Wr;
                    const size_t po = ((size_t)n * p.Hr + oy) * p.Wr + ox;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = co0 + 16 * i + q4;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float u = acc[c][i][j][r] + bv[i][r];
                            v[r] = u > 0.f ? u : u * ns;
                        }
                        if (p.addend != nullptr && live && co < p.Cout) {          // (Cout and ld_add are multiples of 8 / 4 on this path)
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const bf16_t *>(p.addend) + po * p.ld_add + co);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t4[r]);
                        }
                        if (p.addend2 != nullptr && live && co < p.Cout) {
                            const bf16x4 t4 = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const bf16_t *>(p.addend2) + po * p.ld_add2 + co);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t4[r]);
                        }
                        rs_put<DT>(Ot, 32 * j + 2 * (lane & 15) + rx, 16 * i + q4, v);
                    }
                }
            }
            rs_flush(Ot, lane, reinterpret_cast<bf16_t *>(yout) + co0, p.Cout - co0, [&](int px) -> long {
                const int j = px >> 5, oy = 2 * (y0 + JR * wave + j) + ry, ox = 2 * x0 + (px & 31);
                if (oy >= p.Hr || ox >= p.Wr) return -1;
                return (((long)n * p.Hr + oy) * p.Wr + ox) * p.ld_out;
            });
        }
        return;
    }
    with_act(p.act, p.slope, store_all);
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

template <int DT, int KS>
int launch(const TParams &p, const void *x, const void *w, const float *bias, void *y, const bf16_t *zero, hipStream_t st) {
    constexpr int SMEM = Geo<KS>::SMEM;
    static_assert(SMEM >= 4 * RS_TILE, "the row-store epilogue stages one tile per wave in the operand buffers");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)convt3x3s2_glds_kernel<DT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Cout, TCO));
    rdfc_note_kernel("convt3x3s2_glds_kernel<%d, %d>", DT, KS);
    hipLaunchKernelGGL((convt3x3s2_glds_kernel<DT, KS>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)w, bias, y, zero);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's input-gradient entry points ahead of igemm_glds_kernel.  d describes the FORWARD stride-2 conv (result of
// this kernel = d's input map, source = d's output map); returns 1 if it took the launch.  RDFC_T3S2=0 switches it off (A/B), 1: 3x3 only,
// 3: also the launches too small to fill the chip (tests).
int rdfc_try_convt3x3s2_glds(const rdfc_conv_desc *d, const void *src, const void *w_bwd, const float *bias, void *res, hipStream_t st,
                             const void *addend = nullptr, int ld_add = 0, const void *addend2 = nullptr, int ld_add2 = 0) {
    static const int on = getenv("RDFC_T3S2") ? atoi(getenv("RDFC_T3S2")) : 2;
    if (!on || !rdfc_is16(d->dtype)) return 0;
    const int KS = d->KH;
    if (!((KS == 3 || KS == 4) && d->KW == KS && d->sh == 2 && d->sw == 2 && d->ph == 1 && d->pw == 1 && d->dh == 1 && d->dw == 1)) return 0;
    if (KS == 4 && on < 2) return 0;                           // RDFC_T3S2=1: 3x3 only (A/B)
    if (d->Hi + 2 < KS || d->Wi + 2 < KS || d->Ho != (d->Hi + 2 - KS) / 2 + 1 || d->Wo != (d->Wi + 2 - KS) / 2 + 1) return 0;
    if (d->Co % BK != 0 || d->Ci % 64 != 0 || d->ldy % 8 != 0) return 0;
    // (small launches stay on igemm_glds' flat 128-pixel tiles: 8 x 16 source tiles x 64 channels leave most CUs empty there --
    //  256->512 @ 16x30x38: 46 vs 41 us, the 4x4 128->256 @ 8x57x76: 32 vs 28 us)
    if (on < 3 && cdiv(d->Wo, TC) * cdiv(d->Ho, TRS) * d->N * cdiv(d->Ci, TCO) < 384) return 0;
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    TParams p{d->N, d->Ho, d->Wo, d->Hi, d->Wi, d->Co, d->Ci, d->ldy, d->ldx, d->act, d->slope, d->out_f32, cdiv(d->Wo, TC), cdiv(d->Ho, TRS),
              addend, ld_add, 0, addend2, ld_add2, 0, 0};
    {
        const size_t xb = 2 * (size_t)p.N * p.Hs * p.Ws * p.ld_in, wb_ = 2 * (size_t)p.Cout * KS * KS * p.Cin;
        if (xb >= 0x7ffffff0u || wb_ >= 0x7ffffff0u) return 0;                  // the loaders address both through 32-bit byte offsets
        p.x_bytes = (unsigned)xb;
        p.w_bytes = (unsigned)wb_;
    }
    p.row_store = !d->out_f32 && p.Cout % 8 == 0 && p.ld_out % 8 == 0 && (reinterpret_cast<size_t>(res) & 15) == 0 &&
                  (addend == nullptr || (ld_add % 4 == 0 && (reinterpret_cast<size_t>(addend) & 7) == 0)) &&
                  (addend2 == nullptr || (ld_add2 % 4 == 0 && (reinterpret_cast<size_t>(addend2) & 7) == 0)) &&
                  d->act != RDFC_ACT_TANH && d->act != RDFC_ACT_SIGMOID;
    if (p.row_store && rsf_ok(addend, ld_add, addend2, ld_add2)) p.row_store = 2;     // the addends in whole rows, from the flush (common.h)
    if (KS == 4) return d->dtype == RDFC_F16 ? launch<RDFC_F16, 4>(p, src, w_bwd, bias, res, zero, st) : launch<RDFC_BF16, 4>(p, src, w_bwd, bias, res, zero, st);
    return d->dtype == RDFC_F16 ? launch<RDFC_F16, 3>(p, src, w_bwd, bias, res, zero, st) : launch<RDFC_BF16, 3>(p, src, w_bwd, bias, res, zero, st);
}
