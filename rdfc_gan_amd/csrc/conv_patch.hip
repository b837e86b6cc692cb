// conv_patch.hip -- 3x3 / stride 1 / pad 1 convolution (forward and input-gradient) as an
// implicit GEMM whose activation operand is staged ONCE per 32-channel slab as a 2-D patch in LDS
// and re-read nine times (one shifted window per tap).
//
// Why: 3x3 s1 convolutions are ~85 % of the generator's FLOPs.  In the generic kernel
// (conv_igemm.hip) every tap re-fetches its (pixels x 32 ch) tile through the vector L1 -- 16-20 KB
// per 16-MFMA step, i.e. the CU's 64 B/clk L1 path saturates before the matrix pipe does.  Here a
// workgroup owns a TR x TC pixel rectangle (TR*TC = TPX); per channel slab it loads the
// (TR+2) x (TC+2) halo patch (coalesced 16-byte pixel-major rows, zero-filled outside the image) and
// the nine taps become nine LDS address offsets.  Global->LDS traffic per MFMA step drops ~2-3x
// (weights: TCO x 32 per tap; activations: patch / 9).
//
// Same MFMA orientation, LDS padding, epilogue and numerics as conv_igemm.hip (bit-identical
// accumulation order per output: taps outer, channels inner is replaced by channels outer, taps
// inner -- f32 accumulation, so results agree to rounding, not bitwise, with the generic kernel).
#include "common.h"

#include <stdlib.h>

namespace {

struct PatchParams {
    int N, Hin, Win, H, W, Cin, Cout;   // gathered map Hin x Win, produced map H x W (stride 1)
    int ld_in, ld_out;
    int transposed;             // 1: input-gradient form (tap offset 2-k inside the patch instead of k)
    int oy0, ox0;               // patch origin relative to the tile origin: -pad (forward) or pad-2 (transposed)
    int act;
    float slope;
    int out_f32;
    int Krow;                   // 9 * Cin
    int TR, TC;                 // tile rows / cols (TR*TC == TPX, TC % 16 == 0)
    int tiles_x, tiles_y;
};

constexpr int BK = 32;

// LDS tile layout: [row][32 k] with 16-byte chunks.
//   bf16: 64-byte rows, NO padding, chunk index XOR-swizzled with bit 2 of the row:  chunk' = chunk ^ ((row >> 1) & 2).
//         ds_read_b128 fragment reads (16 consecutive rows x one chunk per 16-lane group) are then bank-conflict
//         free for every starting row (brute-forced over all offsets; the padded 80-byte rows used before were a
//         2-way conflict on every read: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).
//   f32 (parity mode): 144-byte padded rows, no swizzle.
template <typename T> struct Lay {          // 2-byte elements (bf16 | f16)
    static constexpr int LDK = BK;
    __device__ static int off(int row, int chunk) { return row * LDK + ((chunk ^ ((row >> 1) & 2)) << 3); }
    typedef bf16x8 frag;
    static constexpr int KSUB = 1;
    __device__ static frag ld(const T *tile, int row, int lane, int) {
        return *reinterpret_cast<const bf16x8 *>(tile + off(row, lane >> 4));
    }
    __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return mma16<DtOf<T>::value>(a, b, c); }
};
template <> struct Lay<float> {
    static constexpr int LDK = BK + 4;
    __device__ static int off(int row, int chunk) { return row * LDK + chunk * 4; }
    typedef float frag;
    static constexpr int KSUB = 8;
    __device__ static frag ld(const float *tile, int row, int lane, int kk) { return tile[row * LDK + 4 * kk + (lane >> 4)]; }
    __device__ static f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

template <typename T, int TCO, int TPX, int WCO, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? (NTHREADS == 512 ? 4 : 2) : (NTHREADS / 256)) void conv3x3_patch_kernel(PatchParams p, const T *__restrict__ x, const T *__restrict__ w,
                                                                 const float *__restrict__ bias, void *__restrict__ yout) {
    typedef Lay<T> L;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CPR = BK / VEC;
    constexpr int LDK = L::LDK;
    constexpr int WPX = (NTHREADS / 64) / WCO;
    constexpr int WTCO = TCO / WCO, WTPX = TPX / WPX;
    constexpr int MT = WTCO / 16, NT = WTPX / 16;
    constexpr int MAXPATCH = TPX + 2 * 64 + 2 * (TPX / 64) + 4;            // (TR+2)*(TC+2) <= this for TC in {16, 32, 64}
    constexpr int NP_V = (MAXPATCH * CPR + NTHREADS - 1) / NTHREADS;     // patch chunks / thread
    constexpr int NA_V = (TCO * CPR + NTHREADS - 1) / NTHREADS;          // weight chunks / thread

    __shared__ __attribute__((aligned(16))) T Ws[2][TCO * LDK];
    __shared__ __attribute__((aligned(16))) T Ps[2][MAXPATCH * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.y * TCO;
    const int TR = p.TR, TC = p.TC, PW = TC + 2, PH = TR + 2, npatch = PH * PW;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int y0 = ty * TR, x0 = tx * TC;
    const T *xn = x + (size_t)n * p.Hin * p.Win * p.ld_in;
    const int ncc = p.Cin / BK, nit = ncc * 9;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    Vec16<T> rp[NP_V], wa[NA_V], wb[NA_V];

    // global / LDS offsets are fixed per workgroup up to a uniform (tap, channel-slab) term: compute them once
    int poff[NP_V], plds[NP_V], woff[NA_V], wlds[NA_V];      // element offsets, global -1 = zero-fill
#pragma unroll
    for (int j = 0; j < NP_V; ++j) {
        const int e = tid + j * NTHREADS, q = e / CPR, ch = e % CPR;
        poff[j] = -1;
        plds[j] = -1;
        if (q < npatch) {
            const int py = q / PW, px = q - py * PW;
            const int iy = y0 + p.oy0 + py, ix = x0 + p.ox0 + px;
            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) poff[j] = (iy * p.Win + ix) * p.ld_in + ch * VEC;
            plds[j] = L::off(q, ch);
        }
    }
#pragma unroll
    for (int j = 0; j < NA_V; ++j) {
        const int e = tid + j * NTHREADS, row = e / CPR, ch = e % CPR;
        woff[j] = (row < TCO && co0 + row < p.Cout) ? (co0 + row) * p.Krow + ch * VEC : -1;
        wlds[j] = row < TCO ? L::off(row, ch) : -1;
    }

    auto load_patch = [&](int cc) {
#pragma unroll
        for (int j = 0; j < NP_V; ++j) {
            rp[j].zero();
            if (poff[j] >= 0) rp[j].load(xn + poff[j] + cc * BK);
        }
    };
    auto store_patch = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NP_V; ++j)
            if (plds[j] >= 0) rp[j].store(&Ps[buf][plds[j]]);
    };
    auto load_w = [&](Vec16<T> (&r)[NA_V], int it) {          // weight tile of global step `it` (= cc*9 + tap)
        const int cc = it / 9, tap = it - cc * 9;
        const int u = tap * p.Cin + cc * BK;
#pragma unroll
        for (int j = 0; j < NA_V; ++j) {
            r[j].zero();
            if (woff[j] >= 0) r[j].load(w + woff[j] + u);
        }
    };
    auto store_w = [&](Vec16<T> (&r)[NA_V], int buf) {
#pragma unroll
        for (int j = 0; j < NA_V; ++j)
            if (wlds[j] >= 0) r[j].store(&Ws[buf][wlds[j]]);
    };

    const int wr = wave / WPX, wc = wave % WPX;
    int bpp[NT];                                 // patch pixel index of this lane's pixel in 16-pixel group j
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = wc * WTPX + 16 * j + (lane & 15);
        bpp[j] = (q / TC) * PW + (q % TC);
    }
    const int arow = wr * WTCO + (lane & 15);

    auto compute = [&](const T *Wt, const T *Pt, int shift) {
#pragma unroll
        for (int kk = 0; kk < L::KSUB; ++kk) {
            typename L::frag a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = L::ld(Wt, arow + 16 * i, lane, kk);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = L::ld(Pt, bpp[j] + shift, lane, kk);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = L::mma(a[i], b[j], acc[i][j]);
        }
    };

    // ---- pipeline: weights two steps ahead (register sets wa / wb alternate), patch one slab ahead ------------
    load_patch(0);
    load_w(wa, 0);
    store_patch(0);
    store_w(wa, 0);
    if (nit > 1) load_w(wa, 1);                  // wa holds step it+1 at even taps, wb at odd taps
    __syncthreads();
    for (int cc = 0; cc < ncc; ++cc) {
        if (cc + 1 < ncc) load_patch(cc + 1);
        const T *Pt = Ps[cc & 1];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int it = cc * 9 + tap;
            const int kh = tap / 3, kw = tap % 3;
            if (it + 2 < nit) {
                if (tap & 1) load_w(wa, it + 2); else load_w(wb, it + 2);
            }
            const int shift = p.transposed ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;
            compute(Ws[it & 1], Pt, shift);
            if (it + 1 < nit) {
                if (tap & 1) store_w(wb, (it + 1) & 1); else store_w(wa, (it + 1) & 1);
            }
            if (tap == 8 && cc + 1 < ncc) store_patch((cc + 1) & 1);
            __syncthreads();
        }
        // nine taps is odd: the two register sets have exchanged roles -- rename them back
#pragma unroll
        for (int j = 0; j < NA_V; ++j) {
            Vec16<T> tmp = wa[j];
            wa[j] = wb[j];
            wb[j] = tmp;
        }
    }

    // ---- epilogue (as conv_igemm.hip): 4 consecutive output channels per lane -------------------------------
    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    with_act(p.act, p.slope, [&](auto actf) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = wc * WTPX + 16 * j + (lane & 15);
            const int oy = y0 + q / TC, ox = x0 + q % TC;
            if (oy >= p.H || ox >= p.W) continue;
            const size_t po = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int co = co0 + wr * WTCO + 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u = acc[i][j][r];
                    if (bias != nullptr && co + r < p.Cout) u += bias[co + r];
                    v[r] = actf(u);
                }
                const size_t off = po * p.ld_out + co;
                if (p.out_f32 || sizeof(T) == 4) {
                    float *y = reinterpret_cast<float *>(yout) + off;
                    if (vec_ok && co + 3 < p.Cout) *reinterpret_cast<f32x4 *>(y) = f32x4{v[0], v[1], v[2], v[3]};
                    else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = v[r];
                } else {
                    bf16_t *y = reinterpret_cast<bf16_t *>(yout) + off;
                    if (vec_ok && co + 3 < p.Cout) {
                        bf16x4 o;
                        for (int r = 0; r < 4; ++r) o[r] = cvt16<DtOf<T>::value>(v[r]);
                        *reinterpret_cast<bf16x4 *>(y) = o;
                    } else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = cvt16<DtOf<T>::value>(v[r]);
                }
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// Slab-resident variant: per 32-channel slab the workgroup stages the halo patch AND the weights of all nine
// taps ([tap][co][32 k], same swizzled rows) in LDS, then runs the nine taps back to back with no barrier
// in between (144 MFMAs per wave on the 64x256 tile).  The next slab is prefetched into registers while the
// current one is being consumed: two barriers per slab instead of nine, and a whole slab of MFMA time to
// cover the global-load latency.
template <typename T, int TCO, int TPX, int WCO, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, sizeof(T) == 2 ? (NTHREADS == 512 ? 2 : 2) : 1) void conv3x3_slab_kernel(
    PatchParams p, const T *__restrict__ x, const T *__restrict__ w, const float *__restrict__ bias, void *__restrict__ yout) {
    typedef Lay<T> L;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CPR = BK / VEC;
    constexpr int LDK = L::LDK;
    constexpr int WPX = (NTHREADS / 64) / WCO;
    constexpr int WTCO = TCO / WCO, WTPX = TPX / WPX;
    constexpr int MT = WTCO / 16, NT = WTPX / 16;
    constexpr int MAXPATCH = TPX + 2 * 64 + 2 * (TPX / 64) + 4;
    constexpr int NP_V = (MAXPATCH * CPR + NTHREADS - 1) / NTHREADS;
    constexpr int NW_V = (9 * TCO * CPR + NTHREADS - 1) / NTHREADS;

    __shared__ __attribute__((aligned(16))) T Ws[9 * TCO * LDK];
    __shared__ __attribute__((aligned(16))) T Ps[MAXPATCH * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.y * TCO;
    const int TR = p.TR, TC = p.TC, PW = TC + 2, PH = TR + 2, npatch = PH * PW;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int y0 = ty * TR, x0 = tx * TC;
    const T *xn = x + (size_t)n * p.Hin * p.Win * p.ld_in;
    const int ncc = p.Cin / BK;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    Vec16<T> rp[NP_V], rw[NW_V];
    int poff[NP_V], plds[NP_V], woff[NW_V], wlds[NW_V];
#pragma unroll
    for (int j = 0; j < NP_V; ++j) {
        const int e = tid + j * NTHREADS, q = e / CPR, ch = e % CPR;
        poff[j] = -1;
        plds[j] = -1;
        if (q < npatch) {
            const int py = (int)(((float)q + 0.5f) * (1.0f / (float)PW)), px = q - py * PW;
            const int iy = y0 + p.oy0 + py, ix = x0 + p.ox0 + px;
            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) poff[j] = (iy * p.Win + ix) * p.ld_in + ch * VEC;
            plds[j] = L::off(q, ch);
        }
    }
#pragma unroll
    for (int j = 0; j < NW_V; ++j) {        // entry e -> (tap, row, chunk)
        const int e = tid + j * NTHREADS, ch = e % CPR, rr = e / CPR, tap = rr / TCO, row = rr - tap * TCO;
        woff[j] = (tap < 9 && co0 + row < p.Cout) ? (co0 + row) * p.Krow + tap * p.Cin + ch * VEC : -1;
        wlds[j] = tap < 9 ? tap * TCO * LDK + L::off(row, ch) : -1;
    }
    auto load_slab = [&](int cc) {
#pragma unroll
        for (int j = 0; j < NP_V; ++j) {
            rp[j].zero();
            if (poff[j] >= 0) rp[j].load(xn + poff[j] + cc * BK);
        }
#pragma unroll
        for (int j = 0; j < NW_V; ++j) {
            rw[j].zero();
            if (woff[j] >= 0) rw[j].load(w + woff[j] + cc * BK);
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int j = 0; j < NP_V; ++j)
            if (plds[j] >= 0) rp[j].store(&Ps[plds[j]]);
#pragma unroll
        for (int j = 0; j < NW_V; ++j)
            if (wlds[j] >= 0) rw[j].store(&Ws[wlds[j]]);
    };

    const int wr = wave / WPX, wc = wave % WPX;
    int bpp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = wc * WTPX + 16 * j + (lane & 15);
        bpp[j] = (q / TC) * PW + (q % TC);
    }
    const int arow = wr * WTCO + (lane & 15);

    load_slab(0);
    store_slab();
    __syncthreads();
    for (int cc = 0; cc < ncc; ++cc) {
        if (cc + 1 < ncc) load_slab(cc + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap % 3;
            const int shift = p.transposed ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;
            const T *Wt = Ws + tap * TCO * LDK;
#pragma unroll
            for (int kk = 0; kk < L::KSUB; ++kk) {
                typename L::frag a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = L::ld(Wt, arow + 16 * i, lane, kk);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = L::ld(Ps, bpp[j] + shift, lane, kk);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = L::mma(a[i], b[j], acc[i][j]);
            }
        }
        if (cc + 1 < ncc) {
            __syncthreads();          // everyone is done reading this slab
            store_slab();
            __syncthreads();
        }
    }

    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    with_act(p.act, p.slope, [&](auto actf) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int q = wc * WTPX + 16 * j + (lane & 15);
            const int oy = y0 + q / TC, ox = x0 + q % TC;
            if (oy >= p.H || ox >= p.W) continue;
            const size_t po = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int co = co0 + wr * WTCO + 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u = acc[i][j][r];
                    if (bias != nullptr && co + r < p.Cout) u += bias[co + r];
                    v[r] = actf(u);
                }
                const size_t off = po * p.ld_out + co;
                if (p.out_f32 || sizeof(T) == 4) {
                    float *y = reinterpret_cast<float *>(yout) + off;
                    if (vec_ok && co + 3 < p.Cout) *reinterpret_cast<f32x4 *>(y) = f32x4{v[0], v[1], v[2], v[3]};
                    else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = v[r];
                } else {
                    bf16_t *y = reinterpret_cast<bf16_t *>(yout) + off;
                    if (vec_ok && co + 3 < p.Cout) {
                        bf16x4 o;
                        for (int r = 0; r < 4; ++r) o[r] = cvt16<DtOf<T>::value>(v[r]);
                        *reinterpret_cast<bf16x4 *>(y) = o;
                    } else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) y[r] = cvt16<DtOf<T>::value>(v[r]);
                }
            }
        }
    });
}

template <typename T, int TCO, int TPX, int WCO, int NTHREADS>
void launch_slab(PatchParams p, const void *x, const void *w, const float *bias, void *y, hipStream_t st);

// tile shape with the least padding waste for an H x W map
void pick_tile(int H, int W, int TPX, int &TR, int &TC) {
    double best = -1;
    for (int tc = 16; tc <= TPX && tc <= 64; tc *= 2) {
        const int tr = TPX / tc;
        const double eff = (double)H * W / ((double)cdiv(H, tr) * tr * cdiv(W, tc) * tc);
        if (eff > best + 1e-9) { best = eff; TR = tr; TC = tc; }
    }
}

template <typename T, int TCO, int TPX, int WCO, int NTHREADS>
void launch_cfg(PatchParams p, const void *x, const void *w, const float *bias, void *y, hipStream_t st) {
    pick_tile(p.H, p.W, TPX, p.TR, p.TC);
    p.tiles_x = cdiv(p.W, p.TC);
    p.tiles_y = cdiv(p.H, p.TR);
    dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Cout, TCO));
    rdfc_note_kernel("conv3x3_patch_kernel<%s, %d, %d, %d, %d>", tname<T>(), TCO, TPX, WCO, NTHREADS);
    hipLaunchKernelGGL((conv3x3_patch_kernel<T, TCO, TPX, WCO, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, (const T *)x, (const T *)w, bias, y);
}

template <typename T, int TCO, int TPX, int WCO, int NTHREADS>
void launch_slab(PatchParams p, const void *x, const void *w, const float *bias, void *y, hipStream_t st) {
    pick_tile(p.H, p.W, TPX, p.TR, p.TC);
    p.tiles_x = cdiv(p.W, p.TC);
    p.tiles_y = cdiv(p.H, p.TR);
    dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Cout, TCO));
    rdfc_note_kernel("conv3x3_slab_kernel<%s, %d, %d, %d, %d>", tname<T>(), TCO, TPX, WCO, NTHREADS);
    hipLaunchKernelGGL((conv3x3_slab_kernel<T, TCO, TPX, WCO, NTHREADS>), grid, dim3(NTHREADS), 0, st, p, (const T *)x, (const T *)w, bias, y);
}

template <typename T>
void launch_T(const PatchParams &p, const void *x, const void *w, const float *bias, void *y, hipStream_t st) {
    // bf16: 8 waves / workgroup (4 waves per SIMD at 2 workgroups per CU) where it measured faster in an
    // interleaved A/B on one device: +15-18 % on the 128x128 tile, +8 % on 32x256, -5..-13 % on 64x256 (spills)
    // Variant choice from interleaved A/B runs on one device (tools/ab_env.sh): the slab-resident kernel wins for
    // Cout <= 64 (+7..13 %, 1-channel heads +68 %) and for the 512-channel layers (+13 %); the per-tap kernel with
    // 8-wave workgroups (2 workgroups / CU) wins for 128- and 256-channel outputs (+6..9 %).
    constexpr int slab_env = -1;
    const bool slab = slab_env >= 0 ? slab_env != 0 : (p.Cout <= 64 || p.Cin >= 512);
    if constexpr (sizeof(T) == 2) if (slab) {
        if (p.Cout > 64) launch_slab<T, 128, 256, 2, 512>(p, x, w, bias, y, st);
        else if (p.Cout > 32) launch_slab<T, 64, 256, 1, 256>(p, x, w, bias, y, st);
        else if (p.Cout > 16) launch_slab<T, 32, 256, 1, 256>(p, x, w, bias, y, st);
        else launch_slab<T, 16, 256, 1, 256>(p, x, w, bias, y, st);
        return;
    }
    constexpr int NTH = sizeof(T) == 2 ? 512 : 256;
    if (p.Cout > 64) launch_cfg<T, 128, 128, 2, NTH>(p, x, w, bias, y, st);
    else if (p.Cout > 32) launch_cfg<T, 64, 256, 1, 256>(p, x, w, bias, y, st);
    else if (p.Cout > 16) launch_cfg<T, 32, 256, 1, NTH>(p, x, w, bias, y, st);
    else launch_cfg<T, 16, 256, 1, 256>(p, x, w, bias, y, st);      // 1-channel heads: activation-traffic bound
}

}  // namespace

// Called by conv_igemm.hip's entry points when the shape qualifies. Returns 1 if it took the launch.
int rdfc_try_conv3x3_patch(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias,
                           void *out, hipStream_t st) {
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->ph == d->pw && d->ph <= 1 && d->dh == 1 && d->dw == 1)) return 0;
    if (Cin % BK != 0 || d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return 0;
    const int pad = d->ph;
    // forward: gather x (Hi x Wi) -> y (Ho x Wo), patch origin -pad; transposed: gather dy (Ho x Wo) -> dx (Hi x Wi), origin pad-2
    PatchParams p{d->N, transposed ? d->Ho : d->Hi, transposed ? d->Wo : d->Wi, transposed ? d->Hi : d->Ho, transposed ? d->Wi : d->Wo,
                  Cin, Cout, transposed ? d->ldy : d->ldx, transposed ? d->ldx : d->ldy, transposed,
                  transposed ? pad - 2 : -pad, transposed ? pad - 2 : -pad, d->act, d->slope, d->out_f32, 9 * Cin, 0, 0, 0, 0};
    if (d->dtype == RDFC_BF16) launch_T<bf16_t>(p, in, w, bias, out, st);
    else if (d->dtype == RDFC_F16) launch_T<f16_t>(p, in, w, bias, out, st);
    else launch_T<float>(p, in, w, bias, out, st);
    return 1;
}
