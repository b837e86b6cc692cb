// conv_igemm_glds.hip -- generic implicit-GEMM convolution (forward and input-gradient), bf16, staged by LDS-DMA.
//
// conv_igemm.hip's decomposition -- flat tiles of 128 (or 256) output pixels x 128 (or 64) output channels, one
// K-step per (tap, 32-channel slab), stride-parity classes for the transposed form -- with conv_glds.hip's
// pipeline: both operand tiles go global -> LDS by global_load_lds_dwordx4 into a ring of four K-step slots filled
// three steps ahead; one raw barrier and one counted s_waitcnt vmcnt(N) per K-step; a wave owns 64 co x 64 px.
// The activation tile is a per-lane gather (every lane's source pixel moves with the tap), which LDS-DMA allows:
// only its destination is lane-linear.  A lane's pixels are fixed for the whole kernel, so the bounds test and the
// source OFFSET (32 bits, through a buffer descriptor of the map: round 5) are recomputed once per tap; the slab
// advances through the DMA's scalar offset, and an out-of-image pixel or a channel row beyond Cout is an offset
// beyond the descriptor's extent, which arrives as zeros -- no vector instruction and no select per DMA, every wave
// issues the same number of DMAs per step.  Unlike the 2-D patch kernels the flat pixel tiling
// wastes nothing on small or odd-sized maps (PatchGAN's 27x37, the 15x19 / 29x38 encoder stages).
#include "common.h"

#include <stdlib.h>

namespace {

constexpr int BK = 32, NT = 256, MAXTAPS = 64;

struct IGParams {
    int N, Hin, Win, Cin, Hout, Wout, Cout;
    int KH, KW, sh, sw, ph, pw, dh, dw;
    int ld_in, ld_out, transposed;
    int act;
    float slope;
    int out_f32;
    const void *addend;          // optional map added to the result before the store (rdfc_conv2d_dgrad_add)
    int ld_add;
    int row_store;               // epilogue through LDS: whole 128-byte channel rows per store instruction (common.h)
    unsigned x_bytes, w_bytes;   // extents of the input map and of the packed weights (buffer descriptors of the LDS-DMA loads; < 2^31)
    float r_sh, r_sw, r_kw;      // 1 / sh, 1 / sw, 1 / KW (qdiv)
};

// a / d and a % d for 0 <= a < 2^21, d > 0 through a float reciprocal rd ~ 1.f / d (1 ulp is enough): the quotient estimate is off by at
// most one either way and is corrected from the remainder (conv_glds.hip: hipcc expands a division by a run-time value into ~45 vector
// instructions, and this kernel's set-up held 4-8 of them per thread plus a serial tap loop on thread 0: its first DMA was issued
// 900-1 200 instructions into the kernel, which the short reductions of the stride-2 and 1x1 layers -- 6 to 18 K-steps -- could not hide)
__device__ __forceinline__ void qdiv(int a, int d, float rd, int &q, int &r) {
    q = (int)((float)a * rd);
    r = a - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;
    q += hi - lo;
    r += (lo - hi) * d;
}

// one LDS-DMA instruction through a buffer descriptor: lane l reads 16 bytes at base + soff + voff (voff per lane, soff wave-uniform)
// and lands at lds_base + 16 l; a lane whose voff is not below num_records arrives as zeros (conv_glds.hip: bdma16)
// the same for 0 <= a < 2^31 through a double reciprocal (the flat pixel index of a large batch: 8 x 480 x 640 is beyond qdiv's 2^21; once per
// gathered pixel in the set-up, three f64 instructions)
__device__ __forceinline__ void qdiv_wide(int a, int d, double rd, int &q, int &r) {
    q = (int)((double)a * rd);
    r = a - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;
    q += hi - lo;
    r += (lo - hi) * d;
}

constexpr unsigned DEAD = 0x80000000u;
__device__ inline void bdma16(unsigned voff, u32x4 rsrc, unsigned soff, unsigned lds_base) {
    // m0 is declared clobbered, not saved and restored (the compiler uses it for nothing else in this kernel; clang warns that m0 is
    // reserved): at three waves per SIMD and 16 MFMAs per step this loop is short of scalar issue slots
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "memory", "m0");
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
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TCO, int DT = RDFC_BF16>
__global__ __launch_bounds__(NT, 2) void igemm_glds_kernel(IGParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                           const float *__restrict__ bias, void *__restrict__ yout,
                                                           const bf16_t *__restrict__ zero) {
    constexpr int TPX = TCO == 128 ? 128 : 256;                 // pixels per tile
    constexpr int WCO = TCO / 64, WPX = 4 / WCO;                // waves along co / px: each 64 co x 64 px
    constexpr int NA = TCO * 4 / NT, NB = TPX * 4 / NT;         // DMAs per thread and K-step: weights (2 / 1), pixels (2 / 4)
    constexpr int AB = TCO * BK * 2, SLOT = AB + TPX * BK * 2;  // 16 KB / 20 KB
    // three slots either way: 48 KB (three workgroups per CU; with four slots and two workgroups 4x4/s2 64->128 @ 228x304 ran 80.9 us,
    // now 68.5; 128->256 @ 114x152 81.1 -> 74.6) / 60 KB
    constexpr int RING = 3, LOOK = RING - 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    __shared__ int tap_w[MAXTAPS], tap_dy[MAXTAPS], tap_dx[MAXTAPS], ntaps_s;
    __shared__ int pxo[TPX];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * TCO;

    // ---- pixel enumeration (as conv_igemm.hip) --------------------------------------------------------------------
    int cy = 0, cx = 0, Ha = p.Hout, Wb = p.Wout;
    if (p.transposed) {                                          // (wave-uniform)
        int t_;
        qdiv((int)blockIdx.z, p.sw, p.r_sw, cy, cx);
        Ha = Wb = 0;
        if (cy < p.Hout) qdiv(p.Hout - cy + p.sh - 1, p.sh, p.r_sh, Ha, t_);
        if (cx < p.Wout) qdiv(p.Wout - cx + p.sw - 1, p.sw, p.r_sw, Wb, t_);
    }
    const int HW = Ha * Wb, Pc = p.N * HW;
    const int m0 = blockIdx.x * TPX;
    if (m0 >= Pc) return;
    // the taps of this parity class, one lane per (kh, kw), compacted in tap order by a ballot (a serial loop on thread 0 with its
    // divisions cost ~1 500 instructions while the other 255 threads waited)
    if (wave == 0) {
        const int ntap_all = p.KH * p.KW;
        int kh, kw, dy = 0, dx = 0;
        qdiv(lane < ntap_all ? lane : 0, p.KW, p.r_kw, kh, kw);
        bool valid = lane < ntap_all;
        if (!p.transposed) {
            dy = kh * p.dh;
            dx = kw * p.dw;
        } else {
            int ry, rx, t_, qy, qx, rem_y, rem_x;
            qdiv(cy + p.ph, p.sh, p.r_sh, t_, ry);
            qdiv(cx + p.pw, p.sw, p.r_sw, t_, rx);
            // ey = ry - kh dh and ex = rx - kw dw must be multiples of the stride (they may be negative: shifted by 1024 strides)
            qdiv(ry - kh * p.dh + 1024 * p.sh, p.sh, p.r_sh, qy, rem_y);
            qdiv(rx - kw * p.dw + 1024 * p.sw, p.sw, p.r_sw, qx, rem_x);
            valid = valid && rem_y == 0 && rem_x == 0;
            dy = qy - 1024;
            dx = qx - 1024;
        }
        const unsigned long long mask = __ballot(valid);
        if (valid) {
            const int at = __popcll(mask & ((1ull << lane) - 1ull));
            tap_w[at] = lane;
            tap_dy[at] = dy;
            tap_dx[at] = dx;
        }
        if (lane == 0) ntaps_s = __popcll(mask);
    }
    // this thread's NB gathered pixels: tile rows (k * NT + tid) >> 2, 16-byte slot tid & 3
    const double r_hw = 1.0 / (double)HW;                        // (m runs over the whole batch: qdiv_wide; the remainder is < HW < 2^21)
    const float r_wb = __builtin_amdgcn_rcpf((float)Wb);
    int pnb[NB], pyy[NB], pxx[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int row = (k * NT + tid) >> 2, m = m0 + row;
        const bool in = m < Pc;
        int n, r, a, b;
        qdiv_wide(in ? m : 0, HW, r_hw, n, r);
        qdiv(r, Wb, r_wb, a, b);
        pnb[k] = in ? n * p.Hin * p.Win : -1;
        int po;
        if (!p.transposed) {
            pyy[k] = a * p.sh - p.ph;
            pxx[k] = b * p.sw - p.pw;
            po = m;
        } else {
            const int oy = cy + a * p.sh, ox = cx + b * p.sw;
            int t_;
            qdiv(oy + p.ph, p.sh, p.r_sh, pyy[k], t_);
            qdiv(ox + p.pw, p.sw, p.r_sw, pxx[k], t_);
            po = (n * p.Hout + oy) * p.Wout + ox;
        }
        if ((tid & 3) == 0) pxo[row] = in ? po : -1;
    }
    const int bslot = tid & 3;
    const u32x4 xsrc = make_rsrc(x, p.x_bytes), wrs = make_rsrc(w, p.w_bytes);
    unsigned woff[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int row = (k * NT + tid) >> 2;
        woff[k] = co0 + row < p.Cout ? 2u * (unsigned)((co0 + row) * (p.KH * p.KW) * p.Cin + ((bslot ^ ((row >> 1) & 2)) << 3)) : DEAD;
    }
    __syncthreads();
    const int ntaps = ntaps_s, ncc = p.Cin / BK, G = ntaps * ncc;

    // ---- issue side: K-step h = (tap ht, slab hc), slabs inner ----------------------------------------------------------
    unsigned boff[NB];
    int ht = 0, hc = 0, hw = 0;                                  // hw = weight tap index of ht
    auto set_tap = [&](int t) {
        const int dy = tap_dy[t], dx = tap_dx[t];
        hw = __builtin_amdgcn_readfirstlane(tap_w[t]);          // (wave-uniform: it goes into the DMAs' scalar offset)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int row = (k * NT + tid) >> 2;
            const int iy = pyy[k] + dy, ix = pxx[k] + dx;
            const bool ok = pnb[k] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            boff[k] = ok ? 2u * (unsigned)((pnb[k] + iy * p.Win + ix) * p.ld_in + ((bslot ^ ((row >> 1) & 2)) << 3)) : DEAD;       // (< 2^31: the launcher checks x_bytes)
        }
    };
    auto issue = [&](int h) {
        // (a step beyond the last, h >= G, re-reads the last tap's first slab -- valid memory -- into a slot nobody reads: the DMA count
        //  per step stays static without a second descriptor or a select)
        const bool live = h < G;
        const unsigned slot = lds0 + (h % RING) * SLOT;
        const unsigned so_w = 2u * (unsigned)(hw * p.Cin + hc * BK), so_x = 2u * (unsigned)(hc * BK);
#pragma unroll
        for (int k = 0; k < NA; ++k) bdma16(woff[k], wrs, so_w, slot + (k * NT + wave * 64) * 16);
#pragma unroll
        for (int k = 0; k < NB; ++k) bdma16(boff[k], xsrc, so_x, slot + AB + (k * NT + wave * 64) * 16);
        if (live) {
            if (++hc == ncc) {
                hc = 0;
                if (++ht < ntaps) set_tap(ht);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int wr = wave / WPX, wc = wave % WPX;

    if (G > 0) {
        set_tap(0);
#pragma unroll
        for (int h = 0; h < LOOK; ++h) issue(h);
        const int arow = wr * 64 + (lane & 15), brow = wc * 64 + (lane & 15), kg = lane >> 4;
        const unsigned a_off = (unsigned)((arow << 6) | (kg << 4)) ^ (unsigned)((arow & 4) << 3);     // rows + 16 i share bit 2
        const unsigned b_off = AB + ((unsigned)((brow << 6) | (kg << 4)) ^ (unsigned)((brow & 4) << 3));
        for (int g = 0; g < G; ++g) {
            wait_vm_and_reads<(LOOK - 1) * (NA + NB)>();                  // the K-steps after g that are already issued may stay in flight
            __builtin_amdgcn_s_barrier();
            issue(g + LOOK);                                    // its slot was last read in step g - 1: every wave is past the barrier
            const unsigned char *S = smem + (g % RING) * SLOT;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(S + a_off + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8 *>(S + b_off + j * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mma16<DT>(a[i], b[j], acc[i][j]);
        }
        wait_vm<0>();
    }

    // ---- epilogue: bias + activation + store (4 consecutive channels per lane) -----------------------------------------
    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    float bv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                               // four consecutive channels per lane: one 16-byte load, not four scalar ones
        const int co = co0 + wr * 64 + 16 * i + q4;
        if (bias != nullptr && co + 3 < p.Cout) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(bias + co);
            bv[i][0] = t4[0]; bv[i][1] = t4[1]; bv[i][2] = t4[2]; bv[i][3] = t4[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[i][r] = (bias != nullptr && co + r < p.Cout) ? bias[co + r] : 0.f;
        }
    }
    // ---- whole-row stores (common.h: rs_put / rs_flush); an addend is added here, in f32, from its 8-byte pieces -----------------------
    if (p.row_store == 2) {
        // launches with an addend (rdfc_conv2d_dgrad_add): f32 staging, 32 of the wave's 64 pixels per pass, the addend fetched in whole
        // 128-byte rows a pass ahead and added by the flush (common.h: rsf_put / rsf_fetch / rsf_flush); same bits as the path below
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        const int cow = co0 + wr * 64;
        const bf16_t *addw = reinterpret_cast<const bf16_t *>(p.addend) + cow;
        auto ppix = [&](int j0, int px) -> long { return (long)pxo[wc * 64 + 16 * j0 + px]; };
        RsfAdd cur, nxt;
        rsf_fetch(cur, lane, addw, p.ld_add, p.Cout - cow, [&](int px) { return ppix(0, px); });
        __syncthreads();
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += 2) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = acc[i][j0 + jj][r] + bv[i][r];
                        v[r] = u > 0.f ? u : u * ns;
                    }
                    rsf_put(Ot, 16 * jj + (lane & 15), 16 * i + q4, v);
                }
            if (j0 == 0) rsf_fetch(nxt, lane, addw, p.ld_add, p.Cout - cow, [&](int px) { return ppix(2, px); });
            rsf_flush<DT>(Ot, lane, reinterpret_cast<bf16_t *>(yout) + cow, p.ld_out, cur, nullptr, p.Cout - cow, [&](int px) { return ppix(j0, px); });
            cur = nxt;
        }
        return;
    }
    if (p.row_store) {
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        __syncthreads();                                           // every wave is done with the operand tiles
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int po = pxo[wc * 64 + 16 * j + (lane & 15)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + wr * 64 + 16 * i + q4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = acc[i][j][r] + bv[i][r];
                    v[r] = u > 0.f ? u : u * ns;
                }
                if (p.addend != nullptr && po >= 0 && co < p.Cout) {       // (Cout and ld_add are multiples of 8 on this path)
                    const bf16x4 t = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const bf16_t *>(p.addend) + (size_t)po * p.ld_add + co);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t[r]);
                }
                rs_put<DT>(Ot, 16 * j + (lane & 15), 16 * i + q4, v);
            }
        }
        const int cow = co0 + wr * 64;
        rs_flush(Ot, lane, reinterpret_cast<bf16_t *>(yout) + cow, p.Cout - cow, [&](int px) -> long {
            const int po = pxo[wc * 64 + px];
            return po < 0 ? -1 : (long)po * p.ld_out;
        });
        return;
    }
    with_act(p.act, p.slope, [&](auto actf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int po = pxo[wc * 64 + 16 * j + (lane & 15)];
            if (po < 0) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + wr * 64 + 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = actf(acc[i][j][r] + bv[i][r]);
                const size_t off = (size_t)po * p.ld_out + co;
                if (p.addend != nullptr) {
                    const bf16_t *ad = reinterpret_cast<const bf16_t *>(p.addend) + (size_t)po * p.ld_add + co;
                    if ((p.ld_add & 3) == 0 && co + 3 < p.Cout) {
                        const bf16x4 t = *reinterpret_cast<const bf16x4 *>(ad);
                        for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t[r]);
                    } else
                        for (int r = 0; r < 4; ++r)
                            if (co + r < p.Cout) v[r] += f32of<DT>(ad[r]);
                }
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
    });
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

template <int TCO, int DT = RDFC_BF16>
int launch(const IGParams &p, const void *x, const void *w, const float *bias, void *y, const bf16_t *zero, hipStream_t st) {
    constexpr int TPX = TCO == 128 ? 128 : 256;
    constexpr int OPER = 3 * (TCO + TPX) * BK * 2;
    constexpr int SMEM = OPER > (NT / 64) * RS_TILE ? OPER : (NT / 64) * RS_TILE;          // the whole-row epilogue reuses it: 9 KB per wave
    static_assert(SMEM >= (NT / 64) * RS_TILE, "rs_put / rs_flush write wave * RS_TILE ... + RS_TILE");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)igemm_glds_kernel<TCO, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    int Pmax, nz = 1;
    if (!p.transposed) {
        Pmax = p.N * p.Hout * p.Wout;
    } else {
        nz = p.sh * p.sw;
        Pmax = p.N * ((p.Hout + p.sh - 1) / p.sh) * ((p.Wout + p.sw - 1) / p.sw);
    }
    rdfc_note_kernel("igemm_glds_kernel<%d, %d>", TCO, DT);
    hipLaunchKernelGGL((igemm_glds_kernel<TCO, DT>), dim3(cdiv(Pmax, TPX), cdiv(p.Cout, TCO), nz), dim3(NT), SMEM, st, p, (const bf16_t *)x,
                       (const bf16_t *)w, bias, y, zero);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's entry points after the 3x3 / stride-1 kernels. Returns 1 if it took the launch.
int rdfc_try_igemm_glds(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                        hipStream_t st, const void *addend = nullptr, int ld_add = 0) {
    if (!rdfc_is16(d->dtype)) return 0;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    const int ld_in = transposed ? d->ldy : d->ldx;
    if (Cin % BK != 0 || ld_in % 8 != 0 || Cout < 64 || d->KH * d->KW > MAXTAPS) return 0;
    // K-steps per output tile.  With four ring slots (two workgroups per CU) short reductions were better off on the register-staged
    // kernel (1x1 192->384, 6 steps: 83 vs 69 us); with three slots and three workgroups per CU this kernel wins from 6 steps up
    // (1x1 192->384 67 -> 63 us forward, 67 -> 57 input gradient; 1x1 384->768 57 -> 49 / 56 -> 47; 1x1 768->1536 60 -> 46;
    // 3x3/s2 64->128 54 -> 46); 3-step reductions (1x1 96->192) stay where they were.
    // (round 6 re-measured the short reductions with this kernel's round-5 set-up: 1x1 / stride 2 64 -> 128 @ 16 x 228 x 304 32.9 -> 27.2 us, its
    //  input gradient 79.3 -> 59.5, 1x1 96 -> 192 59.1 -> 52.3 alone -- and nothing in the step: 265.4 / 264.0 / 262.6 against 263.8 / 263.2 / 262.0
    //  pairs/s with min_steps = 2, three interleaved pairs; the threshold stays)
    constexpr int min_steps = 6;
    if (d->KH * d->KW * (Cin / BK) < min_steps) return 0;
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    IGParams p{d->N, transposed ? d->Ho : d->Hi, transposed ? d->Wo : d->Wi, Cin, transposed ? d->Hi : d->Ho, transposed ? d->Wi : d->Wo, Cout,
               d->KH, d->KW, d->sh, d->sw, d->ph, d->pw, d->dh, d->dw, ld_in, transposed ? d->ldx : d->ldy, transposed,
               d->act, d->slope, d->out_f32, addend, ld_add, 0, 0, 0};
    {
        const size_t xb = 2 * (size_t)p.N * p.Hin * p.Win * p.ld_in, wb_ = 2 * (size_t)Cout * d->KH * d->KW * Cin;
        if (xb >= 0x7ffffff0u || wb_ >= 0x7ffffff0u) return 0;                  // the loaders address both through 32-bit byte offsets
        p.x_bytes = (unsigned)xb;
        p.w_bytes = (unsigned)wb_;
    }
    p.row_store = !d->out_f32 && Cout % 8 == 0 && p.ld_out % 8 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
                  (addend == nullptr || (ld_add % 4 == 0 && (reinterpret_cast<size_t>(addend) & 7) == 0)) &&
                  d->act != RDFC_ACT_TANH && d->act != RDFC_ACT_SIGMOID;
    // qdiv's range: one image's (parity class of) output pixels and the stride / tap arithmetic below 2^21; the batch's flat pixel index goes
    // through qdiv_wide (round 5 refused N * Hout * Wout >= 2^21 here -- batch 8 at 480 x 640 fell back to the register-staged kernel silently)
    if (p.row_store && rsf_ok(addend, ld_add)) p.row_store = 2;        // the addend in whole rows, from the flush (common.h)
    if ((long)p.Hout * p.Wout >= (1L << 21) || (long)p.N * p.Hout * p.Wout >= 0x7fffff00L || d->KH * d->dh + d->ph > 1000 || d->KW * d->dw + d->pw > 1000 ||
        d->sh > 1000 || d->sw > 1000) return 0;
    p.r_sh = 1.f / (float)d->sh;
    p.r_sw = 1.f / (float)d->sw;
    p.r_kw = 1.f / (float)d->KW;
    if (d->dtype == RDFC_F16)
        return Cout > 64 ? launch<128, RDFC_F16>(p, in, w, bias, out, zero, st) : launch<64, RDFC_F16>(p, in, w, bias, out, zero, st);
    if (Cout > 64) return launch<128>(p, in, w, bias, out, zero, st);
    return launch<64>(p, in, w, bias, out, zero, st);
}
