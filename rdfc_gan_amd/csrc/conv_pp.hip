// conv_pp.hip -- 3x3 / stride-1 convolution (forward and input gradient), 2-byte elements, LDS-DMA staging, PING-PONG schedule.
//
// Same decomposition as conv_glds.hip (halo patch of a 32-channel slab in LDS, nine taps = nine LDS offsets, weights through a
// ring of tap slots, both operands global -> LDS by global_load_lds_dwordx4 with the bank swizzle on the source side), but one
// 8-wave workgroup per CU whose two halves alternate roles: while waves 0-3 (group A) run a tap's 16 MFMAs, waves 4-7 (group B,
// their SIMD partners) issue their DMAs and read the next tap's fragments, and vice versa.  conv_glds.hip's workgroups all run
// the same program against one barrier per tap: its SIMD partners reach their MFMAs together and their LDS reads together, and
// in-kernel stamps showed the matrix pipe busy 43 % of the time.  Here every phase has TWO barriers and group B runs one barrier
// behind group A (MI355X_MICROARCH.md, "Two waves per SIMD", item 9; the 8-phase GEMM template of cdna_hip_programming.md):
//
//     barrier event   e(2p)      e(2p+1)     e(2p+2)     e(2p+3)
//     group A          | L(p)      | M(p)      | L(p+1)    | M(p+1)         L(p): issue DMAs of phase p+3, read phase p's fragments
//     group B          | M(p-1)    | L(p)      | M(p)      | L(p+1)         M(p): 16 MFMAs per wave (64 co x 64 px)
//
// Hazards (every wave issues DMAs; "landed" = the issuing wave's counted s_waitcnt vmcnt, "published" = a barrier after it):
//   RAW  phase p+1's data is first read by group A behind e(2p+2): every wave waits for its own pieces of phase p+1 before it
//        arrives there -- group A at the end of M(p), group B at the end of L(p).  Pieces are issued LOOK = 3 phases ahead and
//        retire in order, so the wait leaves the pieces of the last two L segments in flight: vmcnt(n(p-1) + n(p)).
//   WAR  phase q's fragments are in registers once group B has passed its lgkmcnt(0) behind e(2q+2), i.e. before it arrives at
//        e(2q+3): a slot may be refilled from L(q+2) on -> ring of LOOK + 2 = 5 weight slots; the patch buffer of slab cc-1 from
//        tap 1 of slab cc on (group B: tap 0).
// Every wave issues the same number of DMAs in a given phase (n = 1, or 2 on the taps that also carry a patch piece; pieces that
// do not exist read a zero page into a junk slot), so all counts are compile-time constants.
#include "common.h"

#include <stdlib.h>

namespace {

struct PParams {
    int N, Hin, Win, H, W, Cin, Cout;
    int ld_in, ld_out;
    int transposed, oy0, ox0;
    int act;
    float slope;
    int out_f32;
    int tiles_x, tiles_y;
    const void *addend;
    int ld_add;
    int reflect;
};

constexpr int BK = 32, TC = 16, PW = TC + 2, LOOK = 3, RING = LOOK + 2;

__device__ inline void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// DMAs per wave in the L segment of tap t: the regular piece, plus a patch piece on the taps listed for the tile shape
template <int TCO> constexpr int n_dma(int tap) { return TCO == 128 ? ((tap >= 1 && tap <= 3) ? 2 : 1) : ((tap == 1 || tap == 2) ? 2 : 1); }

template <int TCO, int DT>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(PParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                            const float *__restrict__ bias, void *__restrict__ yout,
                                                            const bf16_t *__restrict__ zero) {
    constexpr int NW = 8, WCO = TCO / 64, WPX = NW / WCO;          // waves along co / along pixel rows (64 co x 4 rows x 16 px each)
    constexpr int TR = 4 * WPX, PH = TR + 2, NPATCH = PH * PW;      // tile rows (16 | 32), patch rows, patch pixels (324 | 612)
    constexpr int NPIECE = (NPATCH * 4 + 63) / 64;                  // patch DMA instructions per slab (21 | 39), 1 KB each
    constexpr int PATCH_BYTES = NPIECE * 1024;
    constexpr int WSLOT = TCO * BK * 2;                             // bytes per weight slot (8 KB | 4 KB)
    constexpr int OFF_W = 2 * PATCH_BYTES, OFF_JUNK = OFF_W + RING * WSLOT;
    // patch pieces a wave carries (all issued at least two phases before the slab ends: the wait of its last phase leaves the
    // pieces of the last two L segments in flight):
    //   TCO = 128: piece 8 (t - 1) + wave in the extra slot of taps 1..3;
    //   TCO = 64:  waves 4-7: piece 4 t + (wave - 4) in the regular slot of taps 0..6 (waves 0-3 carry the weights there);
    //              every wave: piece 28 + wave in the extra slot of tap 1, piece 36 + wave in that of tap 2
    constexpr int NREG = TCO == 64 ? 7 : 0, NEX = TCO == 128 ? 3 : 2, NPP = NREG + NEX;
    static_assert(TCO == 128 ? NPIECE <= 24 : NPIECE <= 44, "patch pieces must fit the DMA slots of a slab");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool grpB = wave >= 4;
    const int co0 = blockIdx.y * TCO;
    int t = blockIdx.x;
    {   // XCD-contiguous tile numbering (bijective for any grid size)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7, k = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, n = t / p.tiles_y;
    const int y0 = ty * TR, x0 = tx * TC;
    const int ncc = p.Cin / BK;

    // ---- per-lane DMA sources (computed once) -------------------------------------------------------------------------------
    auto patch_src = [&](int piece, bool &live) -> const bf16_t * {       // source of this lane's 16 bytes of patch piece `piece`
        const int e = piece * 64 + lane, q = e >> 2, slot = e & 3;
        live = false;
        if (piece >= NPIECE || q >= NPATCH) return zero;
        const int py = q / PW, px = q - py * PW;
        int iy = y0 + p.oy0 + py, ix = x0 + p.ox0 + px;
        if (p.reflect) {
            iy = iy < 0 ? -iy : (iy >= p.Hin ? 2 * p.Hin - 2 - iy : iy);
            ix = ix < 0 ? -ix : (ix >= p.Win ? 2 * p.Win - 2 - ix : ix);
        }
        if ((unsigned)iy >= (unsigned)p.Hin || (unsigned)ix >= (unsigned)p.Win) return zero;
        live = true;
        return x + ((size_t)(n * p.Hin + iy) * p.Win + ix) * p.ld_in + ((slot ^ ((q >> 1) & 2)) << 3);
    };
    const bf16_t *psrc[NPP];
    unsigned plive = 0;
    int ppiece[NPP];                                                        // wave-uniform piece index (>= NPIECE: junk)
#pragma unroll
    for (int k = 0; k < NPP; ++k) {
        int piece;
        if (k < NREG) piece = grpB ? 4 * k + (wave - 4) : NPIECE;           // TCO = 64, regular slot of tap k (group B only)
        else piece = TCO == 128 ? 8 * (k - NREG) + wave : 28 + 8 * (k - NREG) + wave;        // extra slots
        bool live;
        psrc[k] = patch_src(piece, live);
        plive |= (live ? 1u : 0u) << k;
        ppiece[k] = piece < NPIECE ? piece : -1;
    }
    const bool has_w = TCO == 128 || !grpB;                                 // this wave's regular slot carries weights
    const bf16_t *wsrc = zero;
    bool wlive = false;
    {
        const int e = (TCO == 128 ? wave : (wave & 3)) * 64 + lane, row = e >> 2, slot = e & 3;
        if (has_w && co0 + row < p.Cout) {
            wsrc = w + (size_t)(co0 + row) * 9 * p.Cin + ((slot ^ ((row >> 1) & 2)) << 3);
            wlive = true;
        }
    }
    const unsigned wdst = (TCO == 128 ? wave : (wave & 3)) * 1024;          // this wave's 1 KB of a weight slot
    auto issue_patch = [&](int k, int cc) {                                 // this wave's patch piece k of slab cc -> patch buffer cc & 1
        const bool on = cc < ncc && ppiece[k] >= 0;
        const unsigned dst = on ? lds0 + (cc & 1) * PATCH_BYTES + ppiece[k] * 1024 : lds0 + OFF_JUNK;
        glds16(on ? psrc[k] + ((plive >> k) & 1 ? cc * BK : 0) : zero, dst);
    };
    auto issue_weights = [&](int it) {                                      // weights of global phase `it` -> ring slot it % RING
        const int cc = it / 9, tap = it - cc * 9;
        const bool on = has_w && cc < ncc;
        glds16(on ? wsrc + (wlive ? tap * p.Cin + cc * BK : 0) : zero, on ? lds0 + OFF_W + (it % RING) * WSLOT + wdst : lds0 + OFF_JUNK);
    };
    // the DMAs of the L segment of (slab cc, tap TAP): patch piece(s) FIRST (they are then older than this phase's weights and are
    // covered by any wait that covers the weights), then the regular slot
    auto issue_phase = [&](int cc, int tap) {
        if (TCO == 128) {
            if (tap >= 1 && tap <= 3) issue_patch(tap - 1, cc + 1);
            issue_weights(cc * 9 + tap + LOOK);
        } else {
            if (tap == 1) issue_patch(NREG, cc + 1);
            if (tap == 2) issue_patch(NREG + 1, cc + 1);
            if (grpB) {
                if (tap < NREG) issue_patch(tap < NREG ? tap : 0, cc + 1);
                else glds16(zero, lds0 + OFF_JUNK);
            } else
                issue_weights(cc * 9 + tap + LOOK);
        }
    };

    // ---- fragment coordinates -------------------------------------------------------------------------------------------------
    const int wr = wave / WPX, wc = wave % WPX;
    int bpp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bpp[j] = (4 * wc + j) * PW + (lane & 15);
    const int arow = wr * 64 + (lane & 15), kg = lane >> 4;
    const unsigned a_off = (unsigned)((arow << 6) | (kg << 4)) ^ (unsigned)((arow & 4) << 3);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    auto read_frags = [&](int it, int cc, int tap, bf16x8 (&a)[4], bf16x8 (&b)[4]) {
        const int kh = tap / 3, kw = tap - kh * 3;
        const int shift = p.transposed ? (2 - kh) * PW + (2 - kw) : kh * PW + kw;
        const unsigned char *Wt = smem + OFF_W + (it % RING) * WSLOT + a_off;
        const unsigned char *Pt = smem + (cc & 1) * PATCH_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = bpp[j] + shift;
            b[j] = *reinterpret_cast<const bf16x8 *>(Pt + (((row << 6) | (kg << 4)) ^ ((row & 4) << 3)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(Wt + i * 1024);
    };

    // ---- prologue: slab 0's patch (every piece, spread over the waves), the weights of phases 0 .. LOOK-1 -----------------------
    {
        constexpr int ROUNDS = (NPIECE + 7) / 8;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int piece = r * 8 + wave;
            bool live;
            const bf16_t *src = patch_src(piece, live);
            glds16(src, piece < NPIECE ? lds0 + piece * 1024 : lds0 + OFF_JUNK);
        }
        if (TCO == 128) {
#pragma unroll
            for (int h = 0; h < LOOK; ++h) issue_weights(h);
        } else {
#pragma unroll
            for (int h = 0; h < LOOK; ++h) {
                if (grpB) glds16(zero, lds0 + OFF_JUNK);
                else issue_weights(h);
            }
        }
        wait_vm<LOOK - 1>();                                      // patch 0 and the weights of phase 0 have landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                             // e(0)
        if (grpB) __builtin_amdgcn_s_barrier();                   // group B runs one barrier behind
    }

#define RDFC_PP_PHASE(TAP)                                                                                     \
    {                                                                                                          \
        constexpr int ALLOW = n_dma<TCO>((TAP + 8) % 9) + n_dma<TCO>(TAP);                                     \
        bf16x8 a_[4], b_[4];                                                                                   \
        issue_phase(cc, TAP);                                                                                  \
        read_frags(it0 + TAP, cc, TAP, a_, b_);                                                                \
        if (grpB) wait_vm<ALLOW>();                                                                            \
        __builtin_amdgcn_s_barrier();                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                       \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[i_][j_] = mma16<DT>(a_[i_], b_[j_], acc[i_][j_]); \
        __builtin_amdgcn_s_setprio(0);                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        if (!grpB) wait_vm<ALLOW>();                                                                           \
        __builtin_amdgcn_s_barrier();                                                                          \
    }
    for (int cc = 0; cc < ncc; ++cc) {
        const int it0 = cc * 9;
        RDFC_PP_PHASE(0)
        RDFC_PP_PHASE(1)
        RDFC_PP_PHASE(2)
        RDFC_PP_PHASE(3)
        RDFC_PP_PHASE(4)
        RDFC_PP_PHASE(5)
        RDFC_PP_PHASE(6)
        RDFC_PP_PHASE(7)
        RDFC_PP_PHASE(8)
    }
#undef RDFC_PP_PHASE
    if (!grpB) __builtin_amdgcn_s_barrier();                      // group A's barrier count catches up with group B's
    wait_vm<0>();                                                 // (trailing zero-page pieces land in LDS this workgroup owns)

    // ---- epilogue: bias + activation (+ addend) + store, 4 consecutive output channels per lane ------------------------------------
    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    float bv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + wr * 64 + 16 * i + q4;
        if (bias != nullptr && co + 3 < p.Cout) {
            const f32x4 t4 = *reinterpret_cast<const f32x4 *>(bias + co);
            bv[i][0] = t4[0]; bv[i][1] = t4[1]; bv[i][2] = t4[2]; bv[i][3] = t4[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[i][r] = (bias != nullptr && co + r < p.Cout) ? bias[co + r] : 0.f;
        }
    }
    auto store_all = [&](auto actf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oy = y0 + 4 * wc + j, ox = x0 + (lane & 15);
            if (oy >= p.H || ox >= p.W) continue;
            const size_t po = ((size_t)n * p.H + oy) * p.W + ox;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + wr * 64 + 16 * i + q4;
                if (co >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = actf(acc[i][j][r] + bv[i][r]);
                const size_t off = po * p.ld_out + co;
                if (p.addend != nullptr) {
                    const bf16_t *ad = reinterpret_cast<const bf16_t *>(p.addend) + po * p.ld_add + co;
                    if ((p.ld_add & 3) == 0 && co + 3 < p.Cout) {
                        const bf16x4 t4 = *reinterpret_cast<const bf16x4 *>(ad);
                        for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t4[r]);
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
    };
    if (p.act == RDFC_ACT_TANH) store_all([](float u) { return tanhf(u); });
    else if (p.act == RDFC_ACT_SIGMOID) store_all([](float u) { return 1.f / (1.f + expf(-u)); });
    else {
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        store_all([ns](float u) { return u > 0.f ? u : u * ns; });
    }
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

template <int TCO, int DT>
int launch(const PParams &p, const void *x, const void *w, const float *bias, void *y, const bf16_t *zero, hipStream_t st) {
    constexpr int TR = 4 * (8 / (TCO / 64)), NPIECE = ((TR + 2) * PW * 4 + 63) / 64;
    constexpr int SMEM = 2 * NPIECE * 1024 + RING * TCO * BK * 2 + 1024;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)conv3x3_pp_kernel<TCO, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * p.N, cdiv(p.Cout, TCO));
    rdfc_note_kernel("conv3x3_pp_kernel<%d, %d>", TCO, DT);
    hipLaunchKernelGGL((conv3x3_pp_kernel<TCO, DT>), grid, dim3(512), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)w, bias, y, zero);
    return 1;
}

}  // namespace

// Called by conv_igemm.hip's entry points ahead of conv_glds.hip's kernels; returns 1 if it took the launch.
// RDFC_PP: 0 = off, 1 (default) = the 128-channel tile where Cout >= 128, 2 = also the 64-channel tile.
int rdfc_try_conv3x3_pp(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                        hipStream_t st, const void *addend, int ld_add, int reflect) {
    const char *env_ = getenv("RDFC_PP");            // (read per call: tools/pp_bench.py switches it between launches of one process)
    const int mode = env_ ? atoi(env_) : 1;
    if (!mode || !rdfc_is16(d->dtype)) return 0;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->ph == d->pw && d->ph <= 1 && d->dh == 1 && d->dw == 1)) return 0;
    if (Cin % BK != 0 || d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return 0;
    const int ld_in = transposed ? d->ldy : d->ldx;
    if (ld_in % 8 != 0 || Cout < 64) return 0;
    if (Cout <= 64 && mode < 2) return 0;
    if (reflect && (transposed || d->ph < 1 || d->ph >= d->Hi || d->pw >= d->Wi)) return 0;
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    const int pad = d->ph;
    PParams p{d->N, transposed ? d->Ho : d->Hi, transposed ? d->Wo : d->Wi, transposed ? d->Hi : d->Ho, transposed ? d->Wi : d->Wo,
              Cin, Cout, ld_in, transposed ? d->ldx : d->ldy, transposed, transposed ? pad - 2 : -pad, transposed ? pad - 2 : -pad,
              d->act, d->slope, d->out_f32, 0, 0, addend, ld_add, reflect};
    p.tiles_x = cdiv(p.W, TC);
    if (Cout > 64) {
        p.tiles_y = cdiv(p.H, 16);
        return d->dtype == RDFC_F16 ? launch<128, RDFC_F16>(p, in, w, bias, out, zero, st) : launch<128, RDFC_BF16>(p, in, w, bias, out, zero, st);
    }
    p.tiles_y = cdiv(p.H, 32);
    return d->dtype == RDFC_F16 ? launch<64, RDFC_F16>(p, in, w, bias, out, zero, st) : launch<64, RDFC_BF16>(p, in, w, bias, out, zero, st);
}
