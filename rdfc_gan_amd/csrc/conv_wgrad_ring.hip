// conv_wgrad_ring.hip -- weight gradient of 3x3 / 4x4 convolutions of stride 1 or 2 (2-byte elements), LDS-DMA staging, flat pixel
// walk, deterministic split-K (partial slabs + a fixed-order reduction instead of float atomics).
//
//   dW[co][ky][kx][ci] += sum over output pixels m of dY[m][co] * X'[m + (ky, kx)][ci]        X' = X zero-padded by `pad`   (stride 1)
//
// Stride 2 (the encoders' down-sampling 3x3 convs, the transposed decoder convs seen from their output, PatchGAN's 4x4 layers):
// input column 2 x + kx - 1 = 2 (x + sx) + px, so tap kx reads column x + sx of the PARITY PLANE px of its input row -- within a
// plane the taps are shifts of one flat sequence again.  A kernel row needs both planes: two X rings, each filled with a lane
// stride of two pixels; taps (plane, shift): 3x3: (1,0) (0,1) (1,1); 4x4: (1,0) (0,1) (1,1) (0,2).
//
// GEMM view as conv_wgrad_glds.hip: M = co, N = (kx, ci), K = pixels; a workgroup (4 waves) owns one kernel row ky, a 128 | 64 co x
// 64 ci tile and a slice of K.  What is different:
//   * K walks a FLAT pixel index over rows padded to L pixels (L = the next multiple of 8 >= Wo + 2): output position f = row * L + x
//     reads X' at f + ky * L + kx.  No k-step is part of an "empty" row segment -- conv_wgrad_glds.hip's 32-pixel segments of one
//     row leave 21 % (W = 76) / 41 % (W = 38) of the pixels of its k-steps empty.
//   * X' is staged ONCE, as a ring of consecutive 32-position blocks contiguous in LDS (block b = positions 32 b .. 32 b + 31 of the
//     workgroup's range, landed one step ahead of dY's block b): tap kx of step s reads ring rows 32 s + kx .. + 31 -- the 34-row
//     image per step of the old kernel re-fetched two halo pixels per step and needed a per-lane row / column for every piece.
//   * L is a multiple of 8 and a DMA instruction covers 4 or 8 consecutive positions, so every instruction's pixels lie in ONE row:
//     row, column and base pointer of a piece are wave-uniform (SALU, advanced incrementally from step to step); a lane adds its
//     constant offset and tests its column (5-6 VALU per piece; the per-lane div / mod walk of the old flat mode cost ~18).
//   * split-K partials go to a scratch slab per slice (plain stores) and a second kernel adds the slices in a fixed order to dW:
//     bitwise reproducible, and 50 MB of stores + a 50 MB read instead of 50 MB of memory-side float atomics (~1.3 TB/s).
#include "common.h"

#include <mutex>
#include <stdlib.h>

namespace {

struct RParams {
    int N, Hi, Wi, Ci, Ho, Wo, Co, pad;
    int ldx, ldy;
    int L, nrows;                // padded row length, N * Ho
    int total_steps, steps_per_slice, nslices;
    int tiles, tiles_ci, xcd_group;
    int reflect;
    int xoff;                    // input column of flat column 0 of plane 0: -pad (stride 1), -2 (stride 2, pad 1: column 2 (j - 1))
    unsigned x_bytes, y_bytes;   // extents of the two maps (wgrad_ring2_kernel's buffer descriptors; < 2^31)
};

constexpr int SEG = 32, NT = 256;

template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// swizzle of the 32-byte channel-pair index by image row (conv_wgrad_glds.hip: conflict-free transposing reads for every kx)
template <int ROWB> __device__ inline int pair_swz(int row) {
    if constexpr (ROWB == 256) return (row & 3) | (((row >> 3) & 1) << 2);
    else return ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
}

// tap kx of a kernel row -> (parity plane, shift within the plane's flat sequence)
template <int S> __device__ constexpr int tap_plane(int kx) { return S == 1 ? 0 : ((kx & 1) ^ 1); }
template <int S> __device__ constexpr int tap_shift(int kx) { return S == 1 ? kx : (kx + 1) / 2; }

// ---- round 5: the walk with (almost) no vector or scalar address arithmetic in the loop ----------------------------------------------
// Round 3's kernel (wgrad_ring_kernel, removed in round 6 after a round as the A/B leg: bit-equal on all 20 benchmark shapes,
// profiles/r05_wgrad_ring2_ab.txt) spent 94 vector and 138 scalar instructions on a k-step of 24 MFMAs (hipcc -S, tools/isa_loop.py):
// per-piece source pointers with a zero-page select per lane, the swizzled address of every transposing read recomputed from the step's
// ring position, the cursors' wrap logic as branches.  Two waves per SIMD cannot issue that beside 2 x 24 MFMAs: it was issue-bound at
// 0.24 of the matrix peak.  wgrad_ring2_kernel keeps the geometry (flat pixel walk, dY slots, X' ring, slab output: same bits) and
// changes how it is addressed:
//   * LDS-DMA through buffer descriptors (conv_glds.hip: bdma16): a piece lies in ONE row of its map, so its descriptor is that row --
//     base = the row's first byte (dY: the piece's first column), num_records = the row's extent -- and every lane's offset is a
//     CONSTANT (dY) or a constant plus one wave-uniform column term (X': one v_add; a column left of the row wraps to a huge offset).
//     Lanes beyond the row, rows outside the image and blocks beyond the slice are out of range and arrive as zeros: no zero page, no
//     per-lane compare / select, no 64-bit vector address.
//   * the k loop is unrolled over the X' ring (8 | 4 steps), so slot and block of every read are immediates and each transposing read's
//     per-lane address is one of 20 registers computed once (the reads of the ring's last block that run over its end have their own).
//   * PIPE (stride 1): the fragments of step s + 1 are read during the MFMAs of step s -- B fragments in place, right after the four
//     MFMAs that end their use, A fragments into the other half of a register pair (the unrolled loop makes its parity static) -- so
//     the reads are off the critical path of a step as in conv3x3_glds.
constexpr unsigned DEAD_OFF = 0x80000000u;
__device__ inline void bdma16(unsigned voff, u32x4 rsrc, unsigned lds_base) {
    // m0 is declared clobbered, not saved and restored (the compiler uses it for nothing else in this kernel; clang warns that m0 is
    // reserved): two scalar instructions fewer per DMA in a loop that was bound by scalar issue
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_base) : "memory", "m0");
}
__device__ inline u32x4 row_rsrc(const void *base, unsigned bytes) {          // base and bytes wave-uniform
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ inline bf16x8 tr_pair(const unsigned char *a0, int half_off) {     // one MFMA fragment = two transposing reads, `half_off` bytes apart
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(a0 + half_off));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ inline bf16x8 tr_pair2(const unsigned char *a0, const unsigned char *a1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)a1);
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    const u32x4 r = {l2[0], l2[1], h2[0], h2[1]};
    return __builtin_bit_cast(bf16x8, r);
}

// KG (round 6): kernel rows per workgroup.  KG = 1: a 256-thread workgroup owns ONE kernel row ky of its (co, ci) tile (blockIdx.y counts
// (tile, ky)).  KG = K: a workgroup is K such four-wave groups side by side -- group g = ky, its own slots and rings in its own third of the
// LDS, the same code -- that share NOTHING BUT THE BARRIER: the K groups walk the slice in lockstep, so the dY block they all stage and the
// X' rows they stage one row apart are fetched from the fabric once and found in the L2 / L1 by the other two.  As separate workgroups
// the three drifted apart: 472 MB fetched per launch on the 64-channel full-resolution layers for 260 MB of operands
// (profiles/r05_pmc_traffic.json), in the phase of the step where the convolutions and normalisation passes of the same maps want the same HBM.
template <int TCO, int DT, int S = 1, int K = 3, bool REFL = false, int KG = 1>
__global__ __launch_bounds__(NT * KG, KG > 1 ? 1 : (TCO == 128 ? 2 : 3)) void wgrad_ring2_kernel(RParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ dy,
                                                                           float *__restrict__ slab) {
    constexpr int TCI = 64, NPL = S;
    constexpr bool PIPE = S == 1;
    constexpr int LOOK = S == 1 ? 3 : 2, RY = 4, XR = S == 1 ? 8 : 4, U = XR;      // U: steps per trip of the unrolled loop (a multiple of RY)
    constexpr int YROW = TCO * 2, XROW = TCI * 2;
    constexpr int YB = SEG * YROW, XBLK = SEG * XROW;
    constexpr int NY = YB / (NT * 16), NX = XBLK / (NT * 16);
    static_assert(NX == 1, "one X' piece per wave, plane and step");
    constexpr int PER_STEP = NY + NPL * NX;
    constexpr int CPY = TCO / 8, CPX = TCI / 8;
    constexpr int OFF_X = RY * YB, XRING = XR * XBLK;
    constexpr int WCO = TCO / 64, WCI = 4 / WCO;
    constexpr int WTCO = TCO / WCO, WTCI = TCI / WCI;
    constexpr int MT = WTCO / 16, NTF = WTCI / 16;
    constexpr int NSH = S == 1 ? K : K / 2 + 1;                 // distinct shifts within a plane: 3 | 4 (stride 1), 2 | 3 (stride 2)
    static_assert(XR >= LOOK + 2 && U % RY == 0, "ring sizes");
    static_assert(KG == 1 || KG == K, "one kernel row per workgroup, or all of them");
    constexpr int SMEM_G = RY * YB + NPL * XRING;                // LDS of one four-wave group
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_all[];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
    unsigned char *smem = smem_all + grp * SMEM_G;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem_all + (unsigned)(grp * SMEM_G);

    const int tid = threadIdx.x & (NT - 1), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.y, slice = blockIdx.x;
    if (p.xcd_group) {
        const int k = blockIdx.x >> 3;
        tile = k % p.tiles;
        slice = (k / p.tiles) * 8 + (blockIdx.x & 7);
    }
    if (slice >= p.nslices) return;
    int ky = grp;
    if (KG == 1) {
        ky = tile % K;
        tile /= K;
    }
    const int ci0 = (tile % p.tiles_ci) * TCI, co0 = (tile / p.tiles_ci) * TCO;
    const int seg0 = slice * p.steps_per_slice;
    int nseg = p.total_steps - seg0;
    if (nseg > p.steps_per_slice) nseg = p.steps_per_slice;

    // ---- DMA sources: ONE descriptor per map; per lane a 32-bit byte offset that advances by a wave-uniform increment per step ---------
    // Wave w stages positions 8 w ... 8 w + 7 of every 32-position block, of dY (one or two instructions: 64 / CPY positions each) and
    // of X' (one instruction per plane).  L is a multiple of 8, so these eight positions lie in ONE row of the padded walk: where the
    // wave is -- column j of its first position, output row yy, image img -- is ONE wave-uniform cursor, advanced with scalar adds and
    // selects only (no multiply, no 64-bit arithmetic, no branch in the loop).  A lane adds the step's increment to its offsets and
    // is switched to DEAD (beyond num_records: zeros) while its column is outside the row or the row outside the map.  dY's block is
    // the one the cursor stood on a step earlier (X' is staged a block ahead): its column / image are the saved previous values.
    constexpr int PY = 64 / CPY;                                 // positions per dY instruction (4 | 8)
    const int ylp = lane / CPY, xlp = lane / CPX;
    const u32x4 ysrc = row_rsrc(dy, (unsigned)p.y_bytes), xsrc = row_rsrc(x, (unsigned)p.x_bytes);
    int cj, cy, ci;                                              // the cursor: position 8 w of the next X' block
    {
        const long f = (long)seg0 * SEG + 8 * wave;
        const int row = (int)(f / p.L);
        cj = (int)(f - (long)row * p.L);
        ci = row / p.Ho;
        cy = row - ci * p.Ho;
    }
    unsigned yoff[NY], xoffv[NPL];
    int ycol[NY];                                                // the lane's position inside the wave's eight
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        const int pos = 8 * wave + PY * k + ylp, slot = lane % CPY;
        const int src = slot ^ (2 * pair_swz<YROW>(pos));        // the chunk this lane FETCHES (the swizzle is applied to the source)
        ycol[k] = PY * k + ylp;
        yoff[k] = 2u * (unsigned)(((long)(ci * p.Ho + cy) * p.Wo + cj + ycol[k]) * p.ldy + co0 + (src << 3));
        if (co0 + (src << 3) >= p.Co) yoff[k] = DEAD_OFF;        // (channels beyond Co -- a 128-co tile's second half, the upper 32 of a 32-channel layer: stays beyond num_records whatever is added)
    }
    const int xslot = lane % CPX, xpos = 8 * wave + xlp;
    const int xchunk = ci0 + ((xslot ^ (2 * pair_swz<XROW>(xpos))) << 3);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)                             // (may be "negative" -- a row above the image, a column left of it -- while the lane is DEAD)
        xoffv[pl] = 2u * (unsigned)(((long)(ci * p.Hi + S * cy + ky - p.pad) * p.Wi + S * (cj + xlp) + p.xoff + pl) * p.ldx + xchunk);
    const unsigned ystep = 2u * (unsigned)(SEG * p.ldy), ywrap = 2u * (unsigned)((p.Wo - p.L) * p.ldy);       // + ywrap (mod 2^32) when the column wraps into the next row
    const unsigned xstep = 2u * (unsigned)(SEG * S * p.ldx), xwrap = 2u * (unsigned)(S * (p.Wi - p.L) * p.ldx),
                   ximg = 2u * (unsigned)((p.Hi - S * p.Ho) * p.Wi * p.ldx);
    int pj = cj, pi = ci;                                        // where the cursor stood a block earlier (the next dY block)
    unsigned yinc = 0;                                           // what takes yoff from that block to the one after it
    auto advance = [&]() {                                       // the cursor moves on by one block; -> the X' offsets follow, dY's increment is kept for later
        pj = cj;
        pi = ci;
        cj += SEG;
        const bool w = cj >= p.L;                                // (SEG <= L: at most one wrap per step)
        cj -= w ? p.L : 0;
        cy += w ? 1 : 0;
        const bool wi_ = cy == p.Ho;                             // (cy reaches Ho only by the increment above)
        cy = wi_ ? 0 : cy;
        ci += wi_ ? 1 : 0;
        yinc = ystep + (w ? ywrap : 0u);
        const unsigned xinc = xstep + (w ? xwrap : 0u) + (wi_ ? ximg : 0u);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) xoffv[pl] += xinc;
    };
    auto issue_x = [&](int bx, int blk_) {                       // X' block `bx` of the slice (the cursor's block, every plane) -> ring block blk_; the cursor moves on
        int yin = S * cy + ky - p.pad;
        if (REFL) yin = yin < 0 ? -yin : (yin >= p.Hi ? 2 * p.Hi - 2 - yin : yin);
        const bool rowok = bx <= nseg && ci < p.N && (unsigned)yin < (unsigned)p.Hi;       // (one block beyond the last dY block: its first positions are read)
        const int xcol = S * cj + p.xoff;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            int xin = xcol + S * xlp + pl;                       // the lane's input column
            unsigned v = xoffv[pl];
            if (REFL) {                                          // mirrored halo: the source follows the reflection (its row part is wave-uniform)
                xin = xin < 0 ? -xin : (xin >= p.Wi ? 2 * p.Wi - 2 - xin : xin);
                v = 2u * (unsigned)(((ci * p.Hi + yin) * p.Wi + xin) * p.ldx + xchunk);
            }
            bdma16((rowok && (unsigned)xin < (unsigned)p.Wi) ? v : DEAD_OFF, xsrc, lds0 + OFF_X + pl * XRING + blk_ * XBLK + wave * 1024);
        }
        advance();
    };
    auto issue_y = [&](int by, int slot_) {                      // dY block `by` (where the cursor stood before its last move) -> slot slot_
        const bool rowok = by < nseg && pi < p.N;
        const int left = p.Wo - pj;                              // lanes at positions below it are inside the row
#pragma unroll
        for (int k = 0; k < NY; ++k) {
            bdma16((rowok && ycol[k] < left) ? yoff[k] : DEAD_OFF, ysrc, lds0 + slot_ * YB + (8 * wave + PY * k) * YROW);
            yoff[k] += yinc;
        }
    };

    // ---- per-lane LDS offsets of the transposing reads (computed once) ----------------------------------------------------------------
    const int wr = wave / WCI, wc = wave % WCI;
    const int g_ = lane >> 4, li = lane & 15, q_ = li >> 2, pp = li & 3;
    unsigned aoff[MT];                                           // dY slot: rows 8 g + q (second half: + 4 rows, same swizzle)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r0 = 8 * g_ + q_, cp = (wr * WTCO + 16 * i) >> 4;
        aoff[i] = (unsigned)(r0 * YROW + ((cp ^ pair_swz<YROW>(r0)) << 5) + pp * 8);
    }
    unsigned boff[NSH][NTF][2], bwrap[NSH][NTF];                 // X' block: rows 8 g + q + shift (+ 4); bwrap: the second half in the ring's last block
#pragma unroll
    for (int sh = 0; sh < NSH; ++sh)
#pragma unroll
        for (int j = 0; j < NTF; ++j) {
            const int cp = (wc * WTCI + 16 * j) >> 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r = 8 * g_ + q_ + sh + 4 * h;
                boff[sh][j][h] = (unsigned)(r * XROW + ((cp ^ pair_swz<XROW>(r)) << 5) + pp * 8);
            }
            const int r1 = 8 * g_ + q_ + sh + 4;
            bwrap[sh][j] = boff[sh][j][1] - (r1 >= SEG ? (unsigned)XRING : 0u);
        }

    f32x4 acc[K][MT][NTF];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) acc[t][i][j] = f32x4{0, 0, 0, 0};

    auto read_a = [&](int slot_, int i) -> bf16x8 { return tr_pair(smem + slot_ * YB + aoff[i], 4 * YROW); };
    auto read_b = [&](int blk_, int t, int j) -> bf16x8 {       // tap t of the block at ring position blk_ (compile-time after unrolling)
        const int sh = tap_shift<S>(t);
        const unsigned char *Xp = smem + OFF_X + tap_plane<S>(t) * XRING + blk_ * XBLK;
        return tr_pair2(Xp + boff[sh][j][0], Xp + ((blk_ == XR - 1 && sh > 0) ? bwrap[sh][j] : boff[sh][j][1]));
    };

    if (nseg > 0) {
        issue_x(0, 0);                                           // (issue_y(b) follows issue_x(b): it stages the block the cursor has just left)
#pragma unroll
        for (int h = 0; h < LOOK; ++h) {
            issue_y(h, h % RY);
            issue_x(h + 1, (h + 1) % XR);
        }
        bf16x8 a[2][MT], b[K][NTF];
        if constexpr (PIPE) {
            wait_vm<(LOOK - 1) * PER_STEP>();                   // dY(0), X'(0), X'(1)
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < MT; ++i) a[0][i] = read_a(0, i);
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int j = 0; j < NTF; ++j) b[t][j] = read_b(0, t, j);
        }
#define RDFC_WR_STEP(UU)                                                                                                   \
        if (s0 + UU < nseg) {                                                                                              \
            constexpr int CUR = UU & 1;                                                                                    \
            if constexpr (PIPE) wait_vm_and_reads<(LOOK - 2) * PER_STEP>();       /* dY(s + 1), X'(s + 1), X'(s + 2) have landed */ \
            else wait_vm_and_reads<(LOOK - 1) * PER_STEP>();                      /* dY(s), X'(s), X'(s + 1) */          \
            __builtin_amdgcn_s_barrier();                                                                                  \
            issue_y(s0 + UU + LOOK, (UU + LOOK) % RY);                                                                     \
            issue_x(s0 + UU + LOOK + 1, (UU + LOOK + 1) % XR);                                                             \
            if constexpr (!PIPE) {                                                                                         \
                _Pragma("unroll") for (int i = 0; i < MT; ++i) a[0][i] = read_a(UU % RY, i);                             \
                _Pragma("unroll") for (int t = 0; t < K; ++t) {                                                            \
                    _Pragma("unroll") for (int j = 0; j < NTF; ++j) b[t][j] = read_b(UU % XR, t, j);                     \
                    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                         \
                        _Pragma("unroll") for (int j = 0; j < NTF; ++j) acc[t][i][j] = mma16<DT>(a[0][i], b[t][j], acc[t][i][j]); \
                }                                                                                                          \
            } else {                                                                                                       \
                _Pragma("unroll") for (int t = 0; t < K; ++t)                                                              \
                    _Pragma("unroll") for (int j = 0; j < NTF; ++j) {                                                      \
                        _Pragma("unroll") for (int i = 0; i < MT; ++i) acc[t][i][j] = mma16<DT>(a[CUR][i], b[t][j], acc[t][i][j]); \
                        __builtin_amdgcn_sched_barrier(0);                                                                 \
                        b[t][j] = read_b((UU + 1) % XR, t, j);                   /* next step's fragment, in place */     \
                        _Pragma("unroll") for (int i = 0; i < MT; ++i)          /* ... and the next step's A fragments, spread over the groups */ \
                            if (i % (K * NTF) == t * NTF + j) a[CUR ^ 1][i] = read_a((UU + 1) % RY, i);                    \
                        __builtin_amdgcn_sched_barrier(0);                                                                 \
                    }                                                                                                      \
            }                                                                                                              \
        }
        for (int s0 = 0; s0 < nseg; s0 += U) {
            RDFC_WR_STEP(0) RDFC_WR_STEP(1) RDFC_WR_STEP(2) RDFC_WR_STEP(3)
            if constexpr (U == 8) { RDFC_WR_STEP(4) RDFC_WR_STEP(5) RDFC_WR_STEP(6) RDFC_WR_STEP(7) }
        }
#undef RDFC_WR_STEP
        wait_vm<0>();
    }

    float *out = slab + (size_t)slice * p.Co * (K * K) * p.Ci;
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTF; ++j) {
                const int ci = ci0 + wc * WTCI + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wr * WTCO + 16 * i + 4 * (lane >> 4) + r;
                    if (co < p.Co && ci < p.Ci) out[((size_t)co * (K * K) + ky * K + t) * p.Ci + ci] = acc[t][i][j][r];
                }
            }
}

// dW[i] += slab[0][i] + slab[1][i] + ... in a FIXED order, whatever order the workgroups finished in.  A workgroup owns 16
// consecutive f32x4 columns; its 256 threads are 16 slice groups x 16 columns: group g adds slices g, g + 16, ... (in that order,
// four loads in flight), then the 16 group sums are added in group order through LDS.  (One thread per column summing every slice
// in turn left a 64 -> 64 layer's 256 slices to 9 216 threads: a ~64-deep chain of dependent memory round trips.)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dw, size_t n4, int nslices) {
    __shared__ f32x4 red[16][16];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + col;
    f32x4 s = f32x4{0, 0, 0, 0};
    if (i < n4) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(slab) + i;
        int k = grp;
        for (; k + 48 < nslices; k += 64) {
            const f32x4 v0 = src[(size_t)k * n4], v1 = src[(size_t)(k + 16) * n4], v2 = src[(size_t)(k + 32) * n4], v3 = src[(size_t)(k + 48) * n4];
            s = ((s + v0) + v1) + v2;
            s = s + v3;
        }
        for (; k < nslices; k += 16) s = s + src[(size_t)k * n4];
    }
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < n4) {
        f32x4 t = red[0][col];
#pragma unroll
        for (int g = 1; g < 16; ++g) t = t + red[g][col];
        // (atomic: with RDFC_WGRAD_STREAM=0 two uses of one parameter may run their weight gradients on different streams; in the
        //  default configuration every weight gradient is on the one side stream, i.e. one add per element at a time, in launch order)
        float *d = dw + 4 * i;
        atomicAdd(d, t[0]);
        atomicAdd(d + 1, t[1]);
        atomicAdd(d + 2, t[2]);
        atomicAdd(d + 3, t[3]);
    }
}

// Scratch for the partial slabs: one buffer per (device, stream) -- launches on one stream run one after the other, so the next
// launch's workgroups cannot overwrite slabs a reduction still reads.  Grown on demand, with two exceptions: never while the stream is
// being captured (no allocation inside a capture: the launch then goes to the float-atomics kernels unless the stream's slab was
// reserved beforehand, rdfc_wgrad_ring_reserve), and never once a captured graph holds the pointer (a replay would read freed memory).
namespace {
struct SlabEnt {
    int dev;
    hipStream_t st;
    void *ptr;
    size_t cap;
    bool pinned;                 // a captured graph refers to ptr: neither freed nor moved any more
};
SlabEnt g_slabs[64];
int g_nslabs = 0;
std::mutex g_slab_mu;
}  // namespace

float *slab_for(hipStream_t st, size_t bytes, bool reserve_only = false) {
    std::lock_guard<std::mutex> lock(g_slab_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    SlabEnt *e = nullptr;
    for (int i = 0; i < g_nslabs; ++i)
        if (g_slabs[i].dev == dev && g_slabs[i].st == st) e = &g_slabs[i];
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    if (e != nullptr && e->cap >= bytes) {
        if (capturing) e->pinned = true;
        return (float *)e->ptr;
    }
    if (capturing || (e != nullptr && e->pinned)) return nullptr;           // no allocation inside a capture; a pinned slab stays where it is
    if (e == nullptr) {
        if (g_nslabs == 64) return nullptr;
        e = &g_slabs[g_nslabs++];
        *e = SlabEnt{dev, st, nullptr, 0, false};
    }
    if (e->ptr != nullptr) {
        if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
        (void)hipFree(e->ptr);
        e->ptr = nullptr;
        e->cap = 0;
    }
    size_t cap = bytes < (size_t)(64u << 20) ? (size_t)(64u << 20) : bytes + bytes / 4;
    if (hipMalloc(&e->ptr, cap) != hipSuccess) {
        e->ptr = nullptr;
        return nullptr;
    }
    e->cap = cap;
    (void)reserve_only;
    return (float *)e->ptr;
}

// RDFC_WGRAD_KG=0: one kernel row per 256-thread workgroup everywhere (rounds 3-5; A/B); rdfc_tune_wgrad_kg switches inside a process
// (measurement / test aid like rdfc_tune_wgrad_ring: process-wide, not thread-safe)
static bool g_ring_kg = !(getenv("RDFC_WGRAD_KG") != nullptr && atoi(getenv("RDFC_WGRAD_KG")) == 0);
extern "C" void rdfc_tune_wgrad_kg(int on) { g_ring_kg = on != 0; }

template <int TCO, int DT, int S, int K>
int launch(RParams p, const void *x, const void *dy, float *dw, hipStream_t st) {
    constexpr int XR = S == 1 ? 8 : 4;
    constexpr int SMEM = 4 * (SEG * TCO * 2) + S * XR * (SEG * 64 * 2);
    // the 64-co tile of the 3x3 / stride-1 layers (48 KB, three workgroups per CU): the three kernel rows of a tile as ONE 768-thread
    // workgroup in lockstep (the kernel's KG) -- same LDS per CU, same waves per SIMD, dY and X' fetched from the fabric once
    constexpr bool KG_OK = TCO == 64 && S == 1 && K == 3;
    const bool kg = KG_OK && g_ring_kg && !p.reflect;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)wgrad_ring2_kernel<TCO, DT, S, K, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        if constexpr (S == 1 && K == 3)
            if (hipFuncSetAttribute((const void *)wgrad_ring2_kernel<TCO, DT, S, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        if constexpr (KG_OK)
            if (hipFuncSetAttribute((const void *)wgrad_ring2_kernel<TCO, DT, S, K, false, K>, hipFuncAttributeMaxDynamicSharedMemorySize, K * SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    p.tiles_ci = cdiv(p.Ci, 64);
    const int tiles = cdiv(p.Co, TCO) * p.tiles_ci * (kg ? 1 : K);
    // one round of the chip's workgroup slots (2 per CU with the 128-co tile, 3 with the 64-co one, 1 with three kernel rows per
    // workgroup) within a slab budget: every workgroup writes its tile once (plain stores) and the reduction reads it once
    constexpr double cap_mb = 64.0;
    const int slots = kg ? 256 : (TCO == 64 ? 768 : 512);
    int slices = slots / tiles;
    const size_t dw_bytes = (size_t)4 * p.Co * p.Ci * K * K;
    const int cap = (int)(cap_mb * 1.0e6 / (double)dw_bytes);
    if (slices > cap) slices = cap;
    if (slices > cdiv(p.total_steps, 8)) slices = cdiv(p.total_steps, 8);
    if (slices < 1) slices = 1;
    p.steps_per_slice = cdiv(p.total_steps, slices);
    slices = cdiv(p.total_steps, p.steps_per_slice);
    p.nslices = slices;
    p.tiles = tiles;
    float *slab = slab_for(st, dw_bytes * slices);
    if (slab == nullptr) return 0;
    p.xcd_group = tiles <= 4 && slices >= 8;
    const dim3 grid = p.xcd_group ? dim3(cdiv(slices, 8) * 8 * tiles, 1) : dim3(slices, tiles);
    // (the symbol as rocprofv3 prints it: every template argument, defaults included)
    rdfc_note_kernel("wgrad_ring2_kernel<%d, %d, %d, %d, %s, %d>", TCO, DT, S, K, p.reflect ? "true" : "false", kg ? K : 1);
    if constexpr (KG_OK) {
        if (kg) {
            hipLaunchKernelGGL((wgrad_ring2_kernel<TCO, DT, S, K, false, K>), grid, dim3(NT * K), K * SMEM, st, p, (const bf16_t *)x, (const bf16_t *)dy, slab);
            const size_t n4 = dw_bytes / 16;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, (const float *)slab, dw, n4, slices);
            return 1;
        }
    }
    if constexpr (S == 1 && K == 3) {
        if (p.reflect) hipLaunchKernelGGL((wgrad_ring2_kernel<TCO, DT, S, K, true>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)dy, slab);
        else hipLaunchKernelGGL((wgrad_ring2_kernel<TCO, DT, S, K, false>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)dy, slab);
    } else
        hipLaunchKernelGGL((wgrad_ring2_kernel<TCO, DT, S, K, false>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)dy, slab);
    const size_t n4 = dw_bytes / 16;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, (const float *)slab, dw, n4, slices);
    return 1;
}

template <int S, int K>
int launch_sk(const rdfc_conv_desc *d, const RParams &p, const void *x, const void *dy, float *dw, hipStream_t st) {
    // 128-co tiles from 256 output channels up (as conv_wgrad_glds.hip: below that the 64-co tile's larger grid wins)
    if (d->Co % 128 == 0 && d->Co >= 256)
        return d->dtype == RDFC_F16 ? launch<128, RDFC_F16, S, K>(p, x, dy, dw, st) : launch<128, RDFC_BF16, S, K>(p, x, dy, dw, st);
    return d->dtype == RDFC_F16 ? launch<64, RDFC_F16, S, K>(p, x, dy, dw, st) : launch<64, RDFC_BF16, S, K>(p, x, dy, dw, st);
}

}  // namespace

// whether this kernel takes the launch (shape rules + the measured exceptions); L = padded row length of the flat pixel walk
// 0: off; 1: 3x3 / stride 1 only; 2 (default): also stride 2 and 4x4; 3: 2 without the size exceptions (tests).  Read from RDFC_WGRAD_RING
// ONCE (round 5 called getenv on every weight-gradient launch); tools/wgrad_ring_bench.py switches it between launches through
// rdfc_tune_wgrad_ring (a measurement aid like rdfc_tune_glds_rows: process-wide, not thread-safe).
static int g_ring_mode = getenv("RDFC_WGRAD_RING") != nullptr ? atoi(getenv("RDFC_WGRAD_RING")) : 2;
extern "C" void rdfc_tune_wgrad_ring(int mode) { g_ring_mode = mode; }
static bool ring_takes(const rdfc_conv_desc *d, int reflect, int &L) {
    const int mode = g_ring_mode;
    if (mode == 0 || !rdfc_is16(d->dtype)) return false;
    const int S = d->sh, K = d->KH;
    if (!(d->KH == d->KW && (K == 3 || K == 4) && d->sh == d->sw && (S == 1 || S == 2) && d->dh == 1 && d->dw == 1 && d->ph == d->pw && d->ph <= 1)) return false;
    if ((S != 1 || K != 3) && (mode < 2 || d->ph != 1 || reflect)) return false;
    if (d->Ho != (d->Hi + 2 * d->ph - K) / S + 1 || d->Wo != (d->Wi + 2 * d->pw - K) / S + 1) return false;
    if (d->Ci % 64 || d->ldx % 8 || d->ldy % 8 || d->Wo < 24) return false;
    {   // both maps are addressed through 32-bit byte offsets (buffer descriptors); larger ones go to the tap-gather / register-staged kernels
        const size_t xb = 2 * (size_t)d->N * d->Hi * d->Wi * d->ldx, yb = 2 * (size_t)d->N * d->Ho * d->Wo * d->ldy;
        if (xb >= 0x7ffffff0u || yb >= 0x7ffffff0u) return false;
    }
    // 32 output channels (the 128 -> 32 head convs at full resolution): the 64-co tile with its upper half beyond Cout -- zeros from the
    // descriptor's range check, nothing stored
    if (d->Co % 64 && !(d->Co == 32 && S == 1 && K == 3 && mode >= 2)) return false;
    if (reflect && (d->ph < 1 || d->ph >= d->Hi || d->pw >= d->Wi)) return false;
    const int omax = S == 1 ? K - 1 : K / 2;                   // largest tap shift within a plane's flat sequence
    L = (d->Wo + omax + 7) / 8 * 8;
    if (L < 32) L = 32;
    if ((double)d->N * d->Ho * L >= 2.0e9) return false;
    // Measured exceptions (profiles/r03_wgrad_ring_ab.txt): the 64-channel full-resolution inputs of the stride-2 layers (64 -> 128 / 192 from
    // 16 x 228 x 304: every kernel row and co tile streams both parity planes of the 142 MB input again; the tap-gather kernel is
    // 12-35 % faster there), and launches too small to pay for the second (reduction) launch.
    if (mode < 3 && S == 2 && d->Ci < 128 && (double)d->N * d->Hi * d->Wi > 6.0e5) return false;
    if (mode < 3 && cdiv((long)d->N * d->Ho * L, SEG) < 256) return false;
    return true;
}

// include/rdfc_hip.h: 1 when rdfc_conv2d_wgrad[_reflect] of this layer adds its split-K partials in a fixed order (bitwise
// reproducible results), 0 when the launch goes to a kernel that accumulates with float atomics
int rdfc_wgrad_narrow_covers(const rdfc_conv_desc *d);            // conv_narrow.hip (per-workgroup partials added in workgroup order)
extern "C" int rdfc_conv2d_wgrad_deterministic(const rdfc_conv_desc *d, int reflect) {
    int L = 0;
    return d != nullptr && ((!reflect && rdfc_wgrad_narrow_covers(d)) || ring_takes(d, reflect, L)) ? 1 : 0;
}

// include/rdfc_hip.h: make sure `stream` owns a slab of at least `bytes` (0: the default 64 MB, enough for every layer of the hot path)
// BEFORE the stream is captured into a graph -- inside a capture nothing can be allocated, and a launch without a slab falls back to the
// float-atomics kernels (same values up to the order of the adds, not bitwise reproducible).
extern "C" int rdfc_wgrad_ring_reserve(void *stream, size_t bytes) {
    RDFC_REQUIRE(slab_for((hipStream_t)stream, bytes ? bytes : (size_t)(64u << 20), true) != nullptr,
                 "rdfc_wgrad_ring_reserve: cannot allocate %zu bytes for this stream (capturing already, or out of memory)", bytes);
    return 0;
}

// Called by rdfc_try_wgrad_glds (conv_wgrad_glds.hip) ahead of its own kernels; returns 1 if it took the launch.
// RDFC_WGRAD_RING=0 switches it off (A/B against the segment walk + float atomics).
int rdfc_try_wgrad_ring(const rdfc_conv_desc *d, const void *x, const void *dy, float *dw, hipStream_t st, int reflect) {
    int L = 0;
    if (!ring_takes(d, reflect, L)) return 0;
    const int S = d->sh, K = d->KH;
    RParams p{d->N, d->Hi, d->Wi, d->Ci, d->Ho, d->Wo, d->Co, d->ph, d->ldx, d->ldy, L, d->N * d->Ho, 0, 0, 0, 0, 0, 0, reflect, S == 1 ? -d->ph : -2, 0, 0};
    p.x_bytes = (unsigned)(2 * (size_t)d->N * d->Hi * d->Wi * d->ldx);      // (< 2^31: ring_takes)
    p.y_bytes = (unsigned)(2 * (size_t)d->N * d->Ho * d->Wo * d->ldy);
    p.total_steps = cdiv((long)p.nrows * p.L, SEG);
    if (S == 1 && K == 3) return launch_sk<1, 3>(d, p, x, dy, dw, st);
    if (S == 2 && K == 3) return launch_sk<2, 3>(d, p, x, dy, dw, st);
    if (S == 2 && K == 4) return launch_sk<2, 4>(d, p, x, dy, dw, st);
    return launch_sk<1, 4>(d, p, x, dy, dw, st);
}
