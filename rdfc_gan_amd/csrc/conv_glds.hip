// conv_glds.hip -- 3x3 / stride-1 convolution (forward and input-gradient), 2-byte elements, staged by LDS-DMA.
//
// Same decomposition as conv_patch.hip (a pixel tile per workgroup, the halo patch of a 32-channel slab in LDS, nine taps = nine
// LDS offsets) with a different pipeline.  PMC on the register-staged kernel (256->256 @ 57x76, profiles/): 3 SALU + 2 VALU per
// MFMA (address generation, bounds tests, the ds_write pass), SQ_WAIT_ANY 55 % of wave-cycles, matrix pipe busy 26 %.  Here
//   * both operands go global -> LDS by LDS-DMA through buffer descriptors (round 5: bdma16): a lane's source is a 32-bit byte offset
//     computed once per workgroup plus a wave-uniform per-phase offset in a scalar register; out-of-image pixels, the rows between
//     stacked images and channel rows beyond Cout are offsets beyond the descriptor's extent and arrive as ZEROS -- no zero page, no
//     select, no vector instruction per DMA; every wave issues the same number of DMAs, so the vmcnt arithmetic stays static;
//     the XOR bank swizzle of the LDS image is applied on the SOURCE side (LDS-DMA destinations are lane-linear);
//   * the weights run through a ring of 3 (128-co tiles) or 2 (64-co tiles) tap slots, the patch is double buffered and its DMAs are
//     spread over the first taps of the previous slab; loads stay in flight across the (raw) barrier: each phase waits with a counted
//     s_waitcnt vmcnt(N) for exactly the DMAs the NEXT phase's fragment reads need;
//   * a wave owns 64 co x (JR rows of 16 px); the taps are walked kernel column by kernel column with the column's pixel fragments
//     resident in registers, and every fragment read is issued a phase ahead, in place (see the kernel).
// One barrier per tap.  Workgroups are 4 waves: 128 co x (2 JR x 16) px, JR = 4 | 6 | 8 | 10 (glds_plan picks per launch), or
// 64 co x (16 x 16) px.  JR = 4 and the 64-co tile: 48 / 51 KB of LDS, 160 registers, three workgroups per CU; JR = 6 / 8 / 10:
// 186 / 219 / 253 registers and 57 / 67 / 75 KB, two per CU.
#include "common.h"

#include <stdlib.h>

namespace {

struct GParams {
    int N, Hin, Win, H, W, Cin, Cout;
    int ld_in, ld_out;
    int transposed, oy0, ox0;
    int act;
    float slope;
    int out_f32;
    int tiles_x, tiles_y;
    int vstride;                 // > 0: the batch is tiled as ONE tall map, image n at virtual rows [n * vstride, n * vstride + H) (stack_rows below)
    int gstride, npg;            // fused statistics over G groups of npg consecutive images: group g starts at virtual row g * gstride (a multiple
                                 // of the tile height: no tile straddles two groups), image l of it at + l * vstride; gstride 0: one run of images
    int stats_groups;            // fused statistics: 0 = off, else the number of groups G (1: batch norm, N: instance norm, in between: BatchNorm
                                 // over G stacked batches)
    const void *addend;          // optional map (2-byte elements, pixel stride ld_add) added to the result before the store: the
    int ld_add;                  // input gradient of a residual junction in one pass (rdfc_conv2d_dgrad_add)
    int reflect;                 // forward only: out-of-image patch pixels read their mirror image (ReflectionPad2d folded into the loader)
    int row_store;               // epilogue through LDS: whole 128-byte channel rows per store instruction (see the epilogue)
    unsigned x_bytes, w_bytes;   // extents of the input map and of the packed weights (buffer descriptors of the LDS-DMA loads; < 2^31)
    float r_tiles_x, r_tiles_y, r_vstride, r_gstride;      // 1 / the divisors of the tile and virtual-row arithmetic (qdiv below; 0 where the divisor is)
    int nchunk;                  // fused statistics: tiles per group = (tiles_y / stats_groups) * tiles_x
    float r_nchunk;
};

// a / d and a % d for 0 <= a < 2^22, d > 0 through the float reciprocal rd = 1.f / d (rounded): the quotient estimate is off by at most one
// either way and is corrected from the remainder.  hipcc expands an integer division by a run-time value into ~45 vector instructions;
// a tile's prologue and epilogue held twenty of them (a 2-slab tile's first DMA was issued ~1 000 instructions into the kernel: round 5).
__device__ __forceinline__ void qdiv(int a, int d, float rd, int &q, int &r) {
    q = (int)((float)a * rd);
    r = a - q * d;
    const int lo = r < 0 ? 1 : 0, hi = r >= d ? 1 : 0;          // (selects, no branches: this runs per lane in the DMA-source set-up)
    q += hi - lo;
    r += (lo - hi) * d;
}

constexpr int BK = 32, TC = 16, PW = TC + 2;
// weight ring: a ring of 3 tap slots (weights fetched two phases ahead) makes the 128-co workgroup 48 KB of LDS -> THREE per CU (the
// kernel needs 168 registers): 256->256 @ 57x76 58.4 -> 52.7 us (its 640 workgroups now fit one round), 128->128 @ 114x152 60.2 -> 53.1,
// 256->128 @ 114x152 98.2 -> 94.0.  The 64-co tiles (2 x 21 KB of patch) need a ring of TWO (weights fetched one phase ahead) to get
// three workgroups on a CU (50 KB each): 64->64 @ 16x228x304 160 -> 134 us forward, 163 -> 136 input gradient, 128->64 244 -> 218.
template <int TCO, int JR = 4> constexpr int ring_of() { return TCO == 128 ? 3 : 2; }       // (a ring of 4 for the tall tiles measured 1-3 % slower: r05_experiments)

// LDS-DMA instructions a wave may still have in flight when it enters phase s of a slab (what it reads then -- the weights of this
// phase, at s = 0 the slab's patch -- is older): the weights of the look - 1 phases after this one (nwp instructions each) and one
// patch round for each of the phases s - 1 ... s - max(look - 1, 1) that issued one (phases 0 ... nr - 1 of a slab do).
constexpr int glds_allow(int s, int look, int nwp, int nr) {
    int n = (look - 1) * nwp;
    for (int d = 1; d <= (look > 2 ? look - 1 : 1); ++d) n += ((s - d + 9) % 9) < nr ? 1 : 0;
    return n;
}

__device__ inline int lay_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 2)) << 3); }

// one LDS-DMA instruction: 64 lanes x 16 B, lane l lands at lds_base + 16 l
__device__ inline void glds16(const void *gsrc, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}
// The same through a buffer descriptor: lane l reads 16 bytes at base + soff + voff (voff per lane, soff wave-uniform) and lands at
// lds_base + 16 l.  A lane whose voff is not below the descriptor's num_records is out of range and its 16 bytes arrive as ZEROS: the
// padding around a map, the rows between two stacked images and the channel rows beyond Cout need no zero page and no select --
// their voff is DEAD.  (soff is not part of the range check: a uniform offset that may run off the end -- a slab beyond the last --
// goes with a descriptor of zero records instead.)  No vector instruction per DMA: offsets are 32-bit registers computed once.
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
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);      // stride 0: a raw buffer, offsets in bytes
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Diagnostic build only (tools/glds_stamps.sh, -DRDFC_GLDS_STAMPS): s_memtime stamps of one workgroup's waves around the parts of
// a phase, slab 2 of the K loop; no stamp executes in the shipped library.
#ifdef RDFC_GLDS_STAMPS
__device__ unsigned long long g_stamps[8][9][6];
__device__ unsigned long long g_tile_stamps[512][10];        // sampled workgroups (wave 0): entry | operands of phase 0 landed | K loop done | stores issued | tile id + 1 | s_memrealtime (100 MHz) at entry | at the end | hw id
#define RDFC_TSTAMP(K)                                                                        \
    if (tstamp_on) {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tbuf[K])::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#define RDFC_STAMP(TAP, K)                                                                    \
    if (stamp_on && cc == 2) {                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sbuf[TAP][K])::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    }
#else
#define RDFC_STAMP(TAP, K)
#define RDFC_TSTAMP(K)
#endif

// virtual row of the stacked batch -> (image, row inside it); image = N: the row belongs to no image (it reads zeros / stores nothing)
__device__ __forceinline__ void vrow(const GParams &p, int v, int &img, int &row) {      // branch-free per lane (p.gstride is uniform)
    const bool neg = v < 0;
    v = neg ? 0 : v;
    int g = 0;
    if (p.gstride) qdiv(v, p.gstride, p.r_gstride, g, v);
    int l;
    qdiv(v, p.vstride, p.r_vstride, l, row);
    img = (neg || (p.gstride && l >= p.npg)) ? p.N : g * p.npg + l;
    row = neg ? 0 : row;
}

// Variants measured and removed in round 3 (numbers in DESIGN.md section 7 and profiles/r01_conv_glds_stamps.txt): producer / consumer
// specialisation (4 consumer + 4 | 2 producer waves, fragment reads pipelined one phase ahead), 64 co x 128 px wave tiles, 8-wave
// 128 co x 256 px and 2-wave workgroups, a fifth ring slot with fragment reads one phase ahead, and an 8-wave ping-pong schedule with
// two barriers per tap (profiles/r03_pingpong_experiment.txt): none beat this one on the step's layer sizes.
//
// Tall wave tiles (JR > 4, round 5).  A wave owns 64 co x (JR rows of 16 px), the workgroup 128 co x (2 JR x 16) px: the weight tile
// of a tap (86 % of the staged bytes of the JR = 4 tile) now serves twice to two-and-a-half times the pixels, and the taps are walked
// COLUMN by column (kx outer, ky inner): the B fragment of patch row r + ky at column shift kx is the same 16 x 32 block for every
// (r, ky) with the same r + ky, so a wave reads the JR + 2 patch rows of a kernel column ONCE into registers and the three taps of that
// column take their operands from there -- JR + 2 instead of 3 JR pixel-fragment reads per column.  Per 32-channel slab and wave:
// 9 x 4 weight-fragment reads + 3 (JR + 2) pixel-fragment reads for 36 JR MFMAs (JR = 8: 66 reads per 288 MFMAs = 0.23 per MFMA;
// the JR = 4 tile: 72 per 144 = 0.5), 8 KB + 2.3 KB of LDS-DMA per 4 x 32 MFMAs (83 B per MFMA against 148).  128 accumulator
// registers + 40 resident B + 16 A: two workgroups per CU (the launch picks JR per shape so that the grid fills whole rounds of the
// 512 slots: plan() below).
template <int TCO, int DT = RDFC_BF16, int JR = 4>     // DT: RDFC_BF16 | RDFC_F16 (2-byte elements moved as opaque bits); JR: pixel rows (of 16) per wave
__global__ __launch_bounds__(256, 2) void conv3x3_glds_kernel(GParams p, const bf16_t *__restrict__ x, const bf16_t *__restrict__ w,
                                                              const float *__restrict__ bias, void *__restrict__ yout,
                                                              const bf16_t *__restrict__ zero, float *__restrict__ stats) {
    constexpr int NW = 4;                                      // waves per workgroup; a wave owns 64 co x (JR x 16) px
    constexpr int NT = NW * 64;
    constexpr int WCO = TCO / 64, WPX = NW / WCO;              // waves along co / along pixel rows
    constexpr int TR = JR * WPX, PH = TR + 2, NPATCH = PH * PW; // tile rows, patch rows, patch pixels (180 / 324)
    constexpr int NPIECE = NPATCH * 4;                         // 16-byte pieces per slab
    constexpr int PATCH_BYTES = (NPIECE + 63) / 64 * 1024;     // whole DMA instructions (1 KB each): 12 KB / 21 KB
    constexpr int NR = (NPIECE + NT - 1) / NT;                 // patch DMA rounds per slab (3 / 6)
    constexpr int NWP = TCO * 4 / NT;                          // weight DMAs per thread and tap (2 / 1)
    constexpr int WSLOT = TCO * BK * 2;                        // bytes per weight slot
    constexpr int RING = ring_of<TCO, JR>();
    constexpr int WLOOK = RING - 1;                            // phases between a weight block's DMA and its use
    constexpr int OFF_W = 2 * PATCH_BYTES, OFF_JUNK = OFF_W + RING * WSLOT;
    static_assert(NR <= 9 - WLOOK, "the patch rounds of a slab must be older than the weights of its first phase");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    constexpr int VPW = 1;                                       // DMA shares per issuing wave (every wave issues its own)
    auto dwave = [&](int) { return wave; };
    const int co0 = blockIdx.y * TCO;
    // Workgroups are dealt to the 8 XCDs round-robin; renumber so that each XCD (own L2) gets a contiguous run of
    // tiles: neighbouring tiles share halo rows and then hit in the same L2.
    int t = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = t & 7, k = t >> 3;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int t_all = t;                                        // tile index after the renumbering (unique)
    int tx;
    qdiv(t, p.tiles_x, p.r_tiles_x, t, tx);
    // stacked batch (p.vstride > 0): ty counts tile rows of the tall virtual map and y0 is a virtual row; else a tile belongs to image n
    int ty = t, n = 0;
    if (!p.vstride) qdiv(t, p.tiles_y, p.r_tiles_y, n, ty);
    const int y0 = ty * TR, x0 = tx * TC;
    const int ncc = p.Cin / BK;

    // ---- per-thread DMA sources (computed once) -------------------------------------------------------------
    // 32-bit BYTE offsets from x / w through buffer descriptors (bdma16): a piece outside the image, in the rows between two stacked
    // images or beyond Cout is DEAD and arrives as zeros.
    // The LDS image of a patch position (py, px) is 64 bytes at (py * PW + px) * 64 with its four 16-byte chunks XOR-swizzled by bit 2
    // of px (LDS-DMA destinations are lane-linear, so the swizzle is applied to the SOURCE chunk): a fragment read of 16 neighbouring
    // pixels x 4 chunks is conflict-free for every start column, and its address splits into a per-lane part that depends on the
    // kernel column only (three registers) and a compile-time row offset.
    unsigned poff[NR];
    unsigned pdst[NR];                                           // wave-uniform LDS offset inside a patch buffer, or junk
    unsigned woff[NWP];                                          // weight rows tid / 4 + 64 k
    const u32x4 xsrc = make_rsrc(x, (unsigned)p.x_bytes), wsrc = make_rsrc(w, (unsigned)p.w_bytes);
    {
        const int dtid = wave * 64 + lane;
        const int row = dtid >> 2, slot = dtid & 3;              // (rows 64 k further on share the swizzle: bit 1 of row / 2 is the same)
#pragma unroll
        for (int k = 0; k < NWP; ++k)
            woff[k] = co0 + row + 64 * k < p.Cout ? 2u * (unsigned)((co0 + row + 64 * k) * 9 * p.Cin + ((slot ^ ((row >> 1) & 2)) << 3)) : DEAD;
    }
    // Phase s of a slab multiplies the patch at shift (ky, kx) = (s % 3, s / 3) -- kernel column by kernel column, in EVERY variant, so
    // that each output sums its 9 Cin / 32 partial products in one order whatever tile height the launch picked (same bits) -- with the
    // weights of tap 3 ky + kx, or of the mirrored tap 8 - (3 ky + kx) in the transposed product.
    auto issue_w = [&](int it) {                                 // weights of global phase `it` -> ring slot it % RING
        const int cc = it / 9, s_ = it - cc * 9;
        const int lin = (s_ % 3) * 3 + s_ / 3, tap = p.transposed ? 8 - lin : lin;
        u32x4 rs = wsrc;
        if (cc >= ncc) rs[2] = 0;                                // (a phase beyond the last slab: zeros)
        const unsigned soff = 2u * (unsigned)(tap * p.Cin + cc * BK);
#pragma unroll
        for (int k = 0; k < NWP; ++k) bdma16(woff[k], rs, soff, lds0 + OFF_W + (it % RING) * WSLOT + (k * NT + wave * 64) * 16);
    };
    issue_w(0);                                                  // (in flight while the patch addresses are worked out)
    {
        const int dtid = wave * 64 + lane;
        // The row part of a piece's address -- virtual row -> (image, row) of the stacked batch (two divisions), the reflection, the
        // bounds test -- is computed ONCE per patch row, by PH threads in parallel, and handed around through LDS (the second patch
        // buffer: nothing lands there before phase 0 has passed its barrier); a piece then adds its column.  With every thread doing
        // the row arithmetic for each of its NR pieces, a tile's first DMA was issued ~1 000 instructions into the kernel and a 2-slab
        // tile (64-channel layers at full resolution) spent a fifth of its life there (tools/glds_tile_stamps.py, round 5).
        unsigned *rowtab = reinterpret_cast<unsigned *>(smem + PATCH_BYTES);
        if (tid < PH) {
            int iy = y0 + p.oy0 + tid, ni = n;
            if (p.vstride) vrow(p, iy, ni, iy);                 // virtual row -> (image, row); the rows between two images read zeros (iy >= Hin below)
            if (p.reflect) iy = iy < 0 ? -iy : (iy >= p.Hin ? 2 * p.Hin - 2 - iy : iy);       // LDS-DMA sources are per lane: the halo of a reflection-padded map is just another address
            const bool live = ni < p.N && (unsigned)iy < (unsigned)p.Hin;
            rowtab[tid] = live ? 2u * (unsigned)((ni * p.Hin + iy) * p.Win * p.ld_in) : DEAD;      // byte offset of the row (< 2^31: glds_shape_ok)
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int e = r * NT + dtid, q = e >> 2, slot = e & 3;
            {                              // (no per-lane branches: selects; the offset of a piece that is not live is never used)
                const int py = q / PW, px = q - py * PW;
                int ix = x0 + p.ox0 + px;
                if (p.reflect) ix = ix < 0 ? -ix : (ix >= p.Win ? 2 * p.Win - 2 - ix : ix);
                const unsigned ro = rowtab[py < PH ? py : PH - 1];
                const bool live = q < NPATCH && ro != DEAD && (unsigned)ix < (unsigned)p.Win;     // (what is still outside belongs to output pixels beyond the map: zeros)
                poff[r] = live ? ro + 2u * (unsigned)(ix * p.ld_in + ((slot ^ ((px >> 1) & 2)) << 3)) : DEAD;
            }
            const int wbase = (r * NT + wave * 64) * 16;         // this wave's 1 KB of the patch image
            pdst[r] = wbase + 1024 <= PATCH_BYTES ? (unsigned)wbase : 0xffffffffu;
        }
    }
    auto issue_p = [&](int r, int cc) {                          // round r of slab cc -> patch buffer cc & 1
        const unsigned dst = pdst[r] == 0xffffffffu ? lds0 + OFF_JUNK : lds0 + (cc & 1) * PATCH_BYTES + pdst[r];
        u32x4 rs = xsrc;
        if (cc >= ncc) rs[2] = 0;
        bdma16(poff[r], rs, 2u * (unsigned)(cc * BK), dst);
    };

    // ---- fragment coordinates ---------------------------------------------------------------------------------
    const int wr = wave / WPX, wc = wave % WPX;
    constexpr int NB = JR + 2;                                   // B fragments a wave holds: the patch rows of a kernel column
    const int kg = lane >> 4;
    unsigned bcol[3];                                            // per-lane byte offset of the wave's first patch row at column shift kx
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int px = (lane & 15) + kx;
        bcol[kx] = (unsigned)((JR * wc * PW + px) * 64 + ((kg ^ ((px >> 1) & 2)) << 4));
    }
    const int arow = wr * 64 + (lane & 15);

    f32x4 acc[4][JR];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < JR; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // byte address of (row, 16-byte chunk kg) in a swizzled 64-byte-row image: (row*64 + kg*16) ^ ((row & 4) << 3)
    const unsigned a_off = (unsigned)((arow << 6) | (kg << 4)) ^ (unsigned)((arow & 4) << 3);     // + i*1024: rows arow + 16 i share bit 2
    auto read_a = [&](int it, bf16x8 (&a)[4]) {
        const unsigned char *Wt = smem + OFF_W + (it % RING) * WSLOT + a_off;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8 *>(Wt + i * 1024);
    };
    auto read_b = [&](int cc, int kx, int row) -> bf16x8 {      // patch row `row` of the wave (0 ... JR + 1) at column shift kx
        return *reinterpret_cast<const bf16x8 *>(smem + (cc & 1) * PATCH_BYTES + bcol[kx] + row * (PW * 64));
    };
    bf16x8 bres[NB];                                             // the B fragments of the current kernel column, patch row j in bres[j]
#ifdef RDFC_GLDS_STAMPS
    unsigned long long sbuf[9][6];
    const bool stamp_on = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0;
    const int wave_all = wave;
    unsigned long long tbuf[6];
    const int tstride = (int)((gridDim.x * gridDim.y + 511) / 512), tlin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const bool tstamp_on = wave == 0 && tlin % tstride == 0;
    RDFC_TSTAMP(0)
    unsigned long long treal0 = 0, treal1 = 0;
    if (tstamp_on) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(treal0)::"memory");
#endif

    // ---- the K loop: every fragment read is issued a phase ahead, between the MFMAs of the phase before ----------------------------------
    // With two waves on a SIMD (the tall tiles), a phase that runs wait -> barrier -> DMA issue -> fragment reads -> MFMAs one after the
    // other leaves the matrix pipe idle whenever both are in the front part (stamps of round 5: 56 % busy).  A phase's MFMAs take all
    // their operands from registers that were filled during the phase before, IN PLACE (no second register set):
    //   * the MFMAs run channel block by channel block (i outer, rows inner): weight fragment i is dead after its JR MFMAs, and the
    //     fragment i of the NEXT phase is read into its registers (that ring slot landed a phase earlier than it used to: the waits
    //     below cover the next phase's weights);
    //   * the pixel fragments roll: a column's second and third tap need one more patch row each, read a phase ahead into the registers
    //     of a row that is dead; during a column's last tap the next column's rows (at the slab's last tap: the next slab's patch, which
    //     therefore must have landed one phase before the slab ends) replace the rows as their last MFMA is issued.
    bf16x8 aa[4];
#pragma unroll
    for (int r = 0; r < NR; ++r) issue_p(r, 0);
#pragma unroll
    for (int h = 1; h <= WLOOK; ++h) issue_w(h);                 // (phase 0's weights went out first of all)
    wait_vm<WLOOK * NWP>();                                      // the weights of phase 0 and patch 0
    __builtin_amdgcn_s_barrier();
    RDFC_TSTAMP(1)
    read_a(0, aa);
#pragma unroll
    for (int j = 0; j < JR; ++j) bres[j] = read_b(0, 0, j);
#define RDFC_GLDS_SCHED __builtin_amdgcn_sched_barrier(0);      // (pins the order written here: left alone, hipcc hoists the reads and the registers run out)
#define RDFC_GLDS_PHASE(TAP)                                                                                   \
    {                                                                                                          \
        constexpr int ALLOW = glds_allow(TAP, WLOOK, NWP, NR);    /* the weights of phase it + 1 were issued WLOOK phases ago */ \
        constexpr int KY = TAP % 3, KX = TAP / 3;                                                              \
        constexpr int NKX = (KX + 1) % 3;                         /* KY == 2: the column the prefetched rows belong to */ \
        const int ncc_ = (TAP == 8) ? cc + 1 : cc;                                                             \
        RDFC_STAMP(TAP, 0)                                                                                     \
        wait_vm_and_reads<ALLOW>();                                                                            \
        RDFC_STAMP(TAP, 1)                                                                                     \
        __builtin_amdgcn_s_barrier();                                                                          \
        RDFC_STAMP(TAP, 2)                                                                                     \
        issue_w(it0 + TAP + 1 + WLOOK);                                                                        \
        if (TAP < NR) issue_p(TAP < NR ? TAP : 0, cc + 1);                                                     \
        RDFC_STAMP(TAP, 3)                                                                                     \
        if (KY == 2) {                                                                                         \
            bres[0] = read_b(ncc_, NKX, 0);                                                                    \
            bres[1] = read_b(ncc_, NKX, 1);                                                                    \
        } else                                                                                                 \
            bres[JR + KY] = read_b(cc, KX, JR + KY);                                                           \
        RDFC_STAMP(TAP, 4)                                                                                     \
        RDFC_GLDS_SCHED                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
            _Pragma("unroll") for (int j = 0; j < JR; ++j) {                                                   \
                acc[i][j] = mma16<DT>(aa[i], bres[j + KY], acc[i][j]);                                         \
                if (KY == 2 && i == 3 && j + 2 < JR) {                                                         \
                    RDFC_GLDS_SCHED                                                                            \
                    bres[j + 2] = read_b(ncc_, NKX, j + 2);                                                    \
                    RDFC_GLDS_SCHED                                                                            \
                }                                                                                              \
            }                                                                                                  \
            RDFC_GLDS_SCHED                                                                                    \
            aa[i] = *reinterpret_cast<const bf16x8 *>(smem + OFF_W + ((it0 + TAP + 1) % RING) * WSLOT + a_off + i * 1024); \
            RDFC_GLDS_SCHED                                                                                    \
        }                                                                                                      \
        RDFC_STAMP(TAP, 5)                                                                                     \
    }
    for (int cc = 0; cc < ncc; ++cc) {
        const int it0 = cc * 9;
        RDFC_GLDS_PHASE(0) RDFC_GLDS_PHASE(1) RDFC_GLDS_PHASE(2) RDFC_GLDS_PHASE(3) RDFC_GLDS_PHASE(4)
        RDFC_GLDS_PHASE(5) RDFC_GLDS_PHASE(6) RDFC_GLDS_PHASE(7) RDFC_GLDS_PHASE(8)
    }
#undef RDFC_GLDS_PHASE
    wait_vm<0>();
    RDFC_TSTAMP(2)
#ifdef RDFC_GLDS_STAMPS
    if (stamp_on && lane == 0 && ncc > 2)
        for (int t_ = 0; t_ < 9; ++t_)
            for (int k_ = 0; k_ < 6; ++k_) g_stamps[wave_all & 7][t_][k_] = sbuf[t_][k_];
#endif

    // ---- epilogue: bias + activation + store, 4 consecutive output channels per lane -------------------------------
    // (the activation is selected once, outside the 64-value loop: a per-element switch costs more than the stores)
    const int q4 = (lane >> 4) * 4;
    const bool vec_ok = (p.ld_out & 3) == 0;
    int on[JR], oyr[JR];                                        // image and row of this wave's JR output rows (-1: none)
#pragma unroll
    for (int j = 0; j < JR; ++j) {                              // (wave-uniform: the rows of an image follow each other, the division runs where one ends)
        int v = y0 + JR * wc + j, ni = n;
        if (p.vstride) {
            if (j > 0 && on[j - 1] < p.N && on[0] >= 0 && oyr[j - 1] >= 0 && oyr[j - 1] + 1 < p.H) {
                ni = on[j - 1];
                v = oyr[j - 1] + 1;
            } else
                vrow(p, v, ni, v);
        }
        on[j] = ni;
        oyr[j] = (ni < p.N && v < p.H) ? v : -1;
    }
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
    // ---- fused normalisation statistics: per-tile partial {sum, sum of squares} of the STORED values, one row per tile in
    //      stats[g][tile][2][Cout] (the fixed-order f64 finalize of norm.hip reads them; no atomics) --------------------
    if (stats != nullptr) {
        float s1[4][4], s2[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[i][r] = s2[i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < JR; ++j) {
            const int oy = oyr[j], ox = x0 + (lane & 15);
            if (oy >= 0 && ox < p.W) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[i][j][r] + bv[i][r];
                        if (!p.out_f32) v = f32of<DT>(cvt16<DT>(v));
                        s1[i][r] += v;
                        s2[i][r] += v * v;
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    s1[i][r] += __shfl_xor(s1[i][r], off, 64);
                    s2[i][r] += __shfl_xor(s2[i][r], off, 64);
                }
        __syncthreads();                                         // every wave is done with the LDS tiles
        float *red = reinterpret_cast<float *>(smem);            // [wave][2][64]
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[(wave * 2 + 0) * 64 + 16 * i + q4 + r] = s1[i][r];
                    red[(wave * 2 + 1) * 64 + 16 * i + q4 + r] = s2[i][r];
                }
        }
        __syncthreads();
        if (tid < TCO) {
            const int wrc = tid >> 6, cc = tid & 63, co = co0 + tid;
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < WPX; ++k) {
                a1 += red[((wrc * WPX + k) * 2 + 0) * 64 + cc];
                a2 += red[((wrc * WPX + k) * 2 + 1) * 64 + cc];
            }
            if (co < p.Cout) {
                // (statistics launches tile the stacked batch group by group: tiles_y / G tile rows per group, rows [g][chunk] in tile order)
                const int nchunk = p.nchunk;
                int g, chunk;
                qdiv(t_all, nchunk, p.r_nchunk, g, chunk);
                float *dst = stats + ((size_t)(g * nchunk + chunk) * 2) * p.Cout + co;
                dst[0] = a1;
                dst[p.Cout] = a2;
            }
        }
    }
    auto store_all = [&](auto actf) {
#pragma unroll
        for (int j = 0; j < JR; ++j) {
            const int oy = oyr[j], ox = x0 + (lane & 15);
            if (oy < 0 || ox >= p.W) continue;
            const size_t po = ((size_t)on[j] * p.H + oy) * p.W + ox;
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
    };
    // ---- whole-row stores (common.h: rs_put / rs_flush); an f32 result keeps the direct path ---------------------------------------------
    if (p.row_store == 2) {
        // the launches with an addend (rdfc_conv2d_dgrad_add): f32 staging, 32 pixels (two tile rows) per pass, the addend fetched in whole
        // 128-byte rows a pass ahead and added by the flush (common.h: rsf_put / rsf_fetch / rsf_flush) -- same arithmetic, same order,
        // same bits as the path below
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        const int cow = co0 + wr * 64;
        const bf16_t *addw = reinterpret_cast<const bf16_t *>(p.addend) + cow;
        auto ppix = [&](int j0, int px) -> long {                  // pixel index of pixel px (0 .. 31) of the pass that starts at tile row j0
            const int j = j0 + (px >> 4), c = px & 15;             // (px >> 4: compile-time after unrolling)
            if (oyr[j] < 0 || x0 + c >= p.W) return -1;
            return ((long)on[j] * p.H + oyr[j]) * p.W + x0 + c;
        };
        RsfAdd cur, nxt;
        rsf_fetch(cur, lane, addw, p.ld_add, p.Cout - cow, [&](int px) { return ppix(0, px); });
        __syncthreads();                                           // every wave is done with the operand tiles
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int j0 = 0; j0 < JR; j0 += 2) {
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
            if (j0 + 2 < JR) rsf_fetch(nxt, lane, addw, p.ld_add, p.Cout - cow, [&](int px) { return ppix(j0 + 2 < JR ? j0 + 2 : j0, px); });
            rsf_flush<DT>(Ot, lane, reinterpret_cast<bf16_t *>(yout) + cow, p.ld_out, cur, nullptr, p.Cout - cow, [&](int px) { return ppix(j0, px); });
            cur = nxt;
        }
        return;
    }
    if (p.row_store) {
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        __syncthreads();                                           // every wave is done with the operand tiles
        RDFC_TSTAMP(4)
        unsigned char *Ot = smem + wave * RS_TILE;
#pragma unroll
        for (int j0 = 0; j0 < JR; j0 += 4) {                       // 64 pixels (four tile rows) at a time through the wave's 9 KB
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj < JR ? j0 + jj : JR - 1;     // (compile-time; a last group of fewer than four rows leaves its other rows unwritten and unread)
                if (j0 + jj >= JR) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u = acc[i][j][r] + bv[i][r];
                        v[r] = u > 0.f ? u : u * ns;
                    }
                    const int co = co0 + wr * 64 + 16 * i + q4, ox = x0 + (lane & 15);
                    if (p.addend != nullptr && oyr[j] >= 0 && ox < p.W && co < p.Cout) {        // (the addend, in f32, from its 8-byte pieces)
                        const size_t po = ((size_t)on[j] * p.H + oyr[j]) * p.W + ox;
                        const bf16x4 t = *reinterpret_cast<const bf16x4 *>(reinterpret_cast<const bf16_t *>(p.addend) + po * p.ld_add + co);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += f32of<DT>(t[r]);
                    }
                    rs_put<DT>(Ot, 16 * jj + (lane & 15), 16 * i + q4, v);
                }
            }
            const int cow = co0 + wr * 64;
            if (j0 == 0) { RDFC_TSTAMP(5) }
            long rowbase[4];                                       // element offset of the row group's rows at column x0 (wave-uniform), -1: no such row
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj < JR ? j0 + jj : JR - 1;
                rowbase[jj] = (j0 + jj < JR && oyr[j] >= 0) ? (((long)on[j] * p.H + oyr[j]) * p.W + x0) * p.ld_out : -1;
            }
            rs_flush(Ot, lane, reinterpret_cast<bf16_t *>(yout) + cow, p.Cout - cow, [&](int px) -> long {
                const int jj = px >> 4, c = px & 15;               // (jj: compile-time after unrolling)
                if (rowbase[jj] < 0 || x0 + c >= p.W) return -1;
                return rowbase[jj] + c * p.ld_out;
            });
        }
#ifdef RDFC_GLDS_STAMPS
        RDFC_TSTAMP(3)
        if (tstamp_on && lane == 0) {
            for (int k_ = 0; k_ < 4; ++k_) g_tile_stamps[tlin / tstride][k_] = tbuf[k_];
            g_tile_stamps[tlin / tstride][4] = (unsigned long long)tlin + 1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(treal1)::"memory");
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g_tile_stamps[tlin / tstride][5] = treal0;
            g_tile_stamps[tlin / tstride][6] = treal1;
            g_tile_stamps[tlin / tstride][7] = hwid;
            g_tile_stamps[tlin / tstride][8] = tbuf[4];
            g_tile_stamps[tlin / tstride][9] = tbuf[5];
        }
#endif
        return;
    }
    if (p.act == RDFC_ACT_TANH) store_all([](float u) { return tanhf(u); });
    else if (p.act == RDFC_ACT_SIGMOID) store_all([](float u) { return 1.f / (1.f + expf(-u)); });
    else {
        const float ns = p.act == RDFC_ACT_NONE ? 1.f : (p.act == RDFC_ACT_RELU ? 0.f : p.slope);
        store_all([ns](float u) { return u > 0.f ? u : u * ns; });
    }
}

// 64-byte page of zeros per device (source of every out-of-range DMA piece)
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

template <int TCO, int DT = RDFC_BF16, int JR = 4>
int launch(const GParams &p, const void *x, const void *w, const float *bias, void *y, const bf16_t *zero, float *stats, hipStream_t st) {
    constexpr int NT = 256, TR = JR * 4 / (TCO / 64);
    constexpr int NPIECE = (TR + 2) * PW * 4, PATCH = (NPIECE + 63) / 64 * 1024, NR = (NPIECE + NT - 1) / NT;
    constexpr int OPER = 2 * PATCH + ring_of<TCO, JR>() * TCO * BK * 2 + (NR * NT * 16 > PATCH ? 1024 : 0);      // + scratch for the DMA rounds' overhang
    constexpr int SMEM = OPER > (NT / 64) * RS_TILE ? OPER : (NT / 64) * RS_TILE;                                  // the whole-row epilogue reuses it: 9 KB per wave
    static_assert(SMEM >= (NT / 64) * RS_TILE, "rs_put / rs_flush write wave * RS_TILE ... + RS_TILE");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void *)conv3x3_glds_kernel<TCO, DT, JR>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return 0;
        attr_set = true;
    }
    dim3 grid(p.tiles_x * p.tiles_y * (p.vstride ? 1 : p.N), cdiv(p.Cout, TCO));
    rdfc_note_kernel("conv3x3_glds_kernel<%d, %d, %d>", TCO, DT, JR);            // the symbol as rocprofv3 prints it (every template argument, a default one too: round 6 -- the JR = 4 launches used to be noted without it and missed their counters in profiles/*_pmc_traffic.json)
    hipLaunchKernelGGL((conv3x3_glds_kernel<TCO, DT, JR>), grid, dim3(NT), SMEM, st, p, (const bf16_t *)x, (const bf16_t *)w, bias, y, zero, stats);
    return 1;
}

}  // namespace

static bool glds_shape_ok(const rdfc_conv_desc *d, int transposed) {
    static const int mode = getenv("RDFC_GLDS") ? atoi(getenv("RDFC_GLDS")) : 1;
    if (!mode || !rdfc_is16(d->dtype)) return false;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    if (!(d->KH == 3 && d->KW == 3 && d->sh == 1 && d->sw == 1 && d->ph == d->pw && d->ph <= 1 && d->dh == 1 && d->dw == 1)) return false;
    if (Cin % BK != 0 || d->Ho != d->Hi + 2 * d->ph - 2 || d->Wo != d->Wi + 2 * d->pw - 2) return false;
    const int ld_in = transposed ? d->ldy : d->ldx;
    const size_t in_bytes = 2 * (size_t)d->N * (transposed ? d->Ho : d->Hi) * (transposed ? d->Wo : d->Wi) * ld_in;
    if (in_bytes >= 0x7ffffff0u || 2 * (size_t)Cout * 9 * Cin >= 0x7ffffff0u) return false;     // the loaders address both through 32-bit byte offsets
    {   // qdiv's range (2^22 tiles / virtual rows), bounded from the shape alone -- whatever tile height and stacking glds_plan then picks -- so
        // that this test, the statistics-row query and the launch agree (round 5 tested it at the launch only: rdfc_conv3x3_glds_stats_nchunk
        // could promise rows for a launch that was then refused).  16 x 480 x 640 is 2 % of it.
        const long H = transposed ? d->Hi : d->Ho, W = transposed ? d->Wi : d->Wo;
        if (2 * (long)d->N * (H + 4) + 64 >= (1L << 22) || (long)cdiv(W, TC) * d->N * (H / 8 + 3) >= (1L << 22)) return false;
    }
    return ld_in % 8 == 0 && Cout >= 64;
}

// Number of per-tile statistics rows the fused epilogue writes per group for this shape (G = 1 or N), 0 if the shape does
// not run on this kernel (the caller then uses rdfc_norm_stats).
// Row tiles of 8 / 16 rows waste the last tile row of every image (57 rows: 8 tiles for 7.1, 29 rows: 4 for 3.6).  With N > 1 the
// batch is tiled as ONE tall map instead: image n at virtual rows [n * S, n * S + H), the S - H rows between two images read zeros
// and store nothing (they ARE the zero padding both neighbours need, which is what fixes S), and only the last tile row of the whole
// batch is partial: 128 -> 116 row tiles for 16 x 57, 64 -> 60 for 16 x 29.  Not with the reflection fold (a mirrored halo row differs
// per image) and not with fused statistics (their per-tile rows are consumed per image / per stacked batch, and a tile may straddle
// two images).  -> S, 0 if the launch keeps per-image tiles.
static int stack_rows(int N, int H, int Hin, int oy0, int TR, bool allowed, int &tiles_y) {
    static const int on = getenv("RDFC_GLDS_STACK") ? atoi(getenv("RDFC_GLDS_STACK")) : 1;
    tiles_y = cdiv(H, TR);
    if (!on || !allowed || N < 2) return 0;
    int S = H > Hin - oy0 ? H : Hin - oy0;                      // a row above an image's first must be a zero row of its predecessor
    if (H + 1 + oy0 >= Hin && S < H + 2 + oy0) S = H + 2 + oy0; // rows below an image's last must not be its successor's first
    const int stacked = cdiv((N - 1) * S + H, TR);
    if (stacked >= N * tiles_y) return 0;
    tiles_y = stacked;
    return S;
}
// Launches with fused statistics (round 4: stacked too).  The statistics are per group of npg = N / G consecutive images (G = 1 batch
// norm, N instance norm, in between BatchNorm over stacked batches) and every tile writes ONE row of one group, so a tile must not
// straddle two groups: the images of a group are stacked at stride S as above and every group starts on a tile boundary (gstride, a
// multiple of TR; its unused tail rows belong to no image).  -> S; tiles_y covers all groups.
static int stack_groups(int N, int G, int H, int Hin, int oy0, int TR, int &tiles_y, int &gstride) {
    const int npg = N / G;
    int S = H > Hin - oy0 ? H : Hin - oy0;
    if (H + 1 + oy0 >= Hin && S < H + 2 + oy0) S = H + 2 + oy0;
    static const int on = getenv("RDFC_GLDS_STACK") ? atoi(getenv("RDFC_GLDS_STACK")) : 1;
    if (!on || npg < 2) S = cdiv(S, TR) * TR;                   // per-image tiles: every image starts on a tile boundary
    gstride = cdiv(npg * S, TR) * TR;                           // (S > H: the last row of a group's slot is a zero row, like the rows between two images)
    if (npg == 1) S = gstride;
    tiles_y = G * (gstride / TR);
    return S;
}

// ---- which tile for which launch (round 5) ------------------------------------------------------------------------------------------
// 128-co workgroups come as 128 co x (8 x 16) px at three per CU (JR = 4: 768 slots) or as tall tiles 128 co x (2 JR x 16) px at two per
// CU (JR = 6 / 8 / 10: 512 slots; less staged and read per MFMA, see the kernel).  A grid is dealt to the slots in rounds and its last,
// partly filled round costs nearly a whole one, so the choice is made per launch from the grid each tile height gives: estimated time =
// rounds x rows per workgroup x workgroups per CU / the tile's steady-state rate, a partly filled round counted by how many workgroups
// its CUs hold.  Calibrated on tools/conv_tall_bench.py (profiles/r05_experiments.txt): the tall tiles run 1.13 x the JR = 4 rate when
// the K loop is long enough (Cin >= 128; with two slabs the prologue and the stores are what a workgroup spends its time on), e.g.
// 256 -> 256 @ 16 x 57 x 76: 1160 workgroups of JR = 4 are 1.5 rounds of 768 slots, 470 of JR = 10 one round of 512 (82 -> 73 us);
// 512 -> 512 @ 16 x 29 x 38: JR = 6 gives 480 (91 -> 81 us); 128 -> 128 @ 16 x 114 x 152 stays at JR = 4 (2300 = 2.99 rounds).
// RDFC_GLDS_JR=4|6|8|10 / rdfc_tune_glds_rows force one height.
struct GPlan {
    int jr, tiles_y, vstride, gstride;
};
// (a height the kernel is not instantiated for -- anything but 4, 6, 8, 10 -- means "no preference": glds_plan would otherwise skip every candidate)
static int glds_jr_ok(int jr) { return (jr == 4 || jr == 6 || jr == 8 || jr == 10) ? jr : 0; }
static int g_glds_jr = glds_jr_ok(getenv("RDFC_GLDS_JR") ? atoi(getenv("RDFC_GLDS_JR")) : 0);
extern "C" void rdfc_tune_glds_rows(int jr) { g_glds_jr = glds_jr_ok(jr); }    // measurement aid (tools/conv_tall_bench.py): A/B of the tile heights in one process; not thread-safe
static GPlan glds_plan(int N, int H, int W, int Hin, int oy0, int Cin, int Cout, int G, bool reflect) {
    const int forced = g_glds_jr;
    auto tiles = [&](int TR, GPlan &g) {
        g.gstride = 0;
        if (G > 0) g.vstride = stack_groups(N, G, H, Hin, oy0, TR, g.tiles_y, g.gstride);
        else g.vstride = stack_rows(N, H, Hin, oy0, TR, !reflect, g.tiles_y);
        return (long)cdiv(W, TC) * g.tiles_y * (g.vstride ? 1 : N) * cdiv(Cout, 128);
    };
    GPlan best{4, 0, 0, 0};
    if (Cout <= 64) {                                            // 64 co x (16 x 16) px, four waves side by side in rows
        tiles(16, best);
        return best;
    }
    double best_cost = 0;
    const int cand[4] = {4, 6, 8, 10};
    for (int c = 0; c < 4; ++c) {
        const int jr = cand[c];
        if (forced ? jr != forced : false) continue;
        GPlan g{jr, 0, 0, 0};
        const long wgs = tiles(2 * jr, g);
        const int per_cu = jr == 4 ? 3 : 2, slots = 256 * per_cu;
        const long full = wgs / slots, rest = wgs - full * slots;
        // a partly filled round: a CU with fewer resident workgroups finishes each of them sooner
        double part = 0;
        if (rest > 0) part = per_cu == 3 ? (rest <= 256 ? 0.45 : (rest <= 512 ? 0.75 : 1.0)) : (rest <= 256 ? 0.7 : 1.0);
        const double rate = (jr == 4 || Cin < 128) ? 1.0 : 1.13;
        const double cost = (full + part) * jr * per_cu / rate;
        if (best.tiles_y == 0 || cost < best_cost) {
            best = g;
            best_cost = cost;
        }
    }
    return best;
}

int rdfc_conv3x3_glds_stats_nchunk(const rdfc_conv_desc *d, int G) {
    if (!glds_shape_ok(d, 0) || G < 1 || d->N % G != 0) return 0;
    const GPlan g = glds_plan(d->N, d->Ho, d->Wo, d->Hi, -d->ph, d->Ci, d->Co, G, false);
    return (g.tiles_y / G) * cdiv(d->Wo, TC);                   // rows per group
}

// Called by conv_igemm.hip's entry points before the register-staged 3x3 kernels. Returns 1 if it took the launch.
// stats != nullptr (forward only): also write the per-tile normalisation statistics, G groups.
int rdfc_conv3x3_glds_covers(const rdfc_conv_desc *d) { return glds_shape_ok(d, 0) ? 1 : 0; }

int rdfc_try_conv3x3_glds(const rdfc_conv_desc *d, int transposed, const void *in, const void *w, const float *bias, void *out,
                          hipStream_t st, float *stats = nullptr, int G = 0, const void *addend = nullptr, int ld_add = 0, int reflect = 0) {
    if (!glds_shape_ok(d, transposed)) return 0;
    if (reflect && (transposed || d->ph < 1 || d->ph >= d->Hi || d->pw >= d->Wi)) return 0;
    const int Cin = transposed ? d->Co : d->Ci, Cout = transposed ? d->Ci : d->Co;
    const int ld_in = transposed ? d->ldy : d->ldx;
    const bf16_t *zero = zero_page();
    if (zero == nullptr) return 0;
    const int pad = d->ph;
    GParams p{d->N, transposed ? d->Ho : d->Hi, transposed ? d->Wo : d->Wi, transposed ? d->Hi : d->Ho, transposed ? d->Wi : d->Wo,
              Cin, Cout, ld_in, transposed ? d->ldx : d->ldy, transposed, transposed ? pad - 2 : -pad, transposed ? pad - 2 : -pad,
              d->act, d->slope, d->out_f32, 0, 0, 0, 0, 0, stats != nullptr ? G : 0, addend, ld_add, reflect, 0, 0, 0};
    p.x_bytes = (unsigned)(2 * (size_t)p.N * p.Hin * p.Win * p.ld_in);      // (< 2^31: glds_shape_ok; a channel slice's last pixel ends before that, and no live offset points there)
    p.w_bytes = (unsigned)(2 * (size_t)Cout * 9 * Cin);
    // whole-row stores: 2-byte results without an addend, channel rows of whole 16-byte chunks at 16-byte aligned addresses, a
    // piecewise-linear epilogue
    p.row_store = !d->out_f32 && Cout % 8 == 0 && p.ld_out % 8 == 0 && (reinterpret_cast<size_t>(out) & 15) == 0 &&
                  (addend == nullptr || (ld_add % 4 == 0 && (reinterpret_cast<size_t>(addend) & 7) == 0)) &&
                  d->act != RDFC_ACT_TANH && d->act != RDFC_ACT_SIGMOID;
    if (p.row_store && rsf_ok(addend, ld_add)) p.row_store = 2;         // the addend in whole rows, from the flush (common.h)
    p.tiles_x = cdiv(p.W, TC);                                 // 128 co x (2 JR x 16) px or 64 co x (16 x 16) px per workgroup
    if (stats != nullptr) {
        if (G < 1 || p.N % G != 0 || transposed || reflect) return 0;
        p.npg = p.N / G;
    }
    const GPlan g = glds_plan(p.N, p.H, p.W, p.Hin, p.oy0, Cin, Cout, stats != nullptr ? G : 0, reflect != 0);
    if (g.tiles_y <= 0) return 0;                              // (no plan: an empty grid must not count as "launched")
    p.tiles_y = g.tiles_y;
    p.vstride = g.vstride;
    p.gstride = g.gstride;
    p.r_tiles_x = 1.f / (float)p.tiles_x;
    p.r_tiles_y = 1.f / (float)p.tiles_y;
    p.r_vstride = p.vstride ? 1.f / (float)p.vstride : 0.f;
    p.r_gstride = p.gstride ? 1.f / (float)p.gstride : 0.f;
    p.nchunk = p.stats_groups > 0 ? (p.tiles_y / p.stats_groups) * p.tiles_x : 1;
    p.r_nchunk = 1.f / (float)p.nchunk;
    if ((long)p.tiles_x * p.tiles_y * (p.vstride ? 1 : p.N) >= (1L << 22) || (long)p.N * (p.vstride > 0 ? p.vstride : p.H) + g.gstride >= (1L << 22)) return 0;   // (qdiv's range)
#define RDFC_GLDS_LAUNCH(DT_)                                                                                        \
    if (Cout <= 64) return launch<64, DT_>(p, in, w, bias, out, zero, stats, st);                                  \
    switch (g.jr) {                                                                                                  \
        case 6: return launch<128, DT_, 6>(p, in, w, bias, out, zero, stats, st);                                  \
        case 8: return launch<128, DT_, 8>(p, in, w, bias, out, zero, stats, st);                                  \
        case 10: return launch<128, DT_, 10>(p, in, w, bias, out, zero, stats, st);                                \
        default: return launch<128, DT_>(p, in, w, bias, out, zero, stats, st);                                    \
    }
    if (d->dtype == RDFC_F16) {
        RDFC_GLDS_LAUNCH(RDFC_F16)
    }
    RDFC_GLDS_LAUNCH(RDFC_BF16)
#undef RDFC_GLDS_LAUNCH
}

#ifdef RDFC_GLDS_STAMPS
extern "C" int rdfc_debug_glds_stamps(unsigned long long *out) {
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : 1;
}
extern "C" int rdfc_debug_glds_tile_stamps(unsigned long long *out, int clear) {
    hipDeviceSynchronize();
    if (clear) {
        static unsigned long long z[512][10];
        return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stamps), z, sizeof(z)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stamps), sizeof(g_tile_stamps)) == hipSuccess ? 0 : 1;
}
#endif
