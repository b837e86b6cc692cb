// norm.hip -- HBM-bound per-channel kernels on pixel-major maps [G][P][C] (ld):
// BatchNorm / InstanceNorm statistics, normalise (+residual +activation), their
// backward, W-AdaIN, per-channel sums.
//
// Thread mapping (all kernels): a thread owns one 16-byte channel vector (cv) and
// walks pixels with stride RPB = 256 / CVT, so per-channel constants live in
// registers and every access is a full 16-byte coalesced load/store.
// grid = (pixel chunks, groups, channel slabs of 256 vectors).
// Reductions are two-stage (per-chunk partials, then a fixed-order finalize in f64):
// deterministic, no atomics.
#include "common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int NT = 256;

struct Map {       // geometry shared by the kernels
    int G, P, C;   // groups, pixels per group, channels
    int chunk;     // pixels per workgroup (host: chunk_pixels)
};

// Pixels per workgroup.  Reductions: 32 rows per thread (4 independent 16-byte loads in flight per stream, eight
// batches), halved down to 8 while the map would give fewer than 512 workgroups: the encoder's mid-size maps (128 ch @
// 114x152, 256 ch @ 57x76) used to get 136-271 workgroups and ran at <= 2.7 TB/s (tools/norm_trace.sh).  Going
// further (4 rows, ~1000 workgroups everywhere) measured 30 % SLOWER: every workgroup pays ~2 us of dependent
// per-channel constant loads before it streams.  Capped at 4096 chunks per group.
inline int chunk_pixels(int P, int C, int vec) {
    const int cvt = C / vec < NT ? (C / vec > 0 ? C / vec : 1) : NT;
    const int rpb = NT / cvt;
    constexpr int min_rows = 8, want_wg = 512;
    int rows = 32;
    while (rows > min_rows && P / (rows * rpb) < want_wg) rows >>= 1;
    int chunk = rows * rpb;
    const int lo = cdiv(P, 4096);
    if (chunk < lo) chunk = cdiv(lo, rpb) * rpb;
    return chunk;
}
inline Map make_map(int G, int P, int C, int vec) { return Map{G, P, C, chunk_pixels(P, C, vec)}; }
// streaming (non-reducing) kernels: 8 rows per thread -> ~2000 workgroups on a full-resolution map
inline Map make_map_stream(int G, int P, int C, int vec) {
    const int cvt = C / vec < NT ? (C / vec > 0 ? C / vec : 1) : NT;
    const int rpb = NT / cvt;
    int c = 8 * rpb;
    const int lo = cdiv(P, 8192);
    if (c < lo) c = lo;
    return Map{G, P, C, c};
}

// body(p) for p = begin + r0, +RPB, ... < end, four rows per trip so four independent loads are in flight
// streamed outputs are written once and read by a later kernel: non-temporal stores measured +1..5 % (A/B on one
// device); non-temporal LOADS measured 10-15 % slower and stay off
// (round 4: plain stores again -- inside the step the consumer of a streamed map is the next kernel, and with plain stores it finds more
//  of it in the Infinity Cache: 241.0 vs 240.0 pairs/s, two interleaved pairs on one box; non-temporal stores were the other arm)
#define ST store
#define LD load
#define ROW_LOOP(cx, body)                                                          \
    {                                                                                \
        int p_ = (cx).p_begin + (cx).r0;                                             \
        const int st_ = (cx).RPB;                                                    \
        for (; p_ + 3 * st_ < (cx).p_end; p_ += 4 * st_) {                           \
            body(p_);                                                                \
            body(p_ + st_);                                                          \
            body(p_ + 2 * st_);                                                      \
            body(p_ + 3 * st_);                                                      \
        }                                                                            \
        for (; p_ < (cx).p_end; p_ += st_) body(p_);                                 \
    }

// eight rows per trip (16 independent loads in flight per thread in the two-tensor reductions): A/B, -DRDFC_NORM_ROWS8 --
// measured SLOWER on cache-cold maps (bwd_reduce 21.7 -> 24.6 us @ 256 ch 57x76, 42.5 -> 45.0 us @ 64 ch 228x304)
#define ROW_LOOP8(cx, body)                                                         \
    {                                                                                \
        int p_ = (cx).p_begin + (cx).r0;                                             \
        const int st_ = (cx).RPB;                                                    \
        for (; p_ + 7 * st_ < (cx).p_end; p_ += 8 * st_) {                           \
            body(p_); body(p_ + st_); body(p_ + 2 * st_); body(p_ + 3 * st_);       \
            body(p_ + 4 * st_); body(p_ + 5 * st_); body(p_ + 6 * st_); body(p_ + 7 * st_); \
        }                                                                            \
        for (; p_ < (cx).p_end; p_ += st_) body(p_);                                 \
    }

// Reductions: the same rows in the same order as ROW_LOOP, with the NEXT trip's four loads issued before the current trip's sums
// (hipcc does not pipeline the loop by itself: four loads, wait for all four, ~100 VALU instructions, and only then the next loads --
// a wave had nothing in flight while it computed; the read-only reductions ran at 3.4-4.1 TB/s beside 6 TB/s of the read+write passes)
template <int NR, typename CX, typename L, typename A>
__device__ __forceinline__ void row_loop_pf(const CX &cx, L ld, A acc) {
    int p = cx.p_begin + cx.r0;
    const int st = cx.RPB;
    using V = decltype(ld(0));
    if (p + (NR - 1) * st < cx.p_end) {
        V a[NR], b[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) a[i] = ld(p + i * st);
        p += NR * st;
        for (; p + (NR - 1) * st < cx.p_end; p += NR * st) {
#pragma unroll
            for (int i = 0; i < NR; ++i) b[i] = ld(p + i * st);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NR; ++i) acc(a[i]);
#pragma unroll
            for (int i = 0; i < NR; ++i) a[i] = b[i];
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) acc(a[i]);
    }
    for (; p < cx.p_end; p += st) acc(ld(p));
}

template <typename T>
struct Ctx {
    static constexpr int VEC = Elem<T>::VEC;
    int cv, r0, RPB, c0, p_begin, p_end, g, chunk, nchunk;
    bool active;
    __device__ Ctx(const Map &m) {
        const int CV = m.C / VEC;
        const int slab0 = blockIdx.z * NT;
        const int CVT = min(CV - slab0, NT);
        RPB = NT / CVT;
        const int t = threadIdx.x;
        active = t < RPB * CVT;
        cv = slab0 + t % CVT;
        r0 = t / CVT;
        c0 = cv * VEC;
        g = blockIdx.y;
        chunk = blockIdx.x;
        nchunk = gridDim.x;
        p_begin = chunk * m.chunk;
        p_end = min(p_begin + m.chunk, m.P);
    }
};

inline dim3 grid_for(const Map &m, int vec) {
    return dim3(cdiv(m.P, m.chunk), m.G, cdiv(m.C / vec, NT));
}

// block-level reduction of K*VEC per-thread sums over the RPB rows, then write partial[g][chunk][k][c].
// When the channel-vector count CVT divides 64, rows that share a channel vector sit in the same wave at lane
// stride CVT: butterfly over those lanes first (xor CVT, 2CVT, ...), then one LDS slot per wave -- instead of a
// handful of threads serially adding up to 64 rows out of LDS (that tail used to cost more than the streaming part).
template <typename T, int K>
__device__ void block_reduce_store(const Ctx<T> &cx, const Map &m, float (&s)[K][Elem<T>::VEC], float *partial) {
    // Every thread deposits its K x VEC sums in LDS ([k][thread][v], 16-byte stores), then the K * CVT * VEC outputs are dealt to
    // ALL 256 threads: output o = (k, channel vector, v) sums its RPB rows (fixed order r = 0 .. RPB-1: reproducible) and is stored
    // with its neighbours in one coalesced line.  (Before: a butterfly of 24 runtime-offset shuffles plus a tail in which CVT
    // threads added and stored everything -- 6.6 of bwd_reduce's 21.7 us on a 256-channel 57x76 map.)
    constexpr int VEC = Elem<T>::VEC;
    __shared__ __attribute__((aligned(16))) float red[K * NT * VEC];
    const int t = threadIdx.x;
    const int CVT = min(m.C / VEC - (int)blockIdx.z * NT, NT), RPB = NT / CVT;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q)
            *reinterpret_cast<f32x4 *>(&red[(k * NT + t) * VEC + 4 * q]) =
                cx.active ? f32x4{s[k][4 * q], s[k][4 * q + 1], s[k][4 * q + 2], s[k][4 * q + 3]} : f32x4{0, 0, 0, 0};
    __syncthreads();
    const int per_k = CVT * VEC, nout = K * per_k;
    const int cbase = (int)blockIdx.z * NT * VEC;
    for (int o = t; o < nout; o += NT) {
        const int k = o / per_k, cvv = o - k * per_k;            // cvv = channel-vector * VEC + v: the channel offset inside the slab
        const int cvt = cvv / VEC, v = cvv - cvt * VEC;
        float a = 0.f;
        for (int r = 0; r < RPB; ++r) a += red[(k * NT + r * CVT + cvt) * VEC + v];
        const int c = cbase + cvv;
        partial[(((size_t)cx.g * cx.nchunk + cx.chunk) * K + k) * m.C + c] = a;
    }
}

// VEC consecutive f32 per-channel constants with 16-byte loads (c0 is a multiple of VEC; torch allocations are 256-byte aligned)
template <int VEC>
__device__ __forceinline__ void load_consts(const float *__restrict__ p, int idx, float dflt, float (&o)[VEC]) {
    if (p == nullptr) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) o[v] = dflt;
        return;
    }
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(p + idx + 4 * q);
        o[4 * q] = t[0]; o[4 * q + 1] = t[1]; o[4 * q + 2] = t[2]; o[4 * q + 3] = t[3];
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm over G stacked batches (several calls of the same layer run as one launch, rdfc_gan.py's batched schedule):
// every group gets its own statistics; the running statistics advance once per group, in group order, each with its own
// momentum -- exactly what G separate forwards would leave behind.
constexpr int MAXSEQ = 8;
struct Moms { float m[MAXSEQ]; };

template <typename T>
__global__ __launch_bounds__(NT) void stats_kernel(Map m, const T *__restrict__ x, int ld, float *__restrict__ partial) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    float s[2][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[0][v] = s[1][v] = 0.f;
    if (cx.active) {
        const T *base = x + (size_t)cx.g * m.P * ld + cx.c0;
        auto fetch = [&](int p) {
            Vec16<T> a;
            a.load(base + (size_t)p * ld);
            return a;
        };
        auto add = [&](const Vec16<T> &a) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float f = a.get(v);
                s[0][v] += f;
                s[1][v] += f * f;
            }
        };
        row_loop_pf<4>(cx, fetch, add);
    }
    block_reduce_store<T, 2>(cx, m, s, partial);
}

// fixed-order sum over the chunk partials of one (group, k, channel): FC channels x FLn lanes per workgroup,
// lane l adds chunks l, l+FLn, ... in f64, then the FLn lane sums are added in lane order (deterministic).
// Workgroup shape (round 4).  1024 threads (32 x 32) waited in the step for a CU with 16 free wave slots beside the conv kernels: 512
// threads + 1.5 % on the step.  Then by row count: the conv epilogues' fused statistics leave one row per TILE (4332 on a full-resolution
// map), i.e. 135 dependent-latency rows per lane at 32 x 16 -- 256 threads as 8 x 32 or 4 x 64 (pick_fc) give 4-8 x the workgroups
// and a quarter of the rows per lane.
constexpr int FL = 16, FNT = 32 * FL;
constexpr int FNT2 = 256;                       // threads of the <FC> kernels: FC channels x (256 / FC) lanes
template <int FC = 32, int FLn = FL>
__device__ inline double chunk_sum(const float *__restrict__ partial, int g, int K, int k, int C, int c, int nchunk,
                                   double *red /* [FLn][FC] */) {
    const int lane = threadIdx.x / FC, cl = threadIdx.x % FC;
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};         // fixed association -> still deterministic; 8 loads in flight (a lane has ~17)
    if (c < C) {
        const float *base = partial + ((size_t)g * nchunk * K + k) * C + c;
        const size_t stride = (size_t)K * C;
        int j = lane;
        for (; j + 7 * FLn < nchunk; j += 8 * FLn) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = base[(size_t)(j + q * FLn) * stride];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] += v[q];
        }
        for (int q = 0; j < nchunk; j += FLn, ++q) a[q & 7] += base[(size_t)j * stride];
    }
    const double s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    red[lane * FC + cl] = s;
    __syncthreads();
    double t = 0;
    for (int l = 0; l < FLn; ++l) t += red[l * FC + cl];
    return t;
}
// channels per workgroup of the <FC> finalize kernels by the number of rows to add
inline int pick_fc(int nchunk) {
    return nchunk <= 256 ? 32 : (nchunk <= 1024 ? 8 : 4);
}
#define RDFC_FC_DISPATCH(fc, ...)                                   \
    do {                                                            \
        if ((fc) == 32) { constexpr int FC = 32; __VA_ARGS__; }     \
        else if ((fc) == 8) { constexpr int FC = 8; __VA_ARGS__; }  \
        else { constexpr int FC = 4; __VA_ARGS__; }                 \
    } while (0)

template <int FC>
__global__ __launch_bounds__(FNT2) void stats_finalize_kernel(Map m, int nchunk, const float *__restrict__ partial,
                                                             float *__restrict__ mean, float *__restrict__ rstd, float eps,
                                                             float *run_mean, float *run_var, float momentum) {
    __shared__ double red[FNT2];
    const int g = blockIdx.y, c = blockIdx.x * FC + (threadIdx.x % FC);
    const double s1 = chunk_sum<FC, FNT2 / FC>(partial, g, 2, 0, m.C, c, nchunk, red);
    const double s2 = chunk_sum<FC, FNT2 / FC>(partial, g, 2, 1, m.C, c, nchunk, red);
    if (threadIdx.x >= FC || c >= m.C) return;
    const int i = g * m.C + c;
    const double mu = s1 / m.P;
    double var = s2 / m.P - mu * mu;
    if (var < 0) var = 0;
    mean[i] = (float)mu;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean != nullptr && g == 0) {   // batch norm (G == 1): nn.BatchNorm2d running statistics
        const double unb = m.P > 1 ? var * m.P / (m.P - 1) : var;
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mu);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
}

__global__ __launch_bounds__(FNT) void stats_finalize_seq_kernel(Map m, int nchunk, const float *__restrict__ partial,
                                                                float *__restrict__ mean, float *__restrict__ rstd, float eps,
                                                                float *run_mean, float *run_var, Moms moms) {
    __shared__ double red[FL * 32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const bool writer = threadIdx.x < 32 && c < m.C;
    double rm = 0, rv = 0;
    if (writer && run_mean != nullptr) {
        rm = run_mean[c];
        rv = run_var[c];
    }
    for (int g = 0; g < m.G; ++g) {
        const double s1 = chunk_sum(partial, g, 2, 0, m.C, c, nchunk, red);
        const double s2 = chunk_sum(partial, g, 2, 1, m.C, c, nchunk, red);
        if (!writer) continue;
        const double mu = s1 / m.P;
        double var = s2 / m.P - mu * mu;
        if (var < 0) var = 0;
        mean[g * m.C + c] = (float)mu;
        rstd[g * m.C + c] = (float)(1.0 / sqrt(var + (double)eps));
        const double unb = m.P > 1 ? var * m.P / (m.P - 1) : var, mom = moms.m[g];
        rm = (double)(float)((1.0 - mom) * rm + mom * mu);          // rounded to f32 per update, like G separate launches
        rv = (double)(float)((1.0 - mom) * rv + mom * unb);
    }
    if (writer && run_mean != nullptr) {
        run_mean[c] = (float)rm;
        run_var[c] = (float)rv;
    }
}

// Round 4: the same per (group, 32 channels) -- grid (C / 32, G) instead of one workgroup walking the groups -- which leaves the
// in-order update of the running statistics to ONE thread per channel of the normalise pass that follows (apply_kernel, RunUpd): it
// reads every group's (mean, biased variance) in f64 from muvar[G][2][C].  13.9 -> ~6 us per launch at two stacked batches.
template <int FC>
__global__ __launch_bounds__(FNT2) void stats_finalize_par_kernel(Map m, int nchunk, const float *__restrict__ partial,
                                                                 float *__restrict__ mean, float *__restrict__ rstd, float eps,
                                                                 double *__restrict__ muvar) {
    __shared__ double red[FNT2];
    const int g = blockIdx.y, c = blockIdx.x * FC + (threadIdx.x % FC);
    const double s1 = chunk_sum<FC, FNT2 / FC>(partial, g, 2, 0, m.C, c, nchunk, red);
    const double s2 = chunk_sum<FC, FNT2 / FC>(partial, g, 2, 1, m.C, c, nchunk, red);
    if (threadIdx.x >= FC || c >= m.C) return;
    const double mu = s1 / m.P;
    double var = s2 / m.P - mu * mu;
    if (var < 0) var = 0;
    mean[g * m.C + c] = (float)mu;
    rstd[g * m.C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (muvar != nullptr) {
        muvar[((size_t)g * 2 + 0) * m.C + c] = mu;
        muvar[((size_t)g * 2 + 1) * m.C + c] = var;
    }
}

// running statistics of a BatchNorm over G stacked batches, advanced in group order by the normalise pass (see above)
struct RunUpd {
    float *run_mean, *run_var;      // nullptr: nothing to do
    const double *muvar;            // [G][2][C]
    Moms moms;
};

struct ApplyArgs {
    const float *mean, *rstd, *gamma, *beta, *prelu;
    int act;            // none | relu | lrelu | prelu only: all piecewise linear -> act(u) = u > 0 ? u : u * neg_slope
    float slope;
};
__device__ inline float neg_slope_of(const ApplyArgs &a) {
    return a.act == RDFC_ACT_NONE ? 1.f : (a.act == RDFC_ACT_RELU ? 0.f : (a.act == RDFC_ACT_PRELU ? *a.prelu : a.slope));
}

template <typename T>
__global__ __launch_bounds__(NT) void apply_kernel(Map m, ApplyArgs a, const T *__restrict__ x, const T *__restrict__ res,
                                                   T *__restrict__ z, int ldx, int ldr, int ldz, RunUpd ru) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    if (!cx.active) return;
    if (ru.run_mean != nullptr && cx.chunk == 0 && cx.g == 0 && cx.r0 == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = cx.c0 + v;
            double rm = ru.run_mean[c], rv = ru.run_var[c];
            for (int g = 0; g < m.G; ++g) {
                const double mu = ru.muvar[((size_t)g * 2 + 0) * m.C + c], var = ru.muvar[((size_t)g * 2 + 1) * m.C + c];
                const double unb = m.P > 1 ? var * m.P / (m.P - 1) : var, mom = ru.moms.m[g];
                rm = (double)(float)((1.0 - mom) * rm + mom * mu);          // rounded to f32 per update, like G separate launches
                rv = (double)(float)((1.0 - mom) * rv + mom * unb);
            }
            ru.run_mean[c] = (float)rm;
            ru.run_var[c] = (float)rv;
        }
    }
    float sc[VEC], sh[VEC];
    {                                   // per-channel constants by 16-byte loads
        float gav[VEC], bev[VEC], rsv[VEC], muv[VEC];
        load_consts<VEC>(a.gamma, cx.c0, 1.f, gav);
        load_consts<VEC>(a.beta, cx.c0, 0.f, bev);
        load_consts<VEC>(a.rstd, cx.g * m.C + cx.c0, 1.f, rsv);
        load_consts<VEC>(a.mean, cx.g * m.C + cx.c0, 0.f, muv);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            sc[v] = gav[v] * rsv[v];
            sh[v] = bev[v] - muv[v] * gav[v] * rsv[v];
        }
    }
    const float ns = neg_slope_of(a);
    const size_t gb = (size_t)cx.g * m.P;
    auto body = [&](int p) {
        Vec16<T> xv, rv, o;
        xv.LD(x + (gb + p) * ldx + cx.c0);
        if (res) rv.LD(res + (gb + p) * ldr + cx.c0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float u = sc[v] * xv.get(v) + sh[v];
            if (res) u += rv.get(v);
            o.set(v, u > 0.f ? u : u * ns);
        }
        o.ST(z + (gb + p) * ldz + cx.c0);
    };
        ROW_LOOP(cx, body);
}

// backward pass 1: sums of du, du*xhat, dz*min(u,0)
// RES: a residual map joins the pre-activation; PRELU: the slope's gradient (third sum) is wanted -- compile-time, so the common
// layer (neither) carries no registers for them: 4 waves per SIMD with two trips of rows in flight
template <typename T, bool RES, bool PRELU>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(3))) void bwd_reduce_kernel(Map m, ApplyArgs a, const T *__restrict__ dz, const T *__restrict__ x,
                                                        const T *__restrict__ res, float *__restrict__ partial, int lddz,
                                                        int ldx, int ldr) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    float s[3][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[0][v] = s[1][v] = s[2][v] = 0.f;
    if (cx.active) {
        float sc[VEC], sh[VEC], mu[VEC];
        {
            float gav[VEC], bev[VEC], rs[VEC];
            load_consts<VEC>(a.gamma, cx.c0, 1.f, gav);
            load_consts<VEC>(a.beta, cx.c0, 0.f, bev);
            load_consts<VEC>(a.rstd, cx.g * m.C + cx.c0, 1.f, rs);
            load_consts<VEC>(a.mean, cx.g * m.C + cx.c0, 0.f, mu);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                sc[v] = gav[v] * rs[v];                       // the forward's (apply_kernel) pre-activation, bit for bit,
                sh[v] = bev[v] - mu[v] * gav[v] * rs[v];      // so the activation mask cannot disagree with the forward
            }
        }
        const float ns = neg_slope_of(a);
        const size_t gb = (size_t)cx.g * m.P;
        struct Row2 {
            Vec16<T> dv, xv;
        };
        struct Row3 {
            Vec16<T> dv, xv, rv;
        };
        using Row = typename std::conditional<RES, Row3, Row2>::type;
        auto fetch = [&](int p) {
            Row r;
            r.dv.load(dz + (gb + p) * lddz + cx.c0);
            r.xv.load(x + (gb + p) * ldx + cx.c0);
            if constexpr (RES) r.rv.load(res + (gb + p) * ldr + cx.c0);
            return r;
        };
        auto add = [&](const Row &r) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float xv_ = r.xv.get(v);
                const float xc = xv_ - mu[v];                  // (x - mean); the common factor rstd multiplies the finished sum below
                float u = sc[v] * xv_ + sh[v];
                if constexpr (RES) u += r.rv.get(v);
                const float d = r.dv.get(v);
                const float du = u > 0.f ? d : d * ns;
                s[0][v] += du;
                s[1][v] += du * xc;
                if constexpr (PRELU) s[2][v] += d * fminf(u, 0.f);
            }
        };
        row_loop_pf<2>(cx, fetch, add);
        float rs[VEC];                                        // (loaded here, not held across the loop: registers)
        load_consts<VEC>(a.rstd, cx.g * m.C + cx.c0, 1.f, rs);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[1][v] *= rs[v];
    }
    block_reduce_store<T, 3>(cx, m, s, partial);
}

template <int FC>
__global__ __launch_bounds__(FNT2) void sum_finalize_kernel(int G, int C, int K, int nchunk, const float *__restrict__ partial,
                                                           float *__restrict__ sums, float *__restrict__ acc0, float *__restrict__ acc1) {
    __shared__ double red[FNT2];
    const int g = blockIdx.y / K, k = blockIdx.y % K, c = blockIdx.x * FC + (threadIdx.x % FC);
    const double s = chunk_sum<FC, FNT2 / FC>(partial, g, K, k, C, c, nchunk, red);
    if (threadIdx.x < FC && c < C) {
        sums[((size_t)g * K + k) * C + c] = (float)s;
        // optional in-place accumulation of sum 0 / sum 1 (a bias gradient: ONE group, so one add per channel and launch; atomic
        // because the same parameter's two uses per step may run their backward on different streams)
        if (k == 0 && acc0 != nullptr) atomicAdd(&acc0[c], (float)s);
        if (k == 1 && acc1 != nullptr) atomicAdd(&acc1[c], (float)s);
    }
}

template <typename T, bool DXS = false>        // DXS: also the partial rows of sum(stored dx) -- its own instantiation: the block reduction's 8 KB of LDS stay out of the plain one
__global__ __launch_bounds__(NT) void bwd_apply_kernel(Map m, ApplyArgs a, const T *__restrict__ dz, const T *__restrict__ x,
                                                       const T *__restrict__ res, const float *__restrict__ sums,
                                                       T *__restrict__ dx, T *__restrict__ dres, int lddz, int ldx, int ldr,
                                                       int lddx, int lddr, int frozen, float *__restrict__ dbeta_acc,
                                                       float *__restrict__ dgamma_acc, float *__restrict__ dxsum) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    // dxsum (round 6): per-workgroup partial rows [g][chunk][C] of the sum over pixels of the STORED input gradient -- the gradient of a
    // conv bias in front of the norm layer (ResnetGenerator's convs, resnet_generator.py:60-100: analytically zero, in practice the
    // rounding noise that Adam turns into lr-sized steps, upstream's too); it used to be one more pass over dx (rdfc_channel_sum).
    // With it every thread stays to the end (block_reduce_store has a workgroup barrier)
    if (!cx.active && !DXS) return;
    float sb[1][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) sb[0][v] = 0.f;
    if (cx.active) {
    // d beta / d gamma of an affine norm straight into their .grad: ONE thread per channel adds the groups' sums (BatchNorm over
    // stacked batches has one set per group) in group order and issues one add -- reproducible whatever the launch order of the
    // workgroups; the add is atomic because a parameter's two uses per step may run their backward on different streams
    if (dbeta_acc != nullptr && cx.chunk == 0 && cx.r0 == 0 && cx.g == 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = cx.c0 + v;
            float tb = 0.f, tg = 0.f;
            for (int g = 0; g < m.G; ++g) {
                tb += sums[(g * 3 + 0) * m.C + c];
                tg += sums[(g * 3 + 1) * m.C + c];
            }
            atomicAdd(&dbeta_acc[c], tb);
            if (dgamma_acc != nullptr) atomicAdd(&dgamma_acc[c], tg);
        }
    }
    float ga[VEC], sc[VEC], sh[VEC], mu[VEC], rs[VEC], m1[VEC], m2[VEC];
    const float invP = 1.f / (float)m.P;
    float bev[VEC], s1v[VEC], s2v[VEC];
    load_consts<VEC>(a.gamma, cx.c0, 1.f, ga);
    load_consts<VEC>(a.beta, cx.c0, 0.f, bev);
    load_consts<VEC>(a.rstd, cx.g * m.C + cx.c0, 1.f, rs);
    load_consts<VEC>(a.mean, cx.g * m.C + cx.c0, 0.f, mu);
    load_consts<VEC>(sums, (cx.g * 3 + 0) * m.C + cx.c0, 0.f, s1v);
    load_consts<VEC>(sums, (cx.g * 3 + 1) * m.C + cx.c0, 0.f, s2v);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        sc[v] = ga[v] * rs[v];
        sh[v] = bev[v] - mu[v] * ga[v] * rs[v];
        m1[v] = frozen ? 0.f : s1v[v] * invP;
        m2[v] = frozen ? 0.f : s2v[v] * invP;
    }
    const float ns = neg_slope_of(a);
    const size_t gb = (size_t)cx.g * m.P;
    auto body = [&](int p) {
        Vec16<T> dv, xv, rv, o, orr;
        dv.LD(dz + (gb + p) * lddz + cx.c0);
        xv.LD(x + (gb + p) * ldx + cx.c0);
        if (res) rv.LD(res + (gb + p) * ldr + cx.c0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float xh = (xv.get(v) - mu[v]) * rs[v];
            float u = sc[v] * xv.get(v) + sh[v];
            if (res) u += rv.get(v);
            const float d = dv.get(v);
            const float du = u > 0.f ? d : d * ns;
            o.set(v, sc[v] * (du - m1[v] - xh * m2[v]));
            orr.set(v, du);
        }
        if constexpr (DXS) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) sb[0][v] += o.get(v);     // (the rounded value that is stored)
        }
        o.ST(dx + (gb + p) * lddx + cx.c0);
        if (dres) orr.ST(dres + (gb + p) * lddr + cx.c0);
    };
        ROW_LOOP(cx, body);
    }
    if constexpr (DXS) block_reduce_store<T, 1>(cx, m, sb, dxsum);
}

// ------------------------------------------------------------------------------------------
// W-AdaIN: out = gamma_p * xhat + beta_p, (gamma_p | beta_p) = gb[p][0:C | C:2C]
template <typename T>
__global__ __launch_bounds__(NT) void adain_apply_kernel(Map m, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const T *__restrict__ x, const T *__restrict__ gbm, T *__restrict__ out,
                                                         int ldx, int ldgb, int ldo) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    if (!cx.active) return;
    float mu[VEC], rs[VEC];
    load_consts<VEC>(mean, cx.g * m.C + cx.c0, 0.f, mu);
    load_consts<VEC>(rstd, cx.g * m.C + cx.c0, 1.f, rs);
    const size_t gb = (size_t)cx.g * m.P;
    auto body = [&](int p) {
        Vec16<T> xv, gv, bv, o;
        xv.load(x + (gb + p) * ldx + cx.c0);
        gv.load(gbm + (gb + p) * ldgb + cx.c0);
        bv.load(gbm + (gb + p) * ldgb + m.C + cx.c0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) o.set(v, gv.get(v) * ((xv.get(v) - mu[v]) * rs[v]) + bv.get(v));
        o.store(out + (gb + p) * ldo + cx.c0);
    };
        ROW_LOOP(cx, body);
}

template <typename T>
__global__ __launch_bounds__(NT) void adain_bwd_reduce_kernel(Map m, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                              const T *__restrict__ dout, const T *__restrict__ x,
                                                              const T *__restrict__ gbm, float *__restrict__ partial, int lddo,
                                                              int ldx, int ldgb) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    float s[3][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[0][v] = s[1][v] = s[2][v] = 0.f;
    if (cx.active) {
        float mu[VEC], rs[VEC];
        load_consts<VEC>(mean, cx.g * m.C + cx.c0, 0.f, mu);
        load_consts<VEC>(rstd, cx.g * m.C + cx.c0, 1.f, rs);
        const size_t gb = (size_t)cx.g * m.P;
        auto body = [&](int p) {
            Vec16<T> dv, xv, gv;
            dv.load(dout + (gb + p) * lddo + cx.c0);
            xv.load(x + (gb + p) * ldx + cx.c0);
            gv.load(gbm + (gb + p) * ldgb + cx.c0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float xh = (xv.get(v) - mu[v]) * rs[v];
                const float dxh = dv.get(v) * gv.get(v);
                s[0][v] += dxh;
                s[1][v] += dxh * xh;
            }
        };
        ROW_LOOP(cx, body);
    }
    block_reduce_store<T, 3>(cx, m, s, partial);
}

template <typename T>
__global__ __launch_bounds__(NT) void adain_bwd_apply_kernel(Map m, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                             const T *__restrict__ dout, const T *__restrict__ x,
                                                             const T *__restrict__ gbm, const float *__restrict__ sums,
                                                             T *__restrict__ dx, T *__restrict__ dgb, int lddo, int ldx, int ldgb,
                                                             int lddx, int lddgb) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    if (!cx.active) return;
    float mu[VEC], rs[VEC], m1[VEC], m2[VEC];
    const float invP = 1.f / (float)m.P;
    load_consts<VEC>(mean, cx.g * m.C + cx.c0, 0.f, mu);
    load_consts<VEC>(rstd, cx.g * m.C + cx.c0, 1.f, rs);
    load_consts<VEC>(sums, (cx.g * 3 + 0) * m.C + cx.c0, 0.f, m1);
    load_consts<VEC>(sums, (cx.g * 3 + 1) * m.C + cx.c0, 0.f, m2);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        m1[v] *= invP;
        m2[v] *= invP;
    }
    const size_t gb = (size_t)cx.g * m.P;
    auto body = [&](int p) {
        Vec16<T> dv, xv, gv, o, og, ob;
        dv.load(dout + (gb + p) * lddo + cx.c0);
        xv.load(x + (gb + p) * ldx + cx.c0);
        gv.load(gbm + (gb + p) * ldgb + cx.c0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float xh = (xv.get(v) - mu[v]) * rs[v];
            const float d = dv.get(v);
            const float dxh = d * gv.get(v);
            o.set(v, rs[v] * (dxh - m1[v] - xh * m2[v]));
            og.set(v, d * xh);
            ob.set(v, d);
        }
        o.store(dx + (gb + p) * lddx + cx.c0);
        og.store(dgb + (gb + p) * lddgb + cx.c0);
        ob.store(dgb + (gb + p) * lddgb + m.C + cx.c0);
    };
        ROW_LOOP(cx, body);
}

// ------------------------------------------------------------------------------------------
// per-channel sum.  Vector path (C a multiple of the 16-byte vector): the common Ctx mapping, K = 1.
template <typename T>
__global__ __launch_bounds__(NT) void chsum_vec_kernel(Map m, int ld, const T *__restrict__ x, float *__restrict__ partial) {
    constexpr int VEC = Elem<T>::VEC;
    Ctx<T> cx(m);
    float s[1][VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[0][v] = 0.f;
    if (cx.active) {
        const T *base = x + cx.c0;
        auto body = [&](int p) {
            Vec16<T> a;
            a.load(base + (size_t)p * ld);
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[0][v] += a.get(v);
        };
        ROW_LOOP(cx, body);
    }
    block_reduce_store<T, 1>(cx, m, s, partial);
}
// scalar path (C = 1, 3, ...): thread t -> channel t % Cb, rows t / Cb
template <typename T>
__global__ __launch_bounds__(NT) void chsum_kernel(int P, int C, int ld, int chunk, const T *__restrict__ x, float *__restrict__ partial) {
    const int slab0 = blockIdx.z * NT, Cb = min(C - slab0, NT), RPB = NT / Cb;
    const int t = threadIdx.x, c = slab0 + t % Cb, r0 = t / Cb;
    const bool active = t < RPB * Cb;
    float s = 0.f;
    const int pb = blockIdx.x * chunk, pe = min(pb + chunk, P);
    if (active)
        for (int p = pb + r0; p < pe; p += RPB) s += Elem<T>::ld(x + (size_t)p * ld + c);
    __shared__ float red[NT];
    red[t] = active ? s : 0.f;
    __syncthreads();
    if (active && r0 == 0) {
        float a = 0.f;
        for (int r = 0; r < RPB; ++r) a += red[r * Cb + t];
        partial[(size_t)blockIdx.x * C + c] = a;
    }
}

template <typename T>
__global__ void act_bwd_kernel(size_t total, int C, int lddy, int ldy, int lddx, int y_f32, int act, float slope,
                               const T *__restrict__ dy, const void *__restrict__ y, T *__restrict__ dx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / C;
        const int c = (int)(i - p * C);
        const float yv = y_f32 ? reinterpret_cast<const float *>(y)[p * ldy + c] : Elem<T>::ld(reinterpret_cast<const T *>(y) + p * ldy + c);
        Elem<T>::st(dx + p * lddx + c, Elem<T>::ld(dy + p * lddy + c) * act_grad_out(yv, act, slope));
    }
}

int check_map(const char *who, int G, int P, int C, int dtype, std::initializer_list<int> lds) {
    RDFC_REQUIRE(rdfc_dtype_ok(dtype), "%s: bad dtype %d", who, dtype);
    const int vec = rdfc_vec(dtype);
    RDFC_REQUIRE(G > 0 && P > 0 && C > 0, "%s: non-positive size", who);
    RDFC_REQUIRE(C % vec == 0, "%s: C=%d must be a multiple of %d", who, C, vec);
    for (int ld : lds) RDFC_REQUIRE(ld >= C && ld % vec == 0, "%s: ld=%d must be >= C and a multiple of %d", who, ld, vec);
    return 0;
}

}  // namespace

#define DISPATCH(dtype, ...) RDFC_DISPATCH(dtype, __VA_ARGS__)

// scratch of a two-stage reduction: [partials f32: G * nchunk * K * C | pad to 8 bytes | f64 [G][2][C] (stacked-batch BatchNorm: mean /
// variance per group, handed from the finalize launch to the normalise pass)]
namespace {
inline size_t part_floats(int G, int nchunk, int K, int C) { return ((size_t)G * nchunk * K * C + 1) & ~(size_t)1; }
}  // namespace

extern "C" int rdfc_norm_scratch_floats(int G, int nchunk, int K, int C) {
    return (int)(part_floats(G, nchunk, K, C) + 2 * ((size_t)G * 2 * C));
}

extern "C" int rdfc_norm_nchunk(int pixels_per_group, int C, int dtype) {
    return cdiv(pixels_per_group, chunk_pixels(pixels_per_group, C, rdfc_vec(dtype)));
}

extern "C" int rdfc_norm_stats(const void *x, float *partial, int G, int P, int C, int ld, int dtype, void *stream) {
    if (int e = check_map("rdfc_norm_stats", G, P, C, dtype, {ld})) return e;
    Map m = make_map(G, P, C, rdfc_vec(dtype));
    DISPATCH(dtype, hipLaunchKernelGGL(stats_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, (const T *)x, ld, partial));
    RDFC_LAUNCH_CHECK("rdfc_norm_stats");
    return 0;
}

extern "C" int rdfc_norm_finalize(const float *partial, float *mean, float *rstd, int G, int P, int C, int dtype, float eps,
                                  float *run_mean, float *run_var, float momentum, void *stream) {
    RDFC_REQUIRE(G > 0 && P > 0 && C > 0, "rdfc_norm_finalize: non-positive size");
    RDFC_REQUIRE(run_mean == nullptr || G == 1, "rdfc_norm_finalize: running statistics need G == 1 (batch norm)");
    Map m = make_map(G, P, C, rdfc_vec(dtype));
    {
        const int nchunk = cdiv(P, m.chunk);
        RDFC_FC_DISPATCH(pick_fc(nchunk), hipLaunchKernelGGL(stats_finalize_kernel<FC>, dim3(cdiv(C, FC), G), dim3(FNT2), 0, (hipStream_t)stream, m,
                                                             nchunk, partial, mean, rstd, eps, run_mean, run_var, momentum));
    }
    RDFC_LAUNCH_CHECK("rdfc_norm_finalize");
    return 0;
}

extern "C" int rdfc_norm_finalize_n(const float *partial, int nchunk, float *mean, float *rstd, int G, int P, int C, float eps,
                                    float *run_mean, float *run_var, float momentum, void *stream) {
    RDFC_REQUIRE(G > 0 && P > 0 && C > 0 && nchunk > 0, "rdfc_norm_finalize_n: non-positive size");
    RDFC_REQUIRE(run_mean == nullptr || G == 1, "rdfc_norm_finalize_n: running statistics need G == 1 (batch norm)");
    Map m{G, P, C, 0};
    RDFC_FC_DISPATCH(pick_fc(nchunk), hipLaunchKernelGGL(stats_finalize_kernel<FC>, dim3(cdiv(C, FC), G), dim3(FNT2), 0, (hipStream_t)stream, m, nchunk,
                                                         partial, mean, rstd, eps, run_mean, run_var, momentum));
    RDFC_LAUNCH_CHECK("rdfc_norm_finalize_n");
    return 0;
}

extern "C" int rdfc_norm_finalize_seq(const float *partial, int nchunk, float *mean, float *rstd, int G, int P, int C, int dtype,
                                      float eps, float *run_mean, float *run_var, const float *group_momentum, void *stream) {
    RDFC_REQUIRE(G > 0 && G <= MAXSEQ && P > 0 && C > 0, "rdfc_norm_finalize_seq: 1 <= G <= %d groups", MAXSEQ);
    RDFC_REQUIRE(run_mean == nullptr || group_momentum != nullptr, "rdfc_norm_finalize_seq: running statistics need the per-group momenta");
    Map m = make_map(G, P, C, rdfc_vec(dtype));
    Moms moms{};
    for (int g = 0; g < G; ++g) moms.m[g] = group_momentum ? group_momentum[g] : 0.f;
    hipLaunchKernelGGL(stats_finalize_seq_kernel, dim3(cdiv(C, 32)), dim3(FNT), 0, (hipStream_t)stream, m,
                       nchunk > 0 ? nchunk : cdiv(P, m.chunk), partial, mean, rstd, eps, run_mean, run_var, moms);
    RDFC_LAUNCH_CHECK("rdfc_norm_finalize_seq");
    return 0;
}

extern "C" int rdfc_norm_apply(const void *x, const float *mean, const float *rstd, const float *gamma, const float *beta,
                               const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype,
                               int act, float slope, const float *prelu, void *stream) {
    if (int e = check_map("rdfc_norm_apply", G, P, C, dtype, {ldx, ldz})) return e;
    RDFC_REQUIRE(act != RDFC_ACT_PRELU || prelu != nullptr, "rdfc_norm_apply: PReLU needs the slope pointer");
    RDFC_REQUIRE(act != RDFC_ACT_TANH && act != RDFC_ACT_SIGMOID, "rdfc_norm_apply: tanh / sigmoid never follow a norm on this path (conv epilogue only)");
    Map m = make_map_stream(G, P, C, rdfc_vec(dtype));
    ApplyArgs a{mean, rstd, gamma, beta, prelu, act, slope};
    DISPATCH(dtype, hipLaunchKernelGGL(apply_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, a,
                                       (const T *)x, (const T *)res, (T *)z, ldx, ldr, ldz, RunUpd{}));
    RDFC_LAUNCH_CHECK("rdfc_norm_apply");
    return 0;
}

// BatchNorm over G stacked batches, the part behind the partial rows: statistics per group from `partial` ([G][nchunk][2][C]; nchunk 0:
// derived from (P, C, dtype)), then the normalise pass, which also advances the running statistics once per group, in group order
// (group_momentum: HOST array of G floats; run_mean NULL: no running statistics).  muvar: f64 [G][2][C] scratch.
extern "C" int rdfc_norm_finalize_apply_seq(const float *partial, int nchunk, double *muvar, const void *x, float *mean, float *rstd,
                                            const float *gamma, const float *beta, const void *res, void *z, int G, int P, int C, int ldx,
                                            int ldr, int ldz, int dtype, float eps, float *run_mean, float *run_var,
                                            const float *group_momentum, int act, float slope, const float *prelu, void *stream) {
    if (int e = check_map("rdfc_norm_finalize_apply_seq", G, P, C, dtype, {ldx, ldz})) return e;
    RDFC_REQUIRE(G <= MAXSEQ || run_mean == nullptr, "rdfc_norm_finalize_apply_seq: running statistics need 1 <= G <= %d groups", MAXSEQ);
    RDFC_REQUIRE(run_mean == nullptr || (group_momentum != nullptr && muvar != nullptr), "rdfc_norm_finalize_apply_seq: running statistics need the per-group momenta and the f64 scratch");
    RDFC_REQUIRE(act != RDFC_ACT_PRELU || prelu != nullptr, "rdfc_norm_finalize_apply_seq: PReLU needs the slope pointer");
    RDFC_REQUIRE(act != RDFC_ACT_TANH && act != RDFC_ACT_SIGMOID, "rdfc_norm_finalize_apply_seq: tanh / sigmoid never follow a norm on this path");
    Map mr = make_map(G, P, C, rdfc_vec(dtype));
    {
        const int rows = nchunk > 0 ? nchunk : cdiv(P, mr.chunk);
        RDFC_FC_DISPATCH(pick_fc(rows), hipLaunchKernelGGL(stats_finalize_par_kernel<FC>, dim3(cdiv(C, FC), G), dim3(FNT2), 0, (hipStream_t)stream, mr, rows,
                                                           partial, mean, rstd, eps, run_mean != nullptr ? muvar : nullptr));
    }
    RunUpd ru{run_mean, run_var, muvar, Moms{}};
    for (int g = 0; g < G && g < MAXSEQ && group_momentum != nullptr; ++g) ru.moms.m[g] = group_momentum[g];
    Map m = make_map_stream(G, P, C, rdfc_vec(dtype));
    ApplyArgs a{mean, rstd, gamma, beta, prelu, act, slope};
    DISPATCH(dtype, hipLaunchKernelGGL(apply_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, a,
                                       (const T *)x, (const T *)res, (T *)z, ldx, ldr, ldz, ru));
    RDFC_LAUNCH_CHECK("rdfc_norm_finalize_apply_seq");
    return 0;
}

template <typename T>
static void launch_bwd_reduce_t(const Map &m, const ApplyArgs &a, const void *dz, const void *x, const void *res, float *partial, int lddz,
                                int ldx, int ldr, hipStream_t st) {
    const dim3 grid = grid_for(m, Elem<T>::VEC);
    const bool pr = a.act == RDFC_ACT_PRELU;
#define RDFC_BR(R, P_) hipLaunchKernelGGL((bwd_reduce_kernel<T, R, P_>), grid, dim3(NT), 0, st, m, a, (const T *)dz, (const T *)x, (const T *)res, partial, lddz, ldx, ldr)
    if (res != nullptr) {
        if (pr) RDFC_BR(true, true); else RDFC_BR(true, false);
    } else {
        if (pr) RDFC_BR(false, true); else RDFC_BR(false, false);
    }
#undef RDFC_BR
}
static void launch_bwd_reduce(int dtype, const Map &m, const ApplyArgs &a, const void *dz, const void *x, const void *res, float *partial,
                              int lddz, int ldx, int ldr, void *stream) {
    DISPATCH(dtype, launch_bwd_reduce_t<T>(m, a, dz, x, res, partial, lddz, ldx, ldr, (hipStream_t)stream));
}

extern "C" int rdfc_norm_bwd_reduce(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                    const float *beta, const void *res, float *partial, int G, int P, int C, int lddz,
                                    int ldx, int ldr, int dtype, int act, float slope, const float *prelu, void *stream) {
    if (int e = check_map("rdfc_norm_bwd_reduce", G, P, C, dtype, {lddz, ldx})) return e;
    Map m = make_map(G, P, C, rdfc_vec(dtype));
    ApplyArgs a{mean, rstd, gamma, beta, prelu, act, slope};
    launch_bwd_reduce(dtype, m, a, dz, x, res, partial, lddz, ldx, ldr, stream);
    RDFC_LAUNCH_CHECK("rdfc_norm_bwd_reduce");
    return 0;
}

extern "C" int rdfc_norm_bwd_finalize_acc(const float *partial, float *sums, int G, int P, int C, int dtype, float *dbeta_acc,
                                          float *dgamma_acc, void *stream) {
    // in-place accumulation from here: one statistics group only (several groups: rdfc_norm_backward adds them in group order)
    RDFC_REQUIRE(G == 1 || (dbeta_acc == nullptr && dgamma_acc == nullptr), "rdfc_norm_bwd_finalize_acc: accumulators need G == 1");
    {
        const int nchunk = cdiv(P, chunk_pixels(P, C, rdfc_vec(dtype)));
        RDFC_FC_DISPATCH(pick_fc(nchunk), hipLaunchKernelGGL(sum_finalize_kernel<FC>, dim3(cdiv(C, FC), G * 3), dim3(FNT2), 0, (hipStream_t)stream, G, C, 3,
                                                             nchunk, partial, sums, dbeta_acc, dgamma_acc));
    }
    RDFC_LAUNCH_CHECK("rdfc_norm_bwd_finalize");
    return 0;
}
extern "C" int rdfc_norm_bwd_finalize(const float *partial, float *sums, int G, int P, int C, int dtype, void *stream) {
    return rdfc_norm_bwd_finalize_acc(partial, sums, G, P, C, dtype, nullptr, nullptr, stream);
}

// floats of scratch the dx-sum variants need: the apply pass's partial rows [G][chunks][C] + the finalized sums [C]
extern "C" int rdfc_norm_dxsum_scratch_floats(int G, int P, int C, int dtype) {
    if (!rdfc_dtype_ok(dtype) || G < 1 || P < 1 || C < 1) return 0;
    const Map m = make_map_stream(G, P, C, rdfc_vec(dtype));
    return G * cdiv(P, m.chunk) * C + C;
}
// the apply pass; dxsum_scratch (rdfc_norm_dxsum_scratch_floats) + dbias_acc: also dbias_acc[c] += sum over groups and pixels of the stored dx
// (a fixed-order two-stage sum: partial rows per workgroup, then the finalize launch of rdfc_channel_sum -- bit-reproducible)
static int bwd_apply_launch(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma, const float *beta,
                            const void *res, const float *sums, void *dx, void *dres, int G, int P, int C, int lddz, int ldx, int ldr,
                            int lddx, int lddr, int dtype, int act, float slope, const float *prelu, int frozen_stats, float *dbeta_acc,
                            float *dgamma_acc, float *dxsum_scratch, float *dbias_acc, void *stream) {
    Map m = make_map_stream(G, P, C, rdfc_vec(dtype));
    ApplyArgs a{mean, rstd, gamma, beta, prelu, act, slope};
    float *part = dbias_acc != nullptr ? dxsum_scratch : nullptr;
    if (part == nullptr) {
        DISPATCH(dtype, hipLaunchKernelGGL(bwd_apply_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, a,
                                           (const T *)dz, (const T *)x, (const T *)res, sums, (T *)dx, (T *)dres, lddz, ldx, ldr,
                                           lddx, lddr, frozen_stats, dbeta_acc, dgamma_acc, nullptr));
    } else {
        DISPATCH(dtype, hipLaunchKernelGGL((bwd_apply_kernel<T, true>), grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, a,
                                           (const T *)dz, (const T *)x, (const T *)res, sums, (T *)dx, (T *)dres, lddz, ldx, ldr,
                                           lddx, lddr, frozen_stats, dbeta_acc, dgamma_acc, part));
        const int rows = G * cdiv(P, m.chunk);
        RDFC_FC_DISPATCH(pick_fc(rows), hipLaunchKernelGGL(sum_finalize_kernel<FC>, dim3(cdiv(C, FC), 1), dim3(FNT2), 0, (hipStream_t)stream, 1, C, 1, rows,
                                                          part, part + (size_t)rows * C, dbias_acc, nullptr));
    }
    return 0;
}

extern "C" int rdfc_norm_bwd_apply_acc(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                       const float *beta, const void *res, const float *sums, void *dx, void *dres, int G, int P,
                                       int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                                       const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, void *stream) {
    if (int e = check_map("rdfc_norm_bwd_apply", G, P, C, dtype, {lddz, ldx, lddx})) return e;
    RDFC_REQUIRE(dbeta_acc != nullptr || dgamma_acc == nullptr, "rdfc_norm_bwd_apply_acc: d gamma accumulates together with d beta");
    bwd_apply_launch(dz, x, mean, rstd, gamma, beta, res, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx, lddr, dtype, act, slope, prelu, frozen_stats,
                     dbeta_acc, dgamma_acc, nullptr, nullptr, stream);
    RDFC_LAUNCH_CHECK("rdfc_norm_bwd_apply");
    return 0;
}
extern "C" int rdfc_norm_bwd_apply_dxsum(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                         const float *beta, const void *res, const float *sums, void *dx, void *dres, int G, int P,
                                         int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                                         const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, float *dxsum_scratch,
                                         float *dbias_acc, void *stream) {
    if (int e = check_map("rdfc_norm_bwd_apply_dxsum", G, P, C, dtype, {lddz, ldx, lddx})) return e;
    RDFC_REQUIRE(dbeta_acc != nullptr || dgamma_acc == nullptr, "rdfc_norm_bwd_apply_dxsum: d gamma accumulates together with d beta");
    RDFC_REQUIRE(dbias_acc == nullptr || dxsum_scratch != nullptr, "rdfc_norm_bwd_apply_dxsum: the bias accumulator needs the scratch rows");
    bwd_apply_launch(dz, x, mean, rstd, gamma, beta, res, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx, lddr, dtype, act, slope, prelu, frozen_stats,
                     dbeta_acc, dgamma_acc, dxsum_scratch, dbias_acc, stream);
    RDFC_LAUNCH_CHECK("rdfc_norm_bwd_apply_dxsum");
    return 0;
}

extern "C" int rdfc_norm_bwd_apply(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                   const float *beta, const void *res, const float *sums, void *dx, void *dres, int G, int P,
                                   int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                                   const float *prelu, int frozen_stats, void *stream) {
    return rdfc_norm_bwd_apply_acc(dz, x, mean, rstd, gamma, beta, res, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx, lddr, dtype, act, slope,
                                   prelu, frozen_stats, nullptr, nullptr, stream);
}

// ---- composites: the launches of one layer's forward / backward behind ONE call (the step makes ~360 of them and is partly
//      launch-bound; a ctypes call costs the host ~4-5 us).  Same kernels, same order, same stream.
//      scratch (f32): forward [G * nchunk * 2 * C] partials; backward [G * nchunk * 3 * C] partials.
// statistics + finalize -> mean / rstd (+ running statistics); scratch: rdfc_norm_scratch_floats(G, nchunk, 2, C)
static int stats_final(const void *x, float *scratch, float *mean, float *rstd, int G, int P, int C, int ldx, int dtype, float eps, float *run_mean,
                       float *run_var, float momentum, void *stream) {
    if (int e = check_map("rdfc_norm_forward", G, P, C, dtype, {ldx})) return e;
    RDFC_REQUIRE(run_mean == nullptr || G == 1, "rdfc_norm_forward: running statistics need G == 1 (batch norm)");
    if (int e = rdfc_norm_stats(x, scratch, G, P, C, ldx, dtype, stream)) return e;
    return rdfc_norm_finalize(scratch, mean, rstd, G, P, C, dtype, eps, run_mean, run_var, momentum, stream);
}

extern "C" int rdfc_norm_stats_final(const void *x, float *scratch, float *mean, float *rstd, int G, int P, int C, int ld, int dtype, float eps,
                                     float *run_mean, float *run_var, float momentum, void *stream) {
    return stats_final(x, scratch, mean, rstd, G, P, C, ld, dtype, eps, run_mean, run_var, momentum, stream);
}

extern "C" int rdfc_norm_forward(const void *x, float *scratch, float *mean, float *rstd, const float *gamma, const float *beta,
                                 const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype, float eps,
                                 float *run_mean, float *run_var, float momentum, int act, float slope, const float *prelu,
                                 void *stream) {
    if (int e = stats_final(x, scratch, mean, rstd, G, P, C, ldx, dtype, eps, run_mean, run_var, momentum, stream)) return e;
    return rdfc_norm_apply(x, mean, rstd, gamma, beta, res, z, G, P, C, ldx, ldr, ldz, dtype, act, slope, prelu, stream);
}

extern "C" int rdfc_norm_forward_seq(const void *x, float *scratch, float *mean, float *rstd, const float *gamma, const float *beta,
                                     const void *res, void *z, int G, int P, int C, int ldx, int ldr, int ldz, int dtype, float eps,
                                     float *run_mean, float *run_var, const float *group_momentum, int act, float slope,
                                     const float *prelu, void *stream) {
    static const float zeros[MAXSEQ] = {0};
    if (int e = rdfc_norm_stats(x, scratch, G, P, C, ldx, dtype, stream)) return e;
    const int nchunk = rdfc_norm_nchunk(P, C, dtype);
    double *muvar = reinterpret_cast<double *>(scratch + part_floats(G, nchunk, 2, C));
    return rdfc_norm_finalize_apply_seq(scratch, nchunk, muvar, x, mean, rstd, gamma, beta, res, z, G, P, C, ldx, ldr, ldz, dtype, eps, run_mean,
                                        run_var, group_momentum ? group_momentum : zeros, act, slope, prelu, stream);
}

static int norm_backward_impl(const char *who, const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                              const float *beta, const void *res, float *scratch, float *sums, void *dx, void *dres, int G, int P,
                              int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                              const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, float *dxsum_scratch,
                              float *dbias_acc, void *stream) {
    if (int e = check_map(who, G, P, C, dtype, {lddz, ldx, lddx})) return e;
    RDFC_REQUIRE(dbeta_acc != nullptr || dgamma_acc == nullptr, "%s: d gamma accumulates together with d beta", who);
    RDFC_REQUIRE(dbias_acc == nullptr || dxsum_scratch != nullptr, "%s: the bias accumulator needs the scratch rows", who);
    ApplyArgs a{mean, rstd, gamma, beta, prelu, act, slope};
    {   // pass 1: the three sums (partial rows, then the fixed-order finalize launch)
        Map m = make_map(G, P, C, rdfc_vec(dtype));
        launch_bwd_reduce(dtype, m, a, dz, x, res, scratch, lddz, ldx, ldr, stream);
        RDFC_LAUNCH_CHECK(who);
        if (int e = rdfc_norm_bwd_finalize_acc(scratch, sums, G, P, C, dtype, nullptr, nullptr, stream)) return e;
    }
    // (the affine gradients are accumulated by the apply pass, which reads the sums anyway: the groups in group order)
    bwd_apply_launch(dz, x, mean, rstd, gamma, beta, res, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx, lddr, dtype, act, slope, prelu, frozen_stats,
                     dbeta_acc, dgamma_acc, dxsum_scratch, dbias_acc, stream);
    RDFC_LAUNCH_CHECK(who);
    return 0;
}

extern "C" int rdfc_norm_backward(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                  const float *beta, const void *res, float *scratch, float *sums, void *dx, void *dres, int G, int P,
                                  int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                                  const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, void *stream) {
    return norm_backward_impl("rdfc_norm_backward", dz, x, mean, rstd, gamma, beta, res, scratch, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx, lddr,
                              dtype, act, slope, prelu, frozen_stats, dbeta_acc, dgamma_acc, nullptr, nullptr, stream);
}
// the same + the gradient of a conv bias in front of the norm layer: dbias_acc[c] += sum of the stored dx (see bwd_apply_kernel: dxsum)
extern "C" int rdfc_norm_backward_dxsum(const void *dz, const void *x, const float *mean, const float *rstd, const float *gamma,
                                        const float *beta, const void *res, float *scratch, float *sums, void *dx, void *dres, int G, int P,
                                        int C, int lddz, int ldx, int ldr, int lddx, int lddr, int dtype, int act, float slope,
                                        const float *prelu, int frozen_stats, float *dbeta_acc, float *dgamma_acc, float *dxsum_scratch,
                                        float *dbias_acc, void *stream) {
    return norm_backward_impl("rdfc_norm_backward_dxsum", dz, x, mean, rstd, gamma, beta, res, scratch, sums, dx, dres, G, P, C, lddz, ldx, ldr, lddx,
                              lddr, dtype, act, slope, prelu, frozen_stats, dbeta_acc, dgamma_acc, dxsum_scratch, dbias_acc, stream);
}

extern "C" int rdfc_adain_apply(const void *x, const float *mean, const float *rstd, const void *gb, void *out, int N, int P,
                                int C, int ldx, int ldgb, int ldo, int dtype, void *stream) {
    if (int e = check_map("rdfc_adain_apply", N, P, C, dtype, {ldx, ldo})) return e;
    RDFC_REQUIRE(ldgb >= 2 * C, "rdfc_adain_apply: ldgb must cover 2C channels");
    Map m = make_map_stream(N, P, C, rdfc_vec(dtype));
    DISPATCH(dtype, hipLaunchKernelGGL(adain_apply_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m, mean,
                                       rstd, (const T *)x, (const T *)gb, (T *)out, ldx, ldgb, ldo));
    RDFC_LAUNCH_CHECK("rdfc_adain_apply");
    return 0;
}

extern "C" int rdfc_adain_bwd_reduce(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb,
                                     float *partial, int N, int P, int C, int lddo, int ldx, int ldgb, int dtype, void *stream) {
    if (int e = check_map("rdfc_adain_bwd_reduce", N, P, C, dtype, {lddo, ldx})) return e;
    Map m = make_map(N, P, C, rdfc_vec(dtype));
    DISPATCH(dtype, hipLaunchKernelGGL(adain_bwd_reduce_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m,
                                       mean, rstd, (const T *)dout, (const T *)x, (const T *)gb, partial, lddo, ldx, ldgb));
    RDFC_LAUNCH_CHECK("rdfc_adain_bwd_reduce");
    return 0;
}

extern "C" int rdfc_adain_bwd_apply(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb,
                                    const float *sums, void *dx, void *dgb, int N, int P, int C, int lddo, int ldx, int ldgb,
                                    int lddx, int lddgb, int dtype, void *stream) {
    if (int e = check_map("rdfc_adain_bwd_apply", N, P, C, dtype, {lddo, ldx, lddx})) return e;
    Map m = make_map_stream(N, P, C, rdfc_vec(dtype));
    DISPATCH(dtype, hipLaunchKernelGGL(adain_bwd_apply_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m,
                                       mean, rstd, (const T *)dout, (const T *)x, (const T *)gb, sums, (T *)dx, (T *)dgb, lddo,
                                       ldx, ldgb, lddx, lddgb));
    RDFC_LAUNCH_CHECK("rdfc_adain_bwd_apply");
    return 0;
}

// W-AdaIN backward in one call: the sums + their finalize + the apply pass.  scratch: rdfc_norm_scratch_floats(N, nchunk, 3, C)
extern "C" int rdfc_adain_backward(const void *dout, const void *x, const float *mean, const float *rstd, const void *gb, float *scratch,
                                   float *sums, void *dx, void *dgb, int N, int P, int C, int lddo, int ldx, int ldgb, int lddx, int lddgb,
                                   int dtype, void *stream) {
    if (int e = check_map("rdfc_adain_backward", N, P, C, dtype, {lddo, ldx, lddx})) return e;
    Map m = make_map(N, P, C, rdfc_vec(dtype));
    DISPATCH(dtype, hipLaunchKernelGGL(adain_bwd_reduce_kernel<T>, grid_for(m, Elem<T>::VEC), dim3(NT), 0, (hipStream_t)stream, m,
                                       mean, rstd, (const T *)dout, (const T *)x, (const T *)gb, scratch, lddo, ldx, ldgb));
    RDFC_LAUNCH_CHECK("rdfc_adain_backward (sums)");
    if (int e = rdfc_norm_bwd_finalize(scratch, sums, N, P, C, dtype, stream)) return e;
    return rdfc_adain_bwd_apply(dout, x, mean, rstd, gb, sums, dx, dgb, N, P, C, lddo, ldx, ldgb, lddx, lddgb, dtype, stream);
}

extern "C" int rdfc_channel_sum(const void *x, float *partial, float *out, int P, int C, int ld, int dtype, void *stream) {
    return rdfc_channel_sum_acc(x, partial, out, nullptr, P, C, ld, dtype, stream);
}

static int channel_sum_geometry(int P, int C, int ld, int dtype, bool &vecpath, int &chunk) {
    const int vec = rdfc_vec(dtype);
    vecpath = C % vec == 0 && ld % vec == 0;
    chunk = chunk_pixels(P, C, vecpath ? vec : 1);
    return cdiv(P, chunk);
}
/* floats of scratch rdfc_channel_sum[_acc] needs */
extern "C" int rdfc_channel_sum_scratch_floats(int P, int C, int ld, int dtype) {
    bool vp;
    int chunk;
    return rdfc_norm_scratch_floats(1, channel_sum_geometry(P, C, ld, dtype, vp, chunk), 1, C);
}

extern "C" int rdfc_channel_sum_acc(const void *x, float *partial, float *out, float *acc, int P, int C, int ld, int dtype, void *stream) {
    RDFC_REQUIRE(rdfc_dtype_ok(dtype), "rdfc_channel_sum: bad dtype");
    RDFC_REQUIRE(P > 0 && C > 0 && ld >= C, "rdfc_channel_sum: bad size");
    bool vecpath;
    int chunk;
    const int nchunk = channel_sum_geometry(P, C, ld, dtype, vecpath, chunk), vec = rdfc_vec(dtype);
    if (vecpath) {
        Map m = make_map(1, P, C, vec);
        DISPATCH(dtype, hipLaunchKernelGGL(chsum_vec_kernel<T>, grid_for(m, vec), dim3(NT), 0, (hipStream_t)stream, m, ld, (const T *)x, partial));
    } else {
        dim3 grid(nchunk, 1, cdiv(C, NT));
        DISPATCH(dtype, hipLaunchKernelGGL(chsum_kernel<T>, grid, dim3(NT), 0, (hipStream_t)stream, P, C, ld, chunk, (const T *)x, partial));
    }
    RDFC_FC_DISPATCH(pick_fc(nchunk), hipLaunchKernelGGL(sum_finalize_kernel<FC>, dim3(cdiv(C, FC), 1), dim3(FNT2), 0, (hipStream_t)stream, 1, C, 1, nchunk,
                                                             partial, out, acc, nullptr));
    RDFC_LAUNCH_CHECK("rdfc_channel_sum");
    return 0;
}

extern "C" int rdfc_act_bwd(const void *dy, const void *y, void *dx, int P, int C, int lddy, int ldy, int lddx, int dtype,
                            int y_is_f32, int act, float slope, void *stream) {
    RDFC_REQUIRE(rdfc_dtype_ok(dtype), "rdfc_act_bwd: bad dtype");
    const size_t total = (size_t)P * C;
    const int blocks = (int)min((size_t)2048, (total + 255) / 256);
    DISPATCH(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, total, C, lddy, ldy, lddx,
                                       y_is_f32, act, slope, (const T *)dy, y, (T *)dx));
    RDFC_LAUNCH_CHECK("rdfc_act_bwd");
    return 0;
}
