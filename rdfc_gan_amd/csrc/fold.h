// fold.h -- the second stage of a two-stage reduction INSIDE the launch that produced the partial rows.
//
// Rounds 1-3 ran every per-channel reduction as two launches: N workgroups write one partial row each, then a small "finalize" launch
// adds the rows in a fixed order (bitwise reproducible, no atomics on the values).  A train step had 213 of those launches; alone
// they take 5-14 us each, inside the step 20-30 us (they queue for a CU slot behind the other streams' convolutions), and every one
// is a dependent kernel boundary on the critical stream.  Here the LAST-ARRIVING workgroup does the finalize:
//
//   every workgroup   stores its row with write-through (sc1) stores -> each storing wave drains (s_waitcnt vmcnt(0)) ->
//                     workgroup barrier -> ONE lane takes a ticket from an agent-scope counter (relaxed atomic add);
//   the workgroup whose ticket is the last of its group of `gs` rows: ONE agent-scope acquire (+ wait + barrier), then it adds the
//                     group's rows in row order (f64) -- a level-2 row -- and takes a ticket from the group-of-groups counter;
//   the workgroup that is last there adds the level-2 rows in order and runs the caller's epilogue per channel.
//
// Two levels keep the serial tail short: <= ~64 rows per level whatever the grid (4 096 rows of 2 x 64 floats would be 2 MB for one
// workgroup to read; 64 rows are 32 KB).  Row order and association are fixed by (chunk index, gs) alone -> the result does not depend
// on which workgroup arrives last.  The protocol is cdna_hip_programming.md Guideline 16 in its counter form ("In-launch split-K
// reduction", the sc1-store variant): placement-independent, no release fence (write-through stores + drain), one acquire per episode;
// the reducer additionally reads with sc1 loads.  Counters are zero on entry and the last arriver zeroes them again, so a slot can be
// handed to a later launch without a memset (fold_ws.hip: a ring of slots per device).
#pragma once
#include "common.h"

namespace fold {

struct Args {
    unsigned *cnt;      // [G][1 + ngroups]: [0] = groups done, [1 + q] = rows of group q done; nullptr = no fold (the caller finalizes)
    double *l2;         // [G][ngroups][KC] level-2 rows (unused when ngroups == 1)
    int nchunk, gs, ngroups;
};

// rows per level-1 group: one level up to 64 rows, else ~sqrt(rows) per level
inline void shape(int nchunk, int &gs, int &ngroups) {
    if (nchunk <= 64) {
        gs = nchunk > 0 ? nchunk : 1;
        ngroups = 1;
        return;
    }
    int s = 8;
    while (s * s < nchunk) ++s;
    gs = s;
    ngroups = (nchunk + gs - 1) / gs;
}
inline int counters(int G, int nchunk) {
    int gs, ng;
    shape(nchunk, gs, ng);
    return G * (1 + ng);
}
// doubles of level-2 scratch
inline size_t l2_doubles(int G, int nchunk, int KC) {
    int gs, ng;
    shape(nchunk, gs, ng);
    return ng > 1 ? (size_t)G * ng * KC : 0;
}

__device__ __forceinline__ void st_f32(float *p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f32(const float *p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_f64(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_f64(const double *p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
}

// every wave of the workgroup: its stores are complete; then ONE lane takes a ticket.  -> the ticket (same value in every thread)
__device__ inline unsigned arrive_ticket(unsigned *counter, unsigned *lds_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) *lds_word = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *lds_word;
}
// The last arriver, once.  Every load of handed-off bytes below is an sc1 load (ld_f32 / ld_f64: served by L2 or beyond, never by this
// CU's L1), which is what makes them see the other workgroups' write-through stores; the agent-scope acquire in front of them drops
// this CU's L1 lines as well (the guide keeps it wherever a hand-off is not one of its measured table rows: here several workgroups
// share a CU).  It is issued by one lane and NOT waited for: the invalidate and the sc1 loads go down the same in-order vector-memory
// path, and no plain load of these bytes exists.
__device__ inline void acquire_all() {
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

constexpr int NTF = 256;          // threads of a folding workgroup (every kernel that folds launches 256)

// Sum over `nr` rows (row stride `stride` elements) of outputs o in [0, nout), all 256 threads busy: when nout <= 128 the rows are split
// into nsl = 256 / nout interleaved slices (thread (sl, o) adds rows sl, sl + nsl, ... in order, f64), the slices are then added in
// slice order through LDS; otherwise a thread owns outputs o, o + 256, ... and adds all rows.  The association depends on (nr, nout)
// only.  emit(o, sum) is called once per output by exactly one thread.  LD: element load (sc1), red: LDS scratch of 256 doubles.
template <typename E, typename LD, typename EMIT>
__device__ inline void sum_rows(const E *rows, size_t stride, int nr, int nout, double *red, LD &&ld, EMIT &&emit) {
    const int t = threadIdx.x;
    int nsl = 1;
    while (nsl * 2 * nout <= NTF && nsl * 2 <= nr) nsl *= 2;
    if (nsl > 1) {
        const int sl = t / nout, o = t - sl * nout;
        double a = 0;
        if (sl < nsl) {
            int r = sl;
            for (; r + 15 * nsl < nr; r += 16 * nsl) {
                E v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = ld(rows + (size_t)(r + q * nsl) * stride + o);
#pragma unroll
                for (int q = 0; q < 16; ++q) a += (double)v[q];
            }
            for (; r < nr; r += nsl) a += (double)ld(rows + (size_t)r * stride + o);
        }
        __syncthreads();
        if (sl < nsl) red[t] = a;
        __syncthreads();
        if (sl == 0) {
            double tot = 0;
            for (int q = 0; q < nsl; ++q) tot += red[q * nout + o];
            emit(o, tot);
        }
        return;
    }
    for (int o = t; o < nout; o += NTF) {
        double a = 0;
        int r = 0;
        for (; r + 16 <= nr; r += 16) {
            E v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = ld(rows + (size_t)(r + q) * stride + o);
#pragma unroll
            for (int q = 0; q < 16; ++q) a += (double)v[q];
        }
        for (; r < nr; ++r) a += (double)ld(rows + (size_t)r * stride + o);
        emit(o, a);
    }
}

// Called by EVERY workgroup (256 threads) of the launch after it has stored its row `chunk` of group g (rows: [nchunk][K][C] f32 of
// group g, stored with st_f32).  Returns true in the ONE workgroup per group that ended up with the totals; there fin(c, s) has been
// called once per channel c < C with s[k] = the sum over all rows of [k][c] (f64, fixed order).  tot: LDS scratch of K * C doubles
// when K * C <= 256 (the totals are exchanged through it), unused otherwise.
template <int K, typename F>
__device__ inline bool arrive(const Args &a, int g, int chunk, int C, const float *rows, F &&fin) {
    __shared__ unsigned s_ticket;
    __shared__ double s_red[NTF];
    __shared__ double s_tot[NTF];
    const int KC = K * C;
    const int grp = chunk / a.gs, r0 = grp * a.gs;
    const int nr = min(a.gs, a.nchunk - r0);
    unsigned *cnt = a.cnt + (size_t)g * (1 + a.ngroups);
    if (arrive_ticket(cnt + 1 + grp, &s_ticket) != (unsigned)(nr - 1)) return false;
    acquire_all();
    if (threadIdx.x == 0) __hip_atomic_store(cnt + 1 + grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every row of the group has arrived: free for a later launch
    const float *grows = rows + (size_t)r0 * KC;
    const bool small = KC <= NTF;
    auto finish_small = [&]() {                  // totals in s_tot[k * C + c]
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += NTF) {
            double s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] = s_tot[k * C + c];
            fin(c, s);
        }
    };
    if (a.ngroups == 1) {
        if (small) {
            sum_rows(grows, (size_t)KC, nr, KC, s_red, [](const float *p) { return ld_f32(p); }, [&](int o, double v) { s_tot[o] = v; });
            finish_small();
        } else {
            for (int c = threadIdx.x; c < C; c += NTF) {
                double s[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    double acc = 0;
                    int r = 0;
                    for (; r + 16 <= nr; r += 16) {
                        float v[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) v[q] = ld_f32(grows + (size_t)(r + q) * KC + k * C + c);
#pragma unroll
                        for (int q = 0; q < 16; ++q) acc += (double)v[q];
                    }
                    for (; r < nr; ++r) acc += (double)ld_f32(grows + (size_t)r * KC + k * C + c);
                    s[k] = acc;
                }
                fin(c, s);
            }
        }
        return true;
    }
    double *l2 = a.l2 + (size_t)g * a.ngroups * KC;
    sum_rows(grows, (size_t)KC, nr, KC, s_red, [](const float *p) { return ld_f32(p); }, [&](int o, double v) { st_f64(l2 + (size_t)grp * KC + o, v); });
    if (arrive_ticket(cnt, &s_ticket) != (unsigned)(a.ngroups - 1)) return false;
    acquire_all();
    if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (small) {
        sum_rows(l2, (size_t)KC, a.ngroups, KC, s_red, [](const double *p) { return ld_f64(p); }, [&](int o, double v) { s_tot[o] = v; });
        finish_small();
    } else {
        for (int c = threadIdx.x; c < C; c += NTF) {
            double s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double acc = 0;
                int r = 0;
                for (; r + 16 <= a.ngroups; r += 16) {
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v[q] = ld_f64(l2 + (size_t)(r + q) * KC + k * C + c);
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc += v[q];
                }
                for (; r < a.ngroups; ++r) acc += ld_f64(l2 + (size_t)r * KC + k * C + c);
                s[k] = acc;
            }
            fin(c, s);
        }
    }
    return true;
}

}  // namespace fold

// fold_ws.hip: n zeroed arrival counters for ONE launch on `st` (a slot of a per-device ring; the launch leaves them zeroed again).
// nullptr on failure (rdfc_last_error is set).
unsigned *rdfc_fold_counters(int n, hipStream_t st);
