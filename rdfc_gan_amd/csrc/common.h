// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rdfc_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- error plumbing -----------------------------------------------------------
void rdfc_set_error(const char *fmt, ...);
// name of the kernel instantiation the last conv entry point launched (rdfc_last_kernel(), used by bench.py to
// attribute HIP-event times to one kernel symbol that can be matched against rocprofv3's kernel trace)
void rdfc_note_kernel(const char *fmt, ...);
template <typename T> inline const char *tname() { return sizeof(T) == 4 ? "float" : (__is_same(T, _Float16) ? "_Float16" : "__bf16"); }
// dtype helpers (RDFC_BF16 and RDFC_F16 share every layout: 2-byte elements, 8 per 16-byte vector)
static inline bool rdfc_is16(int dtype) { return dtype == RDFC_BF16 || dtype == RDFC_F16; }
static inline bool rdfc_dtype_ok(int dtype) { return dtype == RDFC_F32 || rdfc_is16(dtype); }
static inline int rdfc_vec(int dtype) { return rdfc_is16(dtype) ? 8 : 4; }
#define RDFC_DISPATCH(dtype, ...)                                      \
    if ((dtype) == RDFC_BF16) { using T = bf16_t; __VA_ARGS__; }      \
    else if ((dtype) == RDFC_F16) { using T = f16_t; __VA_ARGS__; }   \
    else { using T = float; __VA_ARGS__; }
#define RDFC_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            rdfc_set_error(__VA_ARGS__);   \
            return 1;                      \
        }                                  \
    } while (0)
#define RDFC_LAUNCH_CHECK(name)                                                   \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            rdfc_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- element traits -------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;   // elements per 16-byte access
    __device__ static float ld(const float *p) { return *p; }
    __device__ static void st(float *p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static float ld(const bf16_t *p) { return (float)*p; }
    __device__ static void st(bf16_t *p, float v) { *p = (bf16_t)v; }
};

template <> struct Elem<f16_t> {
    static constexpr int VEC = 8;
    __device__ static float ld(const f16_t *p) { return (float)*p; }
    __device__ static void st(f16_t *p, float v) { *p = (f16_t)v; }
};

// The LDS-DMA kernels move 2-byte elements as opaque bits (typed bf16_t) and only care about the format in three places:
// the MFMA instruction, float -> element in the epilogue, element -> float in a prologue.  DT = RDFC_BF16 | RDFC_F16.
template <int DT> __device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (DT == RDFC_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ bf16_t cvt16(float v) {
    if constexpr (DT == RDFC_F16) return __builtin_bit_cast(bf16_t, (f16_t)v);
    else return (bf16_t)v;
}
template <int DT> __device__ __forceinline__ float f32of(bf16_t x) {
    if constexpr (DT == RDFC_F16) return (float)__builtin_bit_cast(f16_t, x);
    else return (float)x;
}
template <typename T> struct DtOf { static constexpr int value = RDFC_BF16; };
template <> struct DtOf<f16_t> { static constexpr int value = RDFC_F16; };
template <> struct DtOf<float> { static constexpr int value = RDFC_F32; };

// 16-byte vector of T <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    f32x4 v;
    __device__ void load(const float *p) { v = *reinterpret_cast<const f32x4 *>(p); }
    __device__ void load_nt(const float *p) { v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)); }
    __device__ void store(float *p) const { *reinterpret_cast<f32x4 *>(p) = v; }
    __device__ void store_nt(float *p) const { __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); }
    __device__ void zero() { v = f32x4{0, 0, 0, 0}; }
    __device__ float get(int i) const { return v[i]; }
    __device__ void set(int i, float f) { v[i] = f; }
};
template <> struct Vec16<bf16_t> {
    bf16x8 v;
    __device__ void load(const bf16_t *p) { v = *reinterpret_cast<const bf16x8 *>(p); }
    __device__ void load_nt(const bf16_t *p) { v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(p)); }
    __device__ void store(bf16_t *p) const { *reinterpret_cast<bf16x8 *>(p) = v; }
    __device__ void store_nt(bf16_t *p) const { __builtin_nontemporal_store(v, reinterpret_cast<bf16x8 *>(p)); }
    __device__ void zero() {
        for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.0f;
    }
    __device__ float get(int i) const { return (float)v[i]; }
    __device__ void set(int i, float f) { v[i] = (bf16_t)f; }
};

template <> struct Vec16<f16_t> {
    f16x8 v;
    __device__ void load(const f16_t *p) { v = *reinterpret_cast<const f16x8 *>(p); }
    __device__ void load_nt(const f16_t *p) { v = __builtin_nontemporal_load(reinterpret_cast<const f16x8 *>(p)); }
    __device__ void store(f16_t *p) const { *reinterpret_cast<f16x8 *>(p) = v; }
    __device__ void store_nt(f16_t *p) const { __builtin_nontemporal_store(v, reinterpret_cast<f16x8 *>(p)); }
    __device__ void zero() {
        for (int i = 0; i < 8; ++i) v[i] = (f16_t)0.0f;
    }
    __device__ float get(int i) const { return (float)v[i]; }
    __device__ void set(int i, float f) { v[i] = (f16_t)f; }
};

// ---- activations ------------------------------------------------------------------
__device__ inline float act_fwd(float u, int act, float slope) {
    switch (act) {
        case RDFC_ACT_RELU: return u > 0.f ? u : 0.f;
        case RDFC_ACT_LRELU:
        case RDFC_ACT_PRELU: return u > 0.f ? u : u * slope;
        case RDFC_ACT_TANH: return tanhf(u);
        case RDFC_ACT_SIGMOID: return 1.f / (1.f + expf(-u));
        default: return u;
    }
}
// Runs body(actf) with the activation chosen ONCE (a per-element switch inside a 64-value epilogue costs more than
// its stores): piecewise-linear activations share u > 0 ? u : u * negative_slope.
template <typename Body> __device__ inline void with_act(int act, float slope, Body &&body) {
    if (act == RDFC_ACT_TANH) body([](float u) { return tanhf(u); });
    else if (act == RDFC_ACT_SIGMOID) body([](float u) { return 1.f / (1.f + expf(-u)); });
    else {
        const float ns = act == RDFC_ACT_NONE ? 1.f : (act == RDFC_ACT_RELU ? 0.f : slope);
        body([ns](float u) { return u > 0.f ? u : u * ns; });
    }
}
// derivative given the pre-activation u
__device__ inline float act_grad_pre(float u, int act, float slope) {
    switch (act) {
        case RDFC_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case RDFC_ACT_LRELU:
        case RDFC_ACT_PRELU: return u > 0.f ? 1.f : slope;
        case RDFC_ACT_TANH: { float t = tanhf(u); return 1.f - t * t; }
        case RDFC_ACT_SIGMOID: { float s = 1.f / (1.f + expf(-u)); return s * (1.f - s); }
        default: return 1.f;
    }
}
// derivative given the output y = act(u)
__device__ inline float act_grad_out(float y, int act, float slope) {
    switch (act) {
        case RDFC_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case RDFC_ACT_LRELU:
        case RDFC_ACT_PRELU: return y > 0.f ? 1.f : slope;
        case RDFC_ACT_TANH: return 1.f - y * y;
        case RDFC_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ---- whole-row epilogue stores (round 4) ------------------------------------------------------
// An MFMA accumulator lane holds 4 consecutive channels (8 bytes) of one pixel: stored directly, a wave's store instruction touches
// 32 bytes of each of 16 pixel rows.  A timing build of conv3x3_glds without its stores showed what that costs: 64 -> 64 @
// 16 x 228 x 304 138 -> 81 us, 128 -> 128 @ 114 x 152 103 -> 80, 256 -> 256 @ 57 x 76 101 -> 83 (the HBM time of those outputs is
// 28 / 14 / 14 us).  The conv kernels therefore pass a wave's 64 px x 64 ch tile through the wave's own 9 KB of the (by then dead)
// staging LDS -- 16 ds_write_b64 + 8 ds_read_b128 per lane -- and every store instruction writes 8 pixels x 128 contiguous bytes:
// a quarter of the write requests, each a whole line (64 -> 64: 139 -> 117 us, 128 -> 128: 105 -> 91, 256 -> 256: 97 -> 87).
// Usage: workgroup barrier (nobody reads the operand tiles any more) -> rs_put for every (pixel, 4-channel group) of the wave ->
// rs_flush.  Both touch the calling wave's tile only: no further workgroup barrier.
constexpr int RS_ROW = 144;                  // bytes per pixel row of the tile: 128 + 16 (bank spread, keeps rows 16-byte aligned)
constexpr int RS_TILE = 64 * RS_ROW;         // bytes per wave
template <int DT> __device__ __forceinline__ void rs_put(unsigned char *Ot, int px, int ch, const float (&v)[4]) {
    bf16x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = cvt16<DT>(v[r]);
    *reinterpret_cast<bf16x4 *>(Ot + px * RS_ROW + ch * 2) = o;
}
// y: the map's element 0 + the wave's first channel; nch: how many of the wave's 64 channels exist (a multiple of 8);
// addr(px) -> element offset of tile pixel px (0 .. 63) in the map, < 0: no such pixel
template <typename A> __device__ __forceinline__ void rs_flush(const unsigned char *Ot, int lane, bf16_t *y, int nch, A &&addr) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int chunk = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int px = 8 * k + (lane >> 3);
        const long off = addr(px);
        if (off < 0 || chunk * 8 >= nch) continue;
        *reinterpret_cast<u32x4 *>(y + off + chunk * 8) = *reinterpret_cast<const u32x4 *>(Ot + px * RS_ROW + chunk * 16);
    }
    __builtin_amdgcn_wave_barrier();         // (the tile may be refilled by the same wave)
}

// ---- the same with ADDENDS (round 6) -------------------------------------------------------------------------------------------------
// rdfc_conv2d_dgrad_add / _add2 add one or two maps to the input gradient in the epilogue (residual junctions of the BasicBlocks, skip
// connections behind the stride-2 stages, chained fan-outs).  Until round 6 a lane fetched its 4 channels of the addend -- 8 bytes --
// on the way INTO the staging tile: the same 32-byte-per-pixel access pattern the whole-row stores were built to get rid of, on the read
// side.  tools/addend_bench.py, alone, f16: 64 <- 64 @ 16 x 228 x 304 110 us plain, 170 with one addend (142 MB: 28 us at 5 TB/s);
// the transposed stride-2 kernel 64 <- 128 77 / 138 / 190 us with none / one / two.  Now the tile is staged in F32 (32 pixels at a
// time: 32 x 272 B fit the same 9 KB per wave) and the flush -- where a lane owns 8 consecutive channels of a pixel and eight lanes a
// whole 128-byte row -- loads the addends as 16-byte pieces, adds them in f32 in the old order (bias, activation, addend, second
// addend, ONE rounding) and stores: same bits as before.
constexpr int RSF_ROW = 272;                 // bytes per pixel row of the f32 tile: 64 channels x 4 + 16 (bank spread)
constexpr int RSF_PX = 32;                   // pixels per pass
static_assert(RSF_PX * RSF_ROW <= RS_TILE, "the f32 tile of a pass lives in the wave's share of the staging LDS");
__device__ __forceinline__ void rsf_put(unsigned char *Ot, int px, int ch, const float (&v)[4]) {
    *reinterpret_cast<f32x4 *>(Ot + px * RSF_ROW + ch * 4) = f32x4{v[0], v[1], v[2], v[3]};
}
// The addend pieces of a pass are FETCHED a pass ahead (rsf_fetch into registers: the first pass's while the workgroup waits at the
// barrier in front of the epilogue, the next one's while the current pass is flushed) -- with the loads issued inside the flush, between
// the staging tile's write and the store, every pass paid a memory round trip (64 <- 64 @ 16 x 228 x 304: 170 -> 158 us only).
struct RsfAdd {
    bf16x8 a[RSF_PX / 8];                    // a lane's four 16-byte pieces of one addend and pass
};
// add: element 0 of the addend + the wave's first channel; nch as in rs_flush; pix(px) -> PIXEL index of tile pixel px (0 .. 31) in the
// maps, < 0: no such pixel; ld: pixel stride (a multiple of 8: every piece is 16-byte aligned)
template <typename A> __device__ __forceinline__ void rsf_fetch(RsfAdd &r, int lane, const bf16_t *add, int ld, int nch, A &&pix) {
    const int chunk = lane & 7;
#pragma unroll
    for (int k = 0; k < RSF_PX / 8; ++k) {
        const long po = pix(8 * k + (lane >> 3));
#pragma unroll
        for (int i = 0; i < 8; ++i) r.a[k][i] = (bf16_t)0.0f;
        if (po >= 0 && chunk * 8 < nch) r.a[k] = *reinterpret_cast<const bf16x8 *>(add + po * ld + chunk * 8);
    }
}
// y: element 0 of the result + the wave's first channel; r1 (and r2, may be null): the fetched pieces of this pass
template <int DT, typename A>
__device__ __forceinline__ void rsf_flush(const unsigned char *Ot, int lane, bf16_t *y, int ld_out, const RsfAdd &r1, const RsfAdd *r2, int nch, A &&pix) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int chunk = lane & 7;
#pragma unroll
    for (int k = 0; k < RSF_PX / 8; ++k) {
        const int px = 8 * k + (lane >> 3);
        const long po = pix(px);
        if (po < 0 || chunk * 8 >= nch) continue;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(Ot + px * RSF_ROW + chunk * 32), hi = *reinterpret_cast<const f32x4 *>(Ot + px * RSF_ROW + chunk * 32 + 16);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += f32of<DT>(r1.a[k][r]);
        if (r2 != nullptr) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += f32of<DT>(r2->a[k][r]);
        }
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) o[r] = cvt16<DT>(v[r]);
        *reinterpret_cast<bf16x8 *>(y + po * ld_out + chunk * 8) = o;
    }
    __builtin_amdgcn_wave_barrier();         // (the tile may be refilled by the same wave)
}
// whether the addends of a launch qualify for it (else: the 8-byte pieces on the way in, as before)
static inline bool rsf_ok(const void *addend, int ld_add, const void *addend2 = nullptr, int ld_add2 = 0) {
    static const bool on = !(getenv("RDFC_RSF") != nullptr && atoi(getenv("RDFC_RSF")) == 0);      // RDFC_RSF=0: A/B against the 8-byte pieces
    return on && addend != nullptr && ld_add % 8 == 0 && (reinterpret_cast<size_t>(addend) & 15) == 0 &&
           (addend2 == nullptr || (ld_add2 % 8 == 0 && (reinterpret_cast<size_t>(addend2) & 15) == 0));
}

// ---- wave / block reductions ---------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// The wait in front of the raw barrier of an LDS-DMA pipeline step: all but this wave's N youngest vector-memory operations are done
// (its share of the DMAs the coming step reads has landed) AND every LDS read it has issued has RETURNED.  The second half is the
// write-after-read side of a ring: the slot (or patch buffer) read in step s is refilled by DMAs any wave issues as soon as it is past
// the barrier of step s + 1, so arriving at that barrier must mean "my reads of step s are complete", not "issued".  hipcc sinks
// the last fragment reads of a step and their MFMAs below the next barrier when that saves registers (its own lgkmcnt wait then
// lands after the barrier); with a CU's LDS queue full -- three workgroups of the kernel beside another stream's kernel, 480x640
// maps -- such a read returned bytes of a later step: rare wrong half-tiles that came and went run to run (found by
// tools/rerun_check.py in round 3; every LDS-DMA kernel of this library had the same one-step distance).
// (-DRDFC_TEST_NO_READ_WAIT builds the kernels as they were, to show that tests/test_fullsize_gpu.py and the 480x640 reproducibility
// test catch the hazard; never in a shipped library.)
#ifdef RDFC_TEST_NO_READ_WAIT
template <int N> __device__ inline void wait_vm_and_reads() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#else
template <int N> __device__ inline void wait_vm_and_reads() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
#endif
