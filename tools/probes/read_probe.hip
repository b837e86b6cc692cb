// Read-only streaming probe (MI355X): which shape of a reduction's read loop reaches the HBM rate the read+write passes get?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_probe tools/probes/read_probe.hip && /tmp/read_probe
// A 142 MB map (16 x 228 x 304 x 64 f16) is summed per thread; six maps are cycled so every launch reads HBM-cold data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: each workgroup owns one contiguous chunk (the normalisation kernels' layout), NL loads per trip, no pipelining
// MODE 1: the same with the next trip's loads issued before the current trip's adds
// MODE 2: workgroups stride over 4 KB blocks (block b, b + nwg, ...): every instant the chip reads one contiguous window
template <int MODE, int NL, bool NT_>
__global__ __launch_bounds__(256) void probe(const f32x4 *__restrict__ x, size_t n16, size_t chunk16, float *__restrict__ out) {
    f32x4 acc = {0, 0, 0, 0};
    auto ld = [&](size_t i) -> f32x4 { return NT_ ? __builtin_nontemporal_load(x + i) : x[i]; };
    if (MODE == 2) {
        const size_t nblk = n16 / 256;
        for (size_t b = blockIdx.x; b < nblk; b += (size_t)gridDim.x * NL) {
            f32x4 v[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const size_t bb = b + (size_t)k * gridDim.x;
                v[k] = bb < nblk ? ld(bb * 256 + threadIdx.x) : f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) acc += v[k];
        }
    } else {
        size_t i = blockIdx.x * chunk16 + threadIdx.x;
        const size_t end = min(n16, (blockIdx.x + 1) * chunk16);
        if (MODE == 0) {
            for (; i + (NL - 1) * 256 < end; i += NL * 256) {
                f32x4 v[NL];
#pragma unroll
                for (int k = 0; k < NL; ++k) v[k] = ld(i + k * 256);
#pragma unroll
                for (int k = 0; k < NL; ++k) acc += v[k];
            }
        } else {
            if (i + (NL - 1) * 256 < end) {
                f32x4 a[NL], b[NL];
#pragma unroll
                for (int k = 0; k < NL; ++k) a[k] = ld(i + k * 256);
                i += NL * 256;
                for (; i + (NL - 1) * 256 < end; i += NL * 256) {
#pragma unroll
                    for (int k = 0; k < NL; ++k) b[k] = ld(i + k * 256);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < NL; ++k) acc += a[k];
#pragma unroll
                    for (int k = 0; k < NL; ++k) a[k] = b[k];
                }
#pragma unroll
                for (int k = 0; k < NL; ++k) acc += a[k];
            }
        }
        for (; i < end; i += 256) acc += ld(i);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the statistics kernel's arithmetic on the chunked layout: 8 f16 -> f32, sum and sum of squares per element slot
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NL, int WORK>
__global__ __launch_bounds__(256) void probe_stats(const h8 *__restrict__ x, size_t n16, size_t chunk16, float *__restrict__ out) {
    float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = blockIdx.x * chunk16 + threadIdx.x;
    const size_t end = min(n16, (blockIdx.x + 1) * chunk16);
    for (; i + (NL - 1) * 256 < end; i += NL * 256) {
        h8 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = x[i + k * 256];
#pragma unroll
        for (int k = 0; k < NL; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[k][e];
                s0[e] += f;
                if (WORK >= 1) s1[e] += f * f;
            }
    }
    float t = 0;
    for (int e = 0; e < 8; ++e) t += s0[e] + s1[e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = t;
}
template <int NL, int WORK>
void run_stats(const char *name, std::vector<f32x4 *> &bufs, size_t n16, size_t chunk_bytes, float *out) {
    const size_t chunk16 = chunk_bytes / 16;
    const int nwg = (int)((n16 + chunk16 - 1) / chunk16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 6; ++w) hipLaunchKernelGGL((probe_stats<NL, WORK>), dim3(nwg), dim3(256), 0, 0, (const h8 *)bufs[w % bufs.size()], n16, chunk16, out);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe_stats<NL, WORK>), dim3(nwg), dim3(256), 0, 0, (const h8 *)bufs[r % bufs.size()], n16, chunk16, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-34s wg %6d  %7.1f us  %5.2f TB/s\n", name, nwg, us, n16 * 16.0 / us / 1e6);
}

template <int MODE, int NL, bool NT_>
void run(const char *name, std::vector<f32x4 *> &bufs, size_t n16, size_t chunk_bytes, int nwg_fixed, float *out) {
    const size_t chunk16 = chunk_bytes / 16;
    const int nwg = MODE == 2 ? nwg_fixed : (int)((n16 + chunk16 - 1) / chunk16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 6; ++w) hipLaunchKernelGGL((probe<MODE, NL, NT_>), dim3(nwg), dim3(256), 0, 0, bufs[w % bufs.size()], n16, chunk16, out);
    const int reps = 30;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, NL, NT_>), dim3(nwg), dim3(256), 0, 0, bufs[r % bufs.size()], n16, chunk16, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-34s wg %6d  %7.1f us  %5.2f TB/s\n", name, nwg, us, n16 * 16.0 / us / 1e6);
}

int main() {
    const size_t bytes = (size_t)16 * 228 * 304 * 64 * 2, n16 = bytes / 16;
    std::vector<f32x4 *> bufs(6);
    for (auto &b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
    float *out;
    hipMalloc(&out, 64 << 20);
    char nm[96];
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {                       // random f16 bit patterns (finite) instead of zeros
            std::vector<unsigned short> h(bytes / 2);
            unsigned r = 12345u;
            for (auto &v : h) { r = r * 1664525u + 1013904223u; v = (unsigned short)((r >> 16) & 0xbbff); }
            for (auto &b : bufs) hipMemcpy(b, h.data(), bytes, hipMemcpyHostToDevice);
        }
        printf("-- %s data\n", pass ? "random" : "zero");
        run<0, 4, false>("chunk 128 KB, 4 loads (plain sum)", bufs, n16, 128 << 10, 0, out);
        run_stats<4, 0>("chunk 128 KB, f16 sum", bufs, n16, 128 << 10, out);
        run_stats<4, 1>("chunk 128 KB, f16 sum + sumsq", bufs, n16, 128 << 10, out);
        run_stats<4, 1>("chunk  32 KB, f16 sum + sumsq", bufs, n16, 32 << 10, out);
    }
    for (size_t kb : {32, 64, 128, 256}) {
        snprintf(nm, sizeof nm, "chunk %3zu KB, 4 loads", kb); run<0, 4, false>(nm, bufs, n16, kb << 10, 0, out);
        snprintf(nm, sizeof nm, "chunk %3zu KB, 8 loads", kb); run<0, 8, false>(nm, bufs, n16, kb << 10, 0, out);
        snprintf(nm, sizeof nm, "chunk %3zu KB, 4 loads pipelined", kb); run<1, 4, false>(nm, bufs, n16, kb << 10, 0, out);
        snprintf(nm, sizeof nm, "chunk %3zu KB, 4 loads nt", kb); run<0, 4, true>(nm, bufs, n16, kb << 10, 0, out);
    }
    for (int nwg : {1024, 2048, 4096}) {
        snprintf(nm, sizeof nm, "strided 4 KB blocks, 4 loads"); run<2, 4, false>(nm, bufs, n16, 0, nwg, out);
        snprintf(nm, sizeof nm, "strided 4 KB blocks, 8 loads"); run<2, 8, false>(nm, bufs, n16, 0, nwg, out);
        snprintf(nm, sizeof nm, "strided 4 KB blocks, 4 loads nt"); run<2, 4, true>(nm, bufs, n16, 0, nwg, out);
    }
    return 0;
}
