// Micro-benchmark: throughput of LDS atomic adds by type (f32 / u32 / u64) and of global f32 atomics, in the access pattern of
// nlspn.hip's scatter (64 lanes -> 64 consecutive cells, 72 adds per thread).   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int NT = 256, WN = 1152, ADDS = 72, REP = 8;
template <int MODE>
__global__ __launch_bounds__(NT) void k(float *gout, unsigned long long *sink) {
    __shared__ unsigned long long buf[WN];
    for (int e = threadIdx.x; e < WN; e += NT) buf[e] = 0;
    __syncthreads();
    float *f = reinterpret_cast<float *>(buf);
    unsigned *u = reinterpret_cast<unsigned *>(buf);
    const int base = threadIdx.x;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int a = 0; a < ADDS; ++a) {
            const int idx = (base + a * 73 + r * 5) % (WN - 80);
            const float v = 1e-3f * (a + 1);
            if (MODE == 0) atomicAdd(f + idx, v);
            else if (MODE == 1) atomicAdd(u + idx, (unsigned)(a + 1));
            else if (MODE == 2) atomicAdd(buf + idx, (unsigned long long)(a + 1) << 20);
            else atomicAdd(gout + (size_t)blockIdx.x % 64 * 4096 + idx, v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = buf[5] + buf[100];
}
template <int MODE> float run(float *g, unsigned long long *s, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NT), 0, 0, g, s);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NT), 0, 0, g, s);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    const int blocks = 2320;
    float *g; unsigned long long *s;
    hipMalloc(&g, 64 * 4096 * 4 + 8192); hipMemset(g, 0, 64 * 4096 * 4 + 8192);
    hipMalloc(&s, blocks * 8);
    const double adds = (double)blocks * NT * ADDS * REP;
    const char *names[] = {"lds f32", "lds u32", "lds u64", "global f32"};
    float ms[4] = {run<0>(g, s, blocks), run<1>(g, s, blocks), run<2>(g, s, blocks), run<3>(g, s, blocks)};
    for (int i = 0; i < 4; ++i) printf("%-10s %8.1f us  %.2f lane-adds/clk/CU (at 2.1 GHz)\n", names[i], ms[i] * 1e3, adds / (ms[i] * 1e-3) / 256 / 2.1e9);
    return 0;
}
