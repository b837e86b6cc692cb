#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const bf16x2 *a, const bf16x2 *b, float *o) {
    float acc = 1.0f;
    acc = __builtin_amdgcn_fdot2_f32_bf16(a[threadIdx.x], b[threadIdx.x], acc, false);
    o[threadIdx.x] = acc;
}
int main() {
    bf16x2 ha[4], hb[4]; float ho[4];
    for (int i = 0; i < 4; ++i) { ha[i][0] = (__bf16)(1.0f + i); ha[i][1] = (__bf16)(0.5f); hb[i][0] = (__bf16)2.0f; hb[i][1] = (__bf16)(-4.0f + i); }
    bf16x2 *a, *b; float *o;
    hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&o, sizeof ho);
    hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 4>>>(a, b, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; ++i) printf("%d: got %f expect %f\n", i, ho[i], 1.0f + (1.0f + i) * 2.0f + 0.5f * (-4.0f + i));
    return 0;
}
