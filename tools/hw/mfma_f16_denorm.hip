// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/t tools/hw/mfma_f16_denorm.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    // lane l holds A[row l&31][k = 8*(l>>5)+e]; put a_val at k=0 of every row, b_val at k=0 of every column
    if ((threadIdx.x >> 5) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float cases[][2] = {{1.0f, 2.0f}, {9.5367431640625e-7f /*2^-20, fp16 subnormal*/, 1024.0f}, {3.0517578125e-5f /*2^-15 subnormal*/, 1.0f},
                              {6.103515625e-5f /*2^-14 smallest normal*/, 1.0f}, {5.9604644775390625e-8f /*2^-24 smallest subnormal*/, 16384.0f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%.10g b=%g -> mfma %.10g (exact %.10g)\n", c[0], c[1], h, (double)c[0] * c[1]);
    }
    return 0;
}
