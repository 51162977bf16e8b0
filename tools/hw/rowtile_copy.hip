// Micro-benchmark: HBM throughput of the access pattern of the 32-channel ResBlock-pair blocks (a block owns C rows x 256 frames of a
// [B][C][T] fp32 tensor, channel stride T) as a plain copy, with one dword per lane (what the kernels do: lane = frame) against
// 16 bytes per lane (lane = 4 consecutive frames).   build: hipcc --offload-arch=gfx950 -O3 tools/hw/rowtile_copy.hip -o tools/hw/rowtile_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int C>
__global__ void __launch_bounds__(256) copy_dword(const float *x, float *y, int T) {
    const int tid = threadIdx.x, f = blockIdx.x * 256 + (tid & 127), cg = tid >> 7;  // 2 passes of 128 frames, 2 channel groups
    const float *xb = x + (size_t)blockIdx.y * C * T;
    float *yb = y + (size_t)blockIdx.y * C * T;
    float v[2][C / 2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < C / 2; ++k) v[p][k] = xb[(size_t)(cg * (C / 2) + k) * T + f + 128 * p];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < C / 2; ++k) yb[(size_t)(cg * (C / 2) + k) * T + f + 128 * p] = v[p][k] + 1.0f;
}

template <int C>
__global__ void __launch_bounds__(256) copy_x4(const float *x, float *y, int T) {
    const int tid = threadIdx.x, f = blockIdx.x * 256 + 4 * (tid & 63), r0 = tid >> 6;  // lane = 4 frames, wave = rows r0, r0 + 4, ...
    const float *xb = x + (size_t)blockIdx.y * C * T;
    float *yb = y + (size_t)blockIdx.y * C * T;
    float4 v[C / 4];
#pragma unroll
    for (int k = 0; k < C / 4; ++k) v[k] = *reinterpret_cast<const float4 *>(xb + (size_t)(r0 + 4 * k) * T + f);
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        float4 o = v[k];
        o.x += 1.0f; o.y += 1.0f; o.z += 1.0f; o.w += 1.0f;
        *reinterpret_cast<float4 *>(yb + (size_t)(r0 + 4 * k) * T + f) = o;
    }
}

template <typename F>
static float timed(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const int B = 64;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int C = cfg ? 64 : 32, T = cfg ? 102400 : 204800;
        const size_t n = (size_t)B * C * T;
        float *x, *y;
        hipMalloc(&x, n * 4); hipMalloc(&y, n * 4);
        hipMemset(x, 0, n * 4);
        dim3 grid(T / 256, B);
        float a, b;
        if (C == 32) {
            a = timed([&] { hipLaunchKernelGGL(copy_dword<32>, grid, dim3(256), 0, 0, x, y, T); });
            b = timed([&] { hipLaunchKernelGGL(copy_x4<32>, grid, dim3(256), 0, 0, x, y, T); });
        } else {
            a = timed([&] { hipLaunchKernelGGL(copy_dword<64>, grid, dim3(256), 0, 0, x, y, T); });
            b = timed([&] { hipLaunchKernelGGL(copy_x4<64>, grid, dim3(256), 0, 0, x, y, T); });
        }
        printf("C=%d T=%d B=%d (%.2f GB read + %.2f GB written): dword per lane %.3f ms = %.2f TB/s | 16 B per lane %.3f ms = %.2f TB/s\n", C, T, B,
               n * 4 / 1e9, n * 4 / 1e9, a, 2 * n * 4 / a / 1e9, b, 2 * n * 4 / b / 1e9);
        hipFree(x); hipFree(y);
    }
    return 0;
}
