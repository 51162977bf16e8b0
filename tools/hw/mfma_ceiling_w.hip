// GEMM-1 inner loops of the Winograd split-operand stack kernel as bare operand-stream models under the chip's power cap (round 6):
// what does a 96-frame tile on 16-wide matrix instructions buy over the shipped 64-frame tile?  One 8-wave block per CU, two-piece fp16
// operands, three products per term, pseudo-random operand bits, ~2 s per mode.
//   mode 0  v_mfma_f32_32x32x16_f16 from registers (bare pipe)
//   mode 1  v_mfma_f32_16x16x32_f16 from registers (bare pipe, 16-wide form)
//   mode 2  shipped tile: per 16-k step 4 x 1 KiB weight fragments from an L2-resident image + 2 ds_read_b128 -> 6 MFMAs 32x32x16 (wave tile 64 rows x 32 pairs)
//   mode 3  96-frame tile: per 32-k step 8 x 1 KiB weight fragments + 6 ds_read_b128 -> 36 MFMAs 16x16x32 (wave tile 64 rows x 48 pairs)
//   mode 4  128-frame tile: per 16-k step 4 x 1 KiB weight fragments + 4 ds_read_b128 -> 12 MFMAs 32x32x16 (wave tile 64 rows x 64 pairs)
//   modes 5 / 6  HiFi-GAN conv loop (4-wave blocks, two per CU, wave tile 32 rows x 128 frames) on the 32-wide / the 16-wide instruction
// prints executed TFLOP/s; socket power / clock are sampled by the caller (tools/power_probe.py).
//   hipcc --offload-arch=gfx950 -O3 -o build/exp/mfma_ceiling_w tools/hw/mfma_ceiling_w.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 rnd4(unsigned h) {
    return (u32x4){(h & 0x83ff83ffu) | 0x38003800u, ((h >> 3) & 0x83ff83ffu) | 0x34003400u, ((h >> 5) & 0x83ff83ffu) | 0x38003800u, ((h >> 7) & 0x83ff83ffu) | 0x34003400u};
}

constexpr int LDS_BYTES = 2 * 2 * 64 * 528;  // two planes x two pieces x 64 rows

template <int MODE>
__global__ void __launch_bounds__(512, 1) ceiling_kernel(const u32x4 *img, float *out, int iters, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 4; i += 512) {
        unsigned h = (unsigned)(i * 2654435761u + blockIdx.x * 40503u);
        reinterpret_cast<unsigned *>(lds)[i] = (h & 0x83ff83ffu) | 0x38003800u;
    }
    __syncthreads();
    const u32x4 *ap = img + lane;
    int blk = w;
    float s = 0.0f;
    if (MODE == 0 || MODE == 2 || MODE == 4) {
        constexpr int NB = MODE == 4 ? 2 : 1;   // 32-pair column blocks
        f32x16 acc[2][NB];
        for (int a = 0; a < 2; ++a)
            for (int n = 0; n < NB; ++n)
                for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.0f;
        u32x4 A[4][2][2], B[NB][2];  // A: ring of 4 k-steps (the kernel's prefetch distance)
        for (int u = 0; u < 4; ++u)
            for (int k = 0; k < 4; ++k) A[u][k >> 1][k & 1] = rnd4((unsigned)((lane + 64 * (k + 4 * u) + 7) * 2246822519u));
        for (int n = 0; n < NB; ++n) { B[n][0] = A[0][0][1]; B[n][1] = A[0][1][0]; }
        const unsigned boff = (unsigned)((lane & 31) * 528 + (lane >> 5) * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (MODE >= 2) {
#pragma unroll
                    for (int n = 0; n < NB; ++n)
#pragma unroll
                        for (int q = 0; q < 2; ++q) B[n][q] = *reinterpret_cast<const u32x4 *>(lds + boff + (n * 32) * 528 + q * 64 * 528 + 32 * u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int n = 0; n < NB; ++n)
                            acc[rb][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[u & 3][rb][p == 0]), __builtin_bit_cast(f16x8, B[n][p == 1]), acc[rb][n], 0, 0, 0);
                if (MODE >= 2) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        A[u & 3][k >> 1][k & 1] = ap[(size_t)blk * 64];
                        blk += 8;
                        if (blk >= nblk) blk -= nblk;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((it & 63) == 63)
                for (int a = 0; a < 2; ++a)
                    for (int n = 0; n < NB; ++n)
                        for (int r = 0; r < 16; ++r) acc[a][n][r] *= 1e-3f;
        }
        for (int a = 0; a < 2; ++a)
            for (int n = 0; n < NB; ++n)
                for (int r = 0; r < 16; ++r) s += acc[a][n][r];
    } else {
        f32x4 acc[4][3];
        for (int a = 0; a < 4; ++a)
            for (int n = 0; n < 3; ++n)
                for (int r = 0; r < 4; ++r) acc[a][n][r] = 0.0f;
        u32x4 A[2][4][2], B[3][2];  // A: ring of 2 k-steps of 32
        for (int u = 0; u < 2; ++u)
            for (int k = 0; k < 8; ++k) A[u][k >> 1][k & 1] = rnd4((unsigned)((lane + 64 * (k + 8 * u) + 7) * 2246822519u));
        for (int n = 0; n < 3; ++n) { B[n][0] = A[0][n][1]; B[n][1] = A[0][n + 1][0]; }
        const unsigned boff = (unsigned)((lane & 15) * 528 + (lane >> 4) * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 3) {
#pragma unroll
                    for (int n = 0; n < 3; ++n)
#pragma unroll
                        for (int q = 0; q < 2; ++q) B[n][q] = *reinterpret_cast<const u32x4 *>(lds + boff + (n * 16) * 528 + q * 64 * 528 + 64 * u);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int n = 0; n < 3; ++n)
                            acc[mb][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[u & 1][mb][p == 0]), __builtin_bit_cast(f16x8, B[n][p == 1]), acc[mb][n], 0, 0, 0);
                if (MODE == 3) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        A[u & 1][k >> 1][k & 1] = ap[(size_t)blk * 64];
                        blk += 8;
                        if (blk >= nblk) blk -= nblk;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((it & 63) == 63)
                for (int a = 0; a < 4; ++a)
                    for (int n = 0; n < 3; ++n)
                        for (int r = 0; r < 4; ++r) acc[a][n][r] *= 1e-3f;
        }
        for (int a = 0; a < 4; ++a)
            for (int n = 0; n < 3; ++n)
                for (int r = 0; r < 4; ++r) s += acc[a][n][r];
    }
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

// HiFi-GAN conv / ResBlock-pair loop models (wave tile 32 rows x 128 frames, two-piece operands): mode 5 = shipped, per k-step of 16 channels
// 2 weight fragments + 8 ds_read_b128 -> 12 MFMAs 32x32x16; mode 6 = the same tile on 16x16x32: per k-step of 32 channels 4 + 16 -> 48 MFMAs
template <int MODE>
__global__ void __launch_bounds__(256, 2) conv_ceiling_kernel(const u32x4 *img, float *out, int iters, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 256 * 96 / 4; i += 256) {
        unsigned h = (unsigned)(i * 2654435761u + blockIdx.x * 40503u);
        reinterpret_cast<unsigned *>(lds)[i] = (h & 0x83ff83ffu) | 0x38003800u;
    }
    __syncthreads();
    const u32x4 *ap = img + lane;
    int blk = w;
    float s = 0.0f;
    if (MODE == 5) {
        f32x16 acc[4];
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
        u32x4 A[2][2], B[4][2];
        for (int u = 0; u < 2; ++u)
            for (int q = 0; q < 2; ++q) A[u][q] = rnd4((unsigned)((lane + 64 * (q + 2 * u) + 7) * 2246822519u));
        const unsigned boff = (unsigned)((lane & 31) * 80 + (lane >> 5) * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int q = 0; q < 2; ++q) B[n][q] = *reinterpret_cast<const u32x4 *>(lds + boff + (n * 32 + (u & 3)) * 80 + q * 128 * 80 + 32 * (u & 1));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[u & 1][p == 0]), __builtin_bit_cast(f16x8, B[n][p == 1]), acc[n], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    A[u & 1][q] = ap[(size_t)blk * 64];
                    blk += 4;
                    if (blk >= nblk) blk -= nblk;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((it & 63) == 63)
                for (int n = 0; n < 4; ++n)
                    for (int r = 0; r < 16; ++r) acc[n][r] *= 1e-3f;
        }
        for (int n = 0; n < 4; ++n)
            for (int r = 0; r < 16; ++r) s += acc[n][r];
    } else {
        f32x4 acc[2][8];
        for (int m = 0; m < 2; ++m)
            for (int n = 0; n < 8; ++n)
                for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.0f;
        u32x4 A[2][2][2], B[8][2];
        for (int u = 0; u < 2; ++u)
            for (int m = 0; m < 2; ++m)
                for (int q = 0; q < 2; ++q) A[u][m][q] = rnd4((unsigned)((lane + 64 * (q + 2 * m + 4 * u) + 7) * 2246822519u));
        const unsigned boff = (unsigned)((lane & 15) * 96 + (lane >> 4) * 16);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int q = 0; q < 2; ++q) B[n][q] = *reinterpret_cast<const u32x4 *>(lds + boff + (n * 16 + u) * 96 + q * 128 * 96);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 8; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[u & 1][m][p == 0]), __builtin_bit_cast(f16x8, B[n][p == 1]), acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        A[u & 1][m][q] = ap[(size_t)blk * 64];
                        blk += 4;
                        if (blk >= nblk) blk -= nblk;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((it & 63) == 63)
                for (int m = 0; m < 2; ++m)
                    for (int n = 0; n < 8; ++n)
                        for (int r = 0; r < 4; ++r) acc[m][n][r] *= 1e-3f;
        }
        for (int m = 0; m < 2; ++m)
            for (int n = 0; n < 8; ++n)
                for (int r = 0; r < 4; ++r) s += acc[m][n][r];
    }
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE>
int run_conv(const u32x4 *img, float *out, int nblk, double seconds) {
    // executed flops per iteration and wave: mode 5: 8 x 12 x 32768; mode 6: 4 x 48 x 16384 (the same)
    const double flop_it = 8 * 12 * 32768.0;
    const int iters = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void *>(conv_ceiling_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 96);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(conv_ceiling_kernel<MODE>, dim3(512), dim3(256), 256 * 96, 0, img, out, iters, nblk);
    hipDeviceSynchronize();
    const int n = (int)(seconds / 3.2e-3) + 2;
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(conv_ceiling_kernel<MODE>, dim3(512), dim3(256), 256 * 96, 0, img, out, iters, nblk);
    hipEventRecord(e0, 0);
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(conv_ceiling_kernel<MODE>, dim3(512), dim3(256), 256 * 96, 0, img, out, iters, nblk);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)(n / 2) * 512 * 4 * (double)iters * flop_it;
    printf("mode %d: %.1f TFLOP/s executed (%.3f ms per launch, %d launches); conv loop model, wave tile 32 rows x 128 frames, %s\n", MODE,
           flop / (ms * 1e-3) / 1e12, ms / (n / 2), n / 2, MODE == 5 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16");
    fflush(stdout);
    return 0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
int run(const u32x4 *img, float *out, int nblk, double seconds) {
    // executed flops per iteration and wave: modes 0/2: 8 x 6 x 32768; mode 4: 8 x 12 x 32768; modes 1/3: 4 x 36 x 16384
    const double flop_it = (MODE == 0 || MODE == 2) ? 8 * 6 * 32768.0 : (MODE == 4 ? 8 * 12 * 32768.0 : 4 * 36 * 16384.0);
    const int iters = (int)(2000 * (8 * 6 * 32768.0) / flop_it);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ceiling_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), LDS_BYTES, 0, img, out, iters, nblk);
    CK(hipDeviceSynchronize());
    const int n = (int)(seconds / 3.2e-3) + 2;
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), LDS_BYTES, 0, img, out, iters, nblk);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), LDS_BYTES, 0, img, out, iters, nblk);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)(n / 2) * 256 * 8 * (double)iters * flop_it;
    const double a_bytes = (MODE == 2 || MODE == 4) ? 8.0 * 4096 : (MODE == 3 ? 4.0 * 8192 : 0.0);  // per iteration and wave
    printf("mode %d: %.1f TFLOP/s executed (%.3f ms per launch, %d launches); weight-fragment stream %.1f TB/s = %.1f B/clk/CU at 2.0 GHz\n", MODE,
           flop / (ms * 1e-3) / 1e12, ms / (n / 2), n / 2, (double)(n / 2) * 256 * 8 * iters * a_bytes / (ms * 1e-3) / 1e12,
           (double)(n / 2) * 8 * iters * a_bytes / (ms * 1e-3) / 2.0e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int nblk = 2048;  // 2 MiB: the Winograd image of one layer's GEMM 1 (L2 resident after the first pass)
    u32x4 *img; float *out;
    CK(hipMalloc(&img, (size_t)nblk * 1024));
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    std::vector<unsigned> h((size_t)nblk * 256);
    for (size_t i = 0; i < h.size(); ++i) { unsigned x = (unsigned)(i * 2654435761u); h[i] = (x & 0x83ff83ffu) | 0x38003800u; }
    CK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    if ((only < 0 || only == 0) && run<0>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 1) && run<1>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 2) && run<2>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 3) && run<3>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 4) && run<4>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 5) && run_conv<5>(img, out, nblk, seconds)) return 1;
    if ((only < 0 || only == 6) && run_conv<6>(img, out, nblk, seconds)) return 1;
    return 0;
}
