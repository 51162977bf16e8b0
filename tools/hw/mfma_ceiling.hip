// MFMA ceiling under the chip's power budget at the fragment intensity of the split-operand stack kernel (round 4).
// One 8-wave block per CU (2 waves per SIMD, as the kernel), v_mfma_f32_32x32x16_f16 on 4 independent accumulators per wave, random
// (non-zero, non-repeating) operand bits, ~2 s per mode so that the clocks settle at the power cap:
//   mode 0  operands in registers: the bare matrix-pipe ceiling at this chip's 1.4 kW cap
//   mode 1  B fragments from LDS: one ds_read_b128 per 3 MFMAs (the kernel: 4 fragment reads per 12 MFMAs of a k-step)
//   mode 2  mode 1 + A fragments streamed from an L2-resident image: one 1 KiB buffer_load per wave per 3 MFMAs (the kernel: 4 per 12)
// prints achieved TFLOP/s (executed MFMA flops), time, and -- read by tools/power_probe.py-style hwmon sampling in the caller -- nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hw/mfma_ceiling tools/hw/mfma_ceiling.hip && gpurun -- tools/hw/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512, 1) ceiling_kernel(const u32x4 *img, float *out, int iters, int img_kib) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // LDS tile: 64 rows x 528 bytes of pseudo-random fp16 bits (|x| < 1)
    for (int i = tid; i < 64 * 528 / 4; i += 512) {
        unsigned h = (unsigned)(i * 2654435761u + blockIdx.x * 40503u);
        h = (h & 0x83ff83ffu) | 0x38003800u;  // sign + mantissa, exponent of 0.5..1
        reinterpret_cast<unsigned *>(lds)[i] = h;
    }
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    u32x4 A[4], B[2];
    for (int k = 0; k < 4; ++k) {
        unsigned h = (unsigned)((lane + 64 * k + 7) * 2246822519u);
        A[k] = (u32x4){(h & 0x83ff83ffu) | 0x38003800u, ((h >> 3) & 0x83ff83ffu) | 0x34003400u, ((h >> 5) & 0x83ff83ffu) | 0x38003800u, ((h >> 7) & 0x83ff83ffu) | 0x34003400u};
    }
    B[0] = A[1]; B[1] = A[2];
    const unsigned boff = (unsigned)((lane & 31) * 528 + (lane >> 5) * 16);
    const u32x4 *ap = img + (size_t)w * 64 + lane;  // wave w: fragment blocks w, w + 8, ... of the image (1 KiB each)
    const int nblk = img_kib;                        // number of 1 KiB fragment blocks
    int blk = w;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 4 "k-steps" of 12 MFMAs
            if (MODE >= 1) {
                B[0] = *reinterpret_cast<const u32x4 *>(lds + boff + 32 * u);
                B[1] = *reinterpret_cast<const u32x4 *>(lds + boff + 32 * 528 + 32 * u);
            }
            if (MODE >= 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    A[k] = __builtin_nontemporal_load(ap + (size_t)blk * 64);
                    blk += 8;
                    if (blk >= nblk) blk -= nblk;
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[(a + p) & 3]), __builtin_bit_cast(f16x8, B[a & 1]), acc[a], 0, 0, 0);
        }
        if ((it & 63) == 63)  // keep the sums finite
            for (int a = 0; a < 4; ++a)
                for (int r = 0; r < 16; ++r) acc[a][r] *= 1e-3f;
    }
    float s = 0.0f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
int run(const u32x4 *img, float *out, int img_kib, double seconds) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(ceiling_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 528));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), 64 * 528, 0, img, out, iters, img_kib);
    CK(hipDeviceSynchronize());
    // one launch is ~2000 x 48 MFMAs x 32 cycles x 2 waves per SIMD ~ 3 ms: launch enough for `seconds`, time the second half
    const int n = (int)(seconds / 3.2e-3) + 2;
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), 64 * 528, 0, img, out, iters, img_kib);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < n / 2; ++i) hipLaunchKernelGGL(ceiling_kernel<MODE>, dim3(256), dim3(512), 64 * 528, 0, img, out, iters, img_kib);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)(n / 2) * 256 * 8 * (double)iters * 48 * 32768.0;
    const double mfma_per_simd = (double)iters * 48 * 2;
    printf("mode %d: %.1f TFLOP/s executed (%.3f ms per launch, %d launches); matrix pipe busy at 2.4 GHz: %.1f %%; implied clock if the pipe never idles: %.2f GHz\n",
           MODE, flop / (ms * 1e-3) / 1e12, ms / (n / 2), n / 2, 100.0 * mfma_per_simd * 32 / (ms / (n / 2) * 1e-3 * 2.4e9),
           mfma_per_simd * 32 / (ms / (n / 2) * 1e-3) / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int img_kib = 2400;  // 2.4 MB: the two-piece image of one layer (L2 resident after the first pass)
    u32x4 *img; float *out;
    CK(hipMalloc(&img, (size_t)img_kib * 1024));
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    std::vector<unsigned> h((size_t)img_kib * 256);
    for (size_t i = 0; i < h.size(); ++i) { unsigned x = (unsigned)(i * 2654435761u); h[i] = (x & 0x83ff83ffu) | 0x38003800u; }
    CK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    if ((only < 0 || only == 0) && run<0>(img, out, img_kib, seconds)) return 1;
    if ((only < 0 || only == 1) && run<1>(img, out, img_kib, seconds)) return 1;
    if ((only < 0 || only == 2) && run<2>(img, out, img_kib, seconds)) return 1;
    return 0;
}
