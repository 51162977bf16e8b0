// Hardware probe (gfx950), round 5: does a 16-byte buffer store with an SGPR soffset read its data registers late enough to see a VALU
// write issued right behind it?  ROCm 7.2's hazard recognizer pads `buffer_store_dwordx4 vdata, ..., soffset` against a following VALU write
// of vdata ONLY when soffset is not a register (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a register in
// the soffset field").  conv1d_mfma_v2_kernel's fast-path store ran into exactly that pair and its outputs differed from run to run.
//
// Every lane stores {1, 2, 3, 4} (as floats, tagged with its index) 16-byte-wise through a raw buffer descriptor; the instruction right
// behind the store overwrites data dword 2 with a poison value; `nops` s_nop wait states sit in between.  The store/overwrite pair is
// written in inline asm (physical registers v[4:7]), so the compiler's own padding plays no role.  A poisoned dword in memory = the store
// read the register AFTER the overwrite.  Many blocks x many stores per lane keep the memory pipeline backed up (the effect needs a busy
// store path: the training step saw a few dozen of 6.5 M elements).
//   hipcc --offload-arch=gfx950 -O2 -o tools/hw/store_data_hazard tools/hw/store_data_hazard.hip && tools/hw/store_data_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: SGPR soffset, no wait state | 1: + s_nop 0 | 2: + s_nop 1 | 3: soffset 0 (offset in the VGPR), no wait state | 4 - 6: the data produced
                     // right in front of the store by packed / packed + s_nop behind the store / plain multiplies
__global__ void __launch_bounds__(256) probe(float *out, int iters, int rows_per_block) {
    const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, (short)0, (int)0x7fffffff, (int)0x00020000);
    const unsigned lane_off = 16u * threadIdx.x;  // 256 lanes x 16 B = one 4 KiB row per store
    for (int it = 0; it < iters; ++it) {
        const unsigned row = (unsigned)blockIdx.x * rows_per_block + (unsigned)(it % rows_per_block);
        unsigned soff = row * 4096u;                       // wave-uniform: lives in an SGPR
        unsigned voff = lane_off;
        if (MODE == 3) { voff += soff; soff = 0u; }
        const f32x4 v = {1.0f + it, 2.0f + it, 3.0f + it, 4.0f + it};
        const float poison = -12345.0f;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 one2 = {1.0f, 1.0f};
        (void)one2;
        if (MODE == 3) {
            asm volatile("buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison) : "memory", "v6");
        } else if (MODE == 0) {
            asm volatile("buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff) : "memory", "v6");
        } else if (MODE == 1) {
            asm volatile("buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\ts_nop 0\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff) : "memory", "v6");
        } else if (MODE == 2) {
            asm volatile("buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\ts_nop 1\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff) : "memory", "v6");
        } else if (MODE == 4) {  // what conv1d_mfma_v2_kernel's epilogue had: packed fp32 ops produce the data right in front of the store
            asm volatile("v_pk_mul_f32 v[6:7], v[6:7], %5\n\tv_pk_mul_f32 v[4:5], v[4:5], %5\n\t"
                         "buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff), "v"(one2) : "memory", "v4", "v5", "v6", "v7");
        } else if (MODE == 5) {
            asm volatile("v_pk_mul_f32 v[6:7], v[6:7], %5\n\tv_pk_mul_f32 v[4:5], v[4:5], %5\n\t"
                         "buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\ts_nop 0\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff), "v"(one2) : "memory", "v4", "v5", "v6", "v7");
        } else {  // 6: plain (unpacked) multiplies in front of the store
            asm volatile("v_mul_f32 v6, v6, %5\n\tv_mul_f32 v4, v4, %5\n\t"
                         "buffer_store_dwordx4 v[4:7], %1, %2, %4 offen\n\tv_mov_b32 v6, %3"
                         :: "{v[4:7]}"(v), "v"(voff), "s"(r), "v"(poison), "s"(soff), "v"(1.0f) : "memory", "v4", "v6");
        }
    }
}
__global__ void count_poison(const float *p, size_t n, unsigned long long *cnt) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] == -12345.0f;
    if (c) atomicAdd(cnt, c);
}
int main() {
    const int blocks = 2048, rows = 64, iters = 4096;     // 2048 x 64 rows x 4 KiB = 512 MiB target, 8.4 M stores per lane-set
    const size_t bytes = (size_t)blocks * rows * 4096;
    float *d = nullptr; unsigned long long *cnt = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&cnt, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const char *names[7] = {"SGPR soffset, VALU write of data dword 2 right behind the store", "SGPR soffset, s_nop 0 (1 wait state) in between",
                            "SGPR soffset, s_nop 1 (2 wait states) in between", "soffset 0 (offset in the VGPR), no wait state",
                            "v_pk_mul_f32 producing the data, SGPR soffset, VALU write right behind", "v_pk_mul_f32 ..., SGPR soffset, s_nop 0 in between",
                            "v_mul_f32 producing the data, SGPR soffset, VALU write right behind"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            hipMemset(d, 0, bytes); hipMemset(cnt, 0, 8);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, d, iters, rows);
            hipLaunchKernelGGL(count_poison, dim3(2048), dim3(256), 0, 0, (const float *)d, bytes / 4, cnt);
            unsigned long long h = 0;
            hipError_t e = hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost);
            printf("%-72s: poisoned dwords in memory %llu (of %llu stored per pass x %d passes)%s\n", names[mode], h,
                   (unsigned long long)(bytes / 4), iters / rows, e == hipSuccess ? "" : "  [HIP error]");
        }
    return 0;
}
