// Hardware probe (gfx950): what happens to a raw-buffer store / load whose per-lane offset is beyond the descriptor's num_records?
// The epilogues mask lanes by giving them the offset 0x80000000 against num_records = 0x7fffffff (common.h: BUF_OOB).  This program
// allocates 3 GiB, stores through such offsets (with and without a scalar offset, 4- and 16-byte forms) and then looks at the words
// the stores would hit if they were NOT dropped (base + 2 GiB + soffset).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ rsrc_t make_rsrc(const void *p, int nrec) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, nrec, (int)0x00020000); }
__global__ void probe(float *base, unsigned soff, int nrec, float *loaded) {
    const rsrc_t r = make_rsrc(base, nrec);
    const unsigned lane = threadIdx.x;
    const unsigned voff = (lane & 1) ? 0x80000000u + 4u * lane : 4u * lane;  // odd lanes masked, even lanes store normally
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1.0f + lane), r, (int)voff, (int)soff, 0);
    const u32x4 v = {7u, 7u, 7u, 7u};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)((lane & 1) ? 0x80000000u + 16u * lane : 4096u + 16u * lane), (int)soff, 0);
    loaded[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)(0x80000000u + 4u * lane), (int)soff, 0));
}
// second part: EVERY lane masked at exactly 0x80000000 (the form the epilogues use), 4- and 16-byte stores, then the WHOLE allocation is
// searched for a non-zero word: a store that is not dropped but redirected anywhere inside the 3 GiB shows up
__global__ void probe_all_masked(float *base, unsigned soff) {
    const rsrc_t r = make_rsrc(base, 0x7fffffff);
    const u32x4 v = {9u, 9u, 9u, 9u};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)0x80000000u, (int)soff, 0);
    __builtin_amdgcn_raw_buffer_store_b32(9u, r, (int)0x80000000u, (int)soff, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)((threadIdx.x & 16) ? 0x80000000u : 0x80000000u + 16u * threadIdx.x), (int)soff, 0);
}
__global__ void count_nonzero(const unsigned *p, size_t n, unsigned long long *cnt) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != 0u;
    if (c) atomicAdd(cnt, c);
}
int main() {
    const size_t bytes = 3ull << 30;
    float *d = nullptr, *ld = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&ld, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int nrec : {0x7fffffff, (int)0xffffffff}) for (unsigned soff : {0u, 64u, 1u << 20}) {
        hipMemset(d, 0, bytes);
        // make the words behind base + 2 GiB recognisable for the load test
        float pat[64]; for (int i = 0; i < 64; ++i) pat[i] = 100.0f + i;
        hipMemcpy((char *)d + (2ull << 30) + soff, pat, sizeof(pat), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, soff, nrec, ld);
        hipDeviceSynchronize();
        float far[64], near_[64], l[64];
        hipMemcpy(far, (char *)d + (2ull << 30) + soff, sizeof(far), hipMemcpyDeviceToHost);
        hipMemcpy(near_, (char *)d + soff, sizeof(near_), hipMemcpyDeviceToHost);
        hipMemcpy(l, ld, sizeof(l), hipMemcpyDeviceToHost);
        int hit = 0, kept = 0, ldz = 0;
        for (int i = 0; i < 64; ++i) { hit += far[i] != 100.0f + i; kept += (i & 1) == 0 ? near_[i] == 1.0f + i : near_[i] == 0.0f; ldz += l[i] == 0.0f; }
        printf("num_records 0x%08x soffset %8u: words behind base + 2 GiB changed by the masked stores: %d of 64 | in-range lanes stored, masked lanes left alone: %d of 64 | out-of-range loads returning 0: %d of 64\n",
               (unsigned)nrec, soff, hit, kept, ldz);
    }
    unsigned long long *cnt = nullptr;
    hipMalloc(&cnt, 8);
    for (unsigned soff : {0u, 614400u, 1u << 30}) {
        hipMemset(d, 0, bytes); hipMemset(cnt, 0, 8);
        hipLaunchKernelGGL(probe_all_masked, dim3(64), dim3(256), 0, 0, d, soff);
        hipLaunchKernelGGL(count_nonzero, dim3(2048), dim3(256), 0, 0, (const unsigned *)d, bytes / 4, cnt);
        unsigned long long h = 0; hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost);
        printf("all lanes masked at 0x80000000, soffset %10u: non-zero words anywhere in the 3 GiB allocation afterwards: %llu\n", soff, h);
    }
    return 0;
}
