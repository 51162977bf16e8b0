// Shared by csrc/diffnet.hip (diffnet_boundary_x2_kernel: the step boundary as a launch of its own) and csrc/diffnet_x3.hip (the whole-loop
// kernel, where the step boundary is a task of the persistent queue): Philox4x32-10 + Box-Muller, and the two-piece fp16 GEMM of the step
// boundary (skip projection, output head, next step's input projection; diffnet.py:118-120,128-131, spec_denoiser.py:86-101).
#pragma once
#include "common.h"

namespace {

constexpr int BX_DC = 256;             // residual channels
// ---- Philox4x32-10 (counter based) + Box-Muller ------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void randn4(uint64_t seed, uint64_t ctr, float out[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // (0,1] uniforms, Box-Muller
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[1] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u1, &s0, &c0);
    sincosf(6.28318530717958647692f * u3, &s1, &c1);
    out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

typedef _Float16 bx_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned bx_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bx_u32x2 __attribute__((ext_vector_type(2)));
constexpr int BX_XR = BX_DC * 2 + 16;     // bytes per row of the s / h tiles [frame][256]
constexpr int BX_PR = 96 * 2 + 16;     // ... of the x' tile [frame][96]
constexpr int BX_PIECE = 64 * BX_XR;   // one piece of a [64][256] tile

__device__ __forceinline__ void bx_split(float v, unsigned short &p0, unsigned short &p1) {
    const _Float16 h0 = (_Float16)v;
    p0 = __builtin_bit_cast(unsigned short, h0);
    p1 = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h0));
}
__device__ __forceinline__ f32x16 bx_mma(bx_u32x4 a, bx_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bx_f16x8, a), __builtin_bit_cast(bx_f16x8, b), c, 0, 0, 0);
}
// acc[NRB][2] += W[32 (rb0 + i) .. ][16 ks ..] * B over nks k-steps; image [rb32][ng16][piece][lane][8] (K = 1 tap);
// B piece q of (ks, cb) at lds + q * piece_bytes + bfrag(ks, cb)
template <int NRB, typename BF>
__device__ __forceinline__ void bx_gemm(f32x16 (&acc)[NRB][2], rsrc_t img, unsigned lane16, int rb0, int ng16, int nks,
                                        const unsigned char *lds, unsigned piece_bytes, BF bfrag) {
    // Weight fragments straight from the packed image through a ring of PF k-steps, the k-step order pinned (round 4, as in gemm_x3 /
    // sx_gemm):  ds_read B(k + 1) | the MFMAs of k-step k straight from their ring slot | refill of that slot | sched_barrier.  Rounds 2-3
    // had `Ac = A[p]; A[p] = load; mma(Ac)` with a ring of 2: 150 of the kernel's 312 MFMAs sat right behind an s_waitcnt vmcnt(0 / 1).
    // Same products in the same order per accumulator: bit-identical.
    constexpr int PF = 4;
    auto a_load = [&](int ks, int i, int q) {
        return (bx_u32x4)__builtin_amdgcn_raw_buffer_load_b128(img, (int)lane16, (int)((((rb0 + i) * ng16 + ks) * 2 + q) * 1024), 0);
    };
    bx_u32x4 A[PF][NRB][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int i = 0; i < NRB; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) A[p][i][q] = a_load(min(p, nks - 1), i, q);
    bx_u32x4 Bv[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const unsigned bo = bfrag(0, cb);
        Bv[cb][0] = *reinterpret_cast<const bx_u32x4 *>(lds + bo);
        Bv[cb][1] = *reinterpret_cast<const bx_u32x4 *>(lds + piece_bytes + bo);
    }
#pragma unroll 1
    for (int kb = 0; kb < nks; kb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ks = kb + p;
            if (ks < nks) {  // (nks need not be a multiple of PF: the in-projection has 5 k-steps)
                bx_u32x4 Bn[2][2];
                const int kq = min(ks + 1, nks - 1);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const unsigned bo = bfrag(kq, cb);
                    Bn[cb][0] = *reinterpret_cast<const bx_u32x4 *>(lds + bo);
                    Bn[cb][1] = *reinterpret_cast<const bx_u32x4 *>(lds + piece_bytes + bo);
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)  // a1 b0, a0 b1, a0 b0
#pragma unroll
                    for (int i = 0; i < NRB; ++i)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) acc[i][cb] = bx_mma(A[p][i][t == 0 ? 1 : 0], Bv[cb][t == 1 ? 1 : 0], acc[i][cb]);
                const int kn = min(ks + PF, nks - 1);
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    A[p][i][0] = a_load(kn, i, 0);
                    A[p][i][1] = a_load(kn, i, 1);
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    Bv[cb][0] = Bn[cb][0];
                    Bv[cb][1] = Bn[cb][1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}


}  // namespace
