// Column sums over the rows of a row-major buffer of per-block partial results, in ONE fixed association (the same bits every run).
// A block is 64 columns x ROWS_RG row groups (1024 threads).  A thread walks rows rg, rg + RG, rg + 2 RG, ... as FOUR independent
// chains, so four loads are in flight per thread instead of one dependent add per memory round trip; the chains are combined pairwise
// and the row groups in group order through LDS.  (The earlier 4-group / one-chain loops took 22 us for 400 rows x 384 columns --
// ~100 dependent round trips on 6 blocks; tools/rocpd_by_grid.py, round 3.)
#pragma once
#include <stdint.h>

constexpr int ROWS_RG = 16;

__device__ __forceinline__ float rows_sum_chains(const float *p, int64_t row_stride, int rg, int rows) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int r = rg;
    for (; r + 3 * ROWS_RG < rows; r += 4 * ROWS_RG) {
        s0 += p[(int64_t)r * row_stride];
        s1 += p[(int64_t)(r + ROWS_RG) * row_stride];
        s2 += p[(int64_t)(r + 2 * ROWS_RG) * row_stride];
        s3 += p[(int64_t)(r + 3 * ROWS_RG) * row_stride];
    }
    for (; r < rows; r += ROWS_RG) s0 += p[(int64_t)r * row_stride];
    return (s0 + s1) + (s2 + s3);
}

// sum of the ROWS_RG group values of column tl, in group order; valid in the threads of group 0
__device__ __forceinline__ float rows_sum_groups(float s, float (*red)[64], int rg, int tl) {
    red[rg][tl] = s;
    __syncthreads();
    float t = 0.0f;
    if (rg == 0) {
#pragma unroll
        for (int g = 0; g < ROWS_RG; ++g) t += red[g][tl];
    }
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------
// In-launch ordered reduction: the producer kernel's LAST block to arrive sums the per-block partial rows itself, instead of a second
// launch (partial_rows_sum_kernel: ~5 us of GPU time behind a launch boundary and ~12 us of host time, ~70 times per training step).
// Recipe of the CDNA programming guide ("in-launch split-K reduction"): every wave drains its partial stores, barrier, thread 0
// releases at agent scope (+ the restated wait), draws a ticket from a relaxed agent-scope counter; the block that draws the last
// ticket acquires at agent scope and reduces with plain loads.  The counter is zero before the first launch (the host allocates the
// scratch zero-filled) and the last arriver resets it: launches that share it are ordered on one stream.  The sum keeps ONE fixed
// association (NT / 64 row groups x 4 chains, groups combined in order), so results are the same bits every run -- but not the bits
// of partial_rows_sum_kernel's 16 x 4 association.
// `flag`: one LDS word; `red`: NT floats of LDS.  Returns after the reduction in the last block, immediately in all others.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void rows_sum_in_launch(unsigned *counter, unsigned nblocks, const float *part, float *out, int rows, int cols,
                                                   bool accumulate, float scale, float *red, unsigned *flag, float *out2 = nullptr,
                                                   int split = 0) {  // out2: columns >= split go to out2[c - split]
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = ticket == nblocks - 1 ? 1u : 0u;
        if (last) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
    constexpr int RG = NT / 64;
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    for (int c0 = 0; c0 < cols; c0 += 64) {
        const int c = c0 + tl;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if (c < cols) {
            const float *p = part + c;
            int r = rg;
            for (; r + 3 * RG < rows; r += 4 * RG) {
                s0 += p[(int64_t)r * cols];
                s1 += p[(int64_t)(r + RG) * cols];
                s2 += p[(int64_t)(r + 2 * RG) * cols];
                s3 += p[(int64_t)(r + 3 * RG) * cols];
            }
            for (; r < rows; r += RG) s0 += p[(int64_t)r * cols];
        }
        red[rg * 64 + tl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rg == 0 && c < cols) {
            float t = 0.0f;
#pragma unroll
            for (int g = 0; g < RG; ++g) t += red[g * 64 + tl];
            float *o = (out2 && c >= split) ? out2 + (c - split) : out + c;
            *o = accumulate ? *o + scale * t : scale * t;
        }
        __syncthreads();
    }
}
