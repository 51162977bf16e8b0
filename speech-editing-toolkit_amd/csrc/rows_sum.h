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
