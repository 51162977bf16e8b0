// Attention building blocks for the CampNet rows: strided batched fp32 GEMM on MFMA
// (QK^T, PV and their four gradient products), masked row softmax fwd/bwd, position numbering, and the two small
// broadcast ops of the masked-mel input.  Everything works directly on the [B][C][T] activation layout of the rest of
// the library: a head is a channel slice, so Q/K/V/O are addressed through strides and never transposed or copied.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// C[b](m,n) = alpha * sum_k A[b](m,k) * B[b](k,n)  (+ C if accumulate), every operand addressed as
//   base + bo*outer + bi*inner + row*rs + col*cs    with b = bo * n_inner + bi.
// 256 threads = 2x2 waves, 64x64 tile, K chunks of 32 through LDS (As[k][m], Bs[k][n]: MFMA operand reads are
// conflict-free), v_mfma_f32_32x32x2_f32.  Loads pick the thread->element map by which operand stride is 1, so the
// global side is coalesced for either orientation; the host wrapper transposes the problem when C is column-major so
// the stores are coalesced too.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BM = 64, BN = 64, BK = 32, LDP = 65;

__device__ __forceinline__ void load_tile(float (*S)[LDP], const float *base, int64_t s_mn, int64_t s_k, int mn0, int k0,
                                          int MN, int K, int tid) {
    // tile element (mn, k) -> S[k][mn]; unconditional clamped loads + select (no `cond ? load : 0`)
    if (s_mn == 1 || s_k != 1) {  // mn fastest across lanes
        const int mn = tid & 63, kb = tid >> 6;
        const int gmn = mn0 + mn;
        const int64_t off_mn = (int64_t)min(gmn, MN - 1) * s_mn;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gk = k0 + kb + 4 * i;
            v[i] = base[off_mn + (int64_t)min(gk, K - 1) * s_k];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gk = k0 + kb + 4 * i;
            S[kb + 4 * i][mn] = (gmn < MN && gk < K) ? v[i] : 0.0f;
        }
    } else {  // k fastest across lanes (rows of 32 consecutive floats)
        const int k = tid & 31, mb = tid >> 5;
        const int gk = k0 + k;
        const int64_t off_k = (int64_t)min(gk, K - 1) * s_k;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gmn = mn0 + mb + 8 * i;
            v[i] = base[off_k + (int64_t)min(gmn, MN - 1) * s_mn];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gmn = mn0 + mb + 8 * i;
            S[k][mb + 8 * i] = (gmn < MN && gk < K) ? v[i] : 0.0f;
        }
    }
}

__global__ void __launch_bounds__(256) bmm_kernel(SetBmmArgs a) {
    __shared__ float As[BK][LDP];
    __shared__ float Bs[BK][LDP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1, half = lane >> 5, l31 = lane & 31;
    const int bz = blockIdx.z, bo = bz / a.n_inner, bi = bz - bo * a.n_inner;
    const float *A = a.A + bo * a.a_bo + bi * a.a_bi;
    const float *B = a.B + bo * a.b_bo + bi * a.b_bi;
    float *C = a.C + bo * a.c_bo + bi * a.c_bi;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc = {0};
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        load_tile(As, A, a.a_ms, a.a_ks, m0, k0, a.M, a.K, tid);
        load_tile(Bs, B, a.b_ns, a.b_ks, n0, k0, a.N, a.K, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) acc = mfma32(As[kk + half][32 * wm + l31], Bs[kk + half][32 * wn + l31], acc);
        __syncthreads();
    }
    const int gn = n0 + 32 * wn + l31;
    if (gn < a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + 32 * wm + mfma32_row(r, lane);
            if (gm < a.M) {
                float *dst = C + (int64_t)gm * a.c_ms + (int64_t)gn * a.c_ns;
                const float v = a.alpha * acc[r];
                *dst = a.accumulate ? *dst + v : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// row softmax in fp32 (utils/nn/seq_utils.py:21-22), one wave per row; optional key-padding mask kpm[b][col] (1 = pad)
// with b = row / rows_per_batch: masked logits are REPLACED by `fill` (-inf: torch's multi_head_attention_forward,
// -1e8: transformer.py:381-386) before the softmax.  A fully masked row with fill = -inf gives NaN, like torch.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void __launch_bounds__(256) softmax_rows_kernel(const float *x, const float *kpm, float *y, int64_t rows, int cols,
                                                          int64_t rows_per_batch, float fill) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *xr = x + row * cols;
    float *yr = y + row * cols;
    const float *mr = kpm ? kpm + (row / rows_per_batch) * cols : nullptr;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c];
        if (mr && mr[c] != 0.0f) v = fill;
        mx = fmaxf(mx, v);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c];
        if (mr && mr[c] != 0.0f) v = fill;
        sum += expf(v - mx);
    }
    sum = wave_sum(sum);
    for (int c = lane; c < cols; c += 64) {
        float v = xr[c];
        if (mr && mr[c] != 0.0f) v = fill;
        yr[c] = expf(v - mx) / sum;
    }
}

// ds = p * (dp - sum_k p*dp)
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float *p, const float *dp, float *ds, int64_t rows, int cols) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float *pr = p + row * cols, *dr = dp + row * cols;
    float dot = 0.0f;
    for (int c = lane; c < cols; c += 64) dot += pr[c] * dr[c];
    dot = wave_sum(dot);
    float *o = ds + row * cols;
    for (int c = lane; c < cols; c += 64) o[c] = pr[c] * (dr[c] - dot);
}

// pos[b][t] = (#non-zero entries in [0, t]) if entry t is non-zero else 0     (utils/nn/seq_utils.py:6-18, padding_idx 0)
// entries: int64 tokens[b][t] (tok != NULL) or fp32 x[b*x_bs + t] (first channel of a [B][C][T] tensor)
__global__ void __launch_bounds__(64) make_positions_kernel(const int64_t *tok, const float *x, int64_t x_bs, int64_t *pos, int T) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        bool nz = false;
        if (t < T) nz = tok ? tok[(int64_t)b * T + t] != 0 : x[(int64_t)b * x_bs + t] != 0.0f;
        const unsigned long long bal = __ballot(nz);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (t < T) pos[(int64_t)b * T + t] = nz ? (int64_t)(base + before + 1) : 0;
        base += __popcll(bal);
    }
}

// out[b][i] = mean_h p[b][h][i]
__global__ void __launch_bounds__(256) head_mean_kernel(const float *p, float *out, int B, int heads, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * n) return;
    const int64_t b = i / n, j = i - b * n;
    float s = 0.0f;
    for (int h = 0; h < heads; ++h) s += p[(b * heads + h) * n + j];
    out[i] = s / (float)heads;
}

// out[b][c][t] = x[b][c][t] * (1 - m[b][t]) + e[c] * m[b][t]        (campnet.py:56 on the [B][C][T] layout)
__global__ void __launch_bounds__(256) mask_fill_chan_kernel(const float *x, const float *e, const float *m, float *out, int B,
                                                            int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int64_t b = i / ((int64_t)T * C);
    const float mv = m[b * T + t];
    out[i] = x[i] * (1.0f - mv) + e[c] * mv;
}

// out[c] += sum_{b,t} d[b][c][t] * m[b][t]
__global__ void __launch_bounds__(256) masked_channel_sum_kernel(const float *d, const float *m, float *out, int B, int C, int T) {
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < (int64_t)B * T; i += 256) {
        const int64_t b = i / T, t = i - b * T;
        s += d[(b * C + c) * T + t] * m[i];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] += red[0] + red[1] + red[2] + red[3];
}

}  // namespace

extern "C" int64_t set_sizeof_bmm_args(void) { return (int64_t)sizeof(SetBmmArgs); }

extern "C" int set_bmm(const SetBmmArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_bmm");
    const SetBmmArgs &a = *args;
    SET_REQUIRE(a.A && a.B && a.C, "set_bmm");
    SET_REQUIRE(a.n_outer > 0 && a.n_inner > 0 && a.M > 0 && a.N > 0 && a.K > 0, "set_bmm");
    SET_REQUIRE((int64_t)a.n_outer * a.n_inner <= 65535, "set_bmm(batch)");
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.n_outer * a.n_inner);
    hipLaunchKernelGGL(bmm_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return set_check_launch("set_bmm");
}

extern "C" int set_softmax_rows(const float *x, const float *key_padding_mask, float *y, int64_t rows, int32_t cols,
                                int64_t rows_per_batch, float fill, void *stream) {
    SET_REQUIRE(x && y && rows > 0 && cols > 0 && rows_per_batch > 0, "set_softmax_rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(set_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, key_padding_mask,
                       y, rows, cols, rows_per_batch, fill);
    return set_check_launch("set_softmax_rows");
}

extern "C" int set_softmax_rows_bwd(const float *p, const float *dp, float *ds, int64_t rows, int32_t cols, void *stream) {
    SET_REQUIRE(p && dp && ds && rows > 0 && cols > 0, "set_softmax_rows_bwd");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(set_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream, p, dp, ds, rows,
                       cols);
    return set_check_launch("set_softmax_rows_bwd");
}

extern "C" int set_make_positions(const int64_t *tokens, const float *x, int64_t x_bs, int64_t *pos, int32_t B, int32_t T,
                                  void *stream) {
    SET_REQUIRE((tokens != nullptr) != (x != nullptr) && pos && B > 0 && T > 0, "set_make_positions");
    hipLaunchKernelGGL(make_positions_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, x, x_bs, pos, T);
    return set_check_launch("set_make_positions");
}

extern "C" int set_head_mean(const float *p, float *out, int32_t B, int32_t heads, int64_t n, void *stream) {
    SET_REQUIRE(p && out && B > 0 && heads > 0 && n > 0, "set_head_mean");
    hipLaunchKernelGGL(head_mean_kernel, dim3(set_blocks((int64_t)B * n, 256)), dim3(256), 0, (hipStream_t)stream, p, out, B,
                       heads, n);
    return set_check_launch("set_head_mean");
}

extern "C" int set_mask_fill_chan(const float *x, const float *e, const float *m, float *out, int32_t B, int32_t C, int32_t T,
                                  void *stream) {
    SET_REQUIRE(x && e && m && out && B > 0 && C > 0 && T > 0, "set_mask_fill_chan");
    hipLaunchKernelGGL(mask_fill_chan_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0, (hipStream_t)stream, x, e,
                       m, out, B, C, T);
    return set_check_launch("set_mask_fill_chan");
}

extern "C" int set_masked_channel_sum(const float *d, const float *m, float *out, int32_t B, int32_t C, int32_t T,
                                      void *stream) {
    SET_REQUIRE(d && m && out && B > 0 && C > 0 && T > 0, "set_masked_channel_sum");
    hipLaunchKernelGGL(masked_channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, d, m, out, B, C, T);
    return set_check_launch("set_masked_channel_sum");
}
