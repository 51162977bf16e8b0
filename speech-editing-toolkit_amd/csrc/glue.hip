// Conditioner glue kernels: LayerNorm over channels, embeddings, masks, alignment gather,
// integer duration / pitch-bin bookkeeping, transposes.  All are tiny, HBM/latency-bound and
// coalesced along T (the contiguous axis of the [B][C][T] layout).  Reference citations are in
// include/set_amd.h next to each prototype.
#include <mutex>

#include "common.h"
#include "pitch_edges.h"

// ---- LayerNorm over C of [B][C][T]: block = 32 frames x 8 channel groups (loads coalesced along t, 128 B per group
//      row), partial sums combined through LDS.  A thread owns ceil(C/8) channels of one frame; up to 32 of them
//      (C <= 256, every LayerNorm of the model) are fetched ONCE, as one batch of clamped loads, and stay in registers
//      for the mean, the variance and the output pass; wider inputs re-read.  (History: one thread per frame walking
//      all C channels took 95 us for [16][192][800]; 64 frames x 4 groups with three dependent passes 27 us.)
constexpr int LN_FT = 32, LN_CG = 8, LN_RC = 32;
__device__ __forceinline__ float ln_block_sum(float v, float (*red)[LN_FT], int cg, int tl) {
    __syncthreads();  // previous use of red[] is over
    red[cg][tl] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < LN_CG; ++g) s += red[g][tl];
    return s;
}

__global__ void __launch_bounds__(256) layernorm_ch_kernel(const float *x, const float *gamma, const float *beta,
                                                           const float *mask, float *out, int B, int C, int T,
                                                           float eps) {
    __shared__ float red[LN_CG][LN_FT];
    const int tl = threadIdx.x % LN_FT, cg = threadIdx.x / LN_FT;
    const int b = blockIdx.y, t = blockIdx.x * LN_FT + tl;
    const bool valid = t < T;
    const int tc = valid ? t : T - 1;
    const int cq = (C + LN_CG - 1) / LN_CG, c0 = cg * cq, c1 = min(C, c0 + cq);
    const float *xp = x + (int64_t)b * C * T + tc;
    float *op = out + (int64_t)b * C * T + tc;
    const float m = mask ? mask[(int64_t)b * T + tc] : 1.0f;
    if (cq <= LN_RC) {  // block-uniform
        float xv[LN_RC];
#pragma unroll
        for (int i = 0; i < LN_RC; ++i) xv[i] = xp[(int64_t)min(c0 + i, C - 1) * T];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_RC; ++i) s += c0 + i < c1 ? xv[i] : 0.0f;
        const float mean = ln_block_sum(s, red, cg, tl) / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < LN_RC; ++i) {
            const float d = c0 + i < c1 ? xv[i] - mean : 0.0f;
            q = fmaf(d, d, q);
        }
        const float rstd = 1.0f / sqrtf(ln_block_sum(q, red, cg, tl) / (float)C + eps);
        float gm[LN_RC], bt[LN_RC];
#pragma unroll
        for (int i = 0; i < LN_RC; ++i) {
            gm[i] = gamma[min(c0 + i, C - 1)];
            bt[i] = beta[min(c0 + i, C - 1)];
        }
        if (!valid) return;
#pragma unroll
        for (int i = 0; i < LN_RC; ++i) {
            const float v = (xv[i] - mean) * rstd * gm[i] + bt[i];
            if (c0 + i < c1) op[(int64_t)(c0 + i) * T] = mask ? v * m : v;
        }
        return;
    }
    float s = 0.0f;
    for (int c = c0; c < c1; ++c) s += xp[(int64_t)c * T];
    const float mean = ln_block_sum(s, red, cg, tl) / (float)C;
    float q = 0.0f;
    for (int c = c0; c < c1; ++c) {
        const float d = xp[(int64_t)c * T] - mean;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(ln_block_sum(q, red, cg, tl) / (float)C + eps);
    if (!valid) return;
    for (int c = c0; c < c1; ++c) {
        const float v = (xp[(int64_t)c * T] - mean) * rstd * gamma[c] + beta[c];
        op[(int64_t)c * T] = mask ? v * m : v;
    }
}
extern "C" int set_layernorm_ch(const float *x, const float *gamma, const float *beta, const float *mask, float *out,
                                int32_t B, int32_t C, int32_t T, float eps, void *stream) {
    SET_REQUIRE(x && gamma && beta && out && B > 0 && C > 0 && T > 0 && B <= 65535, "set_layernorm_ch");
    hipLaunchKernelGGL(layernorm_ch_kernel, dim3((T + LN_FT - 1) / LN_FT, B), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, mask, out, B, C, T, eps);
    return set_check_launch("set_layernorm_ch");
}

// ---- embedding lookup written channel-major --------------------------------------------------
__global__ void __launch_bounds__(256) embedding_bct_kernel(const int64_t *idx, const float *table, float *out, int B,
                                                            int T, int C, int n_rows, float scale, const float *scale_dev,
                                                            int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((int64_t)T * C));
    int64_t row = idx[(int64_t)b * T + t];
    row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
    const float sc = scale_dev ? scale_dev[0] : scale;
    const float v = sc * table[row * C + c];
    out[i] = accumulate ? out[i] + v : v;
}
extern "C" int set_embedding_bct(const int64_t *idx, const float *table, float *out, int32_t B, int32_t T, int32_t C,
                                 int32_t n_rows, float scale, int32_t accumulate, void *stream) {
    SET_REQUIRE(idx && table && out && B > 0 && T > 0 && C > 0 && n_rows > 0, "set_embedding_bct");
    hipLaunchKernelGGL(embedding_bct_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                       (hipStream_t)stream, idx, table, out, B, T, C, n_rows, scale, (const float *)nullptr, accumulate);
    return set_check_launch("set_embedding_bct");
}
extern "C" int set_embedding_bct_dev_scale(const int64_t *idx, const float *table, float *out, int32_t B, int32_t T, int32_t C,
                                           int32_t n_rows, const float *scale_dev, int32_t accumulate, void *stream) {
    SET_REQUIRE(idx && table && out && scale_dev && B > 0 && T > 0 && C > 0 && n_rows > 0, "set_embedding_bct_dev_scale");
    hipLaunchKernelGGL(embedding_bct_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                       (hipStream_t)stream, idx, table, out, B, T, C, n_rows, 1.0f, scale_dev, accumulate);
    return set_check_launch("set_embedding_bct_dev_scale");
}

// ---- masks ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) abs_sum_mask_kernel(const float *x, float *mask, int B, int C, int T) {
    __shared__ float red[4][64];
    const int tl = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int b = blockIdx.y, t = blockIdx.x * 64 + tl;
    const int tc = t < T ? t : T - 1;
    const int cq = (C + 3) / 4, c0 = cg * cq, c1 = min(C, c0 + cq);
    const float *xp = x + (int64_t)b * C * T + tc;
    float s = 0.0f;
    for (int c = c0; c < c1; ++c) s += fabsf(xp[(int64_t)c * T]);
    red[cg][tl] = s;
    __syncthreads();
    if (cg == 0 && t < T) mask[(int64_t)b * T + t] = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) > 0.0f ? 1.0f : 0.0f;
}
extern "C" int set_abs_sum_mask(const float *x, float *mask, int32_t B, int32_t C, int32_t T, void *stream) {
    SET_REQUIRE(x && mask && B > 0 && C > 0 && T > 0 && B <= 65535, "set_abs_sum_mask");
    hipLaunchKernelGGL(abs_sum_mask_kernel, dim3((T + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, x, mask, B, C, T);
    return set_check_launch("set_abs_sum_mask");
}

__global__ void __launch_bounds__(256) index_mask_kernel(const int64_t *idx, float *mask, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) mask[i] = idx[i] > 0 ? 1.0f : 0.0f;
}
extern "C" int set_index_mask(const int64_t *idx, float *mask, int64_t n, void *stream) {
    SET_REQUIRE(idx && mask && n > 0, "set_index_mask");
    hipLaunchKernelGGL(index_mask_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, idx, mask, n);
    return set_check_launch("set_index_mask");
}

// ---- expand_states: gather encoder states by mel2ph (bit-exact integer indexing) -----------------
__global__ void __launch_bounds__(256) expand_states_kernel(const float *enc, const int64_t *mel2ph, float *out, int B,
                                                            int C, int T_txt, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((int64_t)T * C));
    const int64_t m = mel2ph[(int64_t)b * T + t];
    float v = 0.0f;
    if (m > 0 && m <= T_txt) v = enc[((int64_t)b * C + c) * T_txt + (m - 1)];
    out[i] = v;
}
extern "C" int set_expand_states(const float *enc, const int64_t *mel2ph, float *out, int32_t B, int32_t C,
                                 int32_t T_txt, int32_t T, void *stream) {
    SET_REQUIRE(enc && mel2ph && out && B > 0 && C > 0 && T_txt > 0 && T > 0, "set_expand_states");
    hipLaunchKernelGGL(expand_states_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                       (hipStream_t)stream, enc, mel2ph, out, B, C, T_txt, T);
    return set_check_launch("set_expand_states");
}

__global__ void __launch_bounds__(256) add_chan_mask_kernel(const float *x, const float *add, const float *mask,
                                                            float *out, int B, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((int64_t)T * C));
    float v = x[i];
    if (add) v += add[(int64_t)b * C + c];
    if (mask) v *= mask[(int64_t)b * T + t];
    out[i] = v;
}
// four consecutive frames per thread (T % 4 == 0, 16-byte aligned operands): same arithmetic per element
__global__ void __launch_bounds__(256) add_chan_mask_vec4_kernel(const float *x, const float *add, const float *mask, float *out,
                                                                 int64_t n4, int C, int T) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const int64_t i = i4 * 4;
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((int64_t)T * C));
    f32x4 v = *reinterpret_cast<const f32x4 *>(x + i);
    if (add) {
        const float a = add[(int64_t)b * C + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += a;
    }
    if (mask) {
        const f32x4 m = *reinterpret_cast<const f32x4 *>(mask + (int64_t)b * T + t);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= m[e];
    }
    *reinterpret_cast<f32x4 *>(out + i) = v;
}
extern "C" int set_add_chan_mask(const float *x, const float *add, const float *mask, float *out, int32_t B, int32_t C,
                                 int32_t T, void *stream) {
    SET_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "set_add_chan_mask");
    if (T % 4 == 0 && set_aligned16(x, out, mask))
        hipLaunchKernelGGL(add_chan_mask_vec4_kernel, dim3(set_blocks((int64_t)B * C * T / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, add,
                           mask, out, (int64_t)B * C * T / 4, C, T);
    else
    hipLaunchKernelGGL(add_chan_mask_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                       (hipStream_t)stream, x, add, mask, out, B, C, T);
    return set_check_launch("set_add_chan_mask");
}

// ---- masked ground-truth durations: one block per utterance, LDS histogram ---------------------
__global__ void __launch_bounds__(256) masked_dur_kernel(const int64_t *mel2ph, const float *tmask, const int64_t *txt,
                                                         int64_t *out, int T, int T_txt) {
    extern __shared__ int hist[];
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j <= T_txt; j += 256) hist[j] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        // mel2ph * (1 - mask).long()   (fs.py:137: the float mask is truncated to int64 first)
        const int64_t keep = (int64_t)(1.0f - tmask[(int64_t)b * T + t]);
        const int64_t m = mel2ph[(int64_t)b * T + t] * keep;
        if (m >= 0 && m <= T_txt) atomicAdd(&hist[(int)m], 1);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < T_txt; j += 256)
        out[(int64_t)b * T_txt + j] = txt[(int64_t)b * T_txt + j] != 0 ? (int64_t)hist[j + 1] : 0;
}
extern "C" int set_masked_dur(const int64_t *mel2ph, const float *tmask, const int64_t *txt, int64_t *out, int32_t B,
                              int32_t T, int32_t T_txt, void *stream) {
    SET_REQUIRE(mel2ph && tmask && txt && out && B > 0 && T > 0 && T_txt > 0, "set_masked_dur");
    SET_REQUIRE(T_txt < 16000, "set_masked_dur");
    hipLaunchKernelGGL(masked_dur_kernel, dim3(B), dim3(256), (size_t)(T_txt + 1) * sizeof(int), (hipStream_t)stream,
                       mel2ph, tmask, txt, out, T, T_txt);
    return set_check_launch("set_masked_dur");
}

// ---- f0 -> denormalised Hz -> coarse mel-scale bins ---------------------------------------------
__global__ void __launch_bounds__(256) pitch_coarse_kernel(const float *f0_in, const float *uv_in, const float *tmask,
                                                           const int64_t *mel2ph_pad, int uv_from_logit,
                                                           float *f0_denorm, int64_t *coarse, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float f0 = f0_in[i];
    float uv = uv_in ? uv_in[i] : 0.0f;
    if (tmask) {
        const float k = 1.0f - tmask[i];
        f0 *= k;
        uv *= k;
    }
    // denorm_f0 (pitch/utils.py:71-82): 2**f0, clamp, zero where uv>0 / padding
    float f = exp2f(f0);
    f = fminf(fmaxf(f, 50.0f), 900.0f);
    const bool unvoiced = uv_in && (uv_from_logit ? (uv_in[i] > 0.0f) : (uv > 0.0f));
    if (unvoiced) f = 0.0f;
    if (mel2ph_pad && mel2ph_pad[i] == 0) f = 0.0f;
    if (f0_denorm) f0_denorm[i] = f;
    if (coarse) {
        // f0_to_coarse(denorm_f0(f0)) (pitch/utils.py:17-28,71-82) is a monotone step function of the fp32 input whose edges depend on
        // torch-CPU's (not correctly rounded) pow and log: the runs of equal bins were read off the reference on every fp32 in [5, 10.5]
        // (oracle/make_pitch_edges.py -> pitch_edges.h); a binary search over the run starts is bit-exact for every input.  Unvoiced /
        // padded frames: f = 0 -> mel 0 -> bin 1, the bin of run 0.
        int k = 0;
        if (f != 0.0f && f0 > 5.0f) {
            const unsigned u = __float_as_uint(fminf(f0, 10.5f));
            int lo = 0, hi = PITCH_RUNS - 1;  // largest k with PITCH_RUN_BITS[k] <= u (positive floats order like their bit patterns)
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (PITCH_RUN_BITS[mid] <= u) lo = mid; else hi = mid - 1;
            }
            k = lo;
        }
        coarse[i] = (int64_t)PITCH_RUN_BIN[k];
    }
}
extern "C" int set_pitch_coarse(const float *f0_in, const float *uv_in, const float *tmask, const int64_t *mel2ph_pad,
                                int32_t uv_from_logit, float *f0_denorm, int64_t *coarse, int64_t n, void *stream) {
    SET_REQUIRE(f0_in && n > 0 && (f0_denorm || coarse), "set_pitch_coarse");
    hipLaunchKernelGGL(pitch_coarse_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, f0_in, uv_in,
                       tmask, mel2ph_pad, uv_from_logit, f0_denorm, coarse, n);
    return set_check_launch("set_pitch_coarse");
}

// ---- transposes through a padded LDS tile -------------------------------------------------------
// in: [B][R][S] -> out: [B][S][R]
__global__ void __launch_bounds__(256) transpose_kernel(const float *in, float *out, int R, int S) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, s0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float *ib = in + (int64_t)b * R * S;
    float *ob = out + (int64_t)b * R * S;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, s = s0 + tx;
        tile[k][tx] = (r < R && s < S) ? ib[(int64_t)r * S + s] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int s = s0 + k, r = r0 + tx;
        if (r < R && s < S) ob[(int64_t)s * R + r] = tile[tx][k];
    }
}
static int launch_transpose(const float *in, float *out, int B, int R, int S, void *stream, const char *what) {
    SET_REQUIRE(in && out && B > 0 && R > 0 && S > 0, what);
    dim3 grid((S + 31) / 32, (R + 31) / 32, B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, R, S);
    return set_check_launch(what);
}
extern "C" int set_transpose_btc_to_bct(const float *in, float *out, int32_t B, int32_t T, int32_t C, void *stream) {
    return launch_transpose(in, out, B, T, C, stream, "set_transpose_btc_to_bct");
}
extern "C" int set_transpose_bct_to_btc(const float *in, float *out, int32_t B, int32_t C, int32_t T, void *stream) {
    return launch_transpose(in, out, B, C, T, stream, "set_transpose_bct_to_btc");
}

// ---- small elementwise helpers --------------------------------------------------------------------
__global__ void __launch_bounds__(256) sum_scale_kernel(const float *a, const float *b, const float *c, float *out,
                                                        float s, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = a[i];
    if (b) v += b[i];
    if (c) v += c[i];
    out[i] = v / s;
}
// NB: `s` is a DIVISOR here (hifigan.py:137 `xs / self.num_kernels`), named scale in the header for brevity.
extern "C" int set_sum_scale(const float *a, const float *b, const float *c, float *out, float s, int64_t n,
                             void *stream) {
    SET_REQUIRE(a && out && n > 0 && s != 0.0f, "set_sum_scale");
    hipLaunchKernelGGL(sum_scale_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, s,
                       n);
    return set_check_launch("set_sum_scale");
}

__global__ void __launch_bounds__(256) blend_mask_kernel(const float *a, const float *b, const float *m, float *out,
                                                         int64_t n, int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mm = m[i / inner];
    out[i] = a[i] * (1.0f - mm) + b[i] * mm;
}
extern "C" int set_blend_mask(const float *a, const float *b, const float *m, float *out, int64_t n, int64_t inner,
                              void *stream) {
    SET_REQUIRE(a && b && m && out && n > 0 && inner > 0, "set_blend_mask");
    hipLaunchKernelGGL(blend_mask_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, m, out, n,
                       inner);
    return set_check_launch("set_blend_mask");
}

__global__ void __launch_bounds__(256) mul_one_minus_mask_kernel(const float *x, const float *m, float *out, int64_t n,
                                                                 int64_t inner) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[i] * (1.0f - m[i / inner]);
}
extern "C" int set_mul_one_minus_mask(const float *x, const float *m, float *out, int64_t n, int64_t inner,
                                      void *stream) {
    SET_REQUIRE(x && m && out && n > 0 && inner > 0, "set_mul_one_minus_mask");
    hipLaunchKernelGGL(mul_one_minus_mask_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, x, m,
                       out, n, inner);
    return set_check_launch("set_mul_one_minus_mask");
}

// ---- LengthRegulator (integer, bit-exact): one block per utterance ---------------------------------
__device__ __forceinline__ int64_t round_dur(float d, int64_t token) {
    // torch.round (half to even) then .long(); zeroed on padding tokens
    return token == 0 ? 0 : (int64_t)rintf(d);
}
__global__ void __launch_bounds__(64) dur_total_kernel(const float *dur, const int64_t *txt, int64_t *total,
                                                       int T_txt) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    int64_t s = 0;
    for (int j = 0; j < T_txt; ++j) s += round_dur(dur[(int64_t)b * T_txt + j], txt[(int64_t)b * T_txt + j]);
    total[b] = s;
}
extern "C" int set_dur_total(const float *dur, const int64_t *txt, int64_t *total, int32_t B, int32_t T_txt,
                             void *stream) {
    SET_REQUIRE(dur && txt && total && B > 0 && T_txt > 0, "set_dur_total");
    hipLaunchKernelGGL(dur_total_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dur, txt, total, T_txt);
    return set_check_launch("set_dur_total");
}
__global__ void __launch_bounds__(256) length_regulate_kernel(const float *dur, const int64_t *txt, int64_t *mel2ph,
                                                              int T_txt, int T_out) {
    extern __shared__ int64_t cs[];  // inclusive cumsum
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int64_t s = 0;
        for (int j = 0; j < T_txt; ++j) {
            s += round_dur(dur[(int64_t)b * T_txt + j], txt[(int64_t)b * T_txt + j]);
            cs[j] = s;
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < T_out; p += 256) {
        // token j (1-based j+1) covers [cs[j-1], cs[j]); positions past the total get 0
        int lo = 0, hi = T_txt;  // first j with cs[j] > p
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cs[mid] > p) hi = mid; else lo = mid + 1;
        }
        mel2ph[(int64_t)b * T_out + p] = lo < T_txt ? (int64_t)(lo + 1) : 0;
    }
}
extern "C" int set_length_regulate(const float *dur, const int64_t *txt, int64_t *mel2ph, int32_t B, int32_t T_txt,
                                   int32_t T_out, void *stream) {
    SET_REQUIRE(dur && txt && mel2ph && B > 0 && T_txt > 0 && T_out > 0, "set_length_regulate");
    SET_REQUIRE(T_txt <= 8000, "set_length_regulate");
    hipLaunchKernelGGL(length_regulate_kernel, dim3(B), dim3(256), (size_t)T_txt * sizeof(int64_t),
                       (hipStream_t)stream, dur, txt, mel2ph, T_txt, T_out);
    return set_check_launch("set_length_regulate");
}

// Stream ordering for the host side's second ("leaf") stream: everything enqueued on `after` from now on waits for everything enqueued on
// `first` so far.  One event per call site slot (created once, timing disabled), re-recorded at every call; inside a stream capture the pair
// becomes a cross-stream edge of the graph.  Replaces torch's Event.record + Stream.wait_event (two Python -> C++ round trips per fork).
// Events are created once per slot, on the device the slot belongs to (slot s <-> device s / 2: the host side numbers its slots 2 idx and
// 2 idx + 1), under a mutex: the first call may come from the autograd engine's worker thread while the main thread makes its own first
// call, and from a thread whose current device is not the stream's (round-5 advisor item).
static hipEvent_t set_cached_event(hipEvent_t *table, int32_t slot, int32_t dev, const char *what) {
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (table[slot]) return table[slot];
    int cur = 0, n = 0;
    if (hipGetDevice(&cur) != hipSuccess || hipGetDeviceCount(&n) != hipSuccess) return nullptr;
    const bool sw = dev >= 0 && dev < n && dev != cur;
    if (sw && hipSetDevice(dev) != hipSuccess) return nullptr;
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (sw) (void)hipSetDevice(cur);
    if (rc != hipSuccess) {
        (void)set_fail(SET_E_LAUNCH, what, hipGetErrorString(rc));
        return nullptr;
    }
    table[slot] = e;
    return e;
}

extern "C" int set_stream_order(void *first, void *after, int32_t slot) {
    static hipEvent_t ev[32] = {nullptr};
    SET_REQUIRE(slot >= 0 && slot < 32, "set_stream_order");
    const hipEvent_t e = ev[slot] ? ev[slot] : set_cached_event(ev, slot, slot / 2, "set_stream_order(event)");
    SET_REQUIRE(e != nullptr, "set_stream_order(event)");
    SET_HIP(hipEventRecord(e, (hipStream_t)first), "set_stream_order(record)");
    SET_HIP(hipStreamWaitEvent((hipStream_t)after, e, 0), "set_stream_order(wait)");
    return SET_OK;
}

// Progress markers of a stream (round 6): set_stream_mark records marker `slot` (0 .. 63; device `dev`) on `stream`, set_stream_mark_done
// returns 1 once everything enqueued before that mark has finished, 0 while it has not, < 0 on error.  The leaf stream's host side uses
// them to let go of the operands of leaf kernels that have already run instead of holding every operand of a backward pass until its end.
static hipEvent_t g_marks[64] = {nullptr};
extern "C" int set_stream_mark(void *stream, int32_t slot, int32_t dev) {
    SET_REQUIRE(slot >= 0 && slot < 64, "set_stream_mark");
    const hipEvent_t e = g_marks[slot] ? g_marks[slot] : set_cached_event(g_marks, slot, dev, "set_stream_mark(event)");
    SET_REQUIRE(e != nullptr, "set_stream_mark(event)");
    SET_HIP(hipEventRecord(e, (hipStream_t)stream), "set_stream_mark(record)");
    return SET_OK;
}
extern "C" int set_stream_mark_done(int32_t slot) {
    SET_REQUIRE(slot >= 0 && slot < 64 && g_marks[slot] != nullptr, "set_stream_mark_done");
    const hipError_t rc = hipEventQuery(g_marks[slot]);
    if (rc == hipSuccess) return 1;
    if (rc == hipErrorNotReady) {
        (void)hipGetLastError();
        return 0;
    }
    SET_HIP(rc, "set_stream_mark_done");
    return SET_OK;
}

// A non-blocking stream of the LOWEST priority the device offers: the leaf stream's chip-filling weight-gradient GEMMs must not hold
// the workgroup slots the compute stream's short kernels are waiting for (measured with equal priorities: a 5 us ordered sum of the
// compute stream took 110 - 170 us behind a grouped weight-gradient launch).  Never destroyed (one per process and device).
// (Measured and dropped, round 5: a CU-masked leaf stream, hipExtStreamCreateWithCUMask with 128 - 224 of the 256 CUs, so that some CUs stay
// free for the compute stream -- every launch on such a stream takes a slow path: 17 - 18 ms per spec_denoiser step instead of 9.8.)
extern "C" int set_stream_create_low_priority(void **out) {
    SET_REQUIRE(out != nullptr, "set_stream_create_low_priority");
    int least = 0, greatest = 0;
    SET_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest), "set_stream_create_low_priority(range)");
    hipStream_t s = nullptr;
    SET_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least), "set_stream_create_low_priority");
    *out = (void *)s;
    return SET_OK;
}
