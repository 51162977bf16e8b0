// Training-side kernels of the hot path: weight-gradient GEMM on fp32 MFMA,
// reductions, backward of LayerNorm / activations / gate / embedding / alignment gather, the mel / duration /
// pitch losses of tasks/tts/speech_base.py:219-257 and tasks/speech_editing/speech_editing_base.py:58-108 with
// their gradients, fused AdamW and the global grad norm.  Input gradients of convolutions reuse set_conv1d
// (a convolution with transposed weight addressing and negated dilation/padding).
#include "common.h"
#include "rows_sum.h"

namespace {

// -------------------------------------------------------------------------------------------------------
// conv weight gradient:  dW[co][ci][tap] += sum_{b,t} G[b][co][t] * P(X[b][ci][t + tap*dil - pad])
// GEMM M = Cout, N = (tap, ci), reduction over frames; block tile 128 x 64, 32-frame LDS chunks, split over
// (batch, frame-chunk) slices with fp32 atomics into dW.
// -------------------------------------------------------------------------------------------------------
constexpr int WG_KC = 32;       // frames per LDS chunk
constexpr int WG_LD = WG_KC + 1;  // padded row (conflict-free column reads)

struct WgradArgs {
    const float *g;    // [B][Cout][T]
    const float *x;    // [B][Cin][T_in]
    const float *chan_add;  // optional [B][Cin]
    float *dw;         // [Cout][Cin][K] (+= )
    int B, Cin, Cout, K, dil, pad, T, T_in, pro;
    float pro_param;
    int chunks_per_slice, n_chunks_t;  // chunk id = b * n_chunks_t + tc
    int gx, gy, gz, xcd_map;           // logical grid (tap x ci tile, co tile, split-K slice); launch is 1-D
    float *partial;                    // deterministic mode: slice z stores its tile to partial[z][Cout][Cin][K] (no atomics)
};

__global__ void __launch_bounds__(256) conv1d_wgrad_mfma_kernel(WgradArgs a) {
    __shared__ float Gs[2 * 128 * WG_LD];  // two buffers each: one barrier per chunk (see the loop below)
    __shared__ float Xs[2 * 64 * WG_LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // Logical block (bx, by, bz) from the linear workgroup id.  Workgroups go to the 8 XCDs round-robin by that id and
    // every XCD has its own L2: with xcd_map, slice bz (= one range of frames) is given to ONE XCD, so the gz/8 * gx*gy
    // tiles that read the same g / x frames share them in that L2 instead of fetching them once per XCD.
    int bx, by, bz;
    {
        const int L = blockIdx.x, nxy = a.gx * a.gy;
        if (a.xcd_map) {
            const int slot = L >> 3, xy = slot % nxy;
            bz = (L & 7) + 8 * (slot / nxy);
            bx = xy % a.gx;
            by = xy / a.gx;
        } else {
            bx = L % a.gx;
            by = (L / a.gx) % a.gy;
            bz = L / nxy;
        }
        // the divisions run on the vector ALU: bring the results back to scalar registers, or every address derived
        // from them occupies VGPRs (116 -> 140, one wave per SIMD less)
        bx = __builtin_amdgcn_readfirstlane(bx);
        by = __builtin_amdgcn_readfirstlane(by);
        bz = __builtin_amdgcn_readfirstlane(bz);
    }
    const int ci_tiles = (a.Cin + 63) / 64;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 64;
    const int co0 = by * 128;
    const int shift = tap * a.dil - a.pad;
    f32x16 acc0 = {0}, acc1 = {0};
    const int total_chunks = a.B * a.n_chunks_t;
    const int c_begin = bz * a.chunks_per_slice;
    const int c_end = min(c_begin + a.chunks_per_slice, total_chunks);

    // Staging: thread (k = tid & 31 along t, r0 = tid >> 5) owns rows r0, r0+8, ... of both tiles.  All loads of a chunk
    // are UNCONDITIONAL on clamped addresses and issued together, one chunk AHEAD of the MFMAs (the validity select
    // happens at the LDS write): `if (valid) v = load` costs one serialized global round trip per element, 24 per chunk
    // against ~1 us of MFMA work (measured 530 us for the 512x768 dilated-conv gradient at 25.6k frames).
    static_assert(WG_KC == 32, "staging map assumes 32-frame chunks");
    const int sk = tid & 31, sr0 = tid >> 5;
    const bool has_add = a.chan_add != nullptr;
    float gv[16], xv[8], av[8];
    auto issue = [&](int ch) {
        const int b = ch / a.n_chunks_t, t0 = (ch % a.n_chunks_t) * WG_KC;
        const int tc = min(t0 + sk, a.T - 1);
        const int tic = min(max(t0 + sk + shift, 0), a.T_in - 1);
        const float *gb = a.g + (int64_t)b * a.Cout * a.T + tc;
        const float *xb = a.x + (int64_t)b * a.Cin * a.T_in + tic;
        const float *addb = has_add ? a.chan_add + (int64_t)b * a.Cin : xb;  // dummy stays a valid address
#pragma unroll
        for (int j = 0; j < 16; ++j) gv[j] = gb[(int64_t)min(co0 + sr0 + 8 * j, a.Cout - 1) * a.T];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cic = min(ci0 + sr0 + 8 * j, a.Cin - 1);
            xv[j] = xb[(int64_t)cic * a.T_in];
            av[j] = addb[cic];  // unconditional (dummy address when there is no per-channel add); the `has_add ? : 0` select sits in
                                // commit(): done here it consumed the loaded value at once, i.e. the wave waited for the whole chunk's
                                // loads right after issuing them -- nothing was in flight under the MFMAs (found in the ISA, round 4)
        }
    };
    auto commit = [&](int buf, int ch) {  // buf 0/1 -> integer offsets (a selected pointer would lose its LDS address space)
        const int go = buf * 128 * WG_LD, xo = buf * 64 * WG_LD;
        const int t0 = (ch % a.n_chunks_t) * WG_KC;
        const int t = t0 + sk, ti = t + shift;
        const bool tv = t < a.T, tiv = tv && ti >= 0 && ti < a.T_in;
#pragma unroll
        for (int j = 0; j < 16; ++j) Gs[go + (sr0 + 8 * j) * WG_LD + sk] = (tv && co0 + sr0 + 8 * j < a.Cout) ? gv[j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            Xs[xo + (sr0 + 8 * j) * WG_LD + sk] = (tiv && ci0 + sr0 + 8 * j < a.Cin) ? dev_pro(xv[j] + (has_add ? av[j] : 0.0f), a.pro, a.pro_param) : 0.0f;
    };
    // while a wave runs the MFMAs of chunk c out of buffer c & 1 the loads of chunk c + 1 are in flight; it then writes
    // them to the other buffer (last read one barrier ago) and issues the loads of chunk c + 2
    if (c_begin < c_end) {
        issue(c_begin);
        commit(0, c_begin);
        if (c_begin + 1 < c_end) issue(c_begin + 1);
    }
    __syncthreads();
    int cur = 0;
    for (int ch = c_begin; ch < c_end; ++ch, cur ^= 1) {
        const float *ap = Gs + cur * 128 * WG_LD + (32 * w + l31) * WG_LD + half;
        const float *bp = Xs + cur * 64 * WG_LD + l31 * WG_LD + half;
#pragma unroll
        for (int kk = 0; kk < WG_KC; kk += 2) {
            const float av = ap[kk];
            acc0 = mfma32(av, bp[kk], acc0);
            acc1 = mfma32(av, bp[32 * WG_LD + kk], acc1);
        }
        if (ch + 1 < c_end) {
            commit(cur ^ 1, ch + 1);
            if (ch + 2 < c_end) issue(ch + 2);
        }
        __syncthreads();
    }
    // a padding slice of an XCD-mapped grid has nothing to add (tested here, not before the loop: an early exit there
    // changes the register allocation of the whole kernel, 116 -> 140 VGPRs)
    if (c_begin >= c_end && !a.partial) return;  // (deterministic mode: an empty slice stores zeros)
    float *dst = a.partial ? a.partial + (int64_t)bz * a.Cout * a.Cin * a.K : a.dw;
    const bool plain = a.partial != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * w + mfma32_row(r, lane);
        if (co >= a.Cout) continue;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int ci = ci0 + 32 * cb + l31;
            if (ci >= a.Cin) continue;
            float *p = &dst[((int64_t)co * a.Cin + ci) * a.K + tap];
            const float v = cb == 0 ? acc0[r] : acc1[r];
            if (plain) *p = v; else atomicAdd(p, v);
        }
    }
}

__global__ void __launch_bounds__(256) conv1d_wgrad_naive_kernel(WgradArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.Cout * a.Cin * a.K) return;
    const int tap = (int)(idx % a.K), ci = (int)((idx / a.K) % a.Cin), co = (int)(idx / ((int64_t)a.K * a.Cin));
    const int shift = tap * a.dil - a.pad;
    float s = 0.0f;
    for (int b = 0; b < a.B; ++b) {
        const float add = a.chan_add ? a.chan_add[(int64_t)b * a.Cin + ci] : 0.0f;
        for (int t = 0; t < a.T; ++t) {
            const int ti = t + shift;
            if (ti < 0 || ti >= a.T_in) continue;
            const float xv = dev_pro(a.x[((int64_t)b * a.Cin + ci) * a.T_in + ti] + add, a.pro, a.pro_param);
            s = fmaf(a.g[((int64_t)b * a.Cout + co) * a.T + t], xv, s);
        }
    }
    a.dw[idx] += s;
}

// out[c] += sum_{b,t} x[b][c][t]  (bias grads): grid (C, slices over the batch), wave-shuffle + LDS reduce, one atomic per
// block.  (One block per channel left C <= 512 blocks to stream 50 MB: 27 us at B=32, T=800.)
__global__ void __launch_bounds__(256) channel_sum_kernel(const float *x, float *out, int B, int C, int T, float *part = nullptr) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float *row = x + ((int64_t)b * C + c) * T;
        for (int t = threadIdx.x; t < T; t += 256) s += row[t];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = red[0] + red[1] + red[2] + red[3];
        if (part) part[(int64_t)blockIdx.y * C + c] = v; else atomicAdd(&out[c], v);
    }
}
// out[b][c] = sum_t x[b][c][t] (* 1/div); one wave per (b,c)
__global__ void __launch_bounds__(256) row_sum_kernel(const float *x, float *out, int64_t rows, int T, float scale) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    for (int t = lane; t < T; t += 64) s += x[row * T + t];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[row] = s * scale;
}

// G = dY * mask * alpha * [relu: y > 0]      (epilogue backward of set_conv1d; act in {none, relu})
__global__ void __launch_bounds__(256) conv_epilogue_bwd_kernel(const float *dy, const float *y, const float *mask,
                                                                float *g, int B, int C, int T, int act, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T);
    const int b = (int)(i / ((int64_t)C * T));
    float v = dy[i];
    if (mask) v *= mask[(int64_t)b * T + t];
    if (act == SET_ACT_RELU && !(y[i] > 0.0f)) v = 0.0f;
    g[i] = v * alpha;
}

// four consecutive frames per thread (T % 4 == 0, 16-byte aligned operands): the same arithmetic per element, 16-byte accesses and one
// index division per four elements (round 6: the one-element forms of these elementwise kernels ran at 2 - 4 TB/s)
__global__ void __launch_bounds__(256) conv_epilogue_bwd_vec4_kernel(const float *dy, const float *y, const float *mask, float *g,
                                                                     int64_t n4, int64_t CT, int T, int act, float alpha) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const int64_t i = i4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4 *>(dy + i);
    if (mask) {
        const int64_t b = i / CT;
        const int t = (int)(i % T);
        const f32x4 m = *reinterpret_cast<const f32x4 *>(mask + b * T + t);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= m[e];
    }
    if (act == SET_ACT_RELU) {
        const f32x4 yv = *reinterpret_cast<const f32x4 *>(y + i);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(yv[e] > 0.0f)) v[e] = 0.0f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] * alpha;
    *reinterpret_cast<f32x4 *>(g + i) = v;
}

// ---- activations with saved pre-activation z ----------------------------------------------------------------
__device__ __forceinline__ float dev_act_grad(float z, int act, float p) {
    switch (act) {
        case SET_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
        case SET_ACT_GELU: {  // d/dz 0.5 z (1 + erf(z/sqrt2)) = Phi(z) + z phi(z)
            const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
            return cdf + z * pdf;
        }
        case SET_ACT_TANH: { const float th = tanhf(z); return 1.0f - th * th; }
        case SET_ACT_SOFTPLUS: return z > 20.0f ? 1.0f : dev_sigmoid(z);
        case SET_ACT_MISH: {
            const float sp = dev_softplus(z), th = tanhf(sp);
            const float dsp = z > 20.0f ? 1.0f : dev_sigmoid(z);
            return th + z * (1.0f - th * th) * dsp;
        }
        case SET_ACT_LRELU: return z > 0.0f ? 1.0f : p;
        default: return 1.0f;
    }
}
__global__ void __launch_bounds__(256) act_fwd_kernel(const float *z, float *y, int64_t n, int act, float p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = dev_act(z[i], act, p);
}
__global__ void __launch_bounds__(256) act_bwd_kernel(const float *z, const float *dy, float *dz, int64_t n, int act,
                                                      float p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dz[i] = dy[i] * dev_act_grad(z[i], act, p);
}
// the same followed by the conv epilogue's alpha (dz = (dy act'(z)) alpha: the two roundings of act_bwd + conv_epilogue_bwd, one launch)
__global__ void __launch_bounds__(256) act_bwd_scaled_kernel(const float *z, const float *dy, float *dz, int64_t n, int act, float p,
                                                             float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float t = dy[i] * dev_act_grad(z[i], act, p);
        dz[i] = t * scale;
    }
}

__global__ void __launch_bounds__(256) act_fwd_vec4_kernel(const float *z, float *y, int64_t n4, int act, float p) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + i4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = dev_act(zv[e], act, p);
    *reinterpret_cast<f32x4 *>(y + i4 * 4) = o;
}
// scaled != 0: dz = (dy act'(z)) scale (act_bwd_scaled_kernel's two roundings), else dz = dy act'(z)
__global__ void __launch_bounds__(256) act_bwd_vec4_kernel(const float *z, const float *dy, float *dz, int64_t n4, int act, float p,
                                                           int scaled, float scale) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const f32x4 zv = *reinterpret_cast<const f32x4 *>(z + i4 * 4), dv = *reinterpret_cast<const f32x4 *>(dy + i4 * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = dv[e] * dev_act_grad(zv[e], act, p);
        o[e] = scaled ? t * scale : t;
    }
    *reinterpret_cast<f32x4 *>(dz + i4 * 4) = o;
}

// ---- gate / res-skip backward ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gate_bwd_kernel(const float *y, const float *dz, float *dy, int B, int C,
                                                       int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int64_t ct = i % ((int64_t)C * T), b = i / ((int64_t)C * T);
    const float *yb = y + b * 2 * C * T;
    float *dyb = dy + b * 2 * C * T;
    const float s = dev_sigmoid(yb[ct]), th = tanhf(yb[(int64_t)C * T + ct]);
    const float d = dz[i];
    dyb[ct] = d * th * s * (1.0f - s);
    dyb[(int64_t)C * T + ct] = d * s * (1.0f - th * th);
}
// forward: x_out = (x + o[:C]) / sqrt2 ; skip_out = skip_in + o[C:]
// backward: dx = dx_out / sqrt2 ; do[:C] = dx_out / sqrt2 ; do[C:] = dskip_out
// four consecutive elements per thread (16-byte accesses; C * T a multiple of 4, 16-byte aligned operands): the one-element form above moves
// 131 MB in 62 us at B = 32, T = 800 (2.1 TB/s) -- 20 launches per fp32 training step.  Same arithmetic per element: same bits.
__global__ void __launch_bounds__(256) gate_bwd_vec4_kernel(const float *y, const float *dz, float *dy, int64_t n4, int64_t ct4_per_b,
                                                            int64_t CT) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const int64_t b = i4 / ct4_per_b, ct = (i4 - b * ct4_per_b) * 4;
    const float *yb = y + b * 2 * CT + ct;
    float *dyb = dy + b * 2 * CT + ct;
    const f32x4 yg = *reinterpret_cast<const f32x4 *>(yb), yf = *reinterpret_cast<const f32x4 *>(yb + CT);
    const f32x4 d = *reinterpret_cast<const f32x4 *>(dz + i4 * 4);
    f32x4 og, of;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float s = dev_sigmoid(yg[e]), th = tanhf(yf[e]);
        og[e] = d[e] * th * s * (1.0f - s);
        of[e] = d[e] * s * (1.0f - th * th);
    }
    *reinterpret_cast<f32x4 *>(dyb) = og;
    *reinterpret_cast<f32x4 *>(dyb + CT) = of;
}
__global__ void __launch_bounds__(256) res_skip_bwd_kernel(const float *dx_out, const float *dskip, float *dx, float *d_o,
                                                           int B, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int64_t ct = i % ((int64_t)C * T), b = i / ((int64_t)C * T);
    const float v = dx_out[i] / 1.41421356237309504880f;
    dx[i] = v;
    float *ob = d_o + b * 2 * C * T;
    ob[ct] = v;
    ob[(int64_t)C * T + ct] = dskip[i];
}
__global__ void __launch_bounds__(256) res_skip_bwd_vec4_kernel(const float *dx_out, const float *dskip, float *dx, float *d_o, int64_t n4,
                                                                int64_t ct4_per_b, int64_t CT) {
    const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const int64_t b = i4 / ct4_per_b, ct = (i4 - b * ct4_per_b) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4 *>(dx_out + i4 * 4);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g[e] / 1.41421356237309504880f;
    *reinterpret_cast<f32x4 *>(dx + i4 * 4) = v;
    float *ob = d_o + b * 2 * CT + ct;
    *reinterpret_cast<f32x4 *>(ob) = v;
    *reinterpret_cast<f32x4 *>(ob + CT) = *reinterpret_cast<const f32x4 *>(dskip + i4 * 4);
}

// ---- LayerNorm over channels, backward: block = 32 frames x 8 channel groups (as the forward kernel), column sums
//      through LDS; x and dy of the thread's <= 32 channels are fetched once as two batches and stay in registers for
//      all four passes (C <= 256; wider inputs re-read).  dgamma/dbeta: reduce over the 32 frames of the group inside
//      the wave, then one row of per-block partial sums (`partial[blk][2][C]`, reduced by lnb_partial_sum_kernel) --
//      or, without scratch, one atomic per (block, channel): every block hits the same 2C addresses, and same-address
//      device-scope atomics from different XCDs serialise at ~0.7 us each (measured: 416 blocks -> 280 us for 80 MB
//      of traffic), hence the scratch path.
// Three block shapes (template FT frames x CG channel groups, RC channels per thread in registers): 32 x 8 (C > 256), 16 x 16 in 256 threads
// and 32 x 16 in 512 threads (C <= 256; the launch function picks by grid size).  Round 5: the 16-group shapes (twice the blocks of the
// 32 x 8 one, which left CampNet's B = 16, T = 800 at 400 blocks = two rounds on 256 CUs), the residual gradient `add` fetched with x and dy
// instead of in the store loop (a second exposed memory round trip per block), raw-buffer addressing (one per-lane offset + wave-uniform row
// offsets: 148 -> 96 registers, the 64-bit address per load was the rest).  Still ~2.4 TB/s at the large shapes.
constexpr int LNB_FT = 32;  // frames per block of the wide shape (the scratch size and the fallback loop are written for it)
template <int FT, int CG>
__device__ __forceinline__ float lnb_block_sum(float v, float (*red)[FT], int cg, int tl) {
    __syncthreads();
    red[cg][tl] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < CG; ++g) s += red[g][tl];
    return s;
}
template <int FT>
__device__ __forceinline__ void lnb_emit(float dg, float db, int c, int tl, float *dgamma, float *dbeta, float *partial,
                                         int C) {
    // sum over the FT frames of this channel group (FT consecutive lanes: xor offsets stay inside the group)
#pragma unroll
    for (int off = FT / 2; off > 0; off >>= 1) { dg += __shfl_xor(dg, off); db += __shfl_xor(db, off); }
    if (tl == 0) {
        if (partial) {
            float *row = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
            row[c] = dg;
            row[C + c] = db;
        } else {
            atomicAdd(&dgamma[c], dg);
            atomicAdd(&dbeta[c], db);
        }
    }
}

// add != NULL: dx = (LayerNorm gradient) + add -- the residual branch of a pre-LN sub-block joins here instead of in a separate launch
template <int FT, int CG, int RC>
__global__ void __launch_bounds__(FT * CG) layernorm_ch_bwd_kernel(const float *x, const float *gamma, const float *mask,
                                                               const float *dy, float *dx, float *dgamma, float *dbeta,
                                                               float *partial, int B, int C, int T, float eps, const float *add) {
    static_assert(FT * CG == 256 || FT * CG == 512, "one thread per (frame, channel group)");
    __shared__ float red[CG][FT];
    const int tl = threadIdx.x % FT, cg = threadIdx.x / FT;
    const int b = blockIdx.y, t = blockIdx.x * FT + tl;
    const bool valid = t < T;
    const int tc = valid ? t : T - 1;
    const int cq = (C + CG - 1) / CG, c0 = cg * cq, c1 = min(C, c0 + cq);
    const float *xp = x + (int64_t)b * C * T + tc;
    const float *dp = dy + (int64_t)b * C * T + tc;
    float *op = dx + (int64_t)b * C * T + tc;
    const float *ap = add ? add + (int64_t)b * C * T + tc : nullptr;
    const float m = !valid ? 0.0f : (mask ? mask[(int64_t)b * T + t] : 1.0f);  // m == 0 on the frames beyond T
    if (cq <= RC) {  // block-uniform
        constexpr bool PREF = RC <= 16;  // the residual gradient rides with x and dy when the registers allow (32 more made the wide shape slower)
        // raw-buffer addressing: one per-lane byte offset (first owned channel row, frame) + a wave-uniform row offset per channel; rows
        // beyond C are not fetched (per-lane offset out of range: the buffer unit returns 0) -- no 64-bit address per load
        const rsrc_t rx = make_rsrc(x + (int64_t)b * C * T), rdy = make_rsrc(dy + (int64_t)b * C * T);
        const rsrc_t radd = make_rsrc(add ? add + (int64_t)b * C * T : x), rdx = make_rsrc(dx + (int64_t)b * C * T);
        const unsigned vb = (unsigned)(c0 * T + tc) * 4u, T4 = (unsigned)T * 4u;
        float xv[RC], gv[RC], gm[RC], av[PREF ? RC : 1];
#pragma unroll
        for (int i = 0; i < RC; ++i) xv[i] = buf_load(rx, c0 + i < C ? vb : BUF_OOB, (unsigned)i * T4);
#pragma unroll
        for (int i = 0; i < RC; ++i) gv[i] = buf_load(rdy, c0 + i < C ? vb : BUF_OOB, (unsigned)i * T4);
        if constexpr (PREF) {
#pragma unroll
            for (int i = 0; i < RC; ++i) av[i] = buf_load(radd, (add && c0 + i < C) ? vb : BUF_OOB, (unsigned)i * T4);
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) gm[i] = gamma[min(c0 + i, C - 1)];
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < RC; ++i) s += c0 + i < c1 ? xv[i] : 0.0f;
        const float mean = lnb_block_sum<FT, CG>(s, red, cg, tl) / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const float d = c0 + i < c1 ? xv[i] - mean : 0.0f;
            q = fmaf(d, d, q);
        }
        const float rstd = 1.0f / sqrtf(lnb_block_sum<FT, CG>(q, red, cg, tl) / (float)C + eps);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            xv[i] = (xv[i] - mean) * rstd;                 // x-hat
            gv[i] = c0 + i < c1 ? gv[i] * m : 0.0f;        // masked dy (0 for the channels this thread does not own)
            const float g = gv[i] * gm[i];
            s1 += g;
            s2 = fmaf(g, xv[i], s2);
        }
        s1 = lnb_block_sum<FT, CG>(s1, red, cg, tl) / (float)C;
        s2 = lnb_block_sum<FT, CG>(s2, red, cg, tl) / (float)C;
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            if (c0 + i < c1) {  // uniform per channel group
                const float gx = rstd * (gv[i] * gm[i] - s1 - xv[i] * s2);
                float o;
                if constexpr (PREF) o = add ? gx + av[i] : gx;
                else o = add ? gx + buf_load(radd, valid ? vb : BUF_OOB, (unsigned)i * T4) : gx;
                buf_store(o, rdx, valid ? vb : BUF_OOB, (unsigned)i * T4);  // frames beyond T: offset out of range, dropped
                lnb_emit<FT>(gv[i] * xv[i], gv[i], c0 + i, tl, dgamma, dbeta, partial, C);
            }
        }
        return;
    }
    float s = 0.0f;
    for (int c = c0; c < c1; ++c) s += xp[(int64_t)c * T];
    const float mean = lnb_block_sum<FT, CG>(s, red, cg, tl) / (float)C;
    float q = 0.0f;
    for (int c = c0; c < c1; ++c) { const float d = xp[(int64_t)c * T] - mean; q = fmaf(d, d, q); }
    const float rstd = 1.0f / sqrtf(lnb_block_sum<FT, CG>(q, red, cg, tl) / (float)C + eps);
    float s1 = 0.0f, s2 = 0.0f;
    for (int c = c0; c < c1; ++c) {
        const float xh = (xp[(int64_t)c * T] - mean) * rstd;
        const float g = dp[(int64_t)c * T] * m * gamma[c];
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    s1 = lnb_block_sum<FT, CG>(s1, red, cg, tl) / (float)C;
    s2 = lnb_block_sum<FT, CG>(s2, red, cg, tl) / (float)C;
    for (int c = c0; c < c1; ++c) {
        const float xh = (xp[(int64_t)c * T] - mean) * rstd;
        const float dyc = dp[(int64_t)c * T] * m;
        if (valid) {
            const float gx = rstd * (dyc * gamma[c] - s1 - xh * s2);
            op[(int64_t)c * T] = ap ? gx + ap[(int64_t)c * T] : gx;
        }
        lnb_emit<FT>(dyc * xh, dyc, c, tl, dgamma, dbeta, partial, C);
    }
}
// out[j] += sum_r partial[r][j], j < n (= 2C: dgamma then dbeta); block = 64 columns x ROWS_RG row groups (rows_sum.h)
__global__ void __launch_bounds__(64 * ROWS_RG) lnb_partial_sum_kernel(const float *partial, float *dgamma, float *dbeta, int rows,
                                                                       int C) {
    __shared__ float red[ROWS_RG][64];
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tl, n = 2 * C;
    float s = j < n ? rows_sum_chains(partial + j, n, rg, rows) : 0.0f;
    s = rows_sum_groups(s, red, rg, tl);
    if (rg == 0 && j < n) {
        if (j < C) dgamma[j] += s; else dbeta[j - C] += s;
    }
}

// ---- embedding / alignment gather backward (scatter-add) ------------------------------------------------------
// one thread per (b, c) walks t and flushes one atomic per RUN of equal indices: masked regions / sorted alignments
// map long stretches of frames to one row (e.g. every masked frame -> pitch bin 1), which serialises per-frame atomics.
// (Tried: one wave per (b, c, 64-frame segment) with a segmented scan -- coalesced and 13x more parallel, but the frame-
// level pitch bins give one run per frame, and 5 M atomics issued at once on 58 k addresses cost 2.7 ms more per step
// than this slow walk, which spreads them out.  The way forward is an LDS-privatised table per block, not more threads.)
__global__ void __launch_bounds__(256) embedding_bwd_kernel(const int64_t *idx, const float *dout, float *dtable, int B,
                                                            int T, int C, int n_rows, float scale, int padding_idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int c = (int)(i % C), b = (int)(i / C);
    const int64_t *ib = idx + (int64_t)b * T;
    const float *dp = dout + ((int64_t)b * C + c) * T;
    int64_t cur = -1;
    float acc = 0.0f;
    for (int t = 0; t < T; ++t) {
        int64_t row = ib[t];
        row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
        if (row != cur) {
            if (cur >= 0 && cur != padding_idx) atomicAdd(&dtable[cur * C + c], acc);
            cur = row;
            acc = 0.0f;
        }
        acc = fmaf(scale, dp[t], acc);
    }
    if (cur >= 0 && cur != padding_idx) atomicAdd(&dtable[cur * C + c], acc);  // nn.Embedding(padding_idx) row stays 0
}
__global__ void __launch_bounds__(256) expand_states_bwd_kernel(const int64_t *mel2ph, const float *dout, float *denc,
                                                                int B, int C, int T_txt, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T), c = (int)((i / T) % C), b = (int)(i / ((int64_t)T * C));
    const int64_t m = mel2ph[(int64_t)b * T + t];
    if (m > 0 && m <= T_txt) atomicAdd(&denc[((int64_t)b * C + c) * T_txt + (m - 1)], dout[i]);
}

// ---- dropout: keep mask from Philox(seed, offset + i/4); same kernel for forward and backward ------------------
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ void __launch_bounds__(256) dropout_kernel(const float *x, float *y, int64_t n, float p, uint64_t seed,
                                                      uint64_t offset, const uint64_t *seed_delta, int vec) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    if (seed_delta) seed += *seed_delta;  // set_rng_seed_delta: see randn_kernel
    const uint64_t ctr = offset + (uint64_t)q;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x5eedu, 0u};
    philox_round(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float inv = 1.0f / (1.0f - p);
    if (vec && q * 4 + 3 < n) {  // 16-byte aligned operands: one load, one store (same values)
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(x + q * 4);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float u = (float)(c[k] >> 8) * (1.0f / 16777216.0f);
            o[k] = u >= p ? xv[k] * inv : 0.0f;
        }
        *reinterpret_cast<f32x4 *>(y + q * 4) = o;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = q * 4 + k;
        if (i >= n) break;
        const float u = (float)(c[k] >> 8) * (1.0f / 16777216.0f);
        y[i] = u >= p ? x[i] * inv : 0.0f;
    }
}

// ---- losses ------------------------------------------------------------------------------------------------------
// weights_nonzero_speech (utils/nn/seq_utils.py:33-37): w[b][t] = (sum_m |target[b][t][m]|) != 0
__global__ void __launch_bounds__(256) frame_weight_kernel(const float *target, float *w, int64_t frames, int M) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= frames) return;
    float s = 0.0f;
    for (int m = 0; m < M; ++m) s += fabsf(target[i * M + m]);
    w[i] = s != 0.0f ? 1.0f : 0.0f;
}
// sum-reduce helper: out[0] += sum x[i] (* w[i / inner])
__global__ void __launch_bounds__(256) weighted_sum_kernel(const float *x, const float *w, float *out, int64_t n,
                                                           int64_t inner, float *part = nullptr) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += w ? x[i] * w[i / inner] : x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (part) part[blockIdx.x] = red[0]; else atomicAdd(out, red[0]); }
}
// |pred - target| (forward) / sign(pred - target) (backward), [n]
__global__ void __launch_bounds__(256) l1_elem_kernel(const float *pred, const float *target, float *absd, float *sgn,
                                                      int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float d = pred[i] - target[i];
    if (absd) absd[i] = fabsf(d);
    if (sgn) sgn[i] = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
}
// out[i] = a[i] * w[i / inner] * scale   (broadcast multiply used by the loss backward)
__global__ void __launch_bounds__(256) scale_bcast_kernel(const float *a, const float *w, float *out, int64_t n,
                                                          int64_t inner, const float *scale_dev, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = a[i] * scale;
    if (w) v *= w[i / inner];
    if (scale_dev) v *= scale_dev[0];
    out[i] = v;
}

// SSIM (utils/metrics/ssim.py:12-44): 11x11 gaussian (sigma 1.5) zero-padded depthwise filter on [B][H=T][W=M] images
struct SsimArgs {
    const float *img1, *img2;  // [B][H][W]; `bias` is added to in-range pixels, the zero padding stays 0
    float bias;
    float *mu1, *mu2, *s11, *s22, *s12;  // filtered maps (raw second moments, not centred)
    int B, H, W;
};
__constant__ float c_gauss[11];
// Round 5: 16 x 16 output pixels per block from a 26 x 26 LDS tile of (pixel + bias) values (0 outside the image) instead of 121 x 2 global
// loads per pixel (150 us per launch at B = 32, T = 800, M = 80: two launches per loss on the compute stream).  Same taps in the same order, the
// same skipped (out-of-image) taps, the same fmaf chain per accumulator: bit-identical to the per-pixel gather it replaces.
constexpr int SS_T = 16, SS_H = 5, SS_R = SS_T + 2 * SS_H, SS_LD = SS_R + 1;  // tile, halo, tile rows incl. halo, padded LDS row
__global__ void __launch_bounds__(256) ssim_filter_kernel(SsimArgs a) {
    __shared__ float us[SS_R][SS_LD], vs[SS_R][SS_LD];
    const int b = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const float *p1 = a.img1 + (int64_t)b * a.H * a.W, *p2 = a.img2 + (int64_t)b * a.H * a.W;
    for (int k = threadIdx.x; k < SS_R * SS_R; k += 256) {
        const int ly = k / SS_R, lx = k - ly * SS_R;
        const int yy = y0 - SS_H + ly, xx = x0 - SS_H + lx;
        const bool in = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
        us[ly][lx] = in ? p1[(int64_t)yy * a.W + xx] + a.bias : 0.0f;
        vs[ly][lx] = in ? p2[(int64_t)yy * a.W + xx] + a.bias : 0.0f;
    }
    __syncthreads();
    const int tx = threadIdx.x & (SS_T - 1), ty = threadIdx.x >> 4;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.W || y >= a.H) return;
    // no range tests on the taps: an out-of-image cell of the tile holds +0, and fmaf(w, +0, acc) returns acc bit for bit (the accumulators
    // start at +0 and (+0) + (+-0) = +0 in round-to-nearest, so not even the sign of a zero can differ from skipping the tap)
    float m1 = 0, m2 = 0, q11 = 0, q22 = 0, q12 = 0;
#pragma unroll
    for (int dy = -5; dy <= 5; ++dy) {
#pragma unroll
        for (int dx = -5; dx <= 5; ++dx) {
            const float wgt = c_gauss[dy + 5] * c_gauss[dx + 5];
            const float u = us[ty + dy + SS_H][tx + dx + SS_H], v = vs[ty + dy + SS_H][tx + dx + SS_H];
            m1 = fmaf(wgt, u, m1); m2 = fmaf(wgt, v, m2);
            q11 = fmaf(wgt, u * u, q11); q22 = fmaf(wgt, v * v, q22); q12 = fmaf(wgt, u * v, q12);
        }
    }
    const int64_t i = ((int64_t)b * a.H + y) * a.W + x;
    a.mu1[i] = m1; a.mu2[i] = m2; a.s11[i] = q11; a.s22[i] = q22; a.s12[i] = q12;
}
// per pixel: ssim value (-> one_minus) and the partials of ssim w.r.t. (mu1, E[x^2], E[xy])
__global__ void __launch_bounds__(256) ssim_map_kernel(const float *mu1, const float *mu2, const float *s11,
                                                       const float *s22, const float *s12, float *one_minus,
                                                       float *d_mu1, float *d_s11, float *d_s12, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float C1 = 0.0001f, C2 = 0.0009f;
    const float m1 = mu1[i], m2 = mu2[i];
    const float v1 = s11[i] - m1 * m1, v2 = s22[i] - m2 * m2, cv = s12[i] - m1 * m2;
    const float A1 = 2.0f * m1 * m2 + C1, A2 = 2.0f * cv + C2;
    const float B1 = m1 * m1 + m2 * m2 + C1, B2 = v1 + v2 + C2;
    const float s = (A1 * A2) / (B1 * B2);
    one_minus[i] = 1.0f - s;
    if (d_mu1) {
        // ds/dm1 with v1 = q11 - m1^2, cv = q12 - m1 m2 held through q11,q12:
        const float dA1 = 2.0f * m2, dA2 = -2.0f * m2, dB1 = 2.0f * m1, dB2 = -2.0f * m1;
        d_mu1[i] = (dA1 * A2 + A1 * dA2) / (B1 * B2) - s * (dB1 / B1 + dB2 / B2);
        d_s11[i] = -s / B2;             // dB2/dq11 = 1
        d_s12[i] = 2.0f * A1 / (B1 * B2);  // dA2/dq12 = 2
    }
}
// dimg1 = F(g*d_mu1) + 2 img1 F(g*d_s11) + img2 F(g*d_s12), F = the same (symmetric) gaussian filter, g = upstream
__global__ void __launch_bounds__(256) ssim_bwd_kernel(const float *img1, const float *img2, const float *gm,
                                                       const float *g11, const float *g12, float *dimg1, int B, int H,
                                                       int W, float bias) {
    __shared__ float t0[SS_R][SS_LD], t1[SS_R][SS_LD], t2[SS_R][SS_LD];  // tiles of the three upstream maps (as ssim_filter_kernel)
    const int b = blockIdx.z, x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int64_t base = (int64_t)b * H * W;
    for (int k = threadIdx.x; k < SS_R * SS_R; k += 256) {
        const int ly = k / SS_R, lx = k - ly * SS_R;
        const int yy = y0 - SS_H + ly, xx = x0 - SS_H + lx;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const int64_t j = base + (int64_t)(in ? yy : 0) * W + (in ? xx : 0);
        t0[ly][lx] = in ? gm[j] : 0.0f;
        t1[ly][lx] = in ? g11[j] : 0.0f;
        t2[ly][lx] = in ? g12[j] : 0.0f;
    }
    __syncthreads();
    const int tx = threadIdx.x & (SS_T - 1), ty = threadIdx.x >> 4;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) return;
    float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
    for (int dy = -5; dy <= 5; ++dy) {
#pragma unroll
        for (int dx = -5; dx <= 5; ++dx) {  // (no range tests: see ssim_filter_kernel)
            const float wgt = c_gauss[dy + 5] * c_gauss[dx + 5];
            a0 = fmaf(wgt, t0[ty + dy + SS_H][tx + dx + SS_H], a0);
            a1 = fmaf(wgt, t1[ty + dy + SS_H][tx + dx + SS_H], a1);
            a2 = fmaf(wgt, t2[ty + dy + SS_H][tx + dx + SS_H], a2);
        }
    }
    const int64_t i = base + (int64_t)y * W + x;
    dimg1[i] = a0 + 2.0f * (img1[i] + bias) * a1 + (img2[i] + bias) * a2;
}

// duration losses (speech_editing_base.py:58-90): one block per utterance.
// out[0] += sum_j nonpad (log(dp+1) - log(dg+1))^2 ; out[1] += sum nonpad ; out[2] += sum_w wmask (..)^2 ; out[3] += sum wmask
// ddur (optional): gradient of  lam_p * out0/out1 + lam_w * out2/out3  given the FINAL sums in `sums` (second pass).
__global__ void __launch_bounds__(256) dur_loss_kernel(const float *dur_pred, const int64_t *mel2ph, const int64_t *txt,
                                                       const int64_t *word_id, float *sums, const float *final_sums,
                                                       float *ddur, int T, int T_txt, int n_words, float lam_p,
                                                       float lam_w, float gscale, float *part = nullptr) {
    extern __shared__ float sh[];  // dur_gt[T_txt+1] | wp[n_words+1] | wg[n_words+1]
    float *dg = sh, *wp = sh + (T_txt + 1), *wg = wp + (n_words + 1);
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j <= T_txt; j += 256) dg[j] = 0.0f;
    for (int j = threadIdx.x; j <= n_words; j += 256) { wp[j] = 0.0f; wg[j] = 0.0f; }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        const int64_t m = mel2ph[(int64_t)b * T + t];
        if (m >= 0 && m <= T_txt) atomicAdd(&dg[(int)m], 1.0f);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < T_txt; j += 256) {
        const float np = txt[(int64_t)b * T_txt + j] != 0 ? 1.0f : 0.0f;
        const int wi = (int)word_id[(int64_t)b * T_txt + j];
        atomicAdd(&wg[wi], dg[j + 1] * np);  // frame counts: integer-valued, exact in any order
        // predicted word durations: a word is one contiguous run of tokens (word_id = cumsum(sil) * (1 - sil)), summed by
        // the thread of its first token in token order -> the same bits every run (a float atomicAdd is order-dependent)
        if (wi > 0 && (j == 0 || (int)word_id[(int64_t)b * T_txt + j - 1] != wi)) {
            float acc = 0.0f;
            for (int k = j; k < T_txt && (int)word_id[(int64_t)b * T_txt + k] == wi; ++k) acc += dur_pred[(int64_t)b * T_txt + k];
            wp[wi] = acc;
        }
    }
    __syncthreads();
    if (!ddur) {
        float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int j = threadIdx.x; j < T_txt; j += 256) {
            const float np = txt[(int64_t)b * T_txt + j] != 0 ? 1.0f : 0.0f;
            const float d = logf(dur_pred[(int64_t)b * T_txt + j] + 1.0f) - logf(dg[j + 1] * np + 1.0f);
            s0 += d * d * np; s1 += np;
        }
        for (int wi = 1 + threadIdx.x; wi <= n_words; wi += 256) {
            const float wm = wg[wi] > 0.0f ? 1.0f : 0.0f;
            const float d = logf(wp[wi] + 1.0f) - logf(wg[wi] + 1.0f);
            s2 += d * d * wm; s3 += wm;
        }
        // block sum in a fixed order, then one slot per utterance (part) or the order-dependent atomics
        __shared__ float red4[4][256];
        red4[0][threadIdx.x] = s0; red4[1][threadIdx.x] = s1; red4[2][threadIdx.x] = s2; red4[3][threadIdx.x] = s3;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st)
                for (int k = 0; k < 4; ++k) red4[k][threadIdx.x] += red4[k][threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x < 4) {
            if (part) part[(int64_t)b * 4 + threadIdx.x] = red4[threadIdx.x][0];
            else atomicAdd(&sums[threadIdx.x], red4[threadIdx.x][0]);
        }
    } else {
        for (int j = threadIdx.x; j < T_txt; j += 256) {
            const float np = txt[(int64_t)b * T_txt + j] != 0 ? 1.0f : 0.0f;
            const float dp = dur_pred[(int64_t)b * T_txt + j];
            float g = lam_p * np * 2.0f * (logf(dp + 1.0f) - logf(dg[j + 1] * np + 1.0f)) / (dp + 1.0f) / final_sums[1];
            const int wi = (int)word_id[(int64_t)b * T_txt + j];
            if (lam_w > 0.0f && wi > 0 && wg[wi] > 0.0f)
                g += lam_w * 2.0f * (logf(wp[wi] + 1.0f) - logf(wg[wi] + 1.0f)) / (wp[wi] + 1.0f) / final_sums[3];
            ddur[(int64_t)b * T_txt + j] = g * gscale;
        }
    }
}

// pitch losses (speech_editing_base.py:92-108) on channel-major pitch_pred [B][2][T]:
// sums[0] += sum nonpad * bce(logit, uv) ; sums[1] += sum nonpad ; sums[2] += sum nv |f0p - f0| ; sums[3] += sum nv
// second pass (dpp != NULL): dpp[b][0][t] = lam_f0 * nv * sign / sums[3] ; dpp[b][1][t] = lam_uv * nonpad * (sigmoid - uv) / sums[1]
__global__ void __launch_bounds__(256) pitch_loss_kernel(const float *pp, const float *f0, const float *uv,
                                                         const int64_t *mel2ph, float *sums, const float *final_sums,
                                                         float *dpp, int B, int T, float lam_uv, float lam_f0,
                                                         float gscale, float *part = nullptr) {
    __shared__ float wsum[4][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (i < (int64_t)B * T) {
        const int b = (int)(i / T), t = (int)(i % T);
        const float np = mel2ph[i] != 0 ? 1.0f : 0.0f;
        const float lg = pp[((int64_t)b * 2 + 1) * T + t], fp = pp[((int64_t)b * 2) * T + t];
        const float u = uv[i];
        const float nv = np * (u == 0.0f ? 1.0f : 0.0f);
        if (!dpp) {
            // bce_with_logits = max(x,0) - x*u + log(1 + exp(-|x|))
            s0 = np * (fmaxf(lg, 0.0f) - lg * u + log1pf(expf(-fabsf(lg))));
            s1 = np;
            s2 = nv * fabsf(fp - f0[i]);
            s3 = nv;
        } else {
            const float d = fp - f0[i];
            dpp[((int64_t)b * 2) * T + t] = gscale * lam_f0 * nv * (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) / final_sums[3];
            dpp[((int64_t)b * 2 + 1) * T + t] = gscale * lam_uv * np * (dev_sigmoid(lg) - u) / final_sums[1];
        }
    }
    if (!dpp) {
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); s3 += __shfl_xor(s3, off);
        }
        if (part) {  // one slot per block, waves combined in wave order
            if ((threadIdx.x & 63) == 0) {
                const int wv = threadIdx.x >> 6;
                wsum[0][wv] = s0; wsum[1][wv] = s1; wsum[2][wv] = s2; wsum[3][wv] = s3;
            }
            __syncthreads();
            if (threadIdx.x < 4)
                part[(int64_t)blockIdx.x * 4 + threadIdx.x] =
                    ((wsum[threadIdx.x][0] + wsum[threadIdx.x][1]) + wsum[threadIdx.x][2]) + wsum[threadIdx.x][3];
        } else if ((threadIdx.x & 63) == 0) {
            atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[2], s2); atomicAdd(&sums[3], s3);
        }
    }
}

// ---- deterministic scatter-add of [B][T][C] rows into a table (embedding / alignment-gather backward) --------------
// doutT is the gradient TRANSPOSED to [B][T][C] (channels contiguous): wave = (utterance b, frame segment s, 64-channel
// block); it walks its frames in order, keeps the running sum of the current run of equal rows in registers and adds it
// to ITS OWN partial table part[(b * S + s)][row][c] (no other wave touches that slice) -- coalesced 256-byte accesses,
// no atomics.  scatter_reduce_kernel then adds the slices to the table in slice order.  mode 0: row = idx (embedding:
// out-of-range clamped, padding_idx skipped); mode 1: row = idx - 1, idx == 0 skipped (expand_states: mel2ph is 1-based)
__global__ void __launch_bounds__(256) scatter_rows_kernel(const int64_t *__restrict__ idx, const float *__restrict__ doutT,
                                                           float *__restrict__ part, int B, int T, int C, int n_rows, int S, float scale,
                                                           int padding_idx, int mode) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int cblocks = (C + 63) / 64;
    if (wv >= B * S * cblocks) return;
    const int cb = wv % cblocks, s = (wv / cblocks) % S, b = wv / (cblocks * S);
    const int c = cb * 64 + lane;
    const bool cv = c < C;
    const int seg = (T + S - 1) / S, t_begin = s * seg, t_end = min(T, t_begin + seg);
    const int64_t *ib = idx + (int64_t)b * T;
    const float *dp = doutT + (int64_t)b * T * C + (cv ? c : 0);
    float *tab = part + (int64_t)(b * S + s) * n_rows * C + (cv ? c : 0);
    int cur = -1;
    float acc = 0.0f;
    auto row_of = [&](int64_t r64) {
        if (mode == 0) { const int row = (int)(r64 < 0 ? 0 : (r64 >= n_rows ? n_rows - 1 : r64)); return row == padding_idx ? -1 : row; }
        return (r64 > 0 && r64 <= n_rows) ? (int)r64 - 1 : -1;
    };
    // frames in batches of 8: the eight index / gradient loads are issued together (with frame-level indices -- pitch bins -- every
    // frame starts a new run, and one load-then-store round trip per frame made the walk 78 us); the adds keep the frame order
    int t = t_begin;
    for (; t + 8 <= t_end; t += 8) {
        int64_t r64[8];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { r64[e] = ib[t + e]; v[e] = dp[(int64_t)(t + e) * C]; }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = row_of(r64[e]);
            if (row != cur) {  // wave-uniform
                if (cur >= 0 && cv) tab[(int64_t)cur * C] += acc;
                cur = row;
                acc = 0.0f;
            }
            acc = fmaf(scale, v[e], acc);
        }
    }
    for (; t < t_end; ++t) {
        const int row = row_of(ib[t]);
        if (row != cur) {
            if (cur >= 0 && cv) tab[(int64_t)cur * C] += acc;
            cur = row;
            acc = 0.0f;
        }
        acc = fmaf(scale, dp[(int64_t)t * C], acc);
    }
    if (cur >= 0 && cv) tab[(int64_t)cur * C] += acc;
}
// table[g][i] += sum_{k < K} part[g * K + k][i]  (slice order).  Embedding: one group, K = B * S slices; alignment gather:
// one group per utterance (it owns its rows), K = S segment slices.
__global__ void __launch_bounds__(256) scatter_reduce_kernel(const float *part, float *table, int64_t n, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int g = blockIdx.y;
    const float *pp = part + (int64_t)g * K * n + i;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // four interleaved chains over the slices: four loads in flight, one fixed association
    int k = 0;
    for (; k + 3 < K; k += 4) {
        s0 += pp[(int64_t)k * n];
        s1 += pp[(int64_t)(k + 1) * n];
        s2 += pp[(int64_t)(k + 2) * n];
        s3 += pp[(int64_t)(k + 3) * n];
    }
    for (; k < K; ++k) s0 += pp[(int64_t)k * n];
    table[(int64_t)g * n + i] += (s0 + s1) + (s2 + s3);
}

// ---- optimizer ---------------------------------------------------------------------------------------------------
// sum of squares of a flat buffer -> out[0] (atomic)
__global__ void __launch_bounds__(256) sumsq_kernel(const float *g, float *out, int64_t n, float *part = nullptr) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s = fmaf(g[i], g[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) { if (part) part[blockIdx.x] = red[0]; else atomicAdd(out, red[0]); }
}
// AdamW (torch.optim.AdamW semantics, amsgrad off) over a flat buffer; grads are first scaled by
// clip = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))  (torch.nn.utils.clip_grad_norm_), sumsq optional.
// hyper != NULL (set_adamw_dev): lr and the two bias corrections come from device memory [lr, bc1, bc2] -- the step of a captured
// graph, whose kernel arguments are frozen, reads the values of the update it is replayed for
__global__ void __launch_bounds__(256) adamw_kernel(float *p, const float *g, float *m, float *v, int64_t n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    const float *sumsq, float max_norm, float grad_scale, const float *hyper) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }
    float clip = 1.0f;
    if (sumsq && max_norm > 0.0f) {
        const float c = max_norm / (sqrtf(sumsq[0]) * grad_scale + 1e-6f);
        clip = c < 1.0f ? c : 1.0f;
    }
    const float gi = g[i] * grad_scale * clip;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
}

// ---- every residual layer's diffusion_projection of the step embedding at once ---------------------------------------------------
// The reference applies one Linear(C, C) per residual layer to the [N, C] step embedding (diffnet.py:66,72).  As L separate 1x1 convs
// over N = 32 "frames" that was, per training step, L x (forward 17 us + input gradient 17 us + weight gradient 10 us + its slice
// reduce 5 us + bias sums 8 us + the fan-out adds): 2-block launches that wait on memory round trips, ~1.2 ms of a 13.5 ms step.
// Here: one forward launch (a wave per output row, lanes over the input channels), and for the backward one launch for the L partial
// input gradients (+ their ordered sum) and one for all weight / bias gradients.  fp32 FMAs throughout; fixed summation orders.
//   h [C][N], W_l = w + l w_ls ([C][C] row-major), b_l = b + l b_ls, out / g [N][L C]
constexpr int SP_NMAX = 64;

// NT = N rounded up to 32 / 64: the rows n >= N of the LDS tiles are zero, the loops run over NT without per-element tests
template <int NT>
__global__ void __launch_bounds__(256) step_proj_fwd_kernel(const float *h, const float *w, int64_t w_ls, const float *b, int64_t b_ls,
                                                            float *out, int L, int C, int N) {
    extern __shared__ float sp_hs[];  // [NT][C]: h transposed (lanes walk the channels)
    for (int i = threadIdx.x; i < C * NT; i += 256) {
        const int n = i / C, c = i % C;
        sp_hs[i] = n < N ? h[(int64_t)c * N + n] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = 0; q < 4; ++q) {
        const int row = (blockIdx.x * 4 + wave) * 4 + q;  // (l, co)
        if (row >= L * C) return;
        const int l = row / C, co = row % C;
        const float *wr = w + (int64_t)l * w_ls + (int64_t)co * C;
        float acc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = 0.0f;
        for (int c0 = 4 * lane; c0 < C; c0 += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + c0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 hv = *reinterpret_cast<const f32x4 *>(sp_hs + n * C + c0);
                acc[n] += ((wv[0] * hv[0] + wv[1] * hv[1]) + wv[2] * hv[2]) + wv[3] * hv[3];
            }
        }
        float mine = 0.0f;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float v = acc[n];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            mine = lane == n ? v : mine;
        }
        if (lane < N) out[(int64_t)lane * L * C + row] = mine + b[(int64_t)l * b_ls + co];
    }
}

// part[(l * gridDim.z + cq)][ci][n] = sum over the 64 output channels co of block cq of W_l[co][ci] g[n][l C + co]
__global__ void __launch_bounds__(256) step_proj_bwd_dh_kernel(const float *g, const float *w, int64_t w_ls, float *part, int L, int C,
                                                               int N) {
    __shared__ float gs[SP_NMAX][64];
    const int cil = threadIdx.x & 63, ng = threadIdx.x >> 6;  // n = ng, ng + 4, ...
    const int ci = blockIdx.x * 64 + cil, l = blockIdx.y, co0 = blockIdx.z * 64;
    for (int i = threadIdx.x; i < N * 64; i += 256) gs[i >> 6][i & 63] = g[(int64_t)(i >> 6) * L * C + l * C + co0 + (i & 63)];
    float wv[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wv[k] = w[(int64_t)l * w_ls + (int64_t)(co0 + k) * C + ci];
    __syncthreads();
    float *po = part + ((int64_t)(l * gridDim.z + blockIdx.z) * C + ci) * N;
    for (int n = ng; n < N; n += 4) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) s += wv[k] * gs[n][k];
        po[n] = s;
    }
}

// dW_l[co][ci] += sum_n g[n][l C + co] h[ci][n];  db_l[co] += sum_n g[n][l C + co]     (block: layer l, 16 rows co)
template <int NT>
__global__ void __launch_bounds__(256) step_proj_bwd_dw_kernel(const float *h, const float *g, float *dw, int64_t dw_ls, float *db,
                                                               int64_t db_ls, int L, int C, int N) {
    __shared__ float gs[16][NT];
    const int l = blockIdx.x, co0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 16 * NT; i += 256) {
        const int k = i / NT, n = i % NT;
        gs[k][n] = n < N ? g[(int64_t)n * L * C + l * C + co0 + k] : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < NT; ++n) s += gs[threadIdx.x][n];
        db[(int64_t)l * db_ls + co0 + threadIdx.x] += s;
    }
    for (int ci = threadIdx.x; ci < C; ci += 256) {
        float hv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) hv[n] = h[(int64_t)ci * N + (n < N ? n : N - 1)];  // (n >= N: multiplied by a zero of gs)
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            float s = 0.0f;
#pragma unroll
            for (int n = 0; n < NT; ++n) s += gs[k][n] * hv[n];
            dw[(int64_t)l * dw_ls + (int64_t)(co0 + k) * C + ci] += s;
        }
    }
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
// slice plan of the fp32 MFMA weight-gradient kernel (shared by the atomic and the deterministic entry points)
static void wgrad_f32_plan(WgradArgs &a, bool for_atomics) {
    a.n_chunks_t = (a.T + WG_KC - 1) / WG_KC;
    const int total = a.B * a.n_chunks_t;
    const int tiles = a.K * ((a.Cin + 63) / 64) * ((a.Cout + 127) / 128);
    // split-K slices: every block ends with one atomicAdd per weight element, and the adds of different XCDs on one
    // address serialise at the memory side -- ~2.5 blocks per CU beat 8 (swept: 2048 -> 637, 1024 -> 659, 512 -> 663,
    // 256 -> 616 training samples/s)
    int target_blocks = 640;
    int slices = (target_blocks + tiles - 1) / tiles;
    if (slices > total) slices = total;
    if (slices < 1) slices = 1;
    // one XCD per slice needs a multiple of 8 slices (the padding slices, if any, exit at once)
    bool xcd_map = slices >= 8;
    if (xcd_map) slices = (slices + 7) / 8 * 8 <= total ? (slices + 7) / 8 * 8 : slices / 8 * 8;
    a.chunks_per_slice = (total + slices - 1) / slices;
    slices = (total + a.chunks_per_slice - 1) / a.chunks_per_slice;
    if (xcd_map) slices = (slices + 7) / 8 * 8;
    a.gx = a.K * ((a.Cin + 63) / 64);
    a.gy = (a.Cout + 127) / 128;
    a.gz = slices;
    a.xcd_map = xcd_map ? 1 : 0;
}

extern "C" int set_conv1d_wgrad(const float *g, const float *x, const float *chan_add, float *dw, int32_t B, int32_t Cin,
                                int32_t Cout, int32_t K, int32_t dil, int32_t pad, int32_t T, int32_t T_in, int32_t pro,
                                float pro_param, int32_t impl, void *stream) {
    SET_REQUIRE(g && x && dw && B > 0 && Cin > 0 && Cout > 0 && K > 0 && T > 0 && T_in > 0, "set_conv1d_wgrad");
    WgradArgs a = {g, x, chan_add, dw, B, Cin, Cout, K, dil, pad, T, T_in, pro, pro_param, 0, 0, 0, 0, 0, 0, nullptr};
    hipStream_t s = (hipStream_t)stream;
    if (impl != SET_IMPL_MFMA) {
        hipLaunchKernelGGL(conv1d_wgrad_naive_kernel, dim3(set_blocks((int64_t)Cout * Cin * K, 256)), dim3(256), 0, s, a);
        return set_check_launch("set_conv1d_wgrad(naive)");
    }
    wgrad_f32_plan(a, true);
    hipLaunchKernelGGL(conv1d_wgrad_mfma_kernel, dim3((unsigned)a.gx * a.gy * a.gz), dim3(256), 0, s, a);
    return set_check_launch("set_conv1d_wgrad(mfma)");
}

// deterministic fp32 path (set_conv1d_wgrad_det in bf16.hip): number of slices, and the launch that fills partial[S][...]
int wgrad_f32_slices(int B, int Cin, int Cout, int K, int T) {
    WgradArgs a = {nullptr, nullptr, nullptr, nullptr, B, Cin, Cout, K, 1, 0, T, T, 0, 0.0f, 0, 0, 0, 0, 0, 0, nullptr};
    wgrad_f32_plan(a, false);
    return a.gz;
}
int launch_wgrad_f32_partial(const float *g, const float *x, const float *chan_add, float *partial, int B, int Cin, int Cout,
                             int K, int dil, int pad, int T, int T_in, int pro, float pro_param, hipStream_t s) {
    WgradArgs a = {g, x, chan_add, nullptr, B, Cin, Cout, K, dil, pad, T, T_in, pro, pro_param, 0, 0, 0, 0, 0, 0, partial};
    wgrad_f32_plan(a, false);
    hipLaunchKernelGGL(conv1d_wgrad_mfma_kernel, dim3((unsigned)a.gx * a.gy * a.gz), dim3(256), 0, s, a);
    return set_check_launch("set_conv1d_wgrad_det(f32)");
}

extern "C" int set_channel_sum(const float *x, float *out, int32_t B, int32_t C, int32_t T, void *stream) {
    SET_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "set_channel_sum");
    int slices = (2048 + C - 1) / C;  // ~2048 blocks in total
    if (slices > B) slices = B;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, slices), dim3(256), 0, (hipStream_t)stream, x, out, B, C, T);
    return set_check_launch("set_channel_sum");
}
extern "C" int set_partial_rows_sum(const float *part, float *out, int32_t groups, int32_t rows, int32_t cols, int32_t accumulate,
                                    float scale, void *stream);  // diffnet_bf16.hip

extern "C" int set_step_proj_fwd(const float *h, const float *w, int64_t w_ls, const float *b, int64_t b_ls, float *out, int32_t L,
                                 int32_t C, int32_t N, void *stream) {
    SET_REQUIRE(h && w && b && out && L > 0 && C > 0 && N > 0, "set_step_proj_fwd");
    const int NT = N <= 32 ? 32 : 64;
    if (C % 64 != 0 || N > SP_NMAX || (size_t)C * NT * 4 > 64 * 1024)
        return set_fail(SET_E_UNSUPPORTED, "set_step_proj_fwd", "needs C % 64 == 0, N <= 64, C * N <= 16384");
    const dim3 grid(set_blocks((int64_t)L * C, 16));
    if (NT == 32) hipLaunchKernelGGL(step_proj_fwd_kernel<32>, grid, dim3(256), (size_t)C * NT * 4, (hipStream_t)stream, h, w, w_ls, b, b_ls, out, L, C, N);
    else hipLaunchKernelGGL(step_proj_fwd_kernel<64>, grid, dim3(256), (size_t)C * NT * 4, (hipStream_t)stream, h, w, w_ls, b, b_ls, out, L, C, N);
    return set_check_launch("set_step_proj_fwd");
}

extern "C" int64_t set_step_proj_bwd_scratch_floats(int32_t L, int32_t C, int32_t N) { return (int64_t)L * (C / 64) * C * N; }

// the two halves of set_step_proj_bwd as entry points of their own: the input gradient feeds the chain of backward kernels, the weight /
// bias gradients are parameter gradients nothing reads before the optimizer -- the host side launches them on its second stream
extern "C" int set_step_proj_bwd_dh(const float *g, const float *w, int64_t w_ls, float *dh, float *scratch, int32_t L, int32_t C, int32_t N,
                                    void *stream) {
    SET_REQUIRE(g && w && dh && scratch && L > 0 && C > 0 && N > 0, "set_step_proj_bwd_dh");
    if (C % 64 != 0 || N > SP_NMAX) return set_fail(SET_E_UNSUPPORTED, "set_step_proj_bwd_dh", "needs C % 64 == 0, N <= 64");
    hipLaunchKernelGGL(step_proj_bwd_dh_kernel, dim3(C / 64, L, C / 64), dim3(256), 0, (hipStream_t)stream, g, w, w_ls, scratch, L, C, N);
    const int rc = set_check_launch("set_step_proj_bwd_dh");
    return rc ? rc : set_partial_rows_sum(scratch, dh, 1, L * (C / 64), C * N, 0, 1.0f, stream);  // dh = sum of the partials, in (l, quarter) order
}
extern "C" int set_step_proj_bwd_dw(const float *h, const float *g, float *dw, int64_t dw_ls, float *db, int64_t db_ls, int32_t L, int32_t C,
                                    int32_t N, void *stream) {
    SET_REQUIRE(h && g && dw && db && L > 0 && C > 0 && N > 0, "set_step_proj_bwd_dw");
    if (C % 64 != 0 || N > SP_NMAX) return set_fail(SET_E_UNSUPPORTED, "set_step_proj_bwd_dw", "needs C % 64 == 0, N <= 64");
    hipStream_t s = (hipStream_t)stream;
    if (N <= 32) hipLaunchKernelGGL(step_proj_bwd_dw_kernel<32>, dim3(L, C / 16), dim3(256), 0, s, h, g, dw, dw_ls, db, db_ls, L, C, N);
    else hipLaunchKernelGGL(step_proj_bwd_dw_kernel<64>, dim3(L, C / 16), dim3(256), 0, s, h, g, dw, dw_ls, db, db_ls, L, C, N);
    return set_check_launch("set_step_proj_bwd_dw");
}
extern "C" int set_step_proj_bwd(const float *h, const float *g, const float *w, int64_t w_ls, float *dh, float *dw, int64_t dw_ls,
                                 float *db, int64_t db_ls, float *scratch, int32_t L, int32_t C, int32_t N, void *stream) {
    SET_REQUIRE(h && g && w && dh && dw && db && scratch && L > 0 && C > 0 && N > 0, "set_step_proj_bwd");
    const int rc = set_step_proj_bwd_dh(g, w, w_ls, dh, scratch, L, C, N, stream);
    return rc ? rc : set_step_proj_bwd_dw(h, g, dw, dw_ls, db, db_ls, L, C, N, stream);
}

// Deterministic variants: per-block partial results in `scratch`, combined in block order by set_partial_rows_sum.
extern "C" int set_channel_sum_det(const float *x, float *out, int32_t B, int32_t C, int32_t T, float *scratch, void *stream) {
    SET_REQUIRE(x && out && scratch && B > 0 && C > 0 && T > 0, "set_channel_sum_det");
    int slices = (2048 + C - 1) / C;
    if (slices > B) slices = B;
    if (slices < 1) slices = 1;  // scratch: slices * C <= 2048 + C floats
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, slices), dim3(256), 0, (hipStream_t)stream, x, out, B, C, T, scratch);
    const int rc = set_check_launch("set_channel_sum_det");
    return rc ? rc : set_partial_rows_sum(scratch, out, 1, slices, C, 1, 1.0f, stream);
}
extern "C" int set_weighted_sum_det(const float *x, const float *w, float *out, int64_t n, int64_t inner, float *scratch,
                                    void *stream) {
    SET_REQUIRE(x && out && scratch && n > 0 && inner > 0, "set_weighted_sum_det");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;  // scratch: <= 1024 floats
    hipLaunchKernelGGL(weighted_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, out, n, inner, scratch);
    const int rc = set_check_launch("set_weighted_sum_det");
    return rc ? rc : set_partial_rows_sum(scratch, out, 1, (int)blocks, 1, 1, 1.0f, stream);
}
extern "C" int set_sumsq_det(const float *g, float *out, int64_t n, float *scratch, void *stream) {
    SET_REQUIRE(g && out && scratch && n > 0, "set_sumsq_det");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // scratch: <= 2048 floats
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, out, n, scratch);
    const int rc = set_check_launch("set_sumsq_det");
    return rc ? rc : set_partial_rows_sum(scratch, out, 1, (int)blocks, 1, 1, 1.0f, stream);
}
// sums-only passes of the two loss kernels with ordered partials: scratch >= 4 * B (dur) / 4 * ceil(B*T/256) (pitch) floats
extern "C" int set_dur_loss_sums_det(const float *dur_pred, const int64_t *mel2ph, const int64_t *txt, const int64_t *word_id,
                                     float *sums, int32_t B, int32_t T, int32_t T_txt, int32_t n_words, float *scratch,
                                     void *stream) {
    SET_REQUIRE(dur_pred && mel2ph && txt && word_id && sums && scratch && B > 0 && T > 0 && T_txt > 0 && n_words >= 0,
                "set_dur_loss_sums_det");
    const size_t lds = (size_t)(T_txt + 1 + 2 * (n_words + 1)) * sizeof(float);
    SET_REQUIRE(lds < 56000, "set_dur_loss_sums_det(T_txt too large)");
    hipLaunchKernelGGL(dur_loss_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dur_pred, mel2ph, txt, word_id, sums,
                       (const float *)nullptr, (float *)nullptr, T, T_txt, n_words, 0.0f, 0.0f, 1.0f, scratch);
    const int rc = set_check_launch("set_dur_loss_sums_det");
    return rc ? rc : set_partial_rows_sum(scratch, sums, 1, B, 4, 1, 1.0f, stream);
}
extern "C" int set_pitch_loss_sums_det(const float *pp, const float *f0, const float *uv, const int64_t *mel2ph, float *sums,
                                       int32_t B, int32_t T, float *scratch, void *stream) {
    SET_REQUIRE(pp && f0 && uv && mel2ph && sums && scratch && B > 0 && T > 0, "set_pitch_loss_sums_det");
    const unsigned nb = set_blocks((int64_t)B * T, 256);
    hipLaunchKernelGGL(pitch_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pp, f0, uv, mel2ph, sums,
                       (const float *)nullptr, (float *)nullptr, B, T, 0.0f, 0.0f, 1.0f, scratch);
    const int rc = set_check_launch("set_pitch_loss_sums_det");
    return rc ? rc : set_partial_rows_sum(scratch, sums, 1, (int)nb, 4, 1, 1.0f, stream);
}
// table[n_rows][C] (mode 0) / denc^T[B * n_rows][C] (mode 1) += scatter of doutT [B][T][C]; scratch: B * S * n_rows * C floats
// (mode 0) with S = set_scatter_rows_segments(T); it is zeroed here.
// (a kernel, not hipMemsetAsync: as a memset NODE of a captured training step the 6 MB clear was not ordered against eager work of
// another stream between two replays -- the partial tables came out as uninitialised memory; tests/test_gpu_training.py)
__global__ void __launch_bounds__(256) zero_f32x4_kernel(f32x4 *p, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
}
extern "C" int32_t set_scatter_rows_segments(int32_t T) { int s = T / 128; return s < 1 ? 1 : (s > 8 ? 8 : s); }
extern "C" int set_scatter_rows_det(const int64_t *idx, const float *doutT, float *table, int32_t B, int32_t T, int32_t C,
                                    int32_t n_rows, float scale, int32_t padding_idx, int32_t mode, float *scratch,
                                    void *stream) {
    SET_REQUIRE(idx && doutT && table && scratch && B > 0 && T > 0 && C > 0 && n_rows > 0 && (mode == 0 || mode == 1),
                "set_scatter_rows_det");
    const int S = set_scatter_rows_segments(T);
    const int64_t n = (int64_t)n_rows * C;
    {
        const int64_t n4 = ((int64_t)B * S * n + 3) / 4;  // the scratch buffers are allocated with slack: rounding up to 16 bytes stays inside
        hipLaunchKernelGGL(zero_f32x4_kernel, dim3(set_blocks(n4, 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<f32x4 *>(scratch), n4);
    }
    const int waves = B * S * ((C + 63) / 64);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, idx, doutT, scratch, B, T,
                       C, n_rows, S, scale, padding_idx, mode);
    int rc = set_check_launch("set_scatter_rows_det");
    if (rc) return rc;
    if (mode == 0)
        hipLaunchKernelGGL(scatter_reduce_kernel, dim3(set_blocks(n, 256), 1), dim3(256), 0, (hipStream_t)stream, scratch, table,
                           n, B * S);
    else  // every utterance owns its rows: the S segment slices of utterance b go to table[b]
        hipLaunchKernelGGL(scatter_reduce_kernel, dim3(set_blocks(n, 256), B), dim3(256), 0, (hipStream_t)stream, scratch, table,
                           n, S);
    return set_check_launch("set_scatter_rows_det(reduce)");
}

extern "C" int set_row_sum(const float *x, float *out, int64_t rows, int32_t T, float scale, void *stream) {
    SET_REQUIRE(x && out && rows > 0 && T > 0, "set_row_sum");
    hipLaunchKernelGGL(row_sum_kernel, dim3(set_blocks(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, T, scale);
    return set_check_launch("set_row_sum");
}
extern "C" int set_conv_epilogue_bwd(const float *dy, const float *y, const float *mask, float *g, int32_t B, int32_t C,
                                     int32_t T, int32_t act, float alpha, void *stream) {
    SET_REQUIRE(dy && g && B > 0 && C > 0 && T > 0 && (act == SET_ACT_NONE || (act == SET_ACT_RELU && y)),
                "set_conv_epilogue_bwd");
    if (T % 4 == 0 && set_aligned16(dy, y, mask) && set_aligned16(g, g, g))
        hipLaunchKernelGGL(conv_epilogue_bwd_vec4_kernel, dim3(set_blocks((int64_t)B * C * T / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, y,
                           mask, g, (int64_t)B * C * T / 4, (int64_t)C * T, T, act, alpha);
    else
        hipLaunchKernelGGL(conv_epilogue_bwd_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                           (hipStream_t)stream, dy, y, mask, g, B, C, T, act, alpha);
    return set_check_launch("set_conv_epilogue_bwd");
}
extern "C" int set_act_fwd(const float *z, float *y, int64_t n, int32_t act, float p, void *stream) {
    SET_REQUIRE(z && y && n > 0, "set_act_fwd");
    if (n % 4 == 0 && set_aligned16(z, y, y))
        hipLaunchKernelGGL(act_fwd_vec4_kernel, dim3(set_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, z, y, n / 4, act, p);
    else
        hipLaunchKernelGGL(act_fwd_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, z, y, n, act, p);
    return set_check_launch("set_act_fwd");
}
extern "C" int set_act_bwd(const float *z, const float *dy, float *dz, int64_t n, int32_t act, float p, void *stream) {
    SET_REQUIRE(z && dy && dz && n > 0, "set_act_bwd");
    if (n % 4 == 0 && set_aligned16(z, dy, dz))
        hipLaunchKernelGGL(act_bwd_vec4_kernel, dim3(set_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n / 4, act, p, 0, 1.0f);
    else
        hipLaunchKernelGGL(act_bwd_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n, act, p);
    return set_check_launch("set_act_bwd");
}
extern "C" int set_act_bwd_scaled(const float *z, const float *dy, float *dz, int64_t n, int32_t act, float p, float scale, void *stream) {
    SET_REQUIRE(z && dy && dz && n > 0, "set_act_bwd_scaled");
    if (n % 4 == 0 && set_aligned16(z, dy, dz))
        hipLaunchKernelGGL(act_bwd_vec4_kernel, dim3(set_blocks(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n / 4, act, p, 1, scale);
    else
        hipLaunchKernelGGL(act_bwd_scaled_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, z, dy, dz, n, act, p, scale);
    return set_check_launch("set_act_bwd_scaled");
}
extern "C" int set_gate_bwd(const float *y, const float *dz, float *dy, int32_t B, int32_t C, int32_t T, void *stream) {
    SET_REQUIRE(y && dz && dy && B > 0 && C > 0 && T > 0, "set_gate_bwd");
    const int64_t CT = (int64_t)C * T;
    if (CT % 4 == 0 && set_aligned16(y, dz, dy))
        hipLaunchKernelGGL(gate_bwd_vec4_kernel, dim3(set_blocks(B * CT / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, dz, dy, B * CT / 4,
                           CT / 4, CT);
    else
        hipLaunchKernelGGL(gate_bwd_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0, (hipStream_t)stream, y,
                           dz, dy, B, C, T);
    return set_check_launch("set_gate_bwd");
}
extern "C" int set_res_skip_bwd(const float *dx_out, const float *dskip, float *dx, float *d_o, int32_t B, int32_t C,
                                int32_t T, void *stream) {
    SET_REQUIRE(dx_out && dskip && dx && d_o && B > 0 && C > 0 && T > 0, "set_res_skip_bwd");
    const int64_t CT = (int64_t)C * T;
    if (CT % 4 == 0 && set_aligned16(dx_out, dskip, dx) && set_aligned16(d_o, d_o, d_o))
        hipLaunchKernelGGL(res_skip_bwd_vec4_kernel, dim3(set_blocks(B * CT / 4, 256)), dim3(256), 0, (hipStream_t)stream, dx_out, dskip, dx,
                           d_o, B * CT / 4, CT / 4, CT);
    else
        hipLaunchKernelGGL(res_skip_bwd_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0, (hipStream_t)stream,
                       dx_out, dskip, dx, d_o, B, C, T);
    return set_check_launch("set_res_skip_bwd");
}
extern "C" int64_t set_layernorm_ch_bwd_scratch(int32_t B, int32_t C, int32_t T) {
    return (int64_t)B * ((T + 15) / 16) * 2 * C;  // one row of 2 C partial sums per block; the 16-frame block shape has the most blocks
}
static int layernorm_ch_bwd_launch(const float *x, const float *gamma, const float *mask, const float *dy, float *dx, float *dgamma,
                                   float *dbeta, float *partial, int32_t B, int32_t C, int32_t T, float eps, const float *add, void *stream);
extern "C" int set_layernorm_ch_bwd(const float *x, const float *gamma, const float *mask, const float *dy, float *dx,
                                    float *dgamma, float *dbeta, float *partial, int32_t B, int32_t C, int32_t T,
                                    float eps, void *stream) {
    return layernorm_ch_bwd_launch(x, gamma, mask, dy, dx, dgamma, dbeta, partial, B, C, T, eps, nullptr, stream);
}
extern "C" int set_layernorm_ch_bwd_add(const float *x, const float *gamma, const float *mask, const float *dy, const float *add, float *dx,
                                        float *dgamma, float *dbeta, float *partial, int32_t B, int32_t C, int32_t T, float eps,
                                        void *stream) {
    SET_REQUIRE(add != nullptr, "set_layernorm_ch_bwd_add");
    return layernorm_ch_bwd_launch(x, gamma, mask, dy, dx, dgamma, dbeta, partial, B, C, T, eps, add, stream);
}
static int layernorm_ch_bwd_launch(const float *x, const float *gamma, const float *mask, const float *dy, float *dx, float *dgamma,
                                   float *dbeta, float *partial, int32_t B, int32_t C, int32_t T, float eps, const float *add, void *stream) {
    SET_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && B > 0 && C > 0 && T > 0, "set_layernorm_ch_bwd");
    SET_REQUIRE(B <= 65535, "set_layernorm_ch_bwd(B)");
    SET_REQUIRE((int64_t)(C + 32) * T * 4 < ((int64_t)1 << 31), "set_layernorm_ch_bwd (one utterance exceeds the 2 GiB of a buffer offset)");
    // C <= 256: 16 channel groups (<= 16 channels per thread in registers).  32-frame tiles in 512-thread blocks (128-byte row segments, two
    // blocks per CU) once they fill the chip; 16-frame tiles in 256-thread blocks for the small launches (the conditioner's text level:
    // 128 -> 224 blocks).  C > 256: 8 groups of <= 32 channels.  Alone on the GPU, both launches of a call (profiles/r05_ln_bwd_probe.log):
    // B=16 C=256 T=800 32.6 -> 22.1 us, B=32 C=192 T=800 48.7 -> 31.0 us, B=32 C=192 T=100 19.8 -> 11.0 us.
    // The block shape (and with it the partial rows and the summation order of dgamma / dbeta) is a function of B, C and T ALONE -- 256 = the CU
    // count of the part this is written for, as a constant: a cached per-process device query made the gradients depend on which device was
    // current at the first call (round-5 advisor item).
    constexpr int n_cu = 256;
    const int t32 = (T + LNB_FT - 1) / LNB_FT;
    const bool groups16 = C <= 256, big = groups16 && (int64_t)B * t32 >= n_cu;
    const int tiles = (groups16 && !big) ? (T + 15) / 16 : t32;
    if (big)
        hipLaunchKernelGGL((layernorm_ch_bwd_kernel<32, 16, 16>), dim3(tiles, B), dim3(512), 0, (hipStream_t)stream,
                           x, gamma, mask, dy, dx, dgamma, dbeta, partial, B, C, T, eps, add);
    else if (groups16)
        hipLaunchKernelGGL((layernorm_ch_bwd_kernel<16, 16, 16>), dim3(tiles, B), dim3(256), 0, (hipStream_t)stream,
                           x, gamma, mask, dy, dx, dgamma, dbeta, partial, B, C, T, eps, add);
    else
        hipLaunchKernelGGL((layernorm_ch_bwd_kernel<32, 8, 32>), dim3(tiles, B), dim3(256), 0, (hipStream_t)stream,
                           x, gamma, mask, dy, dx, dgamma, dbeta, partial, B, C, T, eps, add);
    if (partial) {
        const int rc = set_check_launch("set_layernorm_ch_bwd");
        if (rc) return rc;
        hipLaunchKernelGGL(lnb_partial_sum_kernel, dim3((2 * C + 63) / 64), dim3(64 * ROWS_RG), 0, (hipStream_t)stream,
                           partial, dgamma, dbeta, tiles * B, C);
    }
    return set_check_launch("set_layernorm_ch_bwd");
}
extern "C" int set_embedding_bwd(const int64_t *idx, const float *dout, float *dtable, int32_t B, int32_t T, int32_t C,
                                 int32_t n_rows, float scale, int32_t padding_idx, void *stream) {
    SET_REQUIRE(idx && dout && dtable && B > 0 && T > 0 && C > 0 && n_rows > 0, "set_embedding_bwd");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(set_blocks((int64_t)B * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, dout, dtable, B, T, C, n_rows, scale, padding_idx);
    return set_check_launch("set_embedding_bwd");
}
extern "C" int set_expand_states_bwd(const int64_t *mel2ph, const float *dout, float *denc, int32_t B, int32_t C,
                                     int32_t T_txt, int32_t T, void *stream) {
    SET_REQUIRE(mel2ph && dout && denc && B > 0 && C > 0 && T_txt > 0 && T > 0, "set_expand_states_bwd");
    hipLaunchKernelGGL(expand_states_bwd_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0,
                       (hipStream_t)stream, mel2ph, dout, denc, B, C, T_txt, T);
    return set_check_launch("set_expand_states_bwd");
}
extern "C" int set_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset, void *stream) {
    SET_REQUIRE(x && y && n > 0 && p >= 0.0f && p < 1.0f, "set_dropout");
    hipLaunchKernelGGL(dropout_kernel, dim3(set_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, p,
                       seed, offset, set_seed_delta_ptr(), set_aligned16(x, y, y) ? 1 : 0);
    return set_check_launch("set_dropout");
}
extern "C" int set_frame_weight(const float *target, float *w, int64_t frames, int32_t M, void *stream) {
    SET_REQUIRE(target && w && frames > 0 && M > 0, "set_frame_weight");
    hipLaunchKernelGGL(frame_weight_kernel, dim3(set_blocks(frames, 256)), dim3(256), 0, (hipStream_t)stream, target, w,
                       frames, M);
    return set_check_launch("set_frame_weight");
}
extern "C" int set_weighted_sum(const float *x, const float *w, float *out, int64_t n, int64_t inner, void *stream) {
    SET_REQUIRE(x && out && n > 0 && inner > 0, "set_weighted_sum");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, w, out, n, inner);
    return set_check_launch("set_weighted_sum");
}
extern "C" int set_l1_elem(const float *pred, const float *target, float *absd, float *sgn, int64_t n, void *stream) {
    SET_REQUIRE(pred && target && (absd || sgn) && n > 0, "set_l1_elem");
    hipLaunchKernelGGL(l1_elem_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, pred, target, absd, sgn, n);
    return set_check_launch("set_l1_elem");
}
extern "C" int set_scale_bcast(const float *a, const float *w, float *out, int64_t n, int64_t inner,
                               const float *scale_dev, float scale, void *stream) {
    SET_REQUIRE(a && out && n > 0 && inner > 0, "set_scale_bcast");
    hipLaunchKernelGGL(scale_bcast_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, a, w, out, n, inner,
                       scale_dev, scale);
    return set_check_launch("set_scale_bcast");
}
static int ssim_upload_window() {
    static bool done = false;
    if (done) return SET_OK;
    // utils/metrics/ssim.py:12-14: gaussian(11, 1.5), computed in fp32 like torch.Tensor([...]) / sum
    float g[11], s = 0.0f;
    for (int x = 0; x < 11; ++x) { g[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); s += g[x]; }
    for (int x = 0; x < 11; ++x) g[x] /= s;
    SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), g, sizeof(g)), "ssim window");
    done = true;
    return SET_OK;
}
extern "C" int set_ssim_filter(const float *img1, const float *img2, float bias, float *mu1, float *mu2, float *s11,
                               float *s22, float *s12, int32_t B, int32_t H, int32_t W, void *stream) {
    SET_REQUIRE(img1 && img2 && mu1 && mu2 && s11 && s22 && s12 && B > 0 && H > 0 && W > 0, "set_ssim_filter");
    int rc = ssim_upload_window();
    if (rc != SET_OK) return rc;
    SsimArgs a = {img1, img2, bias, mu1, mu2, s11, s22, s12, B, H, W};
    SET_REQUIRE(B <= 65535 && (H + SS_T - 1) / SS_T <= 65535, "set_ssim_filter(grid)");
    hipLaunchKernelGGL(ssim_filter_kernel, dim3((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B), dim3(256), 0, (hipStream_t)stream, a);
    return set_check_launch("set_ssim_filter");
}
extern "C" int set_ssim_map(const float *mu1, const float *mu2, const float *s11, const float *s22, const float *s12,
                            float *one_minus, float *d_mu1, float *d_s11, float *d_s12, int64_t n, void *stream) {
    SET_REQUIRE(mu1 && mu2 && s11 && s22 && s12 && one_minus && n > 0, "set_ssim_map");
    hipLaunchKernelGGL(ssim_map_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, mu1, mu2, s11, s22, s12,
                       one_minus, d_mu1, d_s11, d_s12, n);
    return set_check_launch("set_ssim_map");
}
extern "C" int set_ssim_bwd(const float *img1, const float *img2, float bias, const float *gm, const float *g11,
                            const float *g12, float *dimg1, int32_t B, int32_t H, int32_t W, void *stream) {
    SET_REQUIRE(img1 && img2 && gm && g11 && g12 && dimg1 && B > 0 && H > 0 && W > 0, "set_ssim_bwd");
    int rc = ssim_upload_window();
    if (rc != SET_OK) return rc;
    SET_REQUIRE(B <= 65535 && (H + SS_T - 1) / SS_T <= 65535, "set_ssim_bwd(grid)");
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, B), dim3(256), 0, (hipStream_t)stream, img1,
                       img2, gm, g11, g12, dimg1, B, H, W, bias);
    return set_check_launch("set_ssim_bwd");
}
extern "C" int set_dur_loss(const float *dur_pred, const int64_t *mel2ph, const int64_t *txt, const int64_t *word_id,
                            float *sums, const float *final_sums, float *ddur, int32_t B, int32_t T, int32_t T_txt,
                            int32_t n_words, float lam_p, float lam_w, float gscale, void *stream) {
    SET_REQUIRE(dur_pred && mel2ph && txt && word_id && B > 0 && T > 0 && T_txt > 0 && n_words >= 0, "set_dur_loss");
    SET_REQUIRE((ddur && final_sums) || (!ddur && sums), "set_dur_loss");
    const size_t lds = (size_t)(T_txt + 1 + 2 * (n_words + 1)) * sizeof(float);
    SET_REQUIRE(lds < 60000, "set_dur_loss(T_txt too large)");
    hipLaunchKernelGGL(dur_loss_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dur_pred, mel2ph, txt, word_id, sums,
                       final_sums, ddur, T, T_txt, n_words, lam_p, lam_w, gscale);
    return set_check_launch("set_dur_loss");
}
extern "C" int set_pitch_loss(const float *pp, const float *f0, const float *uv, const int64_t *mel2ph, float *sums,
                              const float *final_sums, float *dpp, int32_t B, int32_t T, float lam_uv, float lam_f0,
                              float gscale, void *stream) {
    SET_REQUIRE(pp && f0 && uv && mel2ph && B > 0 && T > 0, "set_pitch_loss");
    SET_REQUIRE((dpp && final_sums) || (!dpp && sums), "set_pitch_loss");
    hipLaunchKernelGGL(pitch_loss_kernel, dim3(set_blocks((int64_t)B * T, 256)), dim3(256), 0, (hipStream_t)stream, pp, f0,
                       uv, mel2ph, sums, final_sums, dpp, B, T, lam_uv, lam_f0, gscale);
    return set_check_launch("set_pitch_loss");
}
extern "C" int set_sumsq(const float *g, float *out, int64_t n, void *stream) {
    SET_REQUIRE(g && out && n > 0, "set_sumsq");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, out, n);
    return set_check_launch("set_sumsq");
}
extern "C" int set_adamw(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int32_t step, const float *sumsq, float max_norm,
                         float grad_scale, void *stream) {
    SET_REQUIRE(p && g && m && v && n > 0 && step >= 1, "set_adamw");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2, sumsq, max_norm, grad_scale, (const float *)nullptr);
    return set_check_launch("set_adamw");
}
extern "C" int set_adamw_hyper(float beta1, float beta2, int32_t step, float *out2) {
    SET_REQUIRE(out2 && step >= 1, "set_adamw_hyper");
    out2[0] = 1.0f - powf(beta1, (float)step);  // exactly what set_adamw computes on the host
    out2[1] = 1.0f - powf(beta2, (float)step);
    return SET_OK;
}
extern "C" int set_adamw_dev(float *p, const float *g, float *m, float *v, int64_t n, const float *hyper, float beta1, float beta2,
                             float eps, float weight_decay, const float *sumsq, float max_norm, float grad_scale, void *stream) {
    SET_REQUIRE(p && g && m && v && n > 0 && hyper, "set_adamw_dev");
    hipLaunchKernelGGL(adamw_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.0f, beta1,
                       beta2, eps, weight_decay, 1.0f, 1.0f, sumsq, max_norm, grad_scale, hyper);
    return set_check_launch("set_adamw_dev");
}
