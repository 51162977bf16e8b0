// Generic stride-1 conv1d with fused prologue/epilogue: a naive one-thread-per-output kernel
// (any shape; device-side cross-check) and an implicit-GEMM kernel on v_mfma_f32_32x32x2_f32.
// Replaces every F.conv1d / nn.Linear / nn.ConvTranspose1d (polyphase) call on the hot path -- see
// include/set_amd.h for the reference citations.
#include "common.h"
#ifndef SET_CONV_V2_STORE_HAZARD
#define SET_CONV_V2_STORE_HAZARD 0
#endif
#include <type_traits>

thread_local char g_set_err[512] = {0};

extern "C" int set_abi_version(void) { return SET_AMD_ABI_VERSION; }
extern "C" const char *set_last_error(void) { return g_set_err; }

// ------------------------------------------------------------------------------------------
// shared epilogue:  v = act((acc + bias) * alpha) ; + res ; * mask ; (accumulate)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_store(const SetConv1dArgs &a, int b, int co, int n, float acc) {
    float v = acc;
    if (a.bias) v += a.bias[co];
    v *= a.alpha;
    v = dev_act(v, a.act, a.act_param);
    if (a.res) v += a.res[(int64_t)b * a.res_bs + (int64_t)co * a.res_cs + n];
    if (a.mask) v *= a.mask[(int64_t)b * a.T_out + n];
    float *o = a.out + (int64_t)b * a.out_bs + (int64_t)co * a.out_cs + n;
    if (a.accumulate) {
        v += *o;
        if (a.out_div != 0.0f) v = v / a.out_div;
    }
    *o = v;
}

// ------------------------------------------------------------------------------------------
// naive: one thread per (b, co, t)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv1d_naive_kernel(SetConv1dArgs a) {
    const int64_t total = (int64_t)a.B * a.Cout * a.T_iter;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int t = (int)(idx % a.T_iter);
    const int co = (int)((idx / a.T_iter) % a.Cout);
    const int b = (int)(idx / ((int64_t)a.T_iter * a.Cout));
    const int n = t * a.out_stride + a.out_off;
    if (n < 0 || n >= a.T_out) return;
    const float *inb = a.in + (int64_t)b * a.in_bs;
    const float *wco = a.w + a.w_base + (int64_t)co * a.w_sco;
    float acc = 0.0f;
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float add = a.in_chan_add ? a.in_chan_add[(int64_t)b * a.Cin + ci] : 0.0f;
        const float *inc = inb + (int64_t)ci * a.in_cs;
        const float *wc = wco + (int64_t)ci * a.w_sci;
        for (int tap = 0; tap < a.K; ++tap) {
            const int ti = t + tap * a.dil - a.pad;
            if (ti >= 0 && ti < a.T_in) {
                const float x = dev_pro(inc[ti] + add, a.pro, a.pro_param);
                acc = fmaf(wc[(int64_t)tap * a.w_stap], x, acc);
            }
        }
    }
    conv_store(a, b, co, n, acc);
}

// ------------------------------------------------------------------------------------------
// MFMA implicit GEMM.  D[co][t] = sum_k A[co][k] * Bm[k][t],  k = (tap, ci).
//   block = 256 threads = 4 waves arranged WM (rows) x WN (cols); wave tile = 32 rows x 64 cols
//   (two 32x32 accumulators).  Input chunk of KC channels staged in LDS as [KC][BN + halo]
//   (prologue applied on the way in); A fragments come straight from the packed image in
//   global memory (256 B coalesced per wave per k-step, L2 resident).
// ------------------------------------------------------------------------------------------
constexpr int KC = 16;  // input channels per LDS chunk (packed images pad Cin to a multiple of this)

template <int WM, int WN>
__global__ void __launch_bounds__(256, 4) conv1d_mfma_kernel(SetConv1dArgs a, int lo, int halo, int CinP, int RBn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 64 * WN;
    const int W = BN + halo;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * BN;
    const int rb = blockIdx.y * WM + wm;
    const bool rb_valid = rb < RBn;
    const float *inb = a.in + (int64_t)b * a.in_bs;
    const int half = lane >> 5, l31 = lane & 31;

    f32x16 acc0 = {0}, acc1 = {0};
    const int cp_total = CinP / 2;

    // Staging: wave w owns rows w, w+4, w+8, w+12 of the KC x W chunk, lanes run along t (coalesced, no div/mod).
    // Loads are UNCONDITIONAL on clamped addresses and issued together (4 rows x 2 column blocks), the validity
    // select happens at the LDS write: an `if (valid) v = load` makes hipcc branch and drain vmcnt(0) per element,
    // i.e. one serialized global round trip per element (64 of them for Cin = 256; measured 88 us for a 256->256 1x1
    // conv on 25.6k frames).  When the chunk fits NJP = WN + 1 column blocks (halo <= 64) all of its loads are issued in
    // one go and one chunk ahead of the MFMAs; wider halos fall back to column-block pairs without prefetch.
    const float *addp = a.in_chan_add ? a.in_chan_add + (int64_t)b * a.Cin : inb;  // dummy stays a valid address
    const bool has_add = a.in_chan_add != nullptr;
    const int nj = (W + 63) >> 6;
    constexpr int NJP = WN + 1;
    float pv[4][NJP], pa[4];
    auto issue = [&](int c0, int jj0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cc = min(c0 + wave + 4 * k, a.Cin - 1);
            pa[k] = addp[cc];
#pragma unroll
            for (int u = 0; u < NJP; ++u) {
                const int ti = t0 + lo + (jj0 + u) * 64 + lane;
                pv[k][u] = inb[(int64_t)cc * a.in_cs + min(max(ti, 0), a.T_in - 1)];
            }
        }
    };
    auto commit = [&](int boff, int c0, int jj0) {  // boff: LDS float offset of the destination buffer
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = wave + 4 * k;
            const bool cok = c0 + row < a.Cin;
#pragma unroll
            for (int u = 0; u < NJP; ++u) {
                const int j = (jj0 + u) * 64 + lane;
                const int ti = t0 + lo + j;
                const float x = has_add ? pv[k][u] + pa[k] : pv[k][u];
                if (j < W) smem[boff + row * W + j] = (cok && ti >= 0 && ti < a.T_in) ? dev_pro(x, a.pro, a.pro_param) : 0.0f;
            }
        }
    };
    auto mfmas = [&](int boff, int c0) {  // offsets, not pointers: a selected pointer loses its LDS address space
        for (int tap = 0; tap < a.K; ++tap) {
            const int off = tap * a.dil - a.pad - lo;  // >= 0
            const float *ap = a.w + (((int64_t)rb * a.K + tap) * cp_total + (c0 >> 1)) * 64 + lane;
            const float *bp = smem + boff + half * W + wn * 64 + l31 + off;
#pragma unroll
            for (int cp = 0; cp < KC / 2; ++cp) {
                const float av = ap[cp * 64];
                const float b0 = bp[(2 * cp) * W];
                const float b1 = bp[(2 * cp) * W + 32];
                acc0 = mfma32(av, b0, acc0);
                acc1 = mfma32(av, b1, acc1);
            }
        }
    };
    const bool piped = nj <= NJP;
    if (piped) {
        // two LDS buffers, ONE barrier per chunk: while a wave runs the MFMAs of chunk i out of buffer i & 1, the loads of
        // chunk i + 1 are in flight; it then writes them to the other buffer (last read one barrier ago) and issues
        // the loads of chunk i + 2.  1x1 and 3-tap convs have only 16 - 48 MFMAs per chunk, so the second barrier and
        // the exposed LDS write were a visible share of the loop.
        const int bsz = KC * W;
        issue(0, 0);
        commit(0, 0, 0);
        if (KC < CinP) issue(KC, 0);
        __syncthreads();
        int cur = 0;
        for (int c0 = 0; c0 < CinP; c0 += KC, cur = bsz - cur) {
            if (rb_valid) mfmas(cur, c0);
            if (c0 + KC < CinP) {
                commit(bsz - cur, c0 + KC, 0);
                if (c0 + 2 * KC < CinP) issue(c0 + 2 * KC, 0);
            }
            __syncthreads();
        }
    } else {
        for (int c0 = 0; c0 < CinP; c0 += KC) {
            __syncthreads();  // previous chunk fully consumed
            for (int jj0 = 0; jj0 < nj; jj0 += NJP) {
                issue(c0, jj0);
                commit(0, c0, jj0);
            }
            __syncthreads();
            if (rb_valid) mfmas(0, c0);
        }
    }
    if (!rb_valid) return;
    // cheap activations get a specialised batched epilogue; the transcendental ones (tanh on the 1-channel conv_post,
    // gelu / mish / softplus on small layers) keep the compact per-element path, as does a ragged last row block
    const bool act_simple = a.act == SET_ACT_NONE || a.act == SET_ACT_RELU || a.act == SET_ACT_LRELU;
    if (rb * 32 + 32 > a.Cout || !act_simple) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = rb * 32 + mfma32_row(r, lane);
            if (co >= a.Cout) continue;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int t = t0 + wn * 64 + cb * 32 + l31;
                if (t >= a.T_iter) continue;
                const int n = t * a.out_stride + a.out_off;
                if (n < 0 || n >= a.T_out) continue;
                conv_store(a, b, co, n, cb == 0 ? acc0[r] : acc1[r]);
            }
        }
        return;
    }
    // Full row blocks.  Every optional operand (bias, residual, mask, previous output) is fetched as ONE batch under a
    // wave-uniform test, on clamped (always valid) addresses, and the activation switch is resolved once per kernel:
    // the per-element `if (a.res) v += a.res[i]` of conv_store makes hipcc emit a branch + s_waitcnt vmcnt(0) per
    // load -- ~64 serialized memory round trips per lane, ~33 us of every block's lifetime in the 32/64-channel
    // HiFi-GAN stages.  Addresses are buffer offsets: one VGPR per column block (the lane's part of the row, 4*half,
    // and its frame) plus a scalar offset per accumulator register (set_conv1d checks they fit 31 bits).
    const int lrow = rb * 32 + 4 * half;  // accumulator register r of this lane is row lrow + (r & 3) + 8 * (r >> 2)
    const rsrc_t d_out = make_rsrc(a.out + (int64_t)b * a.out_bs);
    unsigned vo[2], vr[2];
    bool tv[2];
    float mk[2] = {1.0f, 1.0f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int t = t0 + wn * 64 + cb * 32 + l31;
        const int n = t * a.out_stride + a.out_off;
        tv[cb] = t < a.T_iter && n >= 0 && n < a.T_out;
        const int nc = min(max(n, 0), a.T_out - 1);
        vo[cb] = (unsigned)(lrow * a.out_cs + nc) * 4u;
        vr[cb] = (unsigned)(lrow * a.res_cs + nc) * 4u;
        if (a.mask) mk[cb] = a.mask[(int64_t)b * a.T_out + nc];
    }
    float bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bi[r] = 0.0f;
    if (a.bias) {
        const rsrc_t d_b = make_rsrc(a.bias);
#pragma unroll
        for (int r = 0; r < 16; ++r) bi[r] = buf_load(d_b, (unsigned)lrow * 4u, (unsigned)((r & 3) + 8 * (r >> 2)) * 4u);
    }
    const rsrc_t d_r = make_rsrc(a.res ? a.res + (int64_t)b * a.res_bs : a.out);
    const bool has_div = a.accumulate && a.out_div != 0.0f;  // wave-uniform
    // one column block (16 registers) at a time: 16 residual (+16 previous-output) loads in flight, then compute, store
    auto column = [&](auto ACT, const f32x16 &acc, int cb) __attribute__((always_inline)) {
        constexpr int kAct = decltype(ACT)::value;
        float rv[16], ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = ov[r] = 0.0f;
        if (a.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = buf_load(d_r, vr[cb], (unsigned)(((r & 3) + 8 * (r >> 2)) * a.res_cs) * 4u);
        }
        if (a.accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ov[r] = buf_load(d_out, vo[cb], (unsigned)(((r & 3) + 8 * (r >> 2)) * a.out_cs) * 4u);
        }
        if (!tv[cb]) return;  // frames outside the output: lanes masked off (after the loads: no wait inside a branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = (dev_act((acc[r] + bi[r]) * a.alpha, kAct, a.act_param) + rv[r]) * mk[cb] + ov[r];
            if (has_div) y = y / a.out_div;
            buf_store(y, d_out, vo[cb], (unsigned)(((r & 3) + 8 * (r >> 2)) * a.out_cs) * 4u);
        }
    };
    auto finish = [&](auto ACT) __attribute__((always_inline)) {
        column(ACT, acc0, 0);
        column(ACT, acc1, 1);
    };
    switch (a.act) {
        case SET_ACT_RELU: finish(std::integral_constant<int, SET_ACT_RELU>{}); break;
        case SET_ACT_LRELU: finish(std::integral_constant<int, SET_ACT_LRELU>{}); break;
        default: finish(std::integral_constant<int, SET_ACT_NONE>{}); break;
    }
}

// ------------------------------------------------------------------------------------------
// v2: big-tile implicit GEMM for wide layers (same machinery as the fused DiffNet layer kernel).
//   block = 4 waves; wave w owns RB row blocks x 2 column blocks (32*RB rows x 64 frames) -> block tile
//   128*RB rows x 64 frames; the input is staged per chunk of CH <= 256 channels as xs[CH][64 + halo]
//   (wave-per-row coalesced loads, prologue applied on the way in); A fragments come from a packed image
//   [row group][wave][k-step][lane][RB] (one RB-float vector per lane per k-step) prefetched by gemm_groups.
//   k-step order inside a chunk: tap-major, channel pairs minor.
// ------------------------------------------------------------------------------------------
constexpr int V2_GS = 4;     // k-steps per operand group
constexpr int V2_CH = 256;   // max channels per LDS chunk

template <int RB>
__global__ void __launch_bounds__(256, RB == 1 ? 4 : (RB == 2 ? 3 : 2)) conv1d_mfma_v2_kernel(SetConv1dArgs a, int lo, int halo, int CinP, int ch_max) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typedef typename AVec<RB>::type avec_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, t0 = blockIdx.x * 64, g = blockIdx.y;
    const int XW = 64 + halo;
    const int T_in = a.T_in;
    const float *inb = a.in + (int64_t)b * a.in_bs;
    const int64_t ks_total = (int64_t)a.K * (CinP / 2);
    const avec_t *wp = reinterpret_cast<const avec_t *>(a.w) + (((int64_t)g * 4 + w) * ks_total) * 64 + lane;

    f32x16 acc[RB][2];
#pragma unroll
    for (int r = 0; r < RB; ++r) { acc[r][0] = (f32x16){0}; acc[r][1] = (f32x16){0}; }

    // staging geometry (per lane): columns `lane` and `64 + lane` of the tile; clamped addresses + selects, no branches
    const int tA = t0 + lo + lane, tB = tA + 64;
    const bool vA = tA >= 0 && tA < T_in, vB = tB >= 0 && tB < T_in && lane < halo;
    const unsigned cA = (unsigned)min(max(tA, 0), T_in - 1), cB = (unsigned)min(max(tB, 0), T_in - 1);
    const bool laneB = lane < halo;
    // optional per-(b, channel) add: ALWAYS load (from a valid dummy row when absent) and select afterwards -- a
    // `ptr ? load : 0` inside the unrolled staging loop makes hipcc branch and drain vmcnt(0) per element
    const bool has_add = a.in_chan_add != nullptr;
    const float *addp = has_add ? a.in_chan_add + (int64_t)b * a.Cin : inb;

    for (int c0 = 0; c0 < CinP; c0 += ch_max) {
        const int CH = min(ch_max, CinP - c0);  // multiple of 16
        __syncthreads();                        // previous chunk fully consumed
        // wave w stages channels c0 + w, w+4, ... (16 rows in flight per wave)
        for (int r0 = w; r0 < CH; r0 += 64) {
            float xa[16], xb[16], dd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = c0 + r0 + 4 * u;
                const int cc = min(c, a.Cin - 1);  // clamp: padded channels read a valid row, zeroed below
                const float *row = inb + (int64_t)cc * a.in_cs;
                xa[u] = row[cA];
                xb[u] = row[cB];
                dd[u] = addp[cc];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int cl = r0 + 4 * u;  // channel inside the chunk
                if (cl < CH) {
                    const bool cv = c0 + cl < a.Cin;
                    const float dv = has_add ? dd[u] : 0.0f;
                    smem[cl * XW + lane] = (vA && cv) ? dev_pro(xa[u] + dv, a.pro, a.pro_param) : 0.0f;
                    if (laneB) smem[cl * XW + 64 + lane] = (vB && cv) ? dev_pro(xb[u] + dv, a.pro, a.pro_param) : 0.0f;
                }
            }
        }
        __syncthreads();
        const int gpt = CH / (2 * V2_GS);        // operand groups per tap (even: CH is a multiple of 16)
        const int rstep = 2 * XW;
        const float *bp = smem + half * XW + l31 + (0 * a.dil - a.pad - lo);
        gemm_groups<RB, 2, V2_GS>(acc, wp, bp, rstep, a.K * gpt, [&](int gi) {
            wp += V2_GS * 64;
            bp += V2_GS * rstep;
            if ((gi % gpt) == gpt - 1) bp += a.dil - (CH / 2) * rstep;  // next tap: channel 0, shifted by dil columns
        });
        wp += V2_GS * 64;  // gemm_groups leaves wp on the chunk's last group
    }

    // epilogue: accumulators go through LDS one row-block slice (128 rows x 64 frames) at a time, then a compact
    // cooperative loop applies bias / alpha / act / res / mask with whole-row coalesced global accesses
    // (a fully unrolled per-register epilogue is ~40k instructions here and spills).
    // fast path: 4 consecutive frames per thread (16-byte LDS reads / global loads / stores)
    const bool act_simple = a.act == SET_ACT_NONE || a.act == SET_ACT_RELU || a.act == SET_ACT_LRELU;
    const bool vec4 = act_simple && a.out_stride == 1 && a.out_off == 0 && !a.accumulate && (a.T_out & 3) == 0 && (a.out_cs & 3) == 0 &&
                      (a.out_bs & 3) == 0 && (!a.res || ((a.res_cs & 3) == 0 && (a.res_bs & 3) == 0)) &&
                      t0 + 64 <= a.T_iter && a.T_iter <= a.T_out;
    const int col = vec4 ? (tid & 15) * 4 : (tid & 63);
    const int t = t0 + col;
    const int n = t * a.out_stride + a.out_off;
    const bool tvalid = t < a.T_iter && n >= 0 && n < a.T_out;
    f32x4 msk4 = {1.0f, 1.0f, 1.0f, 1.0f};
    float msk = 1.0f;
    if (a.mask && tvalid) {
        if (vec4) msk4 = *reinterpret_cast<const f32x4 *>(a.mask + (int64_t)b * a.T_out + n);
        else msk = a.mask[(int64_t)b * a.T_out + n];
    }
    // Addresses are buffer offsets (one VGPR for the lane's row / frame part + a scalar offset per row group;
    // launch_conv_v2 checks they fit 31 bits).  Loads go to clamped, always valid addresses; stores are masked.
    const rsrc_t d_out = make_rsrc(a.out + (int64_t)b * a.out_bs);
    const rsrc_t d_res = make_rsrc(a.res ? a.res + (int64_t)b * a.res_bs : a.out);
    const rsrc_t d_b = make_rsrc(a.bias ? a.bias : a.out);
    const int q = vec4 ? (tid >> 4) : 0;  // lane part of the row (fast path); the ragged path's rows are wave-uniform
    const int nc = min(max(n, 0), a.T_out - 1);
    const bool has_bias = a.bias != nullptr, has_res = a.res != nullptr;
    // one row-block slice; a generic lambda over a compile-time rb instead of `#pragma unroll for (rb)`: with the
    // per-activation bodies inside, the optimizer refused to unroll the loop and acc[rb] went to scratch
    auto slice = [&](auto RBI) __attribute__((always_inline)) {
        constexpr int rb = decltype(RBI)::value;
        // fast path: the 8 rows this thread finishes (row = row0(i) + q), bias and residual fetched as one batch each
        // BEFORE the LDS round trip (a per-row `if (a.res) load` costs one serialized memory round trip per row)
        f32x4 rv4[8];
        float bi8[8];
        if (vec4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row0 = g * 128 * RB + (i >> 1) * 32 * RB + rb * 32 + 16 * (i & 1);
                const int rowc = min(row0 + q, a.Cout - 1);  // == row0 + q for every full 16-row group
                bi8[i] = 0.0f;
                rv4[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                if (has_bias) bi8[i] = buf_load(d_b, (unsigned)rowc * 4u, 0u);
                if (has_res) {
                    if (row0 + 16 <= a.Cout) rv4[i] = buf_load4(d_res, (unsigned)(q * a.res_cs + nc) * 4u, (unsigned)(row0 * a.res_cs) * 4u);
                    else rv4[i] = buf_load4(d_res, (unsigned)(rowc * a.res_cs + nc) * 4u, 0u);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) smem[(32 * w + mfma32_row(r, lane)) * 64 + cb * 32 + l31] = acc[rb][cb][r];
        __syncthreads();
        if (vec4) {
            auto finish = [&](auto ACT) __attribute__((always_inline)) {
                constexpr int kAct = decltype(ACT)::value;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row0 = g * 128 * RB + (i >> 1) * 32 * RB + rb * 32 + 16 * (i & 1);
                    f32x4 v = *reinterpret_cast<const f32x4 *>(smem + (q + 16 * i) * 64 + col);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = dev_act((v[k] + bi8[i]) * a.alpha, kAct, a.act_param);
                    // Range-masked like the other epilogues (a lane past Cout stores beyond num_records: dropped; no exec-masked block per
                    // store, so the stores do not wait for each other), with the WHOLE offset in the VGPR (soffset 0).  Round 4 shipped a branch
                    // here because the masked form made the fp32 training step differ from run to run; round 5 found why
                    // (tools/grad_stability_probe.py TRACE=1): unbranched, the compiler rewrites the store's data registers in the
                    // very next instruction (v_add_u32 v36 right behind buffer_store_dwordx4 v[34:37], ..., s10 offen -- 51 such pairs in the three
                    // instantiations), and ROCm 7.2's hazard recognizer pads a > 64-bit buffer store against that only when its soffset is NOT an
                    // SGPR.  On gfx950 the store then sometimes sends the overwritten dword (always component 2 of a few 16-byte stores, a few
                    // dozen of 6.5 M elements per launch).  Measured: SGPR soffset = unstable 4 of 4 runs; + s_nop 1 behind the store, + vmcnt(0)
                    // behind it, or soffset 0 (the compiler pads by itself) = bit-identical 5 of 5; a full wait IN FRONT of the store does not help.
                    // (An inline-asm s_nop is not a fix: the scheduler may still put the VALU write between the store and the asm.)
                    // tests/test_isa_schedule.py scans every kernel of the library for the pattern.
#if SET_CONV_V2_STORE_HAZARD  // -DSET_CONV_V2_STORE_HAZARD=1 (tools/build_exp.sh): the SGPR-soffset form, for reproducing the hazard
                    buf_store4((v + rv4[i]) * msk4, d_out, (row0 + q < a.Cout) ? (unsigned)(q * a.out_cs + nc) * 4u : BUF_OOB, (unsigned)(row0 * a.out_cs) * 4u);
#else
                    buf_store4((v + rv4[i]) * msk4, d_out, (row0 + q < a.Cout) ? (unsigned)((row0 + q) * a.out_cs + nc) * 4u : BUF_OOB, 0u);
#endif
                }
            };
            switch (a.act) {
                case SET_ACT_RELU: finish(std::integral_constant<int, SET_ACT_RELU>{}); break;
                case SET_ACT_LRELU: finish(std::integral_constant<int, SET_ACT_LRELU>{}); break;
                default: finish(std::integral_constant<int, SET_ACT_NONE>{}); break;
            }
        } else {  // ragged tiles, unaligned shapes, transcendental activations: one frame per thread, the 32 rows of
                  // this wave in batches of 8 (a real loop: the activation switch is expanded 8 times, not 32)
#pragma unroll 1
            for (int i0 = 0; i0 < 32; i0 += 8) {
                float bi[8], rv[8], ov[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = w + 4 * (i0 + i);
                    const int coc = min(g * 128 * RB + (rr >> 5) * 32 * RB + rb * 32 + (rr & 31), a.Cout - 1);  // uniform
                    bi[i] = rv[i] = ov[i] = 0.0f;
                    if (has_bias) bi[i] = buf_load(d_b, 0u, (unsigned)coc * 4u);
                    if (has_res) rv[i] = buf_load(d_res, (unsigned)nc * 4u, (unsigned)(coc * a.res_cs) * 4u);
                    if (a.accumulate) ov[i] = buf_load(d_out, (unsigned)nc * 4u, (unsigned)(coc * a.out_cs) * 4u);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rr = w + 4 * (i0 + i);
                    const int co = g * 128 * RB + (rr >> 5) * 32 * RB + rb * 32 + (rr & 31);
                    float v = dev_act((smem[rr * 64 + col] + bi[i]) * a.alpha, a.act, a.act_param);
                    float y = (v + rv[i]) * msk + ov[i];
                    if (a.accumulate && a.out_div != 0.0f) y = y / a.out_div;
                    if (co < a.Cout) buf_store(y, d_out, tvalid ? (unsigned)nc * 4u : BUF_OOB, (unsigned)(co * a.out_cs) * 4u);  // co is wave-uniform; lanes masked by range
                }
            }
        }
    };
    slice(std::integral_constant<int, 0>{});
    if constexpr (RB > 1) slice(std::integral_constant<int, 1>{});
    if constexpr (RB > 2) {
        slice(std::integral_constant<int, 2>{});
        slice(std::integral_constant<int, 3>{});
    }
}

// packed image for v2:  wp[g][w][ks][lane][rb] = W[row][ci][tap],  row = g*128*RB + w*32*RB + rb*32 + (lane&31),
// ks enumerates (chunk, tap, channel pair) in the kernel's consumption order, ci = chunk0 + 2*cp + (lane>>5)
__global__ void __launch_bounds__(256) pack_conv_weight_v2_kernel(const float *w, float *wp, int Cout, int Cin, int K,
                                                                  int CinP, int ch_max, int RB, int64_t total,
                                                                  int64_t w_base, int64_t w_sco, int64_t w_sci,
                                                                  int64_t w_stap) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int rb = (int)(idx % RB);
    const int lane = (int)((idx / RB) & 63);
    int64_t r = idx / RB / 64;
    const int64_t ks_total = (int64_t)K * (CinP / 2);
    int64_t ks = r % ks_total;
    r /= ks_total;
    const int wv = (int)(r & 3), g = (int)(r >> 2);
    // decode ks -> (chunk, tap, cp)
    int c0 = 0;
    for (;;) {
        const int CH = CinP - c0 < ch_max ? CinP - c0 : ch_max;
        const int64_t per_chunk = (int64_t)K * (CH / 2);
        if (ks < per_chunk) {
            const int tap = (int)(ks / (CH / 2)), cp = (int)(ks % (CH / 2));
            const int co = g * 128 * RB + wv * 32 * RB + rb * 32 + (lane & 31);
            const int ci = c0 + 2 * cp + (lane >> 5);
            float v = 0.0f;
            if (co < Cout && ci < Cin) v = w[w_base + (int64_t)co * w_sco + (int64_t)ci * w_sci + (int64_t)tap * w_stap];
            wp[idx] = v;
            return;
        }
        ks -= per_chunk;
        c0 += CH;
    }
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

static inline int v2_rb(int Cout) { return Cout >= 384 ? 4 : (Cout >= 192 ? 2 : 1); }
static inline int v2_ch_max(int halo) { return (64 + halo) * V2_CH * 4 > 96 * 1024 ? 128 : V2_CH; }

extern "C" int64_t set_packed_conv_weight_v2_size(int32_t Cout, int32_t Cin, int32_t K) {
    const int RB = v2_rb(Cout);
    return (int64_t)round_up(Cout, 128 * RB) * K * round_up(Cin, 16);
}

extern "C" int set_pack_conv_weight_v2(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t K, int32_t dil,
                                       int64_t w_base, int64_t w_sco, int64_t w_sci, int64_t w_stap, void *stream) {
    SET_REQUIRE(w && wp && Cout > 0 && Cin > 0 && K > 0, "set_pack_conv_weight_v2");
    const int RB = v2_rb(Cout);
    const int halo = (K - 1) * (dil < 0 ? -dil : dil);
    const int64_t total = set_packed_conv_weight_v2_size(Cout, Cin, K);
    hipLaunchKernelGGL(pack_conv_weight_v2_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp,
                       Cout, Cin, K, round_up(Cin, 16), v2_ch_max(halo), RB, total, w_base, w_sco, w_sci, w_stap);
    return set_check_launch("set_pack_conv_weight_v2");
}

// the epilogues address out / res with 32-bit byte offsets relative to the batch slice
static inline bool epilogue_offsets_fit(const SetConv1dArgs &a, int rows_padded) {
    const int64_t cs = a.res && a.res_cs > a.out_cs ? a.res_cs : a.out_cs;
    return ((int64_t)rows_padded * cs + a.T_out) * 4 < ((int64_t)1 << 31);
}

static int launch_conv_v2(const SetConv1dArgs &a, hipStream_t s) {
    const int o_first = -a.pad, o_last = (a.K - 1) * a.dil - a.pad;
    const int lo = o_first < o_last ? o_first : o_last;
    const int halo = (o_first < o_last ? o_last : o_first) - lo;
    if (halo > 64) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(mfma2)", "receptive field > 64");
    const int RB = v2_rb(a.Cout);
    if (!epilogue_offsets_fit(a, round_up(a.Cout, 128 * RB)))
        return set_fail(SET_E_UNSUPPORTED, "set_conv1d(mfma2)", "one batch slice of out / res exceeds 2 GiB");
    const int CinP = round_up(a.Cin, 16);
    const int ch_max = v2_ch_max(halo);
    const int ch = CinP < ch_max ? CinP : ch_max;
    size_t lds = (size_t)ch * (64 + halo) * sizeof(float);
    if (lds < 128 * 64 * sizeof(float)) lds = 128 * 64 * sizeof(float);  // epilogue slice
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1d_mfma_v2_kernel<1>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), "conv v2 attr");
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1d_mfma_v2_kernel<2>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), "conv v2 attr");
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1d_mfma_v2_kernel<4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024), "conv v2 attr");
        attr_set = true;
    }
    dim3 grid((a.T_iter + 63) / 64, (a.Cout + 128 * RB - 1) / (128 * RB), a.B), block(256);
    if (RB == 4) hipLaunchKernelGGL(conv1d_mfma_v2_kernel<4>, grid, block, lds, s, a, lo, halo, CinP, ch_max);
    else if (RB == 2) hipLaunchKernelGGL(conv1d_mfma_v2_kernel<2>, grid, block, lds, s, a, lo, halo, CinP, ch_max);
    else hipLaunchKernelGGL(conv1d_mfma_v2_kernel<1>, grid, block, lds, s, a, lo, halo, CinP, ch_max);
    return set_check_launch("set_conv1d(mfma2)");
}


extern "C" int64_t set_packed_conv_weight_size(int32_t Cout, int32_t Cin, int32_t K) {
    return (int64_t)round_up(Cout, 32) * K * round_up(Cin, KC);
}

__global__ void __launch_bounds__(256) pack_conv_weight_kernel(const float *w, float *wp, int Cout, int Cin, int K,
                                                               int CinP, int64_t total, int64_t w_base, int64_t w_sco,
                                                               int64_t w_sci, int64_t w_stap) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    int64_t r = idx >> 6;
    const int cp_total = CinP / 2;
    const int cp = (int)(r % cp_total);
    r /= cp_total;
    const int tap = (int)(r % K);
    const int rb = (int)(r / K);
    const int co = rb * 32 + (lane & 31);
    const int ci = 2 * cp + (lane >> 5);
    float v = 0.0f;
    if (co < Cout && ci < Cin) v = w[w_base + (int64_t)co * w_sco + (int64_t)ci * w_sci + (int64_t)tap * w_stap];
    wp[idx] = v;
}

extern "C" int set_pack_conv_weight(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t K, int64_t w_base,
                                    int64_t w_sco, int64_t w_sci, int64_t w_stap, void *stream) {
    SET_REQUIRE(w && wp && Cout > 0 && Cin > 0 && K > 0, "set_pack_conv_weight");
    const int64_t total = set_packed_conv_weight_size(Cout, Cin, K);
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wp, Cout, Cin, K, round_up(Cin, KC), total, w_base, w_sco, w_sci, w_stap);
    return set_check_launch("set_pack_conv_weight");
}

// every fp32 weight image a training step uses in ONE launch (after the optimizer step: ~150 pack launches of ~5 us per fp32 training
// step otherwise -- 1.1 ms of a 32 ms step on the compute stream): element gidx of the concatenated image space -> its descriptor by
// binary search over `start`; kind 0 = the layout of pack_conv_weight_kernel, kind 1 = pack_conv_weight_v2_kernel.  Same values, same places.
__global__ void __launch_bounds__(256) pack_conv_weights_f32_batch_kernel(const SetPackF32Desc *d, int n, int64_t total) {
    const int64_t gidx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= total) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].start <= gidx) lo = mid; else hi = mid - 1;
    }
    const SetPackF32Desc e = d[lo];
    const int64_t idx = gidx - e.start;
    int co, ci, tap;
    if (e.kind == 0) {
        const int lane = (int)(idx & 63);
        int64_t r = idx >> 6;
        const int cp_total = e.CinP / 2;
        const int cp = (int)(r % cp_total);
        r /= cp_total;
        tap = (int)(r % e.K);
        const int rb = (int)(r / e.K);
        co = rb * 32 + (lane & 31);
        ci = 2 * cp + (lane >> 5);
    } else {
        const int rb = (int)(idx % e.RB);
        const int lane = (int)((idx / e.RB) & 63);
        int64_t r = idx / e.RB / 64;
        const int64_t ks_total = (int64_t)e.K * (e.CinP / 2);
        int64_t ks = r % ks_total;
        r /= ks_total;
        const int wv = (int)(r & 3), g = (int)(r >> 2);
        int c0 = 0;
        for (;;) {  // ks -> (chunk, tap, channel pair), the kernel's consumption order
            const int CH = e.CinP - c0 < e.ch_max ? e.CinP - c0 : e.ch_max;
            const int64_t per_chunk = (int64_t)e.K * (CH / 2);
            if (ks < per_chunk) {
                tap = (int)(ks / (CH / 2));
                ci = c0 + 2 * (int)(ks % (CH / 2)) + (lane >> 5);
                break;
            }
            ks -= per_chunk;
            c0 += CH;
        }
        co = g * 128 * e.RB + wv * 32 * e.RB + rb * 32 + (lane & 31);
    }
    float v = 0.0f;
    if (co < e.Cout && ci < e.Cin) v = e.w[e.w_base + (int64_t)co * e.w_sco + (int64_t)ci * e.w_sci + (int64_t)tap * e.w_stap];
    e.wp[idx] = v;
}
extern "C" int64_t set_sizeof_pack_f32_desc(void) { return (int64_t)sizeof(SetPackF32Desc); }
// host side: the layout parameters of an image (CinP, RB, ch_max) and its element count from Cout / Cin / K (and |dil| for kind 1)
extern "C" int64_t set_fill_pack_f32_desc(SetPackF32Desc *d, int32_t kind, int32_t dil) {
    if (d == nullptr || (kind != 0 && kind != 1) || d->Cout <= 0 || d->Cin <= 0 || d->K <= 0) return -1;
    d->kind = kind;
    if (kind == 0) {
        d->CinP = round_up(d->Cin, KC);
        d->RB = 1;
        d->ch_max = 0;
        return set_packed_conv_weight_size(d->Cout, d->Cin, d->K);
    }
    d->CinP = round_up(d->Cin, 16);
    d->RB = v2_rb(d->Cout);
    d->ch_max = v2_ch_max((d->K - 1) * (dil < 0 ? -dil : dil));
    return set_packed_conv_weight_v2_size(d->Cout, d->Cin, d->K);
}
extern "C" int set_pack_conv_weights_f32_batch(const SetPackF32Desc *descs_dev, int32_t n, int64_t total, void *stream) {
    SET_REQUIRE(descs_dev && n > 0 && total > 0, "set_pack_conv_weights_f32_batch");
    hipLaunchKernelGGL(pack_conv_weights_f32_batch_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, descs_dev, n,
                       total);
    return set_check_launch("set_pack_conv_weights_f32_batch");
}

int set_conv1d_bf16_dispatch(const SetConv1dArgs &a, hipStream_t s);  // bf16.hip
int set_conv1d_x2_dispatch(const SetConv1dArgs &a, hipStream_t s);    // csrc/conv_x2.hip

// ------------------------------------------------------------------------------------------
// SET_IMPL_FEWOUT: a convolution with ONE or TWO output channels over a long sequence (HiFi-GAN's conv_post, hifigan.py:123,
// 138-140: 32 -> 1 channels, 7 taps, 13 M output samples at B = 64).  As a GEMM it has one useful row in 32 (the MFMA kernel
// spends 1.9 ms on it); it is a streaming read of the input with 224 multiply-adds per sample, so: one thread = four consecutive
// output samples of one utterance, per input channel three aligned 16-byte loads (the 12 samples around them: pad <= 4 taps to
// the left, K - pad <= 5 to the right), weights through the scalar cache (the address is wave-uniform), prologue on load,
// bias / alpha / activation and one 16-byte store at the end.  HBM-bound: the input is read once.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv1d_fewout_kernel(SetConv1dArgs a) {
    const int b = blockIdx.y;
    const int t4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t4 >= a.T_out) return;
    const int T = a.T_in;
    const float *inb = a.in + (int64_t)b * a.in_bs;
    const bool v0 = t4 >= 4, v2 = t4 + 8 <= T;
    const int o0 = v0 ? t4 - 4 : 0, o2 = v2 ? t4 + 4 : t4;
    float acc[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    for (int ci = 0; ci < a.Cin; ++ci) {
        const float *row = inb + (int64_t)ci * a.in_cs;
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(row + o0);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(row + t4);
        const f32x4 w2 = *reinterpret_cast<const f32x4 *>(row + o2);
        float v[12];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = v0 ? w0[e] : 0.0f;
            v[4 + e] = w1[e];
            v[8 + e] = v2 ? w2[e] : 0.0f;
        }
        if (a.pro == SET_PRO_LRELU) {
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = v[j] > 0.0f ? v[j] : v[j] * a.pro_param;
        } else if (a.pro == SET_PRO_DIV) {
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = v[j] / a.pro_param;
        }
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            if (co < a.Cout) {
                const float *wr = a.w + a.w_base + (int64_t)co * a.w_sco + (int64_t)ci * a.w_sci;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    if (k < a.K) {
                        const float wv = wr[(int64_t)k * a.w_stap];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[co][e] += wv * v[4 + e + k - a.pad];
                    }
                }
            }
        }
    }
    for (int co = 0; co < a.Cout; ++co) {
        const float bias = a.bias ? a.bias[co] : 0.0f;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = dev_act((acc[co][e] + bias) * a.alpha, a.act, a.act_param);
        *reinterpret_cast<f32x4 *>(a.out + (int64_t)b * a.out_bs + (int64_t)co * a.out_cs + t4) = o;
    }
}

static int launch_conv_fewout(const SetConv1dArgs &a, hipStream_t s) {
    const bool ok = a.Cout <= 2 && a.K <= 9 && a.dil == 1 && a.pad >= 0 && a.pad <= 4 && a.K - a.pad <= 5 && a.out_stride == 1 &&
                    a.out_off == 0 && a.T_in == a.T_out && a.T_iter == a.T_out && (a.T_in & 3) == 0 && !a.res && !a.mask &&
                    !a.in_chan_add && !a.accumulate && (a.in_bs & 3) == 0 && (a.in_cs & 3) == 0 && (a.out_bs & 3) == 0 &&
                    (a.out_cs & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out)) & 15) == 0 &&
                    a.B <= 65535;
    if (!ok) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(fewout)", "needs Cout <= 2, K <= 9, dil 1, same-length output, T % 4 == 0, plain epilogue");
    hipLaunchKernelGGL(conv1d_fewout_kernel, dim3((a.T_out / 4 + 255) / 256, a.B), dim3(256), 0, s, a);
    return set_check_launch("set_conv1d(fewout)");
}

extern "C" int set_conv1d(const SetConv1dArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_conv1d");
    const SetConv1dArgs &a = *args;
    SET_REQUIRE(a.in && a.w && a.out, "set_conv1d");
    SET_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.K > 0 && a.T_in > 0 && a.T_out > 0, "set_conv1d");
    SET_REQUIRE(a.out_stride >= 1, "set_conv1d");
    if (a.T_iter <= 0) return SET_OK;
    hipStream_t s = (hipStream_t)stream;
    if (a.impl == SET_IMPL_MFMA2) return launch_conv_v2(a, s);
    if (a.impl == SET_IMPL_BF16) return set_conv1d_bf16_dispatch(a, s);
    if (a.impl == SET_IMPL_F16X2) return set_conv1d_x2_dispatch(a, s);
    if (a.impl == SET_IMPL_FEWOUT) return launch_conv_fewout(a, s);
    if (a.impl != SET_IMPL_MFMA) {
        const int64_t total = (int64_t)a.B * a.Cout * a.T_iter;
        hipLaunchKernelGGL(conv1d_naive_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, s, a);
        return set_check_launch("set_conv1d(naive)");
    }
    const int o_first = -a.pad, o_last = (a.K - 1) * a.dil - a.pad;
    const int lo = o_first < o_last ? o_first : o_last;
    const int hi = o_first < o_last ? o_last : o_first;
    const int halo = hi - lo;
    if (halo > 512) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(mfma)", "receptive field > 512");
    if (!epilogue_offsets_fit(a, round_up(a.Cout, 32)))
        return set_fail(SET_E_UNSUPPORTED, "set_conv1d(mfma)", "one batch slice of out / res exceeds 2 GiB");
    const int CinP = round_up(a.Cin, KC);
    const int RBn = (a.Cout + 31) / 32;
    dim3 block(256);
    if (RBn >= 4) {
        constexpr int WM = 4, WN = 1;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float) * (halo <= 64 ? 2 : 1);  // two buffers when piped
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN>), grid, block, lds, s, a, lo, halo, CinP, RBn);
    } else if (RBn >= 2) {
        constexpr int WM = 2, WN = 2;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float) * (halo <= 64 ? 2 : 1);  // two buffers when piped
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN>), grid, block, lds, s, a, lo, halo, CinP, RBn);
    } else {
        constexpr int WM = 1, WN = 4;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float) * (halo <= 64 ? 2 : 1);  // two buffers when piped
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN>), grid, block, lds, s, a, lo, halo, CinP, RBn);
    }
    return set_check_launch("set_conv1d(mfma)");
}

// ------------------------------------------------------------------------------------------
// weight-norm fold: one block per dim-0 slice
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) weight_norm_fold_kernel(const float *g, const float *v, float *w,
                                                               int64_t inner) {
    __shared__ float red[256];
    const int i = blockIdx.x;
    const float *vi = v + (int64_t)i * inner;
    float s = 0.0f;
    for (int64_t j = threadIdx.x; j < inner; j += 256) s = fmaf(vi[j], vi[j], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const float scale = g[i] / sqrtf(red[0]);
    float *wi = w + (int64_t)i * inner;
    for (int64_t j = threadIdx.x; j < inner; j += 256) wi[j] = vi[j] * scale;
}

extern "C" int set_weight_norm_fold(const float *g, const float *v, float *w, int32_t n0, int64_t inner,
                                    void *stream) {
    SET_REQUIRE(g && v && w && n0 > 0 && inner > 0, "set_weight_norm_fold");
    hipLaunchKernelGGL(weight_norm_fold_kernel, dim3(n0), dim3(256), 0, (hipStream_t)stream, g, v, w, inner);
    return set_check_launch("set_weight_norm_fold");
}
