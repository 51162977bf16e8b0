// Generic stride-1 conv1d (SetConv1dArgs semantics, incl. the strided output of the polyphase transposed convs) with fp32
// operands carried as TWO fp16 pieces and three fp16 MFMAs per product (fp32 accumulate) -- the operand splitting of
// csrc/diffnet_x3.hip (see there for the arithmetic: fp32-equivalent results on the 2.5 PFLOP/s pipe) applied to the wide
// convolutions of the vocoder (hifigan.py:27-58,108-142), which are MFMA-bound on the fp32 pipe.
//
//   A operand (weights): image [32-row block][16-channel group][tap][piece][lane][8 fp16] in global memory, one 16-byte
//       load per lane per (k-step, row block, piece), prefetched 2 k-steps ahead, across the chunk barriers; the weights are
//       multiplied by a power of two 2^k before they are split (residual pieces stay normal) and 2^-k sits at the image's tail.
//   B operand (activations): LDS tile [piece][frame][32 channels] (rows padded to 80 bytes: conflict-free 16-byte fragment
//       reads), one 32-channel chunk at a time, prologue (leaky ReLU / division) applied and the value split once per chunk;
//       the loads of chunk c+1 are in flight (registers) under the MFMAs of chunk c.  The taps are row shifts of the tile.
//   Block = 4 waves as WM x WN; wave tile = 32 RBW rows x 32 NCB frames (RBW x NCB accumulators of 32x32).
// An activation of magnitude >= 32768 (outside the fp16 range of the splitting) raises a sticky device flag that the host
// polls (set_conv_x2_range_flag): the caller then repeats the forward on the fp32 kernels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef _Float16 cx_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cx_u32x4 __attribute__((ext_vector_type(4)));

__device__ int g_x2_range_flag = 0;

namespace {

constexpr int CX_KCH = 32;               // channels per LDS chunk
constexpr int CX_ROWB = CX_KCH * 2 + 16;  // bytes per LDS row

template <int N> using cx_ic = std::integral_constant<int, N>;

__device__ __forceinline__ unsigned short cx_f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float cx_h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ f32x16 cx_mma(cx_u32x4 a, cx_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cx_f16x8, a), __builtin_bit_cast(cx_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ cx_u32x4 cx_load_u4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
template <int PRO>
__device__ __forceinline__ float cx_pro(float v, float p) {
    if constexpr (PRO == SET_PRO_LRELU) return v > 0.0f ? v : v * p;
    else if constexpr (PRO == SET_PRO_DIV) return v / p;
    else return v;
}
static inline int cx_round_up(int x, int m) { return (x + m - 1) / m * m; }

// image: [rb32 < CoutP/32][g16 < CinP/16][tap][piece][lane][8]; lane l holds row 32 rb32 + (l & 31), channel 16 g16 + 8 (l >> 5) + e
// phases = u > 0 (all output phases of a ConvTranspose1d [Cin][Cout_t][kfull] in one image): row R = co * u + p, tap j ->
// Wt[ci][co][u j + p] (zero beyond kfull); Cout counts these rows (= u * Cout_t).
__global__ void __launch_bounds__(256) pack_conv_x2_kernel(const float *w, unsigned short *img, int Cout, int Cin, int K, int CoutP,
                                                           int CinP, int64_t n_frag_elems, int64_t w_base, int64_t w_sco,
                                                           int64_t w_sci, int64_t w_stap, float scale, int phases, int kfull) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per fragment element (both pieces)
    if (idx == 0) {
        float *tail = reinterpret_cast<float *>(img + 2 * n_frag_elems);
        tail[0] = scale; tail[1] = 1.0f / scale; tail[2] = 0.0f; tail[3] = 0.0f;
    }
    if (idx >= n_frag_elems) return;
    int64_t r = idx;
    const int e = r & 7; r >>= 3;
    const int l = r & 63; r >>= 6;
    const int tap = (int)(r % K); r /= K;
    const int ng16 = CinP / 16;
    const int g16 = (int)(r % ng16), rb = (int)(r / ng16);
    const int row = 32 * rb + (l & 31), ci = 16 * g16 + 8 * (l >> 5) + e;
    float v = 0.0f;
    if (row < Cout && ci < Cin) {
        if (phases > 0) {
            const int co = row / phases, kk = phases * tap + row % phases;
            if (kk < kfull) v = scale * w[(int64_t)ci * (Cout / phases) * kfull + (int64_t)co * kfull + kk];
        } else {
            v = scale * w[w_base + (int64_t)row * w_sco + (int64_t)ci * w_sci + (int64_t)tap * w_stap];
        }
    }
    const unsigned short p0 = cx_f2h(v), p1 = cx_f2h(v - cx_h2f(p0));
    unsigned short *base = img + ((((int64_t)rb * ng16 + g16) * K + tap) * 2) * 512 + l * 8 + e;
    base[0] = p0;
    base[512] = p1;
}

// PH > 0: the rows are (output channel, phase) pairs of a transposed conv with stride PH (row R = co * PH + p); sample
// t of row R is output sample n = t * PH + p - ph_pad of channel co.  With PH % 4 == 0 the four rows of a register group are
// four CONSECUTIVE samples of one channel: one 16-byte store instead of four strided 4-byte ones.
template <int WM, int WN, int RBW, int NCB, bool PHASES>
__global__ void __launch_bounds__(256, 2) conv1d_x2_kernel(SetConv1dArgs a, int lo, int halo, int CinP, int CoutP, int ph_u, int ph_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int MB = 32 * RBW * WM, NB = 32 * NCB * WN;
    constexpr int NPASS_MAX = (NB + 128 + 127) / 128;  // frame passes of 128 rows (halo <= 128)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.z, t0 = blockIdx.x * NB, r0 = blockIdx.y * MB;
    const int R = NB + halo;                       // frame rows of the B tile
    const unsigned piece_bytes = (unsigned)(R * CX_ROWB);
    unsigned char *Bs = smem_raw;                  // [2][R][ROWB]
    const unsigned short *wimg = reinterpret_cast<const unsigned short *>(a.w);
    const float *inb = a.in + (int64_t)b * a.in_bs;
    const int ng16 = CinP / 16, nchunks = CinP / CX_KCH, K = a.K;
    const float inv_scale = reinterpret_cast<const float *>(wimg + (int64_t)(CoutP / 32) * ng16 * K * 1024)[1];
    const int npass = (R + 127) / 128;

    f32x16 acc[RBW][NCB];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[i][j] = (f32x16){0};

    // ---- B staging (one chunk ahead, in registers): thread (frame row sf + 128 pass, channel group scg of 16 channels)
    float pv[NPASS_MAX][16];
    const int sf = tid & 127, scg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const rsrc_t d_in = make_rsrc(inb);
    auto issue_b = [&](int c0) {
#pragma unroll
        for (int p = 0; p < NPASS_MAX; ++p) {
            if (p < npass) {
                const int ti = t0 + lo + p * 128 + sf;
                const unsigned vo = (unsigned)min(max(ti, 0), a.T_in - 1) * 4u;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int cc = min(c0 + scg * 16 + k, a.Cin - 1);
                    pv[p][k] = buf_load(d_in, vo, (unsigned)(cc * (int)a.in_cs) * 4u);
                }
            }
        }
    };
    float amax = 0.0f;
    auto commit_b = [&](auto PROC, int c0) __attribute__((always_inline)) {
        constexpr int kPro = decltype(PROC)::value;
#pragma unroll
        for (int p = 0; p < NPASS_MAX; ++p) {
            if (p < npass) {
                const int row = p * 128 + sf;
                const int ti = t0 + lo + row;
                const bool tv = ti >= 0 && ti < a.T_in;
                unsigned h0[8], h1[8];  // channel pairs (2 j, 2 j + 1), packed
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = 2 * j + h;
                        v[h] = cx_pro<kPro>(pv[p][k], a.pro_param);                  // unconditional, straight-line
                        v[h] = (tv && c0 + scg * 16 + k < a.Cin) ? v[h] : 0.0f;      // select, no branch
                    }
                    amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
                    split2_f16(v[0], v[1], h0[j], h1[j]);
                }
                if (row < R) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        cx_u32x4 u0, u1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            u0[e] = h0[4 * q + e];
                            u1[e] = h1[4 * q + e];
                        }
                        *reinterpret_cast<cx_u32x4 *>(Bs + row * CX_ROWB + (scg * 16 + 8 * q) * 2) = u0;
                        *reinterpret_cast<cx_u32x4 *>(Bs + piece_bytes + row * CX_ROWB + (scg * 16 + 8 * q) * 2) = u1;
                    }
                }
            }
        }
    };

    // ---- A ring: k-step s = (chunk, tap, h): fragments of the wave's RBW row blocks, two k-steps ahead
    const rsrc_t d_w = make_rsrc(wimg);
    const unsigned lane16 = 16u * (unsigned)lane;
    const int rb_first = (r0 + wm * 32 * RBW) / 32;
    auto a_off = [&](int chunk, int tap, int h, int i, int piece) {
        const int rb = min(rb_first + i, CoutP / 32 - 1);  // (a wave fully outside CoutP re-reads the last block; never stored)
        return (unsigned)((((rb * ng16 + chunk * 2 + h) * K + tap) * 2 + piece) * 1024);
    };
    cx_u32x4 A[2][RBW][2];
    auto load_a = [&](int slot_h, int chunk, int tap) {
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) A[slot_h][i][q] = cx_load_u4(d_w, lane16, a_off(chunk, tap, slot_h, i, q));
    };

    issue_b(0);
    load_a(0, 0, 0);
    load_a(1, 0, 0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // MFMAs of the previous chunk are done with the tile
        switch (a.pro) {
            case SET_PRO_LRELU: commit_b(cx_ic<SET_PRO_LRELU>{}, c * CX_KCH); break;
            case SET_PRO_DIV: commit_b(cx_ic<SET_PRO_DIV>{}, c * CX_KCH); break;
            default: commit_b(cx_ic<SET_PRO_NONE>{}, c * CX_KCH); break;
        }
        __syncthreads();
        if (c + 1 < nchunks) issue_b((c + 1) * CX_KCH);
        for (int tap = 0; tap < K; ++tap) {
            const int off = tap * a.dil - a.pad - lo;  // >= 0: frame-row shift of this tap inside the B tile
            // the k-step two ahead: next tap of this chunk, or tap 0 of the next chunk (clamped at the very end)
            const int tap_n = tap + 1 < K ? tap + 1 : 0;
            const int c_n = tap + 1 < K ? c : min(c + 1, nchunks - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                cx_u32x4 Bv[NCB][2];
#pragma unroll
                for (int j = 0; j < NCB; ++j) {
                    const unsigned bo = (unsigned)((wn * 32 * NCB + j * 32 + l31 + off) * CX_ROWB + (h * 16 + half * 8) * 2);
                    Bv[j][0] = *reinterpret_cast<const cx_u32x4 *>(Bs + bo);
                    Bv[j][1] = *reinterpret_cast<const cx_u32x4 *>(Bs + piece_bytes + bo);
                }
                cx_u32x4 Ac[RBW][2];
#pragma unroll
                for (int i = 0; i < RBW; ++i) {
                    Ac[i][0] = A[h][i][0];
                    Ac[i][1] = A[h][i][1];
                }
                load_a(h, c_n, tap_n);
                __builtin_amdgcn_s_setprio(1);
                // a1 b0, a0 b1, a0 b0 (small terms first); the RBW x NCB accumulators interleave
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < RBW; ++i)
#pragma unroll
                        for (int j = 0; j < NCB; ++j) acc[i][j] = cx_mma(Ac[i][t == 0 ? 1 : 0], Bv[j][t == 1 ? 1 : 0], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }
    if (!(amax < 32768.0f)) __hip_atomic_store(&g_x2_range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if constexpr (PHASES) {
        // ---- transposed-conv epilogue: out[co][t u + p - P] = acc / scale + bias[co] ----
        const rsrc_t d_o = make_rsrc(a.out + (int64_t)b * a.out_bs);
        const rsrc_t d_pb = make_rsrc(a.bias ? a.bias : a.out);
        const int n_ch = a.Cout / ph_u;
        const bool vec = (ph_u & 3) == 0 && (ph_pad & 3) == 0 && (a.out_cs & 3) == 0 && (a.T_out & 3) == 0;
        // every bias value this lane needs, fetched BEFORE the first store: vmcnt counts loads and stores in one order, so a load issued
        // after a store cannot be waited for without waiting for that store's completion (the epilogue was load -> wait -> store -> load ...)
        float bz[RBW][4][4];
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = (r0 + (wm * RBW + i) * 32 + 8 * g + 4 * half + e) / ph_u;
                    bz[i][g][e] = a.bias ? buf_load(d_pb, (unsigned)min(c, n_ch - 1) * 4u, 0u) : 0.0f;
                }
        // ... and waited for HERE, once, in straight-line code: left to the wait-count pass, every later block (the paths below are
        // wave-uniform branches) re-waits vmcnt(0) because some predecessor path did not consume the values yet
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bz[i][g][e]));
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
            if (r0 + (wm * RBW + i) * 32 >= a.Cout) continue;
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                const int t = t0 + (wn * NCB + j) * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int R0 = r0 + (wm * RBW + i) * 32 + 8 * g + 4 * half;  // rows R0 .. R0 + 3
                    const int co = R0 / ph_u, p0 = R0 % ph_u;
                    const float bias = bz[i][g][0];
                    const int n0 = t * ph_u + p0 - ph_pad;
                    if (vec) {
                        // (stores masked by range instead of per-store branches, see conv_bf16_epilogue in bf16.hip)
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e] * inv_scale + bias;
                        buf_store4(o, d_o, (t < a.T_iter && co < n_ch && n0 >= 0 && n0 + 3 < a.T_out) ? (unsigned)(co * (int)a.out_cs + n0) * 4u : BUF_OOB, 0u);
                    } else if (ph_u == 2) {
                        // stride 2: registers 4 g + {0, 1} are the two phases of channel co at frame t -> one 8-byte store (a wave
                        // then writes 64 consecutive samples of a channel instead of every other one), {2, 3} those of co + 1
                        typedef unsigned cx_u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int c = co + q;
                            const float be = bz[i][g][2 * q];
                            const float v0 = acc[i][j][4 * g + 2 * q] * inv_scale + be, v1 = acc[i][j][4 * g + 2 * q + 1] * inv_scale + be;
                            const unsigned off = (unsigned)(c * (int)a.out_cs + n0) * 4u;
                            // masked by range, no branches (see conv_bf16_epilogue): the 8-byte store when both samples exist, else
                            // the one that does (only the first / last frame of an utterance takes the 4-byte forms)
                            const bool ok = t < a.T_iter && c < n_ch, both = n0 >= 0 && n0 + 1 < a.T_out;
                            cx_u32x2 o;
                            o[0] = __builtin_bit_cast(unsigned, v0); o[1] = __builtin_bit_cast(unsigned, v1);
                            __builtin_amdgcn_raw_buffer_store_b64(o, d_o, (int)((ok && both) ? off : BUF_OOB), 0, 0);
                            if (__builtin_amdgcn_ballot_w64(ok && !both) != 0) {  // wave-uniform: some lane sits on an utterance edge
                                buf_store(v0, d_o, (ok && !both && n0 >= 0 && n0 < a.T_out) ? off : BUF_OOB, 0u);
                                buf_store(v1, d_o, (ok && !both && n0 + 1 >= 0 && n0 + 1 < a.T_out) ? off + 4u : BUF_OOB, 0u);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int R = R0 + e, c = R / ph_u, n = t * ph_u + R % ph_u - ph_pad;
                            const bool okk = t < a.T_iter && c < n_ch && n >= 0 && n < a.T_out;
                            const float be = bz[i][g][e];
                            buf_store(acc[i][j][4 * g + e] * inv_scale + be, d_o, okk ? (unsigned)(c * (int)a.out_cs + n) * 4u : BUF_OOB, 0u);
                        }
                    }
                }
            }
        }
        return;
    }
    // ---- epilogue (fp32): v = act((acc / scale + bias) * alpha) + res ; * mask ; (+ previous output, / out_div) ----
    const bool has_div = a.accumulate && a.out_div != 0.0f;
    const bool has_res = a.res != nullptr, has_bias = a.bias != nullptr, has_acc = a.accumulate != 0;
    const rsrc_t d_out = make_rsrc(a.out + (int64_t)b * a.out_bs);
    const rsrc_t d_res = make_rsrc(has_res ? a.res + (int64_t)b * a.res_bs : a.out + (int64_t)b * a.out_bs);
    const rsrc_t d_bias = make_rsrc(has_bias ? a.bias : a.out);
    auto tile = [&](auto ACT, const f32x16 &av, int i, int j) __attribute__((always_inline)) {
        constexpr int kAct = decltype(ACT)::value;
        const int rbase = r0 + (wm * RBW + i) * 32 + 4 * half;  // register r of this lane is row rbase + (r&3) + 8*(r>>2)
        const int t = t0 + (wn * NCB + j) * 32 + l31;
        const int n = t * a.out_stride + a.out_off;             // output sample (polyphase transposed conv: stride > 1)
        const bool tv = t < a.T_iter && n >= 0 && n < a.T_out;
        const int nc = min(max(n, 0), a.T_out - 1);
        float mk = 1.0f;
        if (a.mask) mk = buf_load(make_rsrc(a.mask + (int64_t)b * a.T_out), (unsigned)nc * 4u, 0u);
        float bi[16], rv[16], ov[16];
        unsigned ro[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ro[r] = (unsigned)min(rbase + (r & 3) + 8 * (r >> 2), a.Cout - 1);
            bi[r] = rv[r] = ov[r] = 0.0f;
        }
        if (has_bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = buf_load(d_bias, ro[r] * 4u, 0u);
        }
        if (has_res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = buf_load(d_res, (ro[r] * (unsigned)a.res_cs + (unsigned)nc) * 4u, 0u);
        }
        if (has_acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ov[r] = buf_load(d_out, (ro[r] * (unsigned)a.out_cs + (unsigned)nc) * 4u, 0u);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            float y = (av[r] * inv_scale + bi[r]) * a.alpha;
            if constexpr (kAct == SET_ACT_RELU) y = y > 0.0f ? y : 0.0f;
            else if constexpr (kAct == SET_ACT_LRELU) y = y > 0.0f ? y : y * a.act_param;
            else if constexpr (kAct != SET_ACT_NONE) y = dev_act(y, a.act, a.act_param);
            y = (y + rv[r]) * mk + ov[r];
            if (has_div) y = y / a.out_div;
            buf_store(y, d_out, (tv && row < a.Cout) ? (ro[r] * (unsigned)a.out_cs + (unsigned)nc) * 4u : BUF_OOB, 0u);  // masked by range, no branch (see conv_bf16_epilogue)
        }
    };
    auto finish = [&](auto ACT) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RBW; ++i) {
            if (r0 + (wm * RBW + i) * 32 >= a.Cout) continue;  // wave-uniform: a fully padded row block
#pragma unroll
            for (int j = 0; j < NCB; ++j) tile(ACT, acc[i][j], i, j);
        }
    };
    switch (a.act) {
        case SET_ACT_NONE: finish(cx_ic<SET_ACT_NONE>{}); break;
        case SET_ACT_RELU: finish(cx_ic<SET_ACT_RELU>{}); break;
        case SET_ACT_LRELU: finish(cx_ic<SET_ACT_LRELU>{}); break;
        default: finish(cx_ic<-1>{}); break;  // gelu / tanh / softplus / mish: run-time dev_act
    }
}

template <int WM, int WN, int RBW, int NCB, bool PHASES = false>
int launch_conv_x2(const SetConv1dArgs &a, int lo, int halo, hipStream_t s, int ph_u = 0, int ph_pad = 0) {
    constexpr int MB = 32 * RBW * WM, NB = 32 * NCB * WN;
    const int CinP = cx_round_up(a.Cin, CX_KCH), CoutP = cx_round_up(a.Cout, 32);
    const size_t lds = (size_t)2 * (NB + halo) * CX_ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv1d_x2_kernel<WM, WN, RBW, NCB, PHASES>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "conv x2 attr");
        attr_set = true;
    }
    if (lds > 96 * 1024) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(f16x2)", "tile does not fit LDS");
    dim3 grid((a.T_iter + NB - 1) / NB, (a.Cout + MB - 1) / MB, a.B), block(256);
    hipLaunchKernelGGL((conv1d_x2_kernel<WM, WN, RBW, NCB, PHASES>), grid, block, lds, s, a, lo, halo, CinP, CoutP, ph_u, ph_pad);
    return set_check_launch("set_conv1d(f16x2)");
}

}  // namespace

extern "C" int64_t set_packed_conv_weight_x2_size(int32_t Cout, int32_t Cin, int32_t K) {
    return (int64_t)2 * (cx_round_up(Cout, 32) / 32) * (cx_round_up(Cin, CX_KCH) / 16) * K * 512 + 8;  // fp16 elements (+ 4 floats)
}

static int pack_x2(const float *w, void *wp, int Cout, int Cin, int K, int64_t w_base, int64_t w_sco, int64_t w_sci, int64_t w_stap,
                   int scale_exp, int phases, int kfull, void *stream, const char *what) {
    SET_REQUIRE(w && wp && Cout > 0 && Cin > 0 && K > 0 && scale_exp >= -60 && scale_exp <= 60, what);
    const int CoutP = cx_round_up(Cout, 32), CinP = cx_round_up(Cin, CX_KCH);
    const int64_t n = (int64_t)(CoutP / 32) * (CinP / 16) * K * 512;
    hipLaunchKernelGGL(pack_conv_x2_kernel, dim3(set_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<unsigned short *>(wp), Cout, Cin, K, CoutP, CinP, n, w_base, w_sco, w_sci, w_stap,
                       ldexpf(1.0f, scale_exp), phases, kfull);
    return set_check_launch(what);
}

extern "C" int set_pack_conv_weight_x2(const float *w, void *wp, int32_t Cout, int32_t Cin, int32_t K, int64_t w_base, int64_t w_sco,
                                       int64_t w_sci, int64_t w_stap, int32_t scale_exp, void *stream) {
    return pack_x2(w, wp, Cout, Cin, K, w_base, w_sco, w_sci, w_stap, scale_exp, 0, 0, stream, "set_pack_conv_weight_x2");
}

// ConvTranspose1d weight [Cin][Cout][k], stride u: image of the u * Cout (channel, phase) rows x ceil(k / u) taps
extern "C" int64_t set_packed_conv_transpose_x2_size(int32_t Cout, int32_t Cin, int32_t k, int32_t u) {
    return set_packed_conv_weight_x2_size(Cout * u, Cin, (k + u - 1) / u);
}
extern "C" int set_pack_conv_transpose_x2(const float *w, void *wp, int32_t Cout, int32_t Cin, int32_t k, int32_t u, int32_t scale_exp,
                                          void *stream) {
    SET_REQUIRE(u >= 1 && k >= u, "set_pack_conv_transpose_x2");
    return pack_x2(w, wp, Cout * u, Cin, (k + u - 1) / u, 0, 0, 0, 0, scale_exp, u, k, stream, "set_pack_conv_transpose_x2");
}

/* nn.ConvTranspose1d(Cin, Cout, k, stride u, padding P) forward, every output phase in ONE launch: out[b][co][n] = bias[co] +
 * sum_ci sum_j Wt[ci][co][u j + p] * pro(in[b][ci][q - j]),  n + P = u q + p  (hifigan.py:114-115).  in [B][Cin][T_in], out
 * [B][Cout][T_out = (T_in - 1) u - 2 P + k] contiguous; wp = image of set_pack_conv_transpose_x2. */
extern "C" int set_conv_transpose1d_x2(const float *in, const void *wp, const float *bias, float *out, int32_t B, int32_t Cin,
                                       int32_t Cout, int32_t k, int32_t u, int32_t P, int32_t T_in, int32_t pro, float pro_param,
                                       void *stream) {
    SET_REQUIRE(in && wp && out && B > 0 && Cin > 0 && Cout > 0 && u >= 1 && k >= u && P >= 0 && T_in > 0, "set_conv_transpose1d_x2");
    const int J = (k + u - 1) / u;
    SetConv1dArgs a = {};
    a.in = in; a.w = reinterpret_cast<const float *>(wp); a.bias = bias; a.out = out;
    a.T_in = T_in; a.T_iter = T_in + J - 1; a.T_out = (T_in - 1) * u - 2 * P + k;
    a.in_bs = (int64_t)Cin * T_in; a.in_cs = T_in; a.out_bs = (int64_t)Cout * a.T_out; a.out_cs = a.T_out;
    a.B = B; a.Cin = Cin; a.Cout = Cout * u; a.K = J; a.dil = -1; a.pad = 0;
    a.out_stride = u; a.out_off = -P; a.pro = pro; a.pro_param = pro_param; a.alpha = 1.0f; a.impl = SET_IMPL_F16X2;
    SET_REQUIRE(a.T_out > 0, "set_conv_transpose1d_x2");
    if (((int64_t)Cout * a.T_out) * 4 >= ((int64_t)1 << 31) || ((int64_t)Cin * T_in) * 4 >= ((int64_t)1 << 31))
        return set_fail(SET_E_UNSUPPORTED, "set_conv_transpose1d_x2", "one batch slice of in / out exceeds 2 GiB");
    const int lo = -(J - 1), halo = J - 1;
    hipStream_t s = (hipStream_t)stream;
    // (round 6: 128-frame wave tiles, see set_conv1d_x2_ below; before: <2, 2, 2, 2> / <1, 4, 2, 2>)
    if (a.Cout > 64) return launch_conv_x2<4, 1, 1, 4, true>(a, lo, halo, s, u, P);
    return launch_conv_x2<2, 2, 1, 4, true>(a, lo, halo, s, u, P);
}

/* *flag = the sticky "an activation left the fp16 range of the splitting" word (synchronises the device); reset != 0 clears it */
int set_resblock_pair_range_flag_(int *flag, int reset);  // csrc/resblock_x2.hip (its own device word)

extern "C" int set_conv_x2_range_flag(int32_t *flag, int32_t reset) {
    int v = 0, vp = 0;
    SET_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_x2_range_flag), sizeof(int)), "set_conv_x2_range_flag");
    if (int rc = set_resblock_pair_range_flag_(&vp, reset)) return rc;
    if (flag) *flag = v | vp;
    if (reset && v) {
        const int z = 0;
        SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_x2_range_flag), &z, sizeof(int)), "set_conv_x2_range_flag");
    }
    return SET_OK;
}

// called by set_conv1d (csrc/conv1d.hip) for impl == SET_IMPL_F16X2
int set_conv1d_x2_dispatch(const SetConv1dArgs &a, hipStream_t s) {
    if (a.in_chan_add) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(f16x2)", "per-channel add is an fp32 / bf16 path");
    const int o_first = -a.pad, o_last = (a.K - 1) * a.dil - a.pad;
    const int lo = o_first < o_last ? o_first : o_last;
    const int halo = (o_first < o_last ? o_last : o_first) - lo;
    if (halo > 128) return set_fail(SET_E_UNSUPPORTED, "set_conv1d(f16x2)", "receptive field > 128");
    const int64_t cs = a.res && a.res_cs > a.out_cs ? a.res_cs : a.out_cs;
    if (((int64_t)cx_round_up(a.Cout, 32) * cs + a.T_out) * 4 >= ((int64_t)1 << 31) ||
        ((int64_t)a.Cin * a.in_cs + a.T_in) * 4 >= ((int64_t)1 << 31))
        return set_fail(SET_E_UNSUPPORTED, "set_conv1d(f16x2)", "one batch slice of in / out / res exceeds 2 GiB");
    // wave tile 64 rows x 64 frames (4 accumulators): with 8 (2 x 4 blocks) the staging registers of the next chunk no
    // longer fit beside them
    // round 6: 128-frame WAVE tiles -- every weight fragment meets four column blocks of one wave instead of two column blocks of two waves
    // (half the fragment loads per MFMA; same k order per output: bit-identical).  With the same change in the ResBlock-pair kernel HiFi-GAN V1 at
    // B = 64 went 103.1 -> 101.5 -> 100.8 ms per forward on one box (profiles/r06_rp_wide_ab.log, r06_cx_wide_ab.log); before: <2, 2, 2, 2> / <1, 4, 2, 2>
    if (a.Cout > 64) return launch_conv_x2<4, 1, 1, 4>(a, lo, halo, s);   // 128 rows x 128 frames, wave tile 32 x 128
    if (a.Cout > 32) return launch_conv_x2<2, 2, 1, 4>(a, lo, halo, s);   //  64 rows x 256 frames, wave tile 32 x 128
    return launch_conv_x2<1, 4, 1, 2>(a, lo, halo, s);                    //  32 rows x 256 frames
}
