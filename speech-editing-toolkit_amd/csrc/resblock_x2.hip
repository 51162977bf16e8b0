// Fused HiFi-GAN ResBlock1 PAIR on the two-piece fp16 operands (hifigan.py:51-58: one iteration of
//     xt = lrelu(x); xt = c1(xt)  [k taps, dilation d]; xt = lrelu(xt); xt = c2(xt)  [k taps, dilation 1]; x = xt + x
// and, for the block's last pair, the MRF sum / mean of hifigan.py:131-137 in the epilogue).
//
// Why: unfused, the intermediate xt makes a round trip through HBM (write 4 C T bytes, read them back) and x is read twice
// (conv input, residual) -- 20 C T bytes per pair against 8 C T fused.  The 3-tap pairs of every stage and all pairs of the
// 32 / 64-channel stages sit below the MFMA ridge of the split-operand kernel (C k / 6 FLOP per byte against ~60 - 140), so
// they ran at ~3 TB/s of HBM traffic instead of at the MFMA rate (profiles/r02_hifigan_stages.log).
//
// One block = all C output channels x NB intermediate frames (C <= 128):
//   phase 1  t = W1 (*) lrelu(x) + b1 on the frames [f0 - h2, f0 - h2 + NB)   (h2 = (k - 1) / 2: what conv 2 needs around the
//            block's outputs) -- exactly the main loop of conv1d_x2_kernel (csrc/conv_x2.hip): x staged in 32-channel chunks,
//            prologue + fp16 split once per chunk, weights from the packed image two k-steps ahead;
//            epilogue 1: lrelu(t), zero outside [0, T) (conv 2 pads with zeros, it does not see conv 1 evaluated out of
//            range), split into two fp16 pieces, written to an LDS tile [piece][frame][C] that overlays the x chunks;
//   phase 2  y = W2 (*) t + b2 + x on the NB - 2 h2 central frames: a GEMM whose B operand is that tile (taps = row shifts),
//            no staging, no barrier in the loop; epilogue 2: + residual x (re-read: L2-hot), MRF accumulate / divide, store.
// The k-step order of both GEMMs is the one of conv1d_x2_kernel (chunk, tap, half-chunk; pieces a1 b0, a0 b1, a0 b0) and the
// intermediate takes the same fp32 value before it is split, so the fused pair is BIT-IDENTICAL to the two launches it
// replaces (tests/test_gpu_x2conv.py::test_fused_resblock_pair_equals_two_convs).
// Measured and dropped (round 3): the residual x of epilogue 2 fetched before GEMM 2 (epilogue 2 is 43 - 55 % of a 32- / 64-channel
// block's life, tools/resblock_phase_probe.py): same-box A/B of the V1 stages 3.96 -> 3.94 ms (C = 32, k = 3), 4.99 -> 5.15 (C = 64),
// the 64 extra live registers cost the wider shapes more than the earlier issue gains -- the blocks wait on memory THROUGHPUT
// (2.6 - 3.8 TB/s of dword-per-lane accesses), not on the position of the loads.
// Measured and dropped (round 3): 128-frame tiles for the 32- / 64-channel shapes (124 / 156 registers: 4 / 3 blocks per CU instead of
// 3 / 2): same-box A/B of the V1 stages 3.9 -> 4.2 ms (C = 32, k = 3), 6.3 -> 7.6 (k = 11), 10.2 -> 12.0 (C = 64, k = 11).  A plain copy
// of the same row tiles runs at 5.7 TB/s (tools/hw/rowtile_copy.hip; 0.59 ms per 32-channel pair against 1.3 - 1.5 ms here): the
// phases of a block (x round trip, GEMM 1, LDS tile, GEMM 2, residual round trip) add up, and co-resident blocks hide little of it.
// Measured and dropped (round 3): a persistent variant (blocks walk tiles; the next tile's first x chunk is fetched as soon as the
// chunk registers are free, under phase 1's MFMAs / epilogue 1 / phase 2 / epilogue 2).  The 48 chunk registers then live across
// the whole tile: 55 - 84 spilled VGPRs for the 64- / 128- / 256-channel shapes, occupancy 3 -> 2 for the 32-channel one, and the
// V1 forward went from 105 to 127 ms.
#include <stdlib.h>

#include "common.h"

typedef _Float16 rp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rp_u32x2 __attribute__((ext_vector_type(2)));

// (device symbols do not link across translation units without -fgpu-rdc: this file keeps its own sticky range word, and
// set_conv_x2_range_flag (csrc/conv_x2.hip) reads both through set_resblock_pair_range_flag_)
__device__ int g_rp_range_flag = 0;
// debug, builds with -DSET_RP_PROBE only (tools/resblock_phase_probe.py): thread 0 of every 16th block of batch row 1 adds the
// s_memtime ticks of its phases to buf[0..4] and counts itself in buf[7], see set_debug_resblock_phase_buffer
__device__ uint64_t *g_rp_phase_buf = nullptr;

namespace {

constexpr int RP_KCH = 32;                // channels per staged chunk of x
constexpr int RP_ROWB = RP_KCH * 2 + 16;  // bytes per LDS row of a chunk

__device__ __forceinline__ unsigned short rp_f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float rp_h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ f32x16 rp_mma(rp_u32x4 a, rp_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rp_f16x8, a), __builtin_bit_cast(rp_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ rp_u32x4 rp_load_u4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
static inline int rp_round_up(int x, int m) { return (x + m - 1) / m * m; }

template <int WM, int WN, int RBW, int NCB>
__global__ void __launch_bounds__(256, 2) resblock_pair_x2_kernel(SetResblockPairArgs a, int CP, int NV) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NB = 32 * NCB * WN;
    constexpr int NPASS_MAX = (NB + 128 + 127) / 128;  // frame passes of 128 rows (2 h1 <= 128)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;
    const int K = a.K, C = a.C, T = a.T;
    const int h2 = (K - 1) / 2, h1 = a.dil * h2;
    const int b = blockIdx.y;
    const int f0 = blockIdx.x * NV - h2;            // first intermediate frame of the block (column 0)
    const int R1 = NB + 2 * h1;                     // frame rows of an x chunk tile; row r <-> frame f0 - h1 + r
    const unsigned piece1 = (unsigned)(R1 * RP_ROWB);
    const int TROW = CP * 2 + 16;                   // bytes per row of the intermediate tile [piece][NB + 2 h2][CP]
    const unsigned piece2 = (unsigned)((NB + 2 * h2) * TROW);
    unsigned char *Bs = smem_raw;
    const unsigned short *w1 = reinterpret_cast<const unsigned short *>(a.w1);
    const unsigned short *w2 = reinterpret_cast<const unsigned short *>(a.w2);
    const float *xb = a.x + (int64_t)b * a.x_bs;
    const int ng16 = CP / 16, nchunks = CP / RP_KCH;
    const int64_t tail = (int64_t)(CP / 32) * ng16 * K * 1024;  // fp16 elements before the image's scale words
    const float inv_s1 = reinterpret_cast<const float *>(w1 + tail)[1];
    const float inv_s2 = reinterpret_cast<const float *>(w2 + tail)[1];
    const int npass = (R1 + 127) / 128;
    const rsrc_t d_x = make_rsrc(xb);
    const unsigned lane16 = 16u * (unsigned)lane;
    const int rb_first = wm * RBW;
#ifdef SET_RP_PROBE
    uint64_t *const pbuf = g_rp_phase_buf;
    const bool probe = pbuf != nullptr && tid == 0 && blockIdx.y == 1 && (blockIdx.x & 15) == 1;
    uint64_t tprev = probe ? __builtin_amdgcn_s_memtime() : 0;
    auto stamp = [&](int k) {
        if (probe) {
            const uint64_t t = __builtin_amdgcn_s_memtime();
            atomicAdd(reinterpret_cast<unsigned long long *>(pbuf + k), (unsigned long long)(t - tprev));
            tprev = t;
        }
    };
#else
    auto stamp = [](int) {};
#endif

    f32x16 acc[RBW][NCB];
#pragma unroll
    for (int i = 0; i < RBW; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc[i][j] = (f32x16){0};

    // ---- x staging (one chunk ahead, in registers): thread (frame row sf + 128 pass, channel group scg of 16 channels)
    float pv[NPASS_MAX][16];
    const int sf = tid & 127, scg = __builtin_amdgcn_readfirstlane(tid >> 7);
    auto issue_b = [&](int c0) {
#pragma unroll
        for (int p = 0; p < NPASS_MAX; ++p) {
            if (p < npass) {
                const int ti = f0 - h1 + p * 128 + sf;
                const unsigned vo = (unsigned)min(max(ti, 0), T - 1) * 4u;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int cc = min(c0 + scg * 16 + k, C - 1);
                    pv[p][k] = buf_load(d_x, vo, (unsigned)(cc * (int)a.x_cs) * 4u);
                }
            }
        }
    };
    float amax = 0.0f;
    auto commit_b = [&](int c0) {
#pragma unroll
        for (int p = 0; p < NPASS_MAX; ++p) {
            if (p < npass) {
                const int row = p * 128 + sf;
                const int ti = f0 - h1 + row;
                const bool tv = ti >= 0 && ti < T;
                unsigned q0[8], q1[8];  // channel pairs (2 j, 2 j + 1), packed
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = 2 * j + h;
                        v[h] = pv[p][k];
                        v[h] = v[h] > 0.0f ? v[h] : v[h] * a.slope;                   // unconditional, straight-line
                        v[h] = (tv && c0 + scg * 16 + k < C) ? v[h] : 0.0f;           // select, no branch
                    }
                    amax = fmaxf(fmaxf(amax, fabsf(v[0])), fabsf(v[1]));
                    split2_f16(v[0], v[1], q0[j], q1[j]);
                }
                if (row < R1) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        rp_u32x4 u0, u1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            u0[e] = q0[4 * q + e];
                            u1[e] = q1[4 * q + e];
                        }
                        *reinterpret_cast<rp_u32x4 *>(Bs + row * RP_ROWB + (scg * 16 + 8 * q) * 2) = u0;
                        *reinterpret_cast<rp_u32x4 *>(Bs + piece1 + row * RP_ROWB + (scg * 16 + 8 * q) * 2) = u1;
                    }
                }
            }
        }
    };

    // ---- A ring: fragments of the wave's RBW row blocks for (chunk, tap, half-chunk), two k-steps ahead
    auto a_off = [&](int chunk, int tap, int h, int i, int piece) {
        const int rb = min(rb_first + i, CP / 32 - 1);  // (a wave fully outside CP re-reads the last block; never stored)
        return (unsigned)((((rb * ng16 + chunk * 2 + h) * K + tap) * 2 + piece) * 1024);
    };
    rp_u32x4 A[2][RBW][2];
    auto load_a = [&](rsrc_t d_w, int slot_h, int chunk, int tap) {
#pragma unroll
        for (int i = 0; i < RBW; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) A[slot_h][i][q] = rp_load_u4(d_w, lane16, a_off(chunk, tap, slot_h, i, q));
    };

    // =========================== phase 1: t = W1 (*) lrelu(x) =====================================================
    {
        const rsrc_t d_w = make_rsrc(w1);
        issue_b(0);
        load_a(d_w, 0, 0, 0);
        load_a(d_w, 1, 0, 0);
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // MFMAs of the previous chunk are done with the tile
            commit_b(c * RP_KCH);
            __syncthreads();
            stamp(0);  // wait for the chunk's loads + convert + LDS write
            if (c + 1 < nchunks) issue_b((c + 1) * RP_KCH);
            for (int tap = 0; tap < K; ++tap) {
                const int off = tap * a.dil;  // frame-row shift of this tap inside the chunk tile
                const int tap_n = tap + 1 < K ? tap + 1 : 0;
                const int c_n = tap + 1 < K ? c : min(c + 1, nchunks - 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    rp_u32x4 Bv[NCB][2];
#pragma unroll
                    for (int j = 0; j < NCB; ++j) {
                        const unsigned bo = (unsigned)((wn * 32 * NCB + j * 32 + l31 + off) * RP_ROWB + (h * 16 + half * 8) * 2);
                        Bv[j][0] = *reinterpret_cast<const rp_u32x4 *>(Bs + bo);
                        Bv[j][1] = *reinterpret_cast<const rp_u32x4 *>(Bs + piece1 + bo);
                    }
                    rp_u32x4 Ac[RBW][2];
#pragma unroll
                    for (int i = 0; i < RBW; ++i) {
                        Ac[i][0] = A[h][i][0];
                        Ac[i][1] = A[h][i][1];
                    }
                    load_a(d_w, h, c_n, tap_n);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int i = 0; i < RBW; ++i)
#pragma unroll
                            for (int j = 0; j < NCB; ++j) acc[i][j] = rp_mma(Ac[i][t == 0 ? 1 : 0], Bv[j][t == 1 ? 1 : 0], acc[i][j]);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
            stamp(1);  // GEMM 1 of the chunk issued
        }
    }
    // the first fragments of W2 fly under epilogue 1
    const rsrc_t d_w2 = make_rsrc(w2);
    load_a(d_w2, 0, 0, 0);
    load_a(d_w2, 1, 0, 0);
    __syncthreads();  // every wave is done reading the last x chunk: the intermediate tile overlays it

    // ---- epilogue 1: lrelu(acc / s1 + b1), zero outside [0, T) and beyond C, split, 4 consecutive channels per 8-byte write;
    //      column c of the block sits in row c + h2 of the tile (h2 guard rows on either side are never written: they only feed
    //      the 2 h2 edge columns of phase 2, which are not stored)
    const rsrc_t d_b1 = make_rsrc(a.b1);
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        const int ch0 = (rb_first + i) * 32;
        if (ch0 >= CP) continue;  // wave-uniform
        float b1v[16];            // bias of register r's channel: ch0 + 4 half + (r & 3) + 8 (r >> 2), fetched as one batch
#pragma unroll
        for (int r = 0; r < 16; ++r) b1v[r] = buf_load(d_b1, (unsigned)min(ch0 + 4 * half + (r & 3) + 8 * (r >> 2), C - 1) * 4u, 0u);
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int col = (wn * NCB + j) * 32 + l31;
            const int fr = f0 + col;
            const bool fv = fr >= 0 && fr < T;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int chb = ch0 + 8 * g + 4 * half;  // channels chb .. chb + 3 (register 4 g + e)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][4 * g + e] * inv_s1 + b1v[4 * g + e];
                    v[e] = v[e] > 0.0f ? v[e] : v[e] * a.slope;
                    v[e] = (fv && chb + e < C) ? v[e] : 0.0f;
                    amax = fmaxf(amax, fabsf(v[e]));
                }
                const unsigned off = (unsigned)((col + h2) * TROW + chb * 2);
                unsigned lo0, lo1, hi0, hi1;
                split2_f16(v[0], v[1], lo0, lo1);
                split2_f16(v[2], v[3], hi0, hi1);
                const rp_u32x2 u0 = {lo0, hi0}, u1 = {lo1, hi1};
                *reinterpret_cast<rp_u32x2 *>(Bs + off) = u0;
                *reinterpret_cast<rp_u32x2 *>(Bs + piece2 + off) = u1;
            }
            acc[i][j] = (f32x16){0};
        }
    }
    if (!(amax < 32768.0f)) __hip_atomic_store(&g_rp_range_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    stamp(2);  // GEMM 1 drained + epilogue 1

    // =========================== phase 2: y = W2 (*) t ============================================================
    for (int c = 0; c < nchunks; ++c) {
        for (int tap = 0; tap < K; ++tap) {
            const int tap_n = tap + 1 < K ? tap + 1 : 0;
            const int c_n = tap + 1 < K ? c : min(c + 1, nchunks - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rp_u32x4 Bv[NCB][2];
#pragma unroll
                for (int j = 0; j < NCB; ++j) {  // output column c needs t at columns c + tap - h2, i.e. rows c + tap
                    const unsigned bo = (unsigned)((wn * 32 * NCB + j * 32 + l31 + tap) * TROW + (c * RP_KCH + h * 16 + half * 8) * 2);
                    Bv[j][0] = *reinterpret_cast<const rp_u32x4 *>(Bs + bo);
                    Bv[j][1] = *reinterpret_cast<const rp_u32x4 *>(Bs + piece2 + bo);
                }
                rp_u32x4 Ac[RBW][2];
#pragma unroll
                for (int i = 0; i < RBW; ++i) {
                    Ac[i][0] = A[h][i][0];
                    Ac[i][1] = A[h][i][1];
                }
                load_a(d_w2, h, c_n, tap_n);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < RBW; ++i)
#pragma unroll
                        for (int j = 0; j < NCB; ++j) acc[i][j] = rp_mma(Ac[i][t == 0 ? 1 : 0], Bv[j][t == 1 ? 1 : 0], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            }
        }
    }

    stamp(3);  // GEMM 2 issued
    // ---- epilogue 2 (fp32): y = ((acc / s2 + b2) + x) (+ previous output) (/ out_div) on the NV central columns ----
    const bool has_acc = a.accumulate != 0, has_div = has_acc && a.out_div != 0.0f;
    const rsrc_t d_out = make_rsrc(a.out + (int64_t)b * a.out_bs);
    const rsrc_t d_b2 = make_rsrc(a.b2);
#pragma unroll
    for (int i = 0; i < RBW; ++i) {
        if ((rb_first + i) * 32 >= C) continue;  // wave-uniform: a fully padded row block
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int col = (wn * NCB + j) * 32 + l31;
            const int fr = f0 + col;
            const bool tv = col >= h2 && col < NB - h2 && fr < T;  // (fr >= 0 follows from col >= h2)
            const int fc = min(max(fr, 0), T - 1);
            const int rbase = (rb_first + i) * 32 + 4 * half;  // register r of this lane is row rbase + (r & 3) + 8 (r >> 2)
            float bi[16], rv[16], ov[16];
            unsigned ro[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ro[r] = (unsigned)min(rbase + (r & 3) + 8 * (r >> 2), C - 1);
                ov[r] = 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = buf_load(d_b2, ro[r] * 4u, 0u);
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = buf_load(d_x, (ro[r] * (unsigned)a.x_cs + (unsigned)fc) * 4u, 0u);
            if (has_acc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = buf_load(d_out, (ro[r] * (unsigned)a.out_cs + (unsigned)fc) * 4u, 0u);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                float y = acc[i][j][r] * inv_s2 + bi[r];  // the order of conv1d_x2_kernel's epilogue (alpha = 1, no mask)
                y = (y + rv[r]) + ov[r];
                if (has_div) y = y / a.out_div;
                buf_store(y, d_out, (tv && row < C) ? (ro[r] * (unsigned)a.out_cs + (unsigned)fc) * 4u : BUF_OOB, 0u);  // masked by range, no branch (see conv_bf16_epilogue)
            }
        }
    }
    stamp(4);  // GEMM 2 drained + epilogue 2 (residual / accumulate loads, stores issued)
#ifdef SET_RP_PROBE
    if (probe) atomicAdd(reinterpret_cast<unsigned long long *>(pbuf + 7), 1ull);
#endif
}

template <int WM, int WN, int RBW, int NCB>
int launch_pair(const SetResblockPairArgs &a, hipStream_t s) {
    constexpr int NB = 32 * NCB * WN;
    const int CP = rp_round_up(a.C, RP_KCH);
    const int h2 = (a.K - 1) / 2, h1 = a.dil * h2;
    const int NV = NB - 2 * h2;
    const size_t lds1 = (size_t)2 * (NB + 2 * h1) * RP_ROWB, lds2 = (size_t)2 * (NB + 2 * h2) * (CP * 2 + 16);
    const size_t lds = lds1 > lds2 ? lds1 : lds2;
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(resblock_pair_x2_kernel<WM, WN, RBW, NCB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024), "resblock pair attr");
        attr_set = true;
    }
    if (lds > 96 * 1024) return set_fail(SET_E_UNSUPPORTED, "set_resblock_pair_x2", "tile does not fit LDS");
    dim3 grid((a.T + NV - 1) / NV, a.B), block(256);
    hipLaunchKernelGGL((resblock_pair_x2_kernel<WM, WN, RBW, NCB>), grid, block, lds, s, a, CP, NV);
    return set_check_launch("set_resblock_pair_x2");
}

}  // namespace

// read (and optionally clear) this file's range word; called by set_conv_x2_range_flag
int set_resblock_pair_range_flag_(int *flag, int reset) {
    int v = 0;
    SET_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_rp_range_flag), sizeof(int)), "set_conv_x2_range_flag");
    *flag = v;
    if (reset && v) {
        const int z = 0;
        SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rp_range_flag), &z, sizeof(int)), "set_conv_x2_range_flag");
    }
    return SET_OK;
}

extern "C" int set_debug_resblock_phase_buffer(uint64_t *buf) {
#ifndef SET_RP_PROBE
    if (buf) return set_fail(SET_E_UNSUPPORTED, "set_debug_resblock_phase_buffer", "library built without -DSET_RP_PROBE");
#endif
    SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rp_phase_buf), &buf, sizeof(buf)), "set_debug_resblock_phase_buffer");
    return SET_OK;
}

extern "C" int64_t set_sizeof_resblock_pair_args(void) { return (int64_t)sizeof(SetResblockPairArgs); }

/* 0 when set_resblock_pair_x2 takes this shape, SET_E_UNSUPPORTED otherwise (callers then run the two convs separately) */
extern "C" int set_resblock_pair_x2_supported(int32_t C, int32_t K, int32_t dil, int32_t T) {
    // up to 128 channels a block holds a 128- or 256-frame tile; 129 .. 256 channels only fit 64-frame tiles, of which
    // K - 1 columns are recomputed by the neighbours: measured (B = 64, C = 256, T = 6400, three pairs) 3.54 vs 4.38 ms for
    // K = 3, 6.62 vs 6.73 for K = 7, 10.64 vs 9.17 for K = 11 -> fused up to 5 taps only
    const int max_c = 256;
    if (C > 128 && K > 5) return SET_E_UNSUPPORTED;
    if (C < 16 || C > max_c || C > 256 || K < 3 || (K & 1) == 0 || K > 15 || dil < 1 || dil * (K - 1) > 128 || T < 64) return SET_E_UNSUPPORTED;
    return SET_OK;
}

extern "C" int set_resblock_pair_x2(const SetResblockPairArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_resblock_pair_x2");
    const SetResblockPairArgs &a = *args;
    SET_REQUIRE(a.x && a.w1 && a.b1 && a.w2 && a.b2 && a.out && a.out != a.x && a.B > 0 && a.T > 0, "set_resblock_pair_x2");
    if (set_resblock_pair_x2_supported(a.C, a.K, a.dil, a.T) != SET_OK)
        return set_fail(SET_E_UNSUPPORTED, "set_resblock_pair_x2", "shape outside the fused kernel (C in 16..256, odd K <= 15 (<= 5 above 128 channels), receptive field <= 128)");
    SET_REQUIRE(!(a.out_div != 0.0f && !a.accumulate), "set_resblock_pair_x2(out_div needs accumulate)");
    if (((int64_t)rp_round_up(a.C, 32) * a.x_cs + a.T) * 4 >= ((int64_t)1 << 31) ||
        ((int64_t)rp_round_up(a.C, 32) * a.out_cs + a.T) * 4 >= ((int64_t)1 << 31))
        return set_fail(SET_E_UNSUPPORTED, "set_resblock_pair_x2", "one batch slice of x / out exceeds 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    if (a.C > 128) return launch_pair<4, 1, 2, 2>(a, s);  // 256 rows x  64 frames
    // round 6: 128-frame WAVE tiles for 33 .. 128 channels -- every weight fragment meets four column blocks of one wave instead of two column
    // blocks of two waves (half the fragment loads per MFMA; same k order per output: bit-identical).  V1, B = 64: 103.1 -> 101.5 ms per forward,
    // the C = 128 / C = 64 stages +2 .. 5 % (profiles/r06_rp_wide_ab.log); before: <2, 2, 2, 2> and <1, 4, 2, 2>.
    if (a.C > 64) return launch_pair<4, 1, 1, 4>(a, s);   // 128 rows x 128 frames, wave tile 32 x 128
    if (a.C > 32) return launch_pair<2, 2, 1, 4>(a, s);   //  64 rows x 256 frames, wave tile 32 x 128
    return launch_pair<1, 4, 1, 2>(a, s);                 //  32 rows x 256 frames
}
