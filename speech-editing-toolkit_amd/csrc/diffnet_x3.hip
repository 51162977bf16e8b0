// DiffNet residual stack (inference) with fp32 operands SPLIT into 16-bit pieces (two fp16 or three bf16) -- the throughput
// kernel of the reverse loop, its 32-frame-tile variant for part-filled chips and the row-split small-batch kernel (spec_denoiser.py:178-184 -> diffnet.py:60-81, 110-132).
//
// Why: the layer is two GEMMs (512 x 768 and 512 x 256 per frame) and the fp32 matrix instruction of gfx950
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) is 16 x slower than its 16-bit ones (v_mfma_f32_32x32x16_{f16,bf16}, 2.5 PFLOP/s).
// An fp32 value is the sum of three bf16 values (8 + 8 + 8 mantissa bits, round-to-nearest pieces) to within 2^-24:
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)
// so a product a*b is  a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |ab|): SIX bf16 MFMAs (each product of two
// bf16 values is exact in fp32, accumulation is fp32) give every term to within one fp32 ulp -- the same size as the
// rounding of a native fp32 FMA -- at 6/16 of the fp32 MFMA time.  Measured against an fp64 reference the split GEMM is
// at least as accurate as the fp32 MFMA chain (tests/test_gpu_parity.py::test_x3_stack_*), and the parity bar of the path
// (|dmel| < 1e-4 against the reference) is met with the same margin.  It is NOT a reduced-precision path: nothing is
// rounded to bf16 that is not also carried by a lower piece; HBM tensors, biases, the conditioner projection, the gate
// and all accumulators are fp32 exactly as in csrc/diffnet.hip.
//
// Two splittings share the kernel (template parameter):
//   SplitBf16x3  a = a0 + a1 + a2 (bf16), six products, fp32 range, error <= ~2^-23 |ab| per product          (mode 3)
//   SplitF16x2   a = a0 + a1 (fp16: 11 + 11 mantissa bits; v_mfma_f32_32x32x16_f16 keeps subnormal inputs, checked on the
//                device: tools/hw/mfma_f16_denorm.hip), three products a0b0 + a0b1 + a1b0, error <= ~2^-21 |ab| per
//                product + 2^-25 absolute per operand -- still below what the fp32 accumulation itself adds over K = 768
//                (measured: same error against fp64 as the fp32 MFMA chain), at HALF the matrix-pipe work of bf16x3.
//                fp16 range: the weights are pre-scaled per layer and GEMM by a power of two (exact; undone on the
//                accumulators) so that their residual pieces stay normal, and an activation of magnitude >= 32768
//                raises the sticky error word (value 2) instead of overflowing silently.                       (mode 2)
//
// Geometry: persistent task queue over (layer, 64-frame tile) exactly as diffnet_stack_kernel (same flags, same publish
// protocol, same sync_ws layout).  Block = 512 threads = 8 waves, one block per CU; wave w owns gate rows [32w, 32w+32) and
// filter rows 256 + [32w, 32w+32) x 64 frames (4 accumulators of 32x32), then residual / skip rows of GEMM 2.
//   B operands: LDS tiles [piece][frame][256 channels] bf16, rows padded by 16 B (conflict-free 16-byte fragment reads);
//               the x tile is split once when it is staged, the gated z tile (which overlays it) once after the gate.
//   A operands: per-layer images [wave][k-step][row block][piece][lane][8 bf16] in global memory (3 MiB per layer, L2),
//               split once per weight version by set_pack_diffnet_layer_x3.
//   Per k-step (16 channels of one tap) a wave issues 6 A loads + 6 B reads of 16 bytes and 24 MFMAs (768 cycles).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

// measurement builds only (tools/build_exp.sh): bit 0 = the A fragments are loaded once per GEMM (no L2 operand stream), bit 1 =
// every k-step reads the B fragment of k-step 0 (the compiler hoists it: no LDS stream), bit 2 = no s_setprio.  WRONG RESULTS.
#ifndef SET_X3_EXP
#define SET_X3_EXP 0
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// debug: lane 0 of block 0 adds the s_memtime ticks of its phases (claim + wait, stage, GEMM 1, gate, GEMM 2, epilogue +
// publish), summed over its tasks, to buf[0..5] (+ sub-phases of the gate in buf[8..10]) and its task count to buf[7]
__device__ uint64_t *g_x3_phase_buf = nullptr;
// SET_X3_PROBE (default 0 since round 4; `tools/build_exp.sh probe diffnet_x3.hip -DSET_X3_PROBE=1` builds the library the phase
// probes need).  The stamps add to counters through a generic pointer, and one flat access inside the task loop makes the wait-count
// pass treat every outstanding load as possibly out of order (s_waitcnt vmcnt(0) at the top of every 4-k-step group of both GEMMs: the
// weight ring drains once per group).  Without them: proper vmcnt(14) / vmcnt(12) ring waits, 1.788 -> 1.763 ms per 20-layer launch at
// B = 32, T = 800 (the clock falls from 2.00 to 1.86 GHz at 1.40 kW: power-limited, profiles/r04_power.log).  The first build without
// stamps faulted at 32-frame-tile shapes: the NCB = 1 instantiation staged its 256 step offsets with all 512 threads and the upper half
// wrote past `dsh`, over the task slots behind it -- harmless only as long as the compiler kept the slot read in front of the staging,
// which the build with stamps happened to do.  Fixed in x3_main (the bound on idx); the whole parity file passes on either build.
#ifndef SET_X3_PROBE
#define SET_X3_PROBE 0
#endif

namespace {

constexpr int XC = 256;            // residual channels
constexpr int X_MAXD = 8;          // largest dilation
constexpr int XR = XC * 2 + 16;    // bytes per LDS row
constexpr int X_KS1 = 48;          // k-steps of GEMM 1 (3 taps x 16)
constexpr int X_KS2 = 16;          // k-steps of GEMM 2
constexpr float RSQRT2 = 0.70710678118654752440f;
constexpr unsigned X_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
// a = p0 + p1 + p2 exactly (up to 2^-24 |a|): the three bf16 pieces of an fp32 value
__device__ __forceinline__ void split3(float a, unsigned short &p0, unsigned short &p1, unsigned short &p2) {
    p0 = f2bf(a);
    const float r1 = a - bf2f(p0);
    p1 = f2bf(r1);
    p2 = f2bf(r1 - bf2f(p1));
}
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float h2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }

struct SplitBf16x3 {
    static constexpr int NP = 3, NPROD = 6, MODE = 3, PF = 2, PF2 = 1;  // PF / PF2: A prefetch distance in k-steps (8- / 4-wave blocks)
    // piece pairs ordered by magnitude: 2^-16, 2^-16, 2^-16, 2^-8, 2^-8, 1
    static __device__ __forceinline__ constexpr int qa(int t) { return t == 0 ? 2 : (t == 1 || t == 3) ? 1 : 0; }
    static __device__ __forceinline__ constexpr int qb(int t) { return t == 2 ? 2 : (t == 1 || t == 4) ? 1 : 0; }
    static __device__ __forceinline__ void split(float a, unsigned short (&p)[3]) { split3(a, p[0], p[1], p[2]); }
    static __device__ __forceinline__ f32x16 mma(u32x4_t a, u32x4_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
struct SplitF16x2 {
    static constexpr int NP = 2, NPROD = 3, MODE = 2, PF = 4, PF2 = 2;
    static __device__ __forceinline__ constexpr int qa(int t) { return t == 0 ? 1 : 0; }  // a1 b0, a0 b1, a0 b0
    static __device__ __forceinline__ constexpr int qb(int t) { return t == 1 ? 1 : 0; }
    static __device__ __forceinline__ void split(float a, unsigned short (&p)[2]) {
        p[0] = f2h(a);
        p[1] = f2h(a - h2f(p[0]));
    }
    // two values at once, pieces packed (a in the low half; common.h)
    static __device__ __forceinline__ void split2(float a, float b, unsigned (&p)[2]) { split2_f16(a, b, p[0], p[1]); }
    static __device__ __forceinline__ f32x16 mma(u32x4_t a, u32x4_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
// bf16/fp16 elements of one layer's image: GEMM 1 + GEMM 2 fragments, then 4 floats {s1, 1/s1, s2, 1/s2} (the power-of-two
// scales the weights of the two GEMMs were multiplied by before splitting)
template <typename S> constexpr int64_t x_n1() { return 8LL * X_KS1 * 2 * S::NP * 512; }
template <typename S> constexpr int64_t x_n2() { return 8LL * X_KS2 * 2 * S::NP * 512; }
// Winograd F(2,3) image of GEMM 1 (round 6; two-piece fp16 only) for the 16-wide matrix instruction (v_mfma_f32_16x16x32_f16;
// diffnet_stack_x3v_kernel): the four transformed weight planes as 32 k-steps of 32 channels in the order the kernel consumes them -- U1, U2, U0,
// -U3 -- [wave][k-step][row block 4: gate 0-15, gate 16-31, filter 0-15, filter 16-31][piece][lane][8], then {s1w, 1 / s1w} + padding (4 floats:
// the last 8 elements of the layer image)
constexpr int X_KSV = 32;
template <typename S> constexpr int64_t x_nv() { return S::MODE == 2 ? 8LL * X_KSV * 4 * S::NP * 512 + 8 : 0; }
template <typename S> constexpr int64_t x_nimg() { return x_n1<S>() + x_n2<S>() + 8 + x_nv<S>(); }
__device__ __forceinline__ u32x4_t buf_load_u4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ int urow(int r) { return (r & 3) + 8 * (r >> 2); }  // + 4 * (lane >> 5)
// read-once streams (the conditioner projection, the running skip sum): nt = evict-first at the L2, which the layer images live in
#ifndef SET_X3W_NT
#define SET_X3W_NT 1  // measured (round 6, profiles/r06_x3w_nt_ab.log): 1.664 against 1.688 ms per launch; 0 = plain loads
#endif
__device__ __forceinline__ float buf_load_nt(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, SET_X3W_NT ? 2 : 0));
}
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- weight images --------------------------------------------------------------------------------------------------
// lane l of a fragment holds row (l & 31), k = 8 (l >> 5) + e, e < 8:
//   w1x[w][ks][rb][p][l][e]  row = (rb ? 256 : 0) + 32 w + (l & 31), tap = ks / 16, channel = 16 (ks % 16) + k : piece p of Wdil
//   w2x[w][ks][rb][p][l][e]  channel = 16 ks + k : piece p of Wout
template <typename S>
__global__ void __launch_bounds__(256) pack_layer_x3_kernel(const float *wdil, const float *wout, unsigned short *img, float s1, float s2) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (fragment element, all pieces)
    constexpr int64_t n1 = x_n1<S>() / S::NP, n2 = x_n2<S>() / S::NP;
    constexpr int64_t nv = x_nv<S>() > 0 ? (x_nv<S>() - 8) / S::NP : 0;
    if (idx == 0) {
        float *tail = reinterpret_cast<float *>(img + x_n1<S>() + x_n2<S>());
        tail[0] = s1; tail[1] = 1.0f / s1; tail[2] = s2; tail[3] = 1.0f / s2;
        if (nv > 0) {  // |U| <= 1.5 max |w|: half the scale of the direct image keeps the top piece in the same binade range
            float *tw = reinterpret_cast<float *>(img + x_nimg<S>() - 8);
            tw[0] = 0.5f * s1; tw[1] = 1.0f / (0.5f * s1); tw[2] = 0.0f; tw[3] = 0.0f;
        }
    }
    if (idx >= n1 + n2 + nv) return;
    if (idx >= n1 + n2) {  // Winograd planes of GEMM 1 (fragments of the 16-wide matrix instruction: lane l holds row (l & 15), k = 8 (l >> 4) + e)
        // y[2p] = m0 + m1 + m2, y[2p+1] = m1 - m2 - m3 with m_j = U_j . V_j over the channels (diffnet.py:70 dilated_conv, dilation 1):
        // U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2; stored plane order: U1, U2 (the first pair of planes the kernel
        // stages), U0, -U3 (accumulates straight into the odd output).  Formed in fp64, rounded to fp32 once.
        int64_t r = idx - n1 - n2;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int mb = r & 3; r >>= 2;
        const int ks = (int)(r % X_KSV), w = (int)(r / X_KSV);
        const int row = (mb >= 2 ? XC : 0) + 32 * w + 16 * (mb & 1) + (l & 15), k = 8 * (l >> 4) + e;
        const int plane = ks >> 3, c = 32 * (ks & 7) + k;
        const float *g = wdil + ((int64_t)row * XC + c) * 3;
        const double g0 = g[0], g1 = g[1], gg2 = g[2];
        const double u = plane == 0 ? 0.5 * (g0 + g1 + gg2) : (plane == 1 ? 0.5 * (g0 - g1 + gg2) : (plane == 2 ? g0 : -gg2));
        unsigned short p[S::NP];
        S::split((0.5f * s1) * (float)u, p);
        unsigned short *base = img + x_n1<S>() + x_n2<S>() + 8 + ((((int64_t)w * X_KSV + ks) * 4 + mb) * S::NP) * 512 + l * 8 + e;
#pragma unroll
        for (int q = 0; q < S::NP; ++q) base[q * 512] = p[q];
        return;
    }
    const bool g2 = idx >= n1;
    int64_t r = g2 ? idx - n1 : idx;
    const int e = r & 7; r >>= 3;
    const int l = r & 63; r >>= 6;
    const int rb = r & 1; r >>= 1;
    const int nks = g2 ? X_KS2 : X_KS1;
    const int ks = (int)(r % nks), w = (int)(r / nks);
    const int row = (rb ? XC : 0) + 32 * w + (l & 31), k = 8 * (l >> 5) + e;
    const float v = g2 ? s2 * wout[(int64_t)row * XC + 16 * ks + k] : s1 * wdil[((int64_t)row * XC + 16 * (ks % 16) + k) * 3 + ks / 16];
    unsigned short p[S::NP];
    S::split(v, p);
    unsigned short *base = img + (g2 ? x_n1<S>() : 0) + ((((int64_t)w * nks + ks) * 2 + rb) * S::NP) * 512 + l * 8 + e;
#pragma unroll
    for (int q = 0; q < S::NP; ++q) base[q * 512] = p[q];
}

// ---- the GEMM: acc[u][rb][cb] += sum over the piece products, smallest terms first ---------------------------------------
//   NU: image row groups ("image waves" w8 = NU w + u: gate rows 32 w8 .., filter rows 256 + 32 w8 ..) this wave owns
//   A: image block of (w8, ks, rb, piece) at byte offset abase + u * ustride + ((ks * 2 + rb) * NP + piece) * 1024 (+ lane * 16)
//   B: piece q of this lane's fragment for (ks, cb) at lds + q * piece_bytes + bfrag(ks, cb)
//   PF: A prefetch distance in k-steps.  Every accumulator sees its products in the same order whatever NU is.
template <typename S, int NKS, int NU, int NCB, int PF, int SLP = 0, typename BF>
__device__ __forceinline__ void gemm_x3(f32x16 (&acc)[NU][2][NCB], rsrc_t img, unsigned lane16, unsigned abase, unsigned ustride,
                                        const unsigned char *lds, unsigned piece_bytes, BF bfrag) {
    constexpr int NP = S::NP;
    static_assert(NKS % PF == 0, "k-steps must be a multiple of the prefetch distance");
    u32x4_t A[PF][NU][2][NP];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    A[p][u][rb][q] = buf_load_u4(img, lane16, abase + (unsigned)u * ustride + (unsigned)(((p * 2 + rb) * NP + q) * 1024));
    for (int kb = 0; kb < NKS; kb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ks = kb + p;
            u32x4_t Bv[NCB][NP];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const unsigned bo = bfrag((SET_X3_EXP & 2) ? 0 : ks, cb);  // (experiment 2: no B stream)
#pragma unroll
                for (int q = 0; q < NP; ++q) Bv[cb][q] = *reinterpret_cast<const u32x4_t *>(lds + q * piece_bytes + bo);
            }
            // pinned order (round 4): B fragments | the k-step's MFMAs straight from the ring slot | the slot's refill PF k-steps ahead.
            // (Copying the slot to temporaries and refilling it BEFORE the MFMAs -- the earlier form -- made the compiler rotate the ring
            // through v_mov chains behind s_waitcnt vmcnt(0) at the end of every PF k-steps: the ring drained once per loop iteration.)
            __builtin_amdgcn_sched_barrier(0);
            if (SLP) __builtin_amdgcn_s_sleep(SLP);  // (x3v: the wave's own rate cap, see SET_X3V_SLEEP)
            if (!(SET_X3_EXP & 4)) __builtin_amdgcn_s_setprio(1);
            // the accumulators interleave, so consecutive MFMAs never depend on each other
#pragma unroll
            for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
                for (int u = 0; u < NU; ++u)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb)
                            acc[u][rb][cb] = S::mma(A[p][u][rb][S::qa(t)], Bv[cb][S::qb(t)], acc[u][rb][cb]);
            if (!(SET_X3_EXP & 4)) __builtin_amdgcn_s_setprio(0);
            const int kn = (SET_X3_EXP & 1) ? p : min(ks + PF, NKS - 1);  // tail: harmless re-load of the last k-step (experiment 1: no A stream)
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        A[p][u][rb][q] = buf_load_u4(img, lane16, abase + (unsigned)u * ustride + (unsigned)(((kn * 2 + rb) * NP + q) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// dynamic LDS of one block: NP pieces of the x tile (every column block with its own halo rows) + step offsets + task slots
template <typename S, int NCB>
constexpr size_t x3_lds_bytes(int max_dil) {
    return (size_t)S::NP * (size_t)(NCB * (32 + 2 * max_dil) * XR) + NCB * XC * sizeof(float) + 16;
}

// A tile = NCB consecutive 32-frame COLUMN BLOCKS of the batch's block list: utterance b owns blocks b nbu .. b nbu + nbu - 1
// (nbu = ceil(T / 32), the last one partial when 32 does not divide T).  Every column block carries its own utterance, its
// own first frame and its own halo rows in LDS, so a tile may straddle an utterance boundary (T = 800: 25 blocks per
// utterance, 400 tiles of 64 frames for B = 32 instead of 32 x 13 = 416) and the conv still sees zeros beyond each
// utterance's ends.
struct X3Tile {
    const float *xin;       // BATCH bases ([B][256][T])
    float *xout, *skp;
    const float *cp;        // conditioner projection of this layer, utterance 0; utterance stride cp_bs
    const float *dstep;     // step offsets of this layer, utterance 0; utterance stride d_bs, channel stride d_cs
    int64_t cp_bs, d_bs, d_cs;
    const unsigned short *img;
    const float *b_dil, *b_out;
    int32_t *err_flag;
    int T, dil, first;
    int q0, nbu, Q;         // first column block of the tile, blocks per utterance, blocks in the batch
};

// column block cb of a tile: utterance and first frame (wave-uniform); `ok` = the block exists
struct X3Col {
    int b, t0;
    bool ok;
};
__device__ __forceinline__ X3Col x3_col(const X3Tile &a, int cb) {
    const int q = a.q0 + cb, qc = min(q, a.Q - 1);
    X3Col c;
    c.b = qc / a.nbu;
    c.t0 = (qc - c.b * a.nbu) * 32;
    c.ok = q < a.Q;
    return c;
}

// pack NP pieces of 8 consecutive channels (p[e][q]: element e, piece q) into one 16-byte word per piece
template <int NP>
__device__ __forceinline__ void pack8(const unsigned short (&p)[8][NP], u32x4_t (&u)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) u[q][e] = (unsigned)p[2 * e][q] | ((unsigned)p[2 * e + 1][q] << 16);
}

// One (layer, 64-frame tile) task in two parts, so that the persistent kernel can order them around its dependency wait:
//   x3_init  accumulators of GEMM 1 = b_dil + conditioner projection (fp32, hoisted out of the loop by the caller).  These
//            loads come from HBM (the projection is streamed once per step) and depend on nothing the previous layer
//            wrote: they are issued BEFORE the previous task's store drain and the wait for the producer tiles.
//   x3_main  stage x, GEMM 1, gate, GEMM 2, epilogue (stores only; the caller drains and publishes)
// lds = NP pieces of (64 + 2 max_dil) rows + 256 floats
//
// NU = image row groups per wave.  NU = 1: 8 waves per block, one block per CU (two waves of ONE block share each SIMD's matrix
// pipe and are in the same phase at all times: while they stage, gate or store, the pipe idles -- ~30 % of a task).
// NU = 2: 4 waves per block, TWO blocks per CU, each wave owns 64 gate + 64 filter rows (8 accumulators): the two waves on a
// SIMD belong to different blocks, i.e. different (layer, tile) tasks at different points of their life, so one block's
// staging / gate / epilogue / dependency wait runs under the other block's GEMMs.  Same images, same products in the same
// order per accumulator: bit-identical results.
template <int NU, int NCB>
__device__ __forceinline__ void x3_init(const X3Tile &a, f32x16 (&acc)[NU][2][NCB]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    const unsigned T4 = 4u * (unsigned)T;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const X3Col c = x3_col(a, cb);
        const rsrc_t rcp = make_rsrc(a.cp + (int64_t)c.b * a.cp_bs);
        const unsigned vo4 = 4u * (unsigned)(4 * half * T + min(c.t0 + l31, T - 1));
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const float *bd = a.b_dil + (rb ? XC : 0) + 32 * (NU * w + u);  // wave-uniform: scalar loads, no vector-memory slots
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned ur = (unsigned)((rb ? XC : 0) + 32 * (NU * w + u) + urow(r));
                    const float blo = bd[urow(r)], bhi = bd[urow(r) + 4];
                    acc[u][rb][cb][r] = (half ? bhi : blo) + buf_load_nt(rcp, vo4, ur * T4);
                }
            }
    }
}

template <typename S, int NU, int NCB>
__device__ __forceinline__ void x3_main(const X3Tile &a, f32x16 (&acc)[NU][2][NCB], unsigned char *lds, unsigned piece_bytes, uint64_t *dbg,
                                        uint64_t &tprev) {
#define X3_PHASE(p)                                           \
    if (dbg) {                                                \
        const uint64_t tn = __builtin_amdgcn_s_memtime();     \
        dbg[p] += tn - tprev;                                 \
        tprev = tn;                                           \
    }
    X3_PHASE(0)
    constexpr int NP = S::NP;
    constexpr int NW = 8 / NU, NT = 64 * NW;  // waves / threads per block
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int T = a.T, d = a.dil;
    const int RB = 32 + 2 * d;  // LDS rows of one column block of the x tile: row jj <-> frame t0 - d + jj
    float *dsh = reinterpret_cast<float *>(lds + NP * piece_bytes);  // [NCB][256] step offsets
    const unsigned T4 = 4u * (unsigned)T;
    const rsrc_t rw = make_rsrc(a.img);
    const unsigned lane16 = 16u * (unsigned)lane;
    auto row0 = [&](int u, int rb) { return (rb ? XC : 0) + 32 * (NU * w + u); };
    // power-of-two scales of the two weight images (1 for bf16x3): {s1, 1/s1, s2, 1/s2}
    const float *sc = reinterpret_cast<const float *>(a.img + x_n1<S>() + x_n2<S>());
    const float s1 = sc[0], is1 = sc[1], s2 = sc[2], is2 = sc[3];
    bool tv[NCB];
    unsigned vo4[NCB];
    int64_t ub[NCB];  // element offset of the column block's utterance in the [B][256][T] tensors
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const X3Col c = x3_col(a, cb);
        const int t = c.t0 + l31;
        tv[cb] = c.ok && t < T;
        vo4[cb] = 4u * (unsigned)(4 * half * T + min(t, T - 1));
        ub[cb] = (int64_t)c.b * XC * T;
    }

    // ---- stage x + d, split into its pieces.  Wave w stages column block w % NCB (one utterance per wave: a wave-uniform
    //      buffer descriptor), frame row fr = lane & 31 (+ the halo rows 32 .. 32 + 2d - 1 on the lanes fr < 2d), channel
    //      group cg of CHG channels in passes of CPT (<= 32: the loads of a pass are all in flight before the first is consumed).
    {
        constexpr int NCG = (NW / NCB) * 2, CHG = XC / NCG, CPT = CHG < 32 ? CHG : 32, NPASS = CHG / CPT;
        const int cbs = w % NCB, fr = lane & 31, cg = (w / NCB) * 2 + (lane >> 5);
        const X3Col c = x3_col(a, cbs);
        const rsrc_t rx = make_rsrc(a.xin + (int64_t)c.b * XC * T);
#pragma unroll
        for (int i0 = 0; i0 < XC * NCB; i0 += NT) {
            const int idx = i0 + tid;
            // NCB = 1 has more threads (512) than step offsets (256): without the bound the upper half of the block wrote past dsh, over
            // the task slots behind it (found in round 4 when the build without phase stamps faulted at 32-frame-tile shapes)
            if (NT <= XC * NCB || idx < XC * NCB) {
                const X3Col cd = x3_col(a, idx >> 8);
                dsh[idx] = a.dstep[(int64_t)cd.b * a.d_bs + (int64_t)(idx & 255) * a.d_cs];
            }
        }
        float amax = 0.0f;
        auto put = [&](int jj, int ch0, const float (&v)[CPT], bool valid) {
#pragma unroll
            for (int q8 = 0; q8 < CPT / 8; ++q8) {  // 8 channels -> one 16-byte write per piece
                unsigned short p[8][NP];
                // step offsets d[c] of these 8 channels, read unconditionally
                const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dsh + XC * cbs + ch0 + 8 * q8);
                const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dsh + XC * cbs + ch0 + 8 * q8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 8 * q8 + e;
                    const float xs = v[ch] + (e < 4 ? d0[e & 3] : d1[e & 3]);  // select after: `valid ? load + .. : 0` branches per element
                    const float xv = valid ? xs : 0.0f;
                    if constexpr (S::MODE == 2) amax = fmaxf(amax, fabsf(xv));
                    S::split(xv, p[e]);
                }
                u32x4_t u[NP];
                pack8<NP>(p, u);
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    *reinterpret_cast<u32x4_t *>(lds + q * piece_bytes + (cbs * RB + jj) * XR + (ch0 + 8 * q8) * 2) = u[q];
            }
        };
        const int t = c.t0 - d + fr, th = c.t0 - d + 32 + fr;
        const bool tvx = c.ok && t >= 0 && t < T, has_h = fr < 2 * d, tvh = c.ok && th >= 0 && th < T;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int ch0 = CHG * cg + CPT * ps;
            const unsigned cgo = (unsigned)ch0 * T4;  // the channel group's rows (per lane: a wave spans two groups)
            const unsigned vox = 4u * (unsigned)min(max(t, 0), T - 1) + cgo, voh = 4u * (unsigned)min(max(th, 0), T - 1) + cgo;
            float vx[CPT], vh[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) vx[k] = buf_load(rx, vox, (unsigned)k * T4);
            if (has_h) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) vh[k] = buf_load(rx, voh, (unsigned)k * T4);
            }
            if (ps == 0) __syncthreads();  // dsh
            put(fr, ch0, vx, tvx);
            if (has_h) put(32 + fr, ch0, vh, tvh);
        }
        if constexpr (S::MODE == 2) {  // fp16 pieces: |x| >= 32768 (or NaN) is outside the range of the splitting -> say so
            if (!(amax < 32768.0f) && a.err_flag) __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // NU = 2: the accumulator start is fetched only now -- 128 more live registers during the staging pass would spill, and
    // the round trip (HBM: the projection is streamed once per step) runs under the other block's GEMMs
    if constexpr (NU != 1) x3_init<NU, NCB>(a, acc);
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][rb][cb][r] *= s1;
    __syncthreads();
    X3_PHASE(1)

    // ---- GEMM 1: y = Wdil (*) (x + d); k-step ks -> tap ks / 16 (a row shift of tap * d), channels 16 (ks % 16) ..
    gemm_x3<S, X_KS1, NU, NCB, (NU == 1 ? S::PF : S::PF2), 0>(
        acc, rw, lane16, (unsigned)(NU * w * X_KS1 * 2 * NP * 1024), (unsigned)(X_KS1 * 2 * NP * 1024), lds, piece_bytes, [&](int ks, int cb) {
            return (unsigned)((cb * RB + l31 + (ks >> 4) * d) * XR + ((ks & 15) * 16 + half * 8) * 2);
        });
    X3_PHASE(2)

    // ---- residual rows of x for GEMM 2's accumulator start: issued here, consumed after the gate
    // (NU = 2: after the gate -- no room for them next to 8 live accumulators and the gate's temporaries)
    float xres[NU][NCB][16];
    auto load_xres = [&]() {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const rsrc_t rx = make_rsrc(a.xin + ub[cb]);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) xres[u][cb][r] = buf_load(rx, vo4[cb], (unsigned)(row0(u, 0) + urow(r)) * T4);
        }
    };
    if constexpr (NU == 1) load_xres();
    __syncthreads();  // every wave is done reading the x tile: the z tile overlays it (row cb * 32 + l31 <-> that block's frame)
    X3_PHASE(8)
    // ---- gate (lane-local: acc[u][0] gate rows, acc[u][1] the matching filter rows), split z, 4 consecutive channels per write
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned short p[4][NP];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float zz = fsig(acc[u][0][cb][r] * is1) * ftanh(acc[u][1][cb][r] * is1);
                    const float z = tv[cb] ? zz : 0.0f;
                    S::split(z, p[e]);
                }
                const unsigned off = (unsigned)((cb * 32 + l31) * XR + (32 * (NU * w + u) + 8 * g + 4 * half) * 2);
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    u32x2_t uu;
                    uu[0] = (unsigned)p[0][q] | ((unsigned)p[1][q] << 16);
                    uu[1] = (unsigned)p[2][q] | ((unsigned)p[3][q] << 16);
                    *reinterpret_cast<u32x2_t *>(lds + q * piece_bytes + off) = uu;
                }
            }
    X3_PHASE(9)
    if constexpr (NU != 1) load_xres();
    // ---- GEMM 2 accumulators: residual rows start at s2 (b_out + x), skip rows at s2 b_out (the running skip sum is added
    //      in the epilogue, after the x' stores are in flight)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const float *bo = a.b_out + row0(u, rb);  // wave-uniform: scalar loads
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float blo = bo[urow(r)], bhi = bo[urow(r) + 4];
                const float bias = half ? bhi : blo;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[u][rb][cb][r] = (rb == 0 ? bias + xres[u][cb][r] : bias) * s2;
            }
        }
    // the running skip sum of this tile (written by this block's predecessor on the tile, one layer ago).  NU = 1: fetched
    // under GEMM 2.  NU = 2: there is no room for 64 more live registers next to 8 accumulators and the operand ring; the
    // loads follow the x' stores and their round trip is covered by the other block on the CU.
    float sk[NU][NCB][16];
    auto load_sk = [&]() {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const rsrc_t rsk = make_rsrc(a.skp + ub[cb]);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) sk[u][cb][r] = buf_load_nt(rsk, vo4[cb], (unsigned)(32 * (NU * w + u) + urow(r)) * T4);
        }
    };
    if constexpr (NU == 1) load_sk();
    X3_PHASE(10)
    __syncthreads();
    X3_PHASE(3)

    // ---- GEMM 2: o = Wout z
    gemm_x3<S, X_KS2, NU, NCB, (NU == 1 ? S::PF : S::PF2), 0>(
        acc, rw, lane16, (unsigned)(x_n1<S>() * 2 + NU * w * X_KS2 * 2 * NP * 1024), (unsigned)(X_KS2 * 2 * NP * 1024), lds, piece_bytes,
        [&](int ks, int cb) { return (unsigned)((cb * 32 + l31) * XR + (ks * 16 + half * 8) * 2); });
    X3_PHASE(4)

    // ---- epilogue: x' (agent-scope write-through: other XCDs read it right after the publish), then the skip sum
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (tv[cb]) {
            const rsrc_t rxo = make_rsrc(a.xout + ub[cb]);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_agent((acc[u][0][cb][r] * is2) * RSQRT2, rxo, vo4[cb], (unsigned)(row0(u, 0) + urow(r)) * T4);
        }
    }
    const bool first = a.first != 0;
    if constexpr (NU != 1) {
        if (!first) load_sk();
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (tv[cb]) {
            const rsrc_t rsk = make_rsrc(a.skp + ub[cb]);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store_agent(first ? acc[u][1][cb][r] * is2 : acc[u][1][cb][r] * is2 + sk[u][cb][r], rsk, vo4[cb],
                                    (unsigned)(32 * (NU * w + u) + urow(r)) * T4);
        }
    }
}
#undef X3_PHASE

// persistent (layer, tile) queue: the protocol of diffnet_stack_kernel (csrc/diffnet.hip)
// NCB: 32-frame column blocks per tile (2: 64-frame tiles, the throughput shape; 1: 32-frame tiles for batches that leave
// most CUs without a 64-frame tile -- the time of a layer is then the time of one task)
template <typename S, int NU, int NCB>
__global__ void __launch_bounds__(512 / NU, NU) diffnet_stack_x3_kernel(SetDiffnetStackArgs a, int tiles_per_utt, int ntiles, int ntasks,
                                                                        unsigned piece_bytes, int fault_tile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int *s_task = reinterpret_cast<int *>(lds + S::NP * piece_bytes + NCB * XC * sizeof(float));  // [0] next task, [1] peek result, [2] wait result
    int *counter = a.sync_ws, *abort_flag = a.sync_ws + 1, *done = a.sync_ws + 4;
    const int tid = threadIdx.x;
    uint64_t *dbg = (SET_X3_PROBE && blockIdx.x == 0 && tid == 0) ? g_x3_phase_buf : nullptr;
    uint64_t tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;
    // Each block claims its NEXT task while the current one runs (the result is read at the end of the task), issues the
    // next task's producer-independent loads before it drains the stores of the finished tile, and publishes that tile
    // before it waits for its own producers.  Claiming ahead is deadlock-free: a block finishes its claims in claim order,
    // and a claim only ever waits on earlier ones (same argument as diffnet_stack_wino_kernel).
    if (tid == 0) s_task[0] = atomicAdd(counter, 1);
    __syncthreads();
    int n = __builtin_amdgcn_readfirstlane(s_task[0]);
    int i_done = -1, l_done = 0;  // finished but not yet published tile of this block
    while (n < ntasks) {
        const int l = n / ntiles, i = n - l * ntiles, j = i;  // one chain of tiles over the batch: neighbours are i - 1 / i + 1
        X3Tile lt;
        lt.xin = (l & 1) ? a.xb : a.xa;
        lt.xout = (l & 1) ? a.xa : a.xb;
        lt.skp = a.skip;
        lt.cp = a.condproj + (int64_t)l * a.cp_ls; lt.cp_bs = a.cp_bs;
        lt.dstep = a.dstep + (int64_t)l * a.d_ls; lt.d_bs = a.d_bs; lt.d_cs = a.d_cs;
        lt.img = reinterpret_cast<const unsigned short *>(a.wx3_all) + (int64_t)l * x_nimg<S>();
        lt.b_dil = a.b_dil_all + (int64_t)l * 512;
        lt.b_out = a.b_out_all + (int64_t)l * 512;
        lt.err_flag = a.err_flag;
        lt.T = a.T; lt.dil = 1 << (l % a.dilation_cycle_length); lt.first = (l == 0);
        lt.nbu = (a.T + 31) / 32; lt.Q = a.B * lt.nbu; lt.q0 = i * NCB;
        f32x16 acc[NU][2][NCB];
        if constexpr (NU == 1) x3_init<NU, NCB>(lt, acc);  // issued before the previous tile's store drain / publish and before the dependency wait
        __builtin_amdgcn_sched_barrier(0);
        // Lane 0 PEEKS at the three producer flags and claims the next task; both round trips overlap the store drain
        // below.  Only a peek: this block's finished tile is not published yet, and a blocking wait here could wait on a
        // tile that (transitively) waits on ours.  If the producers are not there yet, the blocking wait follows the publish.
        const int *f0 = done + i, *fl = done + (j > 0 ? i - 1 : i), *fr = done + (j < tiles_per_utt - 1 ? i + 1 : i);
        int peek = l, claimed = 0;
        if (tid == 0) {
            if (l > 0) peek = min(ld_agent(f0), min(ld_agent(fl), ld_agent(fr)));
            claimed = atomicAdd(counter, 1);
        }
        if (dbg) { const uint64_t tn = __builtin_amdgcn_s_memtime(); dbg[11] += tn - tprev; tprev = tn; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: the finished tile is visible to every XCD
        if (dbg) { const uint64_t tn = __builtin_amdgcn_s_memtime(); dbg[12] += tn - tprev; tprev = tn; }
        if (tid == 0) {
            if (peek >= l) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_task[0] = claimed;
            s_task[1] = peek >= l ? 1 : 2;
        }
        __syncthreads();  // (also: the LDS tile is free)
        if (tid == 0 && i_done >= 0 && !(l_done == 0 && i_done == fault_tile))
            __hip_atomic_store(done + i_done, l_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i_done = -1;
        if (__builtin_amdgcn_readfirstlane(s_task[1]) == 2) {  // producers not finished at the peek: wait for them now
            if (tid == 0) {
                int ok = 1;
                unsigned spins = 0;
                for (;;) {
                    const int v0 = ld_agent(f0), v1 = ld_agent(fl), v2 = ld_agent(fr);
                    if (min(v0, min(v1, v2)) >= l) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > X_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
                        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.err_flag) __hip_atomic_store(a.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                s_task[2] = ok;
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(s_task[2]) == 0) break;
        }
        const int n_next = __builtin_amdgcn_readfirstlane(s_task[0]);
        x3_main<S, NU, NCB>(lt, acc, lds, piece_bytes, dbg, tprev);
        i_done = i;
        l_done = l;
        n = n_next;
        if (dbg) {
            const uint64_t tn = __builtin_amdgcn_s_memtime();
            dbg[5] += tn - tprev;
            dbg[7] += 1;
            tprev = tn;
        }
    }
    // the last finished tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && i_done >= 0 && !(l_done == 0 && i_done == fault_tile))
        __hip_atomic_store(done + i_done, l_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename S, int NU, int NCB>
int launch_x3(const SetDiffnetStackArgs &a, int n_cu, int fault_tile, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_stack_x3_kernel<S, NU, NCB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "set_diffnet_stack(x3 attr)");
        attr_set = true;
    }
    const int Q = a.B * ((a.T + 31) / 32);                    // 32-frame column blocks of the batch
    const int ntiles = (Q + NCB - 1) / NCB, tiles_per_utt = ntiles;  // tiles are cut from the batch's block list (see X3Tile)
    const int64_t ntasks64 = (int64_t)ntiles * a.L;
    SET_REQUIRE(ntasks64 < (1ll << 30), "set_diffnet_stack(task count)");
    SET_REQUIRE((int64_t)2 * XC * a.T * 4 < ((int64_t)1 << 31), "set_diffnet_stack(split-operand kernel: T too large)");
    const int max_dil = 1 << (a.dilation_cycle_length - 1);
    const unsigned piece_bytes = (unsigned)(NCB * (32 + 2 * max_dil) * XR);  // every column block has its own halo rows
    const size_t ldsz = x3_lds_bytes<S, NCB>(max_dil);
    SET_HIP(set_zero_async(a.sync_ws, (size_t)(4 + ntiles) * sizeof(int32_t), s), "set_diffnet_stack(memset)");
    // workers stay below the runnable-task count (a tile's layers form a chain: at most `ntiles` tasks are ever runnable;
    // see diffnet.hip).  NU = 2: two blocks fit a CU, and the ones that do not get a partner still run a task at a time.
    int grid = NU * n_cu;
    if (grid > ntiles * 4 / 5) grid = ntiles * 4 / 5;
    if (const char *e = getenv("SET_AMD_STACK_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
    if ((int64_t)grid > ntasks64) grid = (int)ntasks64;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((diffnet_stack_x3_kernel<S, NU, NCB>), dim3(grid), dim3(512 / NU), ldsz, s, a, tiles_per_utt, ntiles, (int)ntasks64,
                       piece_bytes, fault_tile);
    return set_check_launch("set_diffnet_stack");
}

// =====================================================================================================================
// Winograd F(2,3) for GEMM 1 on the two-piece fp16 operands (round 6).  The kernel is limited by the package power, not by time
// (profiles/r04_power.log, r06_loop_launch_ab.log), and the fp16 MFMA work is the largest part of its energy -- so the first lever is FEWER
// matrix instructions: the k = 3 conv over an output PAIR (frames 2p, 2p + 1) is four channel GEMMs instead of six,
//     V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3          (d0 .. d3 = x + step offset at frames 2p - 1 .. 2p + 2; fp32, then split)
//     y[2p] = U0 V0 + U1 V1 + U2 V2,   y[2p + 1] = U1 V1 - U2 V2 - U3 V3   (U: pack_layer_x3_kernel)
// i.e. 2/3 of GEMM 1's MFMAs, 3/4 of the layer's.  Dilation 1 and even T only (a pair never straddles an utterance end).
// Same accumulation precision (fp32) and piece products; the sums are formed in a different order than the direct form: results agree
// with it to fp32 rounding (like the fp32-pipe Winograd kernel of csrc/diffnet.hip), not bit for bit.
// History: the first form (diffnet_stack_x3w_kernel, commits 4f9a309 .. 1cd6a4e) ran 64-frame tiles on the 32-wide instruction with three
// accumulator sets (even, odd, one temporary) and the whole four-plane V tile in LDS: 1.61 - 1.70 ms per 20-layer launch against the direct
// form's 1.69 - 1.75, bound by its weight-fragment stream (every fragment met ONE 32-pair column block: 2 MiB per task at ~50 B/clk/CU,
// profiles/r06_x3w_no_a_stream_bound.log).  The kernel below replaced it on every shape (same box: B = 32 1.43 against 1.61 ms on 96-frame
// tiles; 64-frame tiles B = 24 / 16 / 12: +5.3 / +3.7 / +3.1 % frames/s, profiles/r06_x3v_ab.log, r06_x3v_nb2_ab.log).
// =====================================================================================================================
typedef unsigned u32x2w_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// (measured and not kept: the epilogue's write-through stores as nt stores too -- 1.719 against 1.659 ms per launch, profiles/r06_x3w_nt_ab.log)
__device__ __forceinline__ f32x2 buf_load2_nt(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, SET_X3W_NT ? 2 : 0));
}
// Measured on the first form and not kept (round 6): touching the NEXT task's conditioner-projection rows under GEMM 2 (1.669 against 1.650 ms,
// profiles/r06_x3w_prefetch_ab.log); XCD-aware task claiming -- the blocks of one XCD (blockIdx.x % 8) taking their tasks from a chunk of
// 16 / 32 / 64 consecutive tiles of ONE layer so that the 32 CUs sharing an L2 stream the same layer image: static chunk ownership 1.718 against
// 1.672 ms (groups drift apart and wait for each other's tiles), chunks handed out dynamically 2.69 - 3.08 ms at 1,050 - 1,180 W and the maximum
// clock -- the CUs of an XCD then run in lockstep and ask the L2 for the SAME weight fragments at the same moment
// (profiles/r06_x3w_xcd_chunks_ab.log).  The one global counter, which spreads the CUs of an XCD over the phases of a task, is the better schedule.

// =====================================================================================================================
// 96-frame tiles on the 16-wide matrix instruction (round 6, second half).  What bounds the 64-frame Winograd form is the weight-fragment
// stream of GEMM 1: 2 MiB per task from the L2 at ~50 B/clk/CU (profiles/r06_x3w_no_a_stream_bound.log), and the only lever on it is the
// number of columns every fragment meets.  128-frame tiles leave 56 CUs without a tile chain at B = 32, T = 800 (200 chains) and need
// 192 accumulator registers; 96 frames = 48 output pairs are 1.5 column blocks of the 32-wide instruction -- but exactly THREE of the 16-wide
// one.  Operand-stream models of the three GEMM loops under the power cap (tools/hw/mfma_ceiling_w.hip, profiles/r06_ceiling_w.log): shipped
// 64-frame loop 1,139 TFLOP/s executed, 128-frame loop 1,373, 96-frame loop on v_mfma_f32_16x16x32_f16 1,434 (the bare 16-wide instruction
// itself sustains 2,140 against the 32-wide one's 1,875 at the same package power).
//   tile   = 3 column blocks of 32 frames (16 pairs each; every block with its own utterance, first frame and halo frames): 267 chains
//   wave   = 64 rows (gate rows 32 w .. + 31, filter rows 256 + 32 w ..) x 48 pairs = 4 x 3 accumulator blocks of 16 x 16
//   k-step = 32 channels of one plane: 8 weight fragments (1 KiB each) + 6 LDS fragment reads -> 36 MFMAs (64-frame form: 8 + 4 -> 12 of twice the size)
//   planes two at a time (LDS: [plane 2][piece 2][pair 48][XRV] = 102 KB; the z tile [piece][96][XR] overlays it):
//            P = (e + o) / 2 + U1 (d1 + d2),  Q = (e - o) / 2 + U2 (d2 - d1)      (e / o = bias + conditioner projection at the even / odd frame)
//            E = P + Q + U0 (d0 - d2),        O = P - Q + (-U3) (d1 - d3)
//          -- two accumulator sets (96 registers) instead of the 64-frame form's three; the second pair of planes is staged (x re-read
//          from the L2, neighbours by DPP row shifts, block ends by halo loads) after the first pair's GEMMs.
//   GEMM 2 = the 32-wide loop of the other forms on three column blocks (gemm_x3<.., NCB = 3>), same image.
// Measured and not kept: 128-frame tiles (NB = 4; 15 spilled registers) at B = 64 / 48, where every CU has a chain of them: +0.2 / -0.3 %
// (profiles/r06_x3v_nb4_ab.log) -- beyond 96 frames the weight-fragment stream is no longer what the time follows; GEMM 2 on the 16-wide
// instruction in its transposed form (lane = one output row at four consecutive frames): -19 % with 8-byte, -3 % with 16-byte epilogue accesses
// (half-line accesses; profiles/r06_x3v_gemm2_16wide_ab.log); skip rows fetched before GEMM 2 / a 4-deep GEMM 2 ring: +-0.3 %.
// Per-task timeline (tools/x3_timeline_probe.py, profiles/r06_x3v_timeline.log; 68 us per task at B = 32, T = 800): the two waves of a SIMD do not share
// the matrix pipe evenly -- the older one (waves 0-3) finishes every GEMM pass ~4 us before the younger one and runs ~10 us ahead at the task boundary,
// so its epilogue and next accumulator-start loads run under the younger waves' GEMM 2 (17 us against 9.4 us of MFMA issue: the younger wave alone is
// latency-bound on its fragment ring while the CU's memory pipe moves ~490 KB of stores and read-once loads).  Kept from that reading: step offsets
// staged in front of the dependency wait, tiles published per wave.  Measured and not kept: a per-wave rate cap (s_sleep in front of every k-step's
// MFMA burst: both waves finish together, GEMM 2 10 us -- and the epilogue then takes the 7 us it had been hidden for; 1.47-1.50 against 1.42-1.43 ms,
// profiles/r06_x3v_dsh_sleep_ab.log); bias + conditioner projection added in the gate from chunks fetched one ahead, accumulators starting at zero
// (the task boundary shrinks from 5.8 to 1.7 us, but the gate grows by 1.7 us of exposed HBM latency, every GEMM pass by ~1 us at 254 registers and
// the dependency wait by 4 us: 1.57 against 1.41 ms, profiles/r06_x3v_late_cp_ab.log); GEMM 1's ring slots refilled row block by row block inside the
// k-step's MFMA burst (up to 27 MFMAs earlier; bit-identical, no difference: profiles/r06_x3v_early_refill_ab.log).
// Results: same piece products and fp32 accumulation; sums in another order than the 64-frame form (equal to fp32 rounding).
// =====================================================================================================================
// NB = column blocks per tile: 3 (96 frames; shapes with a tile chain for every CU) or 2 (64 frames)
// Row stride of the V tile: 512 + 32 bytes.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ..:
// MI355X_MICROARCH.md, LDS), bank = (address / 4) mod 64; a 16-wide fragment read puts lane (l15, kg) at l15 * stride + 16 kg, so a group mixes
// rows {0-3, 12-15} of one k group with rows 4-11 of the next: with the 528-byte rows of the other tiles (16-byte quads 1 apart per row) two of
// its lanes share a bank (PMC: SQ_LDS_BANK_CONFLICT 41 % of SQ_LDS_IDX_ACTIVE, profiles/r06_pmc_x3_lds.log); with quads 2 apart per row none do.
// The z tile keeps 528-byte rows -- GEMM 2 reads it with the 32-wide layout, for which odd quad distances are the conflict-free ones.
#ifndef SET_X3V_XRV
#define SET_X3V_XRV (XC * 2 + 32)
#endif
constexpr int XRV = SET_X3V_XRV;
template <int NB> constexpr unsigned xv_piece() { return 16 * NB * XRV; }     // one piece of one plane of the V tile (16 NB pair rows)
template <int NB> constexpr unsigned xv_plane() { return 2 * xv_piece<NB>(); }
template <int NB> constexpr unsigned xv_zpiece() { return 32 * NB * XR; }     // one piece of the z tile (32 NB frame rows), which overlays the V tile
template <int NB> constexpr unsigned xv_tile() { return 2 * xv_plane<NB>() > 2 * xv_zpiece<NB>() ? 2 * xv_plane<NB>() : 2 * xv_zpiece<NB>(); }  // NB = 3: 104,448 bytes
#ifndef SET_X3V_PF
#define SET_X3V_PF 2                              // fragment ring depth of GEMM 1 in k-steps of 32 (8 fragments = 32 registers each)
#endif
#ifndef SET_X3V_PF2
#define SET_X3V_PF2 2                             // fragment ring depth of GEMM 2 in k-steps of 16
#endif
#ifndef SET_X3V_WAVE_PUBLISH
#define SET_X3V_WAVE_PUBLISH 1                    // 1 = every wave publishes its part of a finished tile (flag = waves done, 8 per layer); 0 = one store per block behind a barrier
#endif
#define X3V_FLAG_UNIT (SET_X3V_WAVE_PUBLISH ? 8 : 1)
#ifndef X3V_LEAD
#define X3V_LEAD 0                                // the thread that claims the next task and looks at the dependency flags.  Lane 0 of wave 7 (448) -- one of the
#endif                                            // waves that reach the task boundary last, so its look is the freshest (spins in 51 of 256 tasks instead of 242)
                                                  // -- waits for its claim behind its own 48 accumulator-start loads: +2.3 us per task (profiles/r06_x3v_e_ab.log)
#ifndef SET_X3V_GATE2
#define SET_X3V_GATE2 1                           // gate as one quotient (3 transcendentals per value) instead of sigmoid x tanh (4)
#endif
#ifndef SET_X3V_SLEEP
#define SET_X3V_SLEEP 0                           // s_sleep argument (x 64 clocks) in front of every k-step's MFMA burst of GEMM 1 (36 MFMAs = 576 clocks)
#endif
#ifndef SET_X3V_SLEEP2
#define SET_X3V_SLEEP2 0                          // same for GEMM 2 (18 MFMAs of 32 clocks)
#endif
#ifndef SET_X3V_SK_EARLY
#define SET_X3V_SK_EARLY 0                        // 1 = the running skip rows are fetched before GEMM 2 (48 more live registers through it)
#endif

__device__ __forceinline__ f32x4 mma16(u32x4_t a, u32x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// lane i <- lane i - 1 / i + 1 of its 16-lane row (DPP row_shr:1 / row_shl:1; the row ends get `old`)
__device__ __forceinline__ float row_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}

// accumulator start: PQ[0] = s1w (e + o) / 2, PQ[1] = s1w (e - o) / 2; lane (l15, kg): pair l15 of column block nb, rows 4 kg .. 4 kg + 3 of block mb
template <int NB>
__device__ __forceinline__ void x3v_init(const X3Tile &a, f32x4 (&PQ)[2][4][NB]) {
    typedef SplitF16x2 S;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int T = a.T;
    const unsigned T4 = 4u * (unsigned)T;
    const float hs = 0.5f * reinterpret_cast<const float *>(a.img + x_nimg<S>() - 8)[0];
    f32x4 bias[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) bias[mb] = *reinterpret_cast<const f32x4 *>(a.b_dil + (mb >= 2 ? XC : 0) + 32 * w + 16 * (mb & 1) + 4 * kg);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const X3Col c = x3_col(a, nb);
        const rsrc_t rcp = make_rsrc(a.cp + (int64_t)c.b * a.cp_bs);
        const unsigned vo = 4u * (unsigned)(4 * kg * T + min(c.t0 + 2 * l15, T - 2));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const unsigned row0 = (unsigned)((mb >= 2 ? XC : 0) + 32 * w + 16 * (mb & 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 v = buf_load2_nt(rcp, vo, (row0 + (unsigned)i) * T4);
                const float e = bias[mb][i] + v[0], o = bias[mb][i] + v[1];
                PQ[0][mb][nb][i] = hs * (e + o);
                PQ[1][mb][nb][i] = hs * (e - o);
            }
        }
    }
}

// one pair of planes of the V tile: PH = 0: d1 + d2 | d2 - d1 (U1, U2);  PH = 1: d0 - d2 | d1 - d3 (U0, -U3).  lane (l15, cs): pair l15 of every
// column block, 8 channels 32 w + 8 cs ..; returns the largest magnitude staged (range check of the fp16 pieces)
// (two halves, so that the second pass's loads can be issued in front of the barrier that retires the first pair of planes)
template <int NB, int PH>
__device__ __forceinline__ void x3v_stage_load(const X3Tile &a, int w, int lane, f32x2 (&x12)[NB][8], float (&xh)[NB][8]) {
    const int l15 = lane & 15, cs = lane >> 4;
    const int ch0 = 32 * w + 8 * cs;
    const int T = a.T;
    const unsigned T4 = 4u * (unsigned)T;
    const bool edge = l15 == 0 || l15 == 15;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const X3Col c = x3_col(a, nb);
        const rsrc_t rx = make_rsrc(a.xin + (int64_t)c.b * XC * T);
        const int t = c.t0 + 2 * l15;  // frames t, t + 1 (T even: both inside or both outside)
        const unsigned vox = 4u * (unsigned)min(t, T - 2) + (unsigned)ch0 * T4;
#pragma unroll
        for (int k = 0; k < 8; ++k) x12[nb][k] = buf_load2(rx, vox, (unsigned)k * T4);
        if (PH == 1) {
            const int th = l15 == 0 ? c.t0 - 1 : c.t0 + 32;  // halo frame of the block's end lanes
            const unsigned voh = 4u * (unsigned)min(max(th, 0), T - 1) + (unsigned)ch0 * T4;
            if (edge) {
#pragma unroll
                for (int k = 0; k < 8; ++k) xh[nb][k] = buf_load(rx, voh, (unsigned)k * T4);
            }
        }
    }
}
template <int NB, int PH>
__device__ __forceinline__ float x3v_stage_store(const X3Tile &a, unsigned char *lds, const float *dsh, int w, int lane, const f32x2 (&x12)[NB][8],
                                                 const float (&xh)[NB][8]) {
    typedef SplitF16x2 S;
    const int l15 = lane & 15, cs = lane >> 4;
    const int ch0 = 32 * w + 8 * cs;
    const int T = a.T;
    const bool edge = l15 == 0 || l15 == 15;
    float amax = 0.0f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const X3Col c = x3_col(a, nb);
        const int t = c.t0 + 2 * l15, th = l15 == 0 ? c.t0 - 1 : c.t0 + 32;
        const bool v12 = c.ok && t < T, vh = c.ok && edge && th >= 0 && th < T;
        const f32x4 dA = *reinterpret_cast<const f32x4 *>(dsh + nb * XC + ch0);
        const f32x4 dB = *reinterpret_cast<const f32x4 *>(dsh + nb * XC + ch0 + 4);
        u32x4_t u[2][2];  // [plane of the pair][piece]: 8 channels, two to a word
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            float V[2][2];  // [plane][channel of the word]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 2 * e2 + h;
                const float ds = e < 4 ? dA[e & 3] : dB[e & 3];
                const float d1 = v12 ? x12[nb][e][0] + ds : 0.0f, d2 = v12 ? x12[nb][e][1] + ds : 0.0f;
                if (PH == 0) {
                    V[0][h] = d1 + d2;
                    V[1][h] = d2 - d1;
                } else {
                    float d0 = row_prev(d2), d3 = row_next(d1);
                    const float hv = vh ? xh[nb][e] + ds : 0.0f;
                    d0 = l15 == 0 ? hv : d0;
                    d3 = l15 == 15 ? hv : d3;
                    V[0][h] = d0 - d2;
                    V[1][h] = d1 - d3;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                amax = fmaxf(fmaxf(amax, fabsf(V[j][0])), fabsf(V[j][1]));
                unsigned pw[2];
                S::split2(V[j][0], V[j][1], pw);
                u[j][0][e2] = pw[0];
                u[j][1][e2] = pw[1];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4_t *>(lds + j * xv_plane<NB>() + q * xv_piece<NB>() + (16 * nb + l15) * XRV + ch0 * 2) = u[j][q];
    }
    return amax;
}

// 8 k-steps (one plane) of GEMM 1 into acc; the fragment ring runs through the 16 k-steps of a plane PAIR (k-steps 16 pair .. 16 pair + 15)
template <int NB, int PFV>
__device__ __forceinline__ void x3v_plane(f32x4 (&acc)[4][NB], u32x4_t (&A)[PFV][4][2], rsrc_t img, unsigned lane16, unsigned abase, int seg,
                                          const unsigned char *bplane, unsigned boff) {
    typedef SplitF16x2 S;
    const int ks_last = 16 * (seg >> 1) + 15;
    for (int kb = 0; kb < 8; kb += PFV) {
#pragma unroll
        for (int p = 0; p < PFV; ++p) {
            const int kc = kb + p, ks = 8 * seg + kc;
            u32x4_t Bv[NB][2];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 2; ++q) Bv[nb][q] = *reinterpret_cast<const u32x4_t *>(bplane + q * xv_piece<NB>() + boff + (unsigned)(nb * 16 * XRV) + (unsigned)kc * 64u);
            __builtin_amdgcn_sched_barrier(0);
            if (SET_X3V_SLEEP) __builtin_amdgcn_s_sleep(SET_X3V_SLEEP);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int t = 0; t < S::NPROD; ++t)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mma16(A[p][mb][S::qa(t)], Bv[nb][S::qb(t)], acc[mb][nb]);
            __builtin_amdgcn_s_setprio(0);
            const int kn = min(ks + PFV, ks_last);  // tail: harmless re-load of the pair's last k-step
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int q = 0; q < 2; ++q) A[p][mb][q] = buf_load_u4(img, lane16, abase + (unsigned)(((kn * 4 + mb) * 2 + q) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// (SET_X3_PROBE builds: lane 0 of block 0 adds its s_memtime ticks per phase to g_x3_phase_buf -- 0 claim + accumulator start + wait, 1 first
// plane pair staged, 2 its GEMMs, 3 E / O + second pair staged, 4 its GEMMs, 8 residual loads + gate, 9 GEMM 2, 10 epilogue issue, 5 publish)
template <int NB>
__device__ __forceinline__ void x3v_main(const X3Tile &a, f32x4 (&PQ)[2][4][NB], unsigned char *lds, uint64_t *dbg, uint64_t &tprev, uint64_t (&ts)[32]) {
#if SET_X3_PROBE == 2  // timeline build: every wave keeps the s_memtime stamps of its task in SGPRs (ts[8 + p]), no memory access here
#define X3V_PHASE(p)                                  \
    {                                                 \
        __builtin_amdgcn_sched_barrier(0);            \
        ts[8 + (p)] = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_sched_barrier(0);            \
    }
#else
#define X3V_PHASE(p)                                          \
    if (dbg) {                                                \
        const uint64_t tn = __builtin_amdgcn_s_memtime();     \
        dbg[p] += tn - tprev;                                 \
        tprev = tn;                                           \
    }
#endif
#if SET_X3_PROBE == 2
#define X3V_TS(p) X3V_PHASE(p)  // stamps of the timeline build only
#else
#define X3V_TS(p)
#endif
    X3V_PHASE(0)
    typedef SplitF16x2 S;
    constexpr int PFV = SET_X3V_PF;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;    // 16-wide fragments (GEMM 1, gate)
    const int half = lane >> 5, l31 = lane & 31;  // 32-wide fragments (GEMM 2, epilogue)
    const int T = a.T;
    const unsigned T4 = 4u * (unsigned)T;
    const unsigned lane16 = 16u * (unsigned)lane;
    float *dsh = reinterpret_cast<float *>(lds + xv_tile<NB>());  // [NB][256] step offsets of the column blocks' utterances
    const rsrc_t rw = make_rsrc(a.img);
    const float *sc = reinterpret_cast<const float *>(a.img + x_n1<S>() + x_n2<S>());
    const float s2 = sc[2], is2 = sc[3];
    const float is1w = reinterpret_cast<const float *>(a.img + x_nimg<S>() - 8)[1];
    const float cg2 = is1w * -1.4426950408889634f, cf2 = is1w * 2.8853900817779268f;  // (is1w: a power of two)
    (void)cg2; (void)cf2;
    // (dsh: staged by the kernel's task loop in front of the dependency wait -- the offsets depend on nothing the layer below wrote)
    // ---- GEMM 1, first pair of planes
    const unsigned abase = (unsigned)((x_n1<S>() + x_n2<S>() + 8) * 2) + (unsigned)(w * X_KSV * 4 * 2 * 1024);
    const unsigned boff = (unsigned)(l15 * XRV + kg * 16);
    u32x4_t A[PFV][4][2];
    f32x2 x12[NB][8];
    float xh[NB][8];
    x3v_stage_load<NB, 0>(a, w, lane, x12, xh);
    float amax = x3v_stage_store<NB, 0>(a, lds, dsh, w, lane, x12, xh);
#pragma unroll
    for (int p = 0; p < PFV; ++p)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 2; ++q) A[p][mb][q] = buf_load_u4(rw, lane16, abase + (unsigned)(((p * 4 + mb) * 2 + q) * 1024));
    X3V_TS(12)
    __syncthreads();
    X3V_PHASE(1)
    x3v_plane<NB, PFV>(PQ[0], A, rw, lane16, abase, 0, lds, boff);
    x3v_plane<NB, PFV>(PQ[1], A, rw, lane16, abase, 1, lds + xv_plane<NB>(), boff);
    X3V_PHASE(2)
    x3v_stage_load<NB, 1>(a, w, lane, x12, xh);  // second pair's x: in flight through the E / O combine and the barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float pp = PQ[0][mb][nb][i], qq = PQ[1][mb][nb][i];
                PQ[0][mb][nb][i] = pp + qq;  // even frame of the pair
                PQ[1][mb][nb][i] = pp - qq;  // odd frame
            }
    __syncthreads();  // every wave is done reading the first pair of planes
    X3V_TS(13)
    // ---- second pair
    amax = fmaxf(amax, x3v_stage_store<NB, 1>(a, lds, dsh, w, lane, x12, xh));
    if (!(amax < 32768.0f) && a.err_flag) __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int p = 0; p < PFV; ++p)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int q = 0; q < 2; ++q) A[p][mb][q] = buf_load_u4(rw, lane16, abase + (unsigned)((((16 + p) * 4 + mb) * 2 + q) * 1024));
    X3V_TS(14)
    __syncthreads();
    X3V_PHASE(3)
    x3v_plane<NB, PFV>(PQ[0], A, rw, lane16, abase, 2, lds, boff);
    x3v_plane<NB, PFV>(PQ[1], A, rw, lane16, abase, 3, lds + xv_plane<NB>(), boff);
    X3V_PHASE(4)
    // ---- residual rows of x for GEMM 2's accumulator start (issued here, consumed after the gate)
    bool tv[NB];
    unsigned vo4[NB];
    int64_t ub[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const X3Col c = x3_col(a, cb);
        const int t = c.t0 + l31;
        tv[cb] = c.ok && t < T;
        vo4[cb] = 4u * (unsigned)(4 * half * T + min(t, T - 1));
        ub[cb] = (int64_t)c.b * XC * T;
    }
    float xres[NB][16];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const rsrc_t rx = make_rsrc(a.xin + ub[cb]);
#pragma unroll
        for (int r = 0; r < 16; ++r) xres[cb][r] = buf_load(rx, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
    }
    __syncthreads();  // every wave is done reading the V tile: the z tile overlays it (row = frame of the tile)
    X3V_TS(15)
    // ---- gate: lane (l15, kg) holds the pair's frames 2 l15 (PQ[0]) and 2 l15 + 1 (PQ[1]) of column block nb, channels 32 w + 16 m + 4 kg ..
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const X3Col c = x3_col(a, nb);
        const bool tvp = c.ok && c.t0 + 2 * l15 < T;
#pragma unroll
        for (int eo = 0; eo < 2; ++eo)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float zz[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#if SET_X3V_GATE2
                    // sigmoid(a) tanh(b) = (v - 1) / ((1 + u) (1 + v)), u = e^-a, v = e^2b: three transcendentals instead of four.  v is capped at 2^64
                    // (tanh = 1 to fp32 from 2^25 on; inf - 1 times 1 / inf would be NaN); u may overflow: the quotient then is the limit, 0.
                    const float u = __builtin_amdgcn_exp2f(PQ[eo][m][nb][i] * cg2);
                    const float v = __builtin_amdgcn_exp2f(fminf(PQ[eo][m + 2][nb][i] * cf2, 64.0f));
                    const float g = (v - 1.0f) * __builtin_amdgcn_rcpf((1.0f + u) * (1.0f + v));
#else
                    const float g = fsig(PQ[eo][m][nb][i] * is1w) * ftanh(PQ[eo][m + 2][nb][i] * is1w);
#endif
                    zz[i] = tvp ? g : 0.0f;
                }
                unsigned plo[2], phi[2];
                S::split2(zz[0], zz[1], plo);
                S::split2(zz[2], zz[3], phi);
                const unsigned off = (unsigned)((32 * nb + 2 * l15 + eo) * XR + (32 * w + 16 * m + 4 * kg) * 2);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x2w_t uu;
                    uu[0] = plo[q];
                    uu[1] = phi[q];
                    *reinterpret_cast<u32x2w_t *>(lds + q * xv_zpiece<NB>() + off) = uu;
                }
            }
    }
    // ---- GEMM 2 accumulators: residual rows start at s2 (b_out + x), skip rows at s2 b_out
    f32x16 acc[1][2][NB];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float *bo = a.b_out + (rb ? XC : 0) + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float blo = bo[urow(r)], bhi = bo[urow(r) + 4];
            const float bias = half ? bhi : blo;
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) acc[0][rb][cb][r] = (rb == 0 ? bias + xres[cb][r] : bias) * s2;
        }
    }
#if SET_X3V_SK_EARLY
    float sk[NB][16];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        const rsrc_t rsk = make_rsrc(a.skp + ub[cb]);
#pragma unroll
        for (int r = 0; r < 16; ++r) sk[cb][r] = buf_load_nt(rsk, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
    }
#endif
    X3V_TS(16)
    __syncthreads();
    X3V_PHASE(8)
    gemm_x3<S, X_KS2, 1, NB, SET_X3V_PF2, SET_X3V_SLEEP2>(acc, rw, lane16, (unsigned)(x_n1<S>() * 2 + w * X_KS2 * 2 * 2 * 1024), (unsigned)(X_KS2 * 2 * 2 * 1024), lds,
                                             xv_zpiece<NB>(), [&](int ks, int cb) { return (unsigned)((cb * 32 + l31) * XR + (ks * 16 + half * 8) * 2); });
    X3V_PHASE(9)
    // ---- epilogue
    const bool first = a.first != 0;
#if !SET_X3V_SK_EARLY
    float sk[NB][16];
    if (!first) {
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const rsrc_t rsk = make_rsrc(a.skp + ub[cb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) sk[cb][r] = buf_load_nt(rsk, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
        }
    }
#endif
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        if (tv[cb]) {
            const rsrc_t rxo = make_rsrc(a.xout + ub[cb]);
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_store_agent((acc[0][0][cb][r] * is2) * RSQRT2, rxo, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
        }
    }
    X3V_TS(17)
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
        if (tv[cb]) {
            const rsrc_t rsk = make_rsrc(a.skp + ub[cb]);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store_agent(first ? acc[0][1][cb][r] * is2 : acc[0][1][cb][r] * is2 + sk[cb][r], rsk, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
        }
    }
    X3V_PHASE(10)
}
#undef X3V_PHASE
#undef X3V_TS

// the persistent (layer, tile) queue of diffnet_stack_x3_kernel (same flags, same publish protocol) on tiles of NB column blocks
#if SET_X3_PROBE == 2
#ifndef X3V_TL_TASK
#define X3V_TL_TASK 10
#endif
#define X3V_KTS(p)                                \
    {                                             \
        __builtin_amdgcn_sched_barrier(0);        \
        ts[p] = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_sched_barrier(0);        \
    }
#else
#define X3V_KTS(p)
#endif
template <int NB>
__global__ void __launch_bounds__(512, 1) diffnet_stack_x3v_kernel(SetDiffnetStackArgs a, int ntiles, int ntasks, int fault_tile) {
    typedef SplitF16x2 S;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int *s_task = reinterpret_cast<int *>(lds + xv_tile<NB>() + NB * XC * sizeof(float));  // [0] next task, [1] peek result, [2] wait result
    int *counter = a.sync_ws, *abort_flag = a.sync_ws + 1, *done = a.sync_ws + 4;
    const int tid = threadIdx.x;
    uint64_t *dbg = (SET_X3_PROBE == 1 && blockIdx.x == 0 && tid == 0) ? g_x3_phase_buf : nullptr;
    uint64_t tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;
    if (tid == 0) s_task[0] = atomicAdd(counter, 1);
    __syncthreads();
    int n = __builtin_amdgcn_readfirstlane(s_task[0]);
    int i_done = -1, l_done = 0;
    uint64_t ts[32] = {};
    int n_mine = 0;
    (void)n_mine;
    while (n < ntasks) {
        const int l = n / ntiles, i = n - l * ntiles;
        X3Tile lt;
        lt.xin = (l & 1) ? a.xb : a.xa;
        lt.xout = (l & 1) ? a.xa : a.xb;
        lt.skp = a.skip;
        lt.cp = a.condproj + (int64_t)l * a.cp_ls; lt.cp_bs = a.cp_bs;
        lt.dstep = a.dstep + (int64_t)l * a.d_ls; lt.d_bs = a.d_bs; lt.d_cs = a.d_cs;
        lt.img = reinterpret_cast<const unsigned short *>(a.wx3_all) + (int64_t)l * x_nimg<S>();
        lt.b_dil = a.b_dil_all + (int64_t)l * 512;
        lt.b_out = a.b_out_all + (int64_t)l * 512;
        lt.err_flag = a.err_flag;
        lt.T = a.T; lt.dil = 1; lt.first = (l == 0);
        lt.nbu = (a.T + 31) / 32; lt.Q = a.B * lt.nbu; lt.q0 = i * NB;  // (x3v)
        f32x4 PQ[2][4][NB];
        X3V_KTS(0)
#if SET_X3V_WAVE_PUBLISH
        // the finished tile is published by every wave on its own, as soon as ITS stores are complete (agent-scope write-through stores: complete =
        // visible to every XCD): the tile's flag counts waves, 8 per layer.  (The block-wide form -- drain, barrier, one store -- published ~5 us
        // after the last wave's stores were issued, behind the next task's accumulator-start loads; profiles/r06_x3v_timeline.log.)
        if (i_done >= 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((tid & 63) == 0 && !(l_done == 0 && i_done == fault_tile))
                (void)__hip_atomic_fetch_add(done + i_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            i_done = -1;
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        // step offsets of the column blocks' utterances: loaded first, stored to LDS behind the wait below, published by the task-slot barrier
        // (the previous task read its offsets for the last time in its second staging pass, several barriers ago)
        float dsv[(NB * XC + 511) / 512];
#pragma unroll
        for (int j = 0; j < (NB * XC + 511) / 512; ++j) {
            const int idx = min(tid + 512 * j, NB * XC - 1);
            dsv[j] = lt.dstep[(int64_t)x3_col(lt, idx >> 8).b * lt.d_bs + (int64_t)(idx & (XC - 1)) * lt.d_cs];
        }
        x3v_init<NB>(lt, PQ);
        __builtin_amdgcn_sched_barrier(0);
        const int *f0 = done + i, *fl = done + (i > 0 ? i - 1 : i), *fr = done + (i < ntiles - 1 ? i + 1 : i);
        int peek = l, claimed = 0;
        if (tid == X3V_LEAD) {
            if (l > 0) peek = min(ld_agent(f0), min(ld_agent(fl), ld_agent(fr))) / X3V_FLAG_UNIT;
            claimed = atomicAdd(counter, 1);
        }
        X3V_KTS(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        X3V_KTS(2)
        {
            float *dsh = reinterpret_cast<float *>(lds + xv_tile<NB>());
#pragma unroll
            for (int j = 0; j < (NB * XC + 511) / 512; ++j)
                if (tid + 512 * j < NB * XC) dsh[tid + 512 * j] = dsv[j];
        }
        if (tid == X3V_LEAD) {
            if (peek >= l) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_task[0] = claimed;
            s_task[1] = peek >= l ? 1 : 2;
        }
        __syncthreads();
        X3V_KTS(3)
#if !SET_X3V_WAVE_PUBLISH
        if (tid == 0 && i_done >= 0 && !(l_done == 0 && i_done == fault_tile))
            __hip_atomic_store(done + i_done, l_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i_done = -1;
#endif
        if (__builtin_amdgcn_readfirstlane(s_task[1]) == 2) {
            if (tid == X3V_LEAD) {
                int ok = 1;
                unsigned spins = 0;
                for (;;) {
                    const int v0 = ld_agent(f0), v1 = ld_agent(fl), v2 = ld_agent(fr);
                    if (min(v0, min(v1, v2)) >= l * X3V_FLAG_UNIT) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > X_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
                        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.err_flag) __hip_atomic_store(a.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                s_task[2] = ok;
            }
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(s_task[2]) == 0) break;
        }
        const int n_next = __builtin_amdgcn_readfirstlane(s_task[0]);
        X3V_KTS(4)
        x3v_main<NB>(lt, PQ, lds, dbg, tprev, ts);
#if SET_X3_PROBE == 2
        // timeline of this block's X3V_TL_TASK-th task, waves 0 and 7: row (2 block + wave / 7) of 32 dwords = ticks since the task's first stamp
        // (slot 5: the task number, 6: 1 if the dependency wait had to spin)
        if (++n_mine == X3V_TL_TASK && (tid == 0 || tid == 448) && g_x3_phase_buf) {
            const rsrc_t rt = make_rsrc(g_x3_phase_buf);
            const unsigned row = (unsigned)(2 * blockIdx.x + (tid ? 1 : 0)) * 128u;
#pragma unroll
            for (int k = 0; k < 32; ++k) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(ts[k] - ts[0]), rt, (int)(row + 4 * k), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)n, rt, (int)(row + 20), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)s_task[1], rt, (int)(row + 24), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)ts[0], rt, (int)(row + 28), 0, 0);
        } else if (n_mine == X3V_TL_TASK + 1 && (tid == 0 || tid == 448) && g_x3_phase_buf) {  // slot 31: the NEXT task's first stamp (absolute, as slot 7)
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)ts[0], make_rsrc(g_x3_phase_buf), (int)((unsigned)(2 * blockIdx.x + (tid ? 1 : 0)) * 128u + 124u), 0, 0);
        }
#endif
        i_done = i;
        l_done = l;
        n = n_next;
        if (dbg) {
            const uint64_t tn = __builtin_amdgcn_s_memtime();
            dbg[5] += tn - tprev;
            dbg[7] += 1;
            tprev = tn;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SET_X3V_WAVE_PUBLISH
    if ((tid & 63) == 0 && i_done >= 0 && !(l_done == 0 && i_done == fault_tile))
        (void)__hip_atomic_fetch_add(done + i_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    __syncthreads();
    if (tid == 0 && i_done >= 0 && !(l_done == 0 && i_done == fault_tile))
        __hip_atomic_store(done + i_done, l_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

template <int NB>
int launch_x3v(const SetDiffnetStackArgs &a, int n_cu, int fault_tile, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_stack_x3v_kernel<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                "set_diffnet_stack(x3v attr)");
        attr_set = true;
    }
    const int Q = a.B * ((a.T + 31) / 32);
    const int ntiles = (Q + NB - 1) / NB;
    const int64_t ntasks64 = (int64_t)ntiles * a.L;
    SET_REQUIRE(ntasks64 < (1ll << 30), "set_diffnet_stack(task count)");
    SET_REQUIRE((int64_t)2 * XC * a.T * 4 < ((int64_t)1 << 31), "set_diffnet_stack(split-operand kernel: T too large)");
    SET_REQUIRE(a.dilation_cycle_length == 1 && a.T % 2 == 0, "set_diffnet_stack(x3v: dilation 1 and even T only)");
    const size_t ldsz = (size_t)xv_tile<NB>() + NB * XC * sizeof(float) + 16;
    SET_HIP(set_zero_async(a.sync_ws, (size_t)(4 + ntiles) * sizeof(int32_t), s), "set_diffnet_stack(memset)");
    int grid = n_cu;
    if (NB == 2 && grid > ntiles * 4 / 5) grid = ntiles * 4 / 5;  // (the 64-frame rule of the earlier forms: workers beyond 0.8 tile chains mostly wait)
    if (const char *e = getenv("SET_AMD_STACK_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
    if ((int64_t)grid > ntasks64) grid = (int)ntasks64;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(diffnet_stack_x3v_kernel<NB>, dim3(grid), dim3(512), ldsz, s, a, ntiles, (int)ntasks64, fault_tile);
    return set_check_launch("set_diffnet_stack");
}

// Measured and not kept (round 6; the code is in commit 9e25393 "... whole-loop kernel", profiles/r06_loop_launch_ab.log): the WHOLE reverse loop
// as one launch -- the step boundary (skip projection -> output head -> posterior -> next step's input projection) as one more task type of
// this queue, flags counting over all 100 steps.  Bit-identical to the per-step launches, and 2.3 % SLOWER at B = 32, T = 800 (179.2 against
// 175.2 ms per 100 steps; 1,364 W / 1.95 GHz against 1,351 W / 1.99 GHz): the launch gaps and the drain of every stack launch are not lost time
// under the package power limit -- the clock recovers in them.

// =====================================================================================================================
// Small batches: the row-split scheme of diffnet_stack_split_kernel (csrc/diffnet.hip: every 32-frame tile computed by FOUR
// co-operating blocks, one 32-row block per wave, two counter rendezvous per layer) on the two-piece fp16 operands.  With a
// handful of utterances the time of a layer is a chain of memory round trips plus ONE accumulator's MFMA chain per wave: on
// the fp32 pipe that chain is 512 dependent MFMAs of 64 cycles (21 us of the 34 us per layer), here 192 of 32 cycles.
//   block (tile i, part h), wave j: image block w8 = 2h + (j & 1), row block rb = j >> 1 (0: gate rows 32 w8 .., then
//   residual rows; 1: the matching filter rows, then skip rows) -- the images of the throughput kernel, as they are.
//   z crosses between the parts already split: z_ws slot of a tile = [piece][frame 32][256 channels] fp16 (32 KiB), written
//   with 8-byte agent-scope stores of 4 channels, copied to LDS with 16-byte loads.
// Same products in the same order per accumulator as diffnet_stack_x3_kernel<SplitF16x2>: bit-identical to it.
// =====================================================================================================================
constexpr unsigned SX_SPIN_LIMIT = 1u << 20;
constexpr int SX_PF = 8;  // A prefetch distance in k-steps (2 x 16 bytes each).  The images are cold in L2 at every layer (42 MB
                          // cycle through 4 MB per XCD), so a memory-side round trip per SX_PF k-steps bounds the GEMMs.  The kernel
                          // only runs with one block per CU (launch bounds (256, 1): the ring may spill into AGPRs): 4 -> 8 k-steps
                          // 66.3 -> 64.4 ms per 100 steps at B = 1 (GEMM 1 11.2 -> 10.4 us, GEMM 2 3.6 -> 2.8 us per layer);
                          // 16 k-steps: GEMM 1 8.8 / GEMM 2 1.1 us but the x staging doubles (AGPR traffic), 80.6 ms.  An
                          // LDS-fragment ring and an explicit L2 prefetch of the next layer's slices were not faster either.

// agent-scope (sc1) loads: coherent with the sc1 write-through stores of other XCDs without an acquire fence in front of them
__device__ __forceinline__ float buf_load_sc1(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 16));
}
__device__ __forceinline__ u32x4_t buf_load_u4_sc1(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 16);
}

// fence = false: the caller reads everything other blocks produced with agent-scope loads (buf_load_sc1): no `buffer_inv sc1`,
// which would also throw this XCD's copy of the weight images out of the L2 -- twice per layer
__device__ __forceinline__ bool sx_wait(const int *f0, const int *f1, const int *f2, int want, int *abort_flag, int *err_flag, bool fence = true) {
    unsigned spins = 0;
    for (;;) {
        const int v0 = ld_agent(f0), v1 = ld_agent(f1), v2 = ld_agent(f2);
        if (min(v0, min(v1, v2)) >= want) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SX_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (err_flag) __hip_atomic_store(err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    if (fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// acc += A B over NKS k-steps, ONE accumulator: A fragments of (k-step, piece) at abase + (ks * 4 + piece) * 1024 (the image's
// [ks][rb][piece] order with this wave's rb folded into abase), ring of SX_PF k-steps (preloaded by the caller into A);
// B piece q of k-step ks at lds + q * piece_bytes + bfrag(ks)
template <int NKS, typename BF>
__device__ __forceinline__ void sx_gemm(f32x16 &acc, u32x4_t (&A)[SX_PF][2], rsrc_t img, unsigned lane16, unsigned abase,
                                        const unsigned char *lds, unsigned piece_bytes, BF bfrag) {
    // The schedule is pinned per k-step (round 4, as in gemm_x3):  ds_read B(k + 1) | the three MFMAs of k-step k straight from their
    // ring slot | buffer_load refill of that slot | sched_barrier.  Rounds 2-3 had `a = A[p]; A[p] = load; mma(a)` in a fully unrolled
    // loop: the compiler rotated the ring through v_movs and waited vmcnt(0) / vmcnt(1) in front of EVERY k-step (70 + 69 of them in
    // the ISA) -- the ring of 8 k-steps never held more than one, which is the "GEMM 1 takes 9.4 us for 2.2 us of MFMAs" that deeper
    // rings, L2 warmers and fence-free waits could not move.  Same products in the same order: bit-identical.
    unsigned bo = bfrag(0);
    u32x4_t b0 = *reinterpret_cast<const u32x4_t *>(lds + bo);
    u32x4_t b1 = *reinterpret_cast<const u32x4_t *>(lds + piece_bytes + bo);
#pragma unroll 1
    for (int kb = 0; kb < NKS; kb += SX_PF) {
#pragma unroll
        for (int p = 0; p < SX_PF; ++p) {
            const int ks = kb + p;
            const unsigned bn = bfrag(min(ks + 1, NKS - 1));
            const u32x4_t nb0 = *reinterpret_cast<const u32x4_t *>(lds + bn);
            const u32x4_t nb1 = *reinterpret_cast<const u32x4_t *>(lds + piece_bytes + bn);
            acc = SplitF16x2::mma(A[p][1], b0, acc);  // the order of SplitF16x2::qa / qb
            acc = SplitF16x2::mma(A[p][0], b1, acc);
            acc = SplitF16x2::mma(A[p][0], b0, acc);
            const int kn = min(ks + SX_PF, NKS - 1);
            A[p][0] = buf_load_u4(img, lane16, abase + (unsigned)(kn * 4 * 1024));
            A[p][1] = buf_load_u4(img, lane16, abase + (unsigned)((kn * 4 + 1) * 1024));
            b0 = nb0;
            b1 = nb1;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ void sx_preload(u32x4_t (&A)[SX_PF][2], rsrc_t img, unsigned lane16, unsigned abase) {
#pragma unroll
    for (int p = 0; p < SX_PF; ++p) {
        A[p][0] = buf_load_u4(img, lane16, abase + (unsigned)(p * 4 * 1024));
        A[p][1] = buf_load_u4(img, lane16, abase + (unsigned)((p * 4 + 1) * 1024));
    }
}

__global__ void __launch_bounds__(256, 1) diffnet_stack_split_x2_kernel(SetDiffnetStackArgs a, int tiles_per_utt, int ntiles,
                                                                         unsigned piece_bytes, int fault_tile, int nofence) {
    typedef SplitF16x2 S;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // x tile pieces [2][32 + 2 maxd][XR]; z tile overlays it
    float *gs = reinterpret_cast<float *>(lds + 2 * piece_bytes);       // [64][32] tanh(filter rows) of this part
    float *dsh = gs + 64 * 32;                                          // [256] step offsets
    int *s_ok = reinterpret_cast<int *>(dsh + XC);
    int *abort_flag = a.sync_ws + 1, *ready = a.sync_ws + 4, *zcnt = a.sync_ws + 4 + ntiles;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= 4 * ntiles) {
        // ---- L2 warmers (launched when CUs are left over: 8 extra blocks, one per XCD -- workgroups go to the XCDs round-robin, so
        // block 4 ntiles + q shares its L2 with the compute blocks of part (4 ntiles + q) & 3).  42 MB of images cycle through 4 MB of
        // L2 per XCD: without them every ring turn of a GEMM is a miss to memory (GEMM 1 10.4 us per layer for 2.2 us of MFMAs).  The
        // warmer reads the part's fragments of layer l (384 + 128 KB, contiguous) while the compute blocks are on layer l - 1; it is
        // paced by tile 0's progress counter and touches nothing the compute blocks write.
        const int hp = blockIdx.x & 3;
        for (int l = 1; l < a.L; ++l) {
            unsigned spins = 0;
            while (l >= 2 && ld_agent(ready) < 4 * (l - 1)) {  // tile 0 has finished layer l - 2: it is on layer l - 1 now
                __builtin_amdgcn_s_sleep(8);
                if (++spins > SX_SPIN_LIMIT || ld_agent(abort_flag) != 0) return;
            }
            const unsigned char *img = reinterpret_cast<const unsigned char *>(a.wx3_all) + (int64_t)l * x_nimg<S>() * 2;
            const rsrc_t r1 = make_rsrc(img + (size_t)(2 * hp) * X_KS1 * 4 * 1024);
            const rsrc_t r2 = make_rsrc(img + x_n1<S>() * 2 + (size_t)(2 * hp) * X_KS2 * 4 * 1024);
            for (unsigned off = 16u * tid; off < 2u * X_KS1 * 4 * 1024; off += 8u * 4096u) {
#pragma unroll
                for (unsigned q = 0; q < 8; ++q) {
                    const u32x4_t v = buf_load_u4(r1, off + q * 4096u, 0u);
                    asm volatile("" ::"v"(v));
                }
            }
            for (unsigned off = 16u * tid; off < 2u * X_KS2 * 4 * 1024; off += 8u * 4096u) {
#pragma unroll
                for (unsigned q = 0; q < 8; ++q) {
                    const u32x4_t v = buf_load_u4(r2, off + q * 4096u, 0u);
                    asm volatile("" ::"v"(v));
                }
            }
        }
        return;
    }
    const int i = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int b = i / tiles_per_utt, jt = i - b * tiles_per_utt;
    const int T = a.T, t0 = jt * 32;
    const int il = jt > 0 ? i - 1 : i, ir = jt < tiles_per_utt - 1 ? i + 1 : i;
    const unsigned T4 = 4u * (unsigned)T;
    const int w8 = 2 * h + (j & 1), rb = j >> 1;       // image block / row block of this wave
    const int ch0 = 32 * w8;                          // its gate / residual / skip channel base; GEMM rows rb * 256 + ch0 ..
    const int tq = t0 + l31;
    const bool tv = tq < T;
    const unsigned vo4 = 4u * (unsigned)(4 * half * T + min(tq, T - 1));
    const unsigned lane16 = 16u * (unsigned)lane;
    unsigned short *zt = reinterpret_cast<unsigned short *>(a.z_ws) + (int64_t)i * (2 * 32 * XC);  // [piece][frame][256]
    const rsrc_t rz = make_rsrc(zt);
    const rsrc_t rsk = make_rsrc(a.skip + (int64_t)b * XC * T);
    uint64_t *dbg = (SET_X3_PROBE && blockIdx.x == 5 && tid == 0) ? g_x3_phase_buf : nullptr;  // debug (probe build only): phase ticks of one block, see below
    uint64_t tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define SX_PHASE(p)                                           \
    if (dbg) {                                                \
        const uint64_t tn = __builtin_amdgcn_s_memtime();     \
        dbg[p] += tn - tprev;                                 \
        tprev = tn;                                           \
    }
    for (int l = 0; l < a.L; ++l) {
        const int d = 1 << (l % a.dilation_cycle_length);
        const unsigned short *img = reinterpret_cast<const unsigned short *>(a.wx3_all) + (int64_t)l * x_nimg<S>();
        const rsrc_t rw = make_rsrc(img);
        const float *sc = reinterpret_cast<const float *>(img + x_n1<S>() + x_n2<S>());
        const float s1 = sc[0], is1 = sc[1], s2 = sc[2], is2 = sc[3];
        const rsrc_t rxin = make_rsrc(((l & 1) ? a.xb : a.xa) + (int64_t)b * XC * T);
        const rsrc_t rxout = make_rsrc(((l & 1) ? a.xa : a.xb) + (int64_t)b * XC * T);
        const rsrc_t rcp = make_rsrc(a.condproj + (int64_t)l * a.cp_ls + (int64_t)b * a.cp_bs);
        const float *dstep = a.dstep + (int64_t)l * a.d_ls + (int64_t)b * a.d_bs;
        const float *bd = a.b_dil_all + (int64_t)l * 512 + rb * XC + ch0, *bo = a.b_out_all + (int64_t)l * 512 + rb * XC + ch0;
        // ---- producer-independent loads first: A ring of GEMM 1, accumulator = b_dil + conditioner projection
        u32x4_t A[SX_PF][2];
        const unsigned ab1 = (unsigned)(((w8 * X_KS1) * 2 + rb) * 2 * 1024);
        sx_preload(A, rw, lane16, ab1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float blo = bd[urow(r)], bhi = bd[urow(r) + 4];
            acc[r] = (half ? bhi : blo) + buf_load(rcp, vo4, (unsigned)(rb * XC + ch0 + urow(r)) * T4);
        }
        if (tid == 0) *s_ok = (l == 0 || sx_wait(ready + i, ready + il, ready + ir, 4 * l, abort_flag, a.err_flag, !nofence)) ? 1 : 0;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*s_ok) == 0) return;
        SX_PHASE(0)
        // ---- stage x + d as two fp16 pieces: thread (frame row f, 32 channels cg); rows 32 .. 32 + 2d - 1 by the lanes f < 2d
        {
            const int f = tid & 31, cg = tid >> 5;  // 8 channel groups
            dsh[tid] = dstep[(int64_t)tid * a.d_cs];
            float amax = 0.0f;
            auto put = [&](int row, const float (&v)[32], bool valid) {
#pragma unroll
                for (int q8 = 0; q8 < 4; ++q8) {
                    unsigned short p[8][2];
                    const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dsh + 32 * cg + 8 * q8);
                    const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dsh + 32 * cg + 8 * q8 + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xs = v[8 * q8 + e] + (e < 4 ? d0[e & 3] : d1[e & 3]);
                        const float xv = valid ? xs : 0.0f;
                        amax = fmaxf(amax, fabsf(xv));
                        S::split(xv, p[e]);
                    }
                    u32x4_t u[2];
                    pack8<2>(p, u);
                    *reinterpret_cast<u32x4_t *>(lds + row * XR + (32 * cg + 8 * q8) * 2) = u[0];
                    *reinterpret_cast<u32x4_t *>(lds + piece_bytes + row * XR + (32 * cg + 8 * q8) * 2) = u[1];
                }
            };
            const int t = t0 - d + f, th = t0 - d + 32 + f;
            const bool tvx = t >= 0 && t < T, has_h = f < 2 * d, tvh = th >= 0 && th < T;
            const unsigned vox = 4u * (unsigned)min(max(t, 0), T - 1), voh = 4u * (unsigned)min(max(th, 0), T - 1);
            float vx[32], vh[32];
            if (nofence) {  // rows written by the neighbouring tiles' blocks (other CUs, other XCDs): agent-scope loads
#pragma unroll
                for (int k = 0; k < 32; ++k) vx[k] = buf_load_sc1(rxin, vox, (unsigned)(32 * cg + k) * T4);
                if (has_h) {
#pragma unroll
                    for (int k = 0; k < 32; ++k) vh[k] = buf_load_sc1(rxin, voh, (unsigned)(32 * cg + k) * T4);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) vx[k] = buf_load(rxin, vox, (unsigned)(32 * cg + k) * T4);
                if (has_h) {
#pragma unroll
                    for (int k = 0; k < 32; ++k) vh[k] = buf_load(rxin, voh, (unsigned)(32 * cg + k) * T4);
                }
            }
            __syncthreads();  // dsh
            put(f, vx, tvx);
            if (has_h) put(32 + f, vh, tvh);
            if (!(amax < 32768.0f) && a.err_flag) __hip_atomic_store(a.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= s1;
        __syncthreads();
        SX_PHASE(1)
        // ---- GEMM 1: one 32-row block of  y = Wdil (*) (x + d)
        sx_gemm<X_KS1>(acc, A, rw, lane16, ab1, lds, piece_bytes, [&](int ks) {
            return (unsigned)((l31 + (ks >> 4) * d) * XR + ((ks & 15) * 16 + half * 8) * 2);
        });
        SX_PHASE(2)
        // ---- gate: the filter waves hand tanh(y_f) to the gate waves through LDS; z leaves already split
        if (rb == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gs[(32 * (j & 1) + urow(r) + 4 * half) * 32 + l31] = ftanh(acc[r] * is1);
        }
        __syncthreads();  // gs complete; every wave is done reading the x tile
        if (rb == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned short p[4][2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float zz = fsig(acc[r] * is1) * gs[(32 * (j & 1) + urow(r) + 4 * half) * 32 + l31];
                    S::split(tv ? zz : 0.0f, p[e]);
                }
                const unsigned off = (unsigned)(l31 * (XC * 2) + (ch0 + 8 * g + 4 * half) * 2);  // [frame][256] fp16, no padding
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    u32x2_t u;
                    u[0] = (unsigned)p[0][q] | ((unsigned)p[1][q] << 16);
                    u[1] = (unsigned)p[2][q] | ((unsigned)p[3][q] << 16);
                    __builtin_amdgcn_raw_buffer_store_b64(u, rz, (int)(off + (unsigned)q * (32u * XC * 2u)), 0, 16);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the z rows of this wave are visible to every XCD
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(zcnt + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SX_PHASE(3)
        // ---- GEMM 2: A ring + accumulator start (residual rows: b_out + x; skip rows: b_out, the running sum joins in the
        //      epilogue) -- rows this very wave wrote one layer ago; the loads fly while lane 0 waits for the other parts' z
        const unsigned ab2 = (unsigned)(x_n1<S>() * 2 + ((w8 * X_KS2) * 2 + rb) * 2 * 1024);
        sx_preload(A, rw, lane16, ab2);
        float prev[16];  // x rows (rb 0) / running skip sum (rb 1) of channels ch0 ..
        {
            const rsrc_t rp = make_rsrc(rb == 0 ? ((l & 1) ? a.xb : a.xa) + (int64_t)b * XC * T : a.skip + (int64_t)b * XC * T);
#pragma unroll
            for (int r = 0; r < 16; ++r) prev[r] = buf_load(rp, vo4, (unsigned)(ch0 + urow(r)) * T4);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float blo = bo[urow(r)], bhi = bo[urow(r) + 4];
            const float bias = half ? bhi : blo;
            acc[r] = (rb == 0 ? bias + prev[r] : bias) * s2;
        }
        if (tid == 0) *s_ok = sx_wait(zcnt + i, zcnt + i, zcnt + i, 4 * (l + 1), abort_flag, a.err_flag, !nofence) ? 1 : 0;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*s_ok) == 0) return;
        SX_PHASE(4)
        // ---- the whole z tile (two pieces, 32 frames x 512 bytes each) -> LDS rows of XR bytes
        {
            u32x4_t zv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                zv[k] = nofence ? buf_load_u4_sc1(rz, 16u * (unsigned)tid, 4096u * (unsigned)k) : buf_load_u4(rz, 16u * (unsigned)tid, 4096u * (unsigned)k);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int unit = tid + 256 * k;            // 16-byte unit of the 32 KiB slot
                const int q = unit >> 10, fr = (unit >> 5) & 31, part = unit & 31;
                *reinterpret_cast<u32x4_t *>(lds + q * piece_bytes + fr * XR + part * 16) = zv[k];
            }
        }
        __syncthreads();
        SX_PHASE(5)
        // ---- GEMM 2: one 32-row block of  o = Wout z
        sx_gemm<X_KS2>(acc, A, rw, lane16, ab2, lds, piece_bytes, [&](int ks) {
            return (unsigned)(l31 * XR + (ks * 16 + half * 8) * 2);
        });
        SX_PHASE(6)
        // ---- epilogue
        if (tv && rb == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) buf_store_agent((acc[r] * is2) * RSQRT2, rxout, vo4, (unsigned)(ch0 + urow(r)) * T4);
        } else if (tv) {
            const bool first = l == 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store(first ? acc[r] * is2 : acc[r] * is2 + prev[r], rsk, vo4, (unsigned)(ch0 + urow(r)) * T4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every store of the block has completed; the LDS tile is free for the next layer
        if (tid == 0 && !(l == 0 && i == fault_tile && h == 0))
            __hip_atomic_fetch_add(ready + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SX_PHASE(7)
    }
#undef SX_PHASE
}

}  // namespace

// called by set_diffnet_stack (csrc/diffnet.hip) for the row-split variant when two-piece fp16 images are given
int set_launch_diffnet_stack_split_x2(const SetDiffnetStackArgs &a, int fault_tile, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_stack_split_x2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "set_diffnet_stack(split x2 attr)");
        attr_set = true;
    }
    const int tiles = (a.T + 31) / 32, nt = a.B * tiles;
    const int max_dil = 1 << (a.dilation_cycle_length - 1);
    const unsigned piece_bytes = (unsigned)((32 + 2 * max_dil) * XR);
    const size_t ldsz = (size_t)2 * piece_bytes + (64 * 32 + XC) * sizeof(float) + 16;
    SET_HIP(set_zero_async(a.sync_ws, (size_t)(4 + 2 * nt) * sizeof(int32_t), s), "set_diffnet_stack(memset)");
    // 8 L2-warmer blocks (one per XCD) when the chip has CUs to spare and the block -> XCD round-robin lines them up with the parts
    static int warm = -1, n_cu = 0;
    if (warm < 0) {
        warm = 1;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 0;
    }
    const int extra = (warm && 4 * nt + 8 <= n_cu && a.L > 1) ? 8 : 0;  // (block q: XCD q % 8, part q & 3 = (q % 8) & 3 -- consistent)
    // agent-scope loads for the tiles other blocks produced instead of an acquire fence after each wait (measured in round 3 against
    // the fences: 62.7 -> 61.4 ms per 100 steps at B = 1, 66.9 -> 62.3 at B = 2 (T = 800), bit-identical either way)
    const int nofence = 1;
    hipLaunchKernelGGL(diffnet_stack_split_x2_kernel, dim3(4 * nt + extra), dim3(256), ldsz, s, a, tiles, nt, piece_bytes, fault_tile, nofence);
    return set_check_launch("set_diffnet_stack");
}

extern "C" int set_debug_x3_phase_buffer(uint64_t *buf) {
    SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_x3_phase_buf), &buf, sizeof(buf)), "set_debug_x3_phase_buffer");
    return SET_OK;
}

extern "C" int64_t set_diffnet_layer_x3_image_size(int32_t mode) {
    return mode == 2 ? x_nimg<SplitF16x2>() : (mode == 3 ? x_nimg<SplitBf16x3>() : -1);
}

extern "C" int set_pack_diffnet_layer_x3(const float *w_dil, const float *w_out, void *img, int32_t mode, int32_t k1, int32_t k2,
                                         void *stream) {
    SET_REQUIRE(w_dil && w_out && img && (mode == 2 || mode == 3), "set_pack_diffnet_layer_x3");
    SET_REQUIRE(k1 >= -60 && k1 <= 60 && k2 >= -60 && k2 <= 60, "set_pack_diffnet_layer_x3(scale exponents)");
    const float s1 = ldexpf(1.0f, k1), s2 = ldexpf(1.0f, k2);
    unsigned short *im = reinterpret_cast<unsigned short *>(img);
    if (mode == 2)
        hipLaunchKernelGGL(pack_layer_x3_kernel<SplitF16x2>, dim3(set_blocks((x_nimg<SplitF16x2>() - 16) / 2, 256)), dim3(256), 0,
                           (hipStream_t)stream, w_dil, w_out, im, s1, s2);
    else
        hipLaunchKernelGGL(pack_layer_x3_kernel<SplitBf16x3>, dim3(set_blocks((x_nimg<SplitBf16x3>() - 8) / 3, 256)), dim3(256), 0,
                           (hipStream_t)stream, w_dil, w_out, im, s1, s2);
    return set_check_launch("set_pack_diffnet_layer_x3");
}

// does set_launch_diffnet_stack_x3 take the Winograd form (diffnet_stack_x3v_kernel) for this shape?  Returns the column blocks per tile:
// 0 = direct form, 2 = 64-frame tiles, 3 = 96-frame tiles.  SET_AMD_X3_WINO=0 pins the direct form, =2 / =3 the tile width.
int set_x3_winograd_selected(int x3_mode, int B, int T, int dilation_cycle_length, int n_cu) {
    if (x3_mode != 2 || dilation_cycle_length != 1 || T % 2 != 0) return 0;
    const int64_t tiles64 = (int64_t)B * ((T + 63) / 64);
    bool narrow = 5 * tiles64 < 3 * (int64_t)n_cu;
    if (const char *e = getenv("SET_AMD_X3_TILE")) narrow = atoi(e) == 32;
    if (narrow) return 0;
    if (const char *e = getenv("SET_AMD_X3_WINO")) {
        const int v = atoi(e);
        if (v <= 0) return 0;
        if (v == 2 || v == 3) return v;
    }
    // 96-frame tiles once every CU has a tile chain of them (B = 32, T = 800: 267 chains for 256 CUs; below that the workers wait for each
    // other: B = 24 123 k frames/s on 96-frame tiles against 149 k on 64-frame ones, profiles/r06_x3v_nb2_ab.log)
    const int64_t tiles96 = ((int64_t)B * ((T + 31) / 32) + 2) / 3;
    return tiles96 >= (int64_t)n_cu ? 3 : 2;
}

// called by set_diffnet_stack (csrc/diffnet.hip) once it has picked this kernel
int set_launch_diffnet_stack_x3(const SetDiffnetStackArgs &a, int n_cu, int fault_tile, hipStream_t s) {
    SET_REQUIRE(a.x3_mode == 2 || a.x3_mode == 3, "set_diffnet_stack(x3_mode must be 2 = f16x2 or 3 = bf16x3)");
    // tile width: 64 frames from ~0.6 tiles per CU on; below that 32-frame tiles (twice the tasks, each about half as long:
    // the time of a layer is the time of one task while the chip is not full).  SET_AMD_X3_TILE=32|64 overrides.
    const int64_t tiles64 = (int64_t)a.B * ((a.T + 63) / 64);
    bool narrow = 5 * tiles64 < 3 * (int64_t)n_cu;
    if (const char *e = getenv("SET_AMD_X3_TILE")) narrow = atoi(e) == 32;
    // block shape: one 8-wave block per CU.  (Round 3 also shipped two 4-wave blocks per CU behind SET_AMD_X3_WAVES=4 -- bit-identical,
    // never faster: B = 32 1.78 ms per 20 layers (8 waves) vs 1.94 - 2.38 ms, B = 64 3.65 vs 3.69 ms with the clock dropping from 1.88
    // to 1.63 GHz; profiles/r03_x3_pair_probe.log -- but its 64-frame instantiation spilled registers, which confounded the
    // comparison; round 4 measured the power limit directly instead (profiles/r04_power.log, r04_mfma_ceiling.log) and removed the
    // variant from the library.  The NU template parameter of the kernel stays for tools/build_exp.sh experiments.)
    // round 6: Winograd F(2,3) form of GEMM 1 on 64- / 96-frame tiles (dilation 1, even T; SET_AMD_X3_WINO=0 keeps the direct form) -- see
    // diffnet_stack_x3v_kernel (its 8-byte loads of frame pairs need 8-byte aligned tensors and even strides; anything else takes the direct form)
    const bool al8 = ((reinterpret_cast<uintptr_t>(a.condproj) | reinterpret_cast<uintptr_t>(a.xa) | reinterpret_cast<uintptr_t>(a.xb)) & 7) == 0 &&
                     ((a.cp_bs | a.cp_ls) & 1) == 0;
    const int wino = al8 ? set_x3_winograd_selected(a.x3_mode, a.B, a.T, a.dilation_cycle_length, n_cu) : 0;
    if (wino == 3) return launch_x3v<3>(a, n_cu, fault_tile, s);
    if (wino == 2) return launch_x3v<2>(a, n_cu, fault_tile, s);
    if (a.x3_mode == 2)
        return narrow ? launch_x3<SplitF16x2, 1, 1>(a, n_cu, fault_tile, s) : launch_x3<SplitF16x2, 1, 2>(a, n_cu, fault_tile, s);
    return narrow ? launch_x3<SplitBf16x3, 1, 1>(a, n_cu, fault_tile, s) : launch_x3<SplitBf16x3, 1, 2>(a, n_cu, fault_tile, s);
}
