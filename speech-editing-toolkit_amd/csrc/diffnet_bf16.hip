// Fused DiffNet residual layer for the TRAINING path with bf16 MFMA operands (BASELINE configs[1]).
//
//   forward  (diffnet.py:60-81):  y = Wdil (*) (x + d) + Wcond cond + b ;  z = sigmoid(y_g) tanh(y_f) ;
//                                 o = Wout z + b ;  x' = (x + o_res) / sqrt2 ;  skip += o_skip
//   backward (same lines, transposed):  d_o = [dx'/sqrt2 ; dskip] ;  dz = Wout^T d_o ;  dy = gate'(y) dz ;
//                                 dx = dx'/sqrt2 + Wdil^T (*) dy ;  dcond += Wcond^T dy ;  dd = sum_t Wdil^T (*) dy
//
// One launch per layer and direction.  The per-op path moves ~0.7 GB (forward) + ~1.5 GB (backward) through HBM per
// layer at B=32, T=800 (conditioner projection, pre-gate, gate, output projection and residual tensors each make a round
// trip); fused, a layer reads x, cond (forward) / dx', dskip, y (backward) once and writes what the next kernel needs:
//   forward   reads x 26 MB + cond 20 MB + skip 26 MB, writes x' 26 MB + skip 26 MB + y (bf16) 26 MB + z (bf16) 13 MB
//   backward  reads dx' 26 + dskip 26 + y 26 + dcond 20, writes dx 26 + dy (bf16) 26 + d_o (bf16) 26 + dcond 20
// y / z / dy / d_o are stored in bf16 [B][C][T]: they are MFMA operands of later GEMMs (weight gradients) or inputs of
// the gate derivative, i.e. exactly the tensors torch.autocast keeps in bf16 in the reference's AMP path
// (utils/commons/trainer.py:325).  x, skip, cond and their gradients stay fp32 in HBM, accumulation is fp32.
//
// Geometry (both kernels): block = 512 threads = 8 waves, tile = 128 frames x all rows; wave w owns 32 gate rows
// [32w, 32w+32) and the matching 32 filter rows 256 + [32w, 32w+32) (forward: also residual / skip rows of GEMM 2), so
// the gate and its derivative are lane-local in the accumulator layout.  B operands are LDS tiles [frame][channel] in
// bf16, rows padded by 16 bytes (a 16-byte fragment read of 16 consecutive frames hits 64 distinct banks); the k = 3 taps
// are row shifts of one tile.  A operands (weights) come from packed bf16 fragment images in global memory (L2
// resident), one 1 KiB coalesced load per wave per (row block, k-step), prefetched 4 k-steps ahead.
#include <stdlib.h>

#include "common.h"
#include "rows_sum.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// debug: phase time stamps (s_memtime) of one block of the layer kernels, see set_debug_bf16_phase_buffer
__device__ uint64_t *g_bf16_phase_buf = nullptr;
// (the per-layer kernels' stamps are stores through a generic pointer: compiled in only with -DSET_BF16_PROBE=1 -- tools/build_exp.sh
// bf16probe diffnet_bf16.hip -DSET_BF16_PROBE=1 --, because one flat access makes the wait-count pass drain the weight ring with
// vmcnt(0) at the top of every k-step group of the PRODUCTION kernel; the fused-layers kernels sum their stamps in scalar registers)
#ifndef SET_BF16_PROBE
#define SET_BF16_PROBE 0
#endif
#define BF16_PHASE(i)                                                                                    \
    if (SET_BF16_PROBE && g_bf16_phase_buf && blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0) g_bf16_phase_buf[i] = __builtin_amdgcn_s_memtime();

namespace {

constexpr int FC = 256;    // residual channels
constexpr int FH = 192;    // conditioner channels (hidden_size)
constexpr int FNT = 128;   // frames per tile
constexpr int XR = FC * 2 + 16;      // bytes per LDS row of a [frame][256] bf16 tile
constexpr int CR = FH * 2 + 16;      // ... of the [frame][192] conditioner tile
constexpr int DR = 2 * FC * 2 + 16;  // ... of a [frame][512] tile (backward)
constexpr int KS_C = FH / 16;        // 12 k-steps of the conditioner projection
constexpr int KS_T = FC / 16;        // 16 k-steps per tap
constexpr int KS1 = KS_C + 3 * KS_T; // 60 k-steps of GEMM 1
constexpr int KS2 = FC / 16;         // 16 k-steps of GEMM 2
constexpr int PF = 4;                // A-fragment prefetch distance (k-steps)
constexpr float RSQRT2 = 0.70710678118654752440f;

__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }
__device__ __forceinline__ f32x16 mma16(u32x4_t a, u32x4_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4_t buf_load_u4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ unsigned short buf_load_u16(rsrc_t r, unsigned voff, unsigned soff) {
    return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store_u16(unsigned short v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b16((short)v, r, (int)voff, (int)soff, 0);
}
// The bf16 tensors the training kernels hand to each other (y16, z16, dy16, do16: [B][C][T] logically) live in a CHANNEL-QUAD-INTERLEAVED
// layout: element (c, t) of a batch slice at byte ((c >> 2) T + t) 8 + (c & 3) 2, i.e. [C / 4][T][4].  The four consecutive accumulator
// registers of a lane are four consecutive channels of one frame, so a lane stores / loads them as ONE 8-byte access (lanes = consecutive
// frames: 512 contiguous bytes per wave) instead of four 2-byte ones -- these kernels issued 192 (forward) / 384 (backward) 2-byte memory
// instructions per lane, and the backward ran 65 instead of 85 us without them.  The weight-gradient loaders (bf16.hip) read the same layout
// in 16-byte units (2 frames x 4 channels) and transpose 8 frames x 4 channels in registers (v_perm_b32).  Same values as before, another place.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void buf_store_q4(u32x2_t v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ u32x2_t buf_load_q4(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ int urow(int r) { return (r & 3) + 8 * (r >> 2); }  // + 4 * (lane >> 5)

// acc[NRB][NCB] += A * B over `nks` k-steps.  A: image [ks][NRB][lane][8 bf16] at `img` (byte offsets), one b128 load
// per (ks, rb), ring of PF k-steps.  B: bfrag(ks, cb) returns the LDS byte address of this lane's 16-byte fragment.
template <int NRB, int NCB, typename BF>
__device__ __forceinline__ void gemm_bf16(f32x16 (&acc)[NRB][NCB], rsrc_t img, unsigned lane16, int ks0, int nks,
                                          const unsigned char *lds, BF bfrag) {
    u32x4_t A[PF][NRB];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
            A[p][rb] = buf_load_u4(img, lane16, (unsigned)(((ks0 + min(p, nks - 1)) * NRB + rb) * 1024));
    for (int kb = 0; kb < nks; kb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int ks = kb + p;  // nks is a multiple of PF
            u32x4_t Bv[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) Bv[cb] = *reinterpret_cast<const u32x4_t *>(lds + bfrag(ks, cb));
            // pinned order (round 4): B fragments | MFMAs straight from the ring slot | the slot's refill PF k-steps ahead.  (Copying the
            // slot and refilling it in front of the MFMAs made the compiler rotate the ring through v_mov chains behind vmcnt(0).)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mma16(A[p][rb], Bv[cb], acc[rb][cb]);
            __builtin_amdgcn_s_setprio(0);
            const int kn = min(ks + PF, nks - 1);  // tail: harmless re-load of the last k-step
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) A[p][rb] = buf_load_u4(img, lane16, (unsigned)(((ks0 + kn) * NRB + rb) * 1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// =====================================================================================================================
// packed images (bf16), lane l of a fragment holds row (l & 31), k = 8 * (l >> 5) + e:
//   w1b[w][ks][rb][l][e]   forward GEMM 1: row = (rb ? 256 : 0) + 32 w + (l&31);  ks < 12: Wcond[row][16 ks + k]
//                          else tap = (ks-12)/16, Wdil[row][16 ((ks-12)%16) + k][tap]
//   w2b[w][ks][rb][l][e]   forward GEMM 2: Wout[row][16 ks + k]
//   wt2[w][ks][l][e]       backward dz:    row = 32 w + (l&31) (z channel), Wout[16 ks + k][row]
//   wt1[w][ks][l][e]       backward dx:    tap = ks/32, row = 32 w + (l&31) (input channel), Wdil[16 (ks%32) + k][row][tap]
//   wtc[g][ks][l][e]       backward dcond: g < 6, row = 32 g + (l&31) (conditioner channel), Wcond[16 ks + k][row]
// =====================================================================================================================
constexpr int64_t N_W1B = 8LL * KS1 * 2 * 64 * 8, N_W2B = 8LL * KS2 * 2 * 64 * 8;
constexpr int64_t N_WT2 = 8LL * 32 * 64 * 8, N_WT1 = 8LL * 96 * 64 * 8, N_WTC = 6LL * 32 * 64 * 8;
constexpr int64_t OFF_W2B = N_W1B, OFF_WT2 = OFF_W2B + N_W2B, OFF_WT1 = OFF_WT2 + N_WT2, OFF_WTC = OFF_WT1 + N_WT1;
constexpr int64_t N_IMG = OFF_WTC + N_WTC;  // bf16 elements per layer

// blockIdx.y = layer: layer q reads w* + q * its element stride and writes image q (one launch re-rounds the whole stack after an update)
__global__ void __launch_bounds__(256) pack_layer_bf16_kernel(const float *wdil, const float *wcond, const float *wout,
                                                              unsigned short *img, int64_t sdil, int64_t scond, int64_t sout) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= N_IMG) return;
    wdil += (int64_t)blockIdx.y * sdil; wcond += (int64_t)blockIdx.y * scond; wout += (int64_t)blockIdx.y * sout;
    img += (int64_t)blockIdx.y * N_IMG;
    float v;
    if (idx < OFF_W2B) {
        int64_t r = idx;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int rb = r & 1; r >>= 1;
        const int ks = (int)(r % KS1), w = (int)(r / KS1);
        const int row = (rb ? FC : 0) + 32 * w + (l & 31), k = 8 * (l >> 5) + e;
        if (ks < KS_C) v = wcond[(int64_t)row * FH + 16 * ks + k];
        else { const int tap = (ks - KS_C) / KS_T, ch = 16 * ((ks - KS_C) % KS_T) + k; v = wdil[((int64_t)row * FC + ch) * 3 + tap]; }
    } else if (idx < OFF_WT2) {
        int64_t r = idx - OFF_W2B;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int rb = r & 1; r >>= 1;
        const int ks = (int)(r % KS2), w = (int)(r / KS2);
        const int row = (rb ? FC : 0) + 32 * w + (l & 31), k = 8 * (l >> 5) + e;
        v = wout[(int64_t)row * FC + 16 * ks + k];
    } else if (idx < OFF_WT1) {
        int64_t r = idx - OFF_WT2;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int ks = (int)(r % 32), w = (int)(r / 32);
        const int row = 32 * w + (l & 31), k = 16 * ks + 8 * (l >> 5) + e;
        v = wout[(int64_t)k * FC + row];
    } else if (idx < OFF_WTC) {
        int64_t r = idx - OFF_WT1;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int ks = (int)(r % 96), w = (int)(r / 96);
        const int tap = ks / 32, row = 32 * w + (l & 31), k = 16 * (ks % 32) + 8 * (l >> 5) + e;
        v = wdil[((int64_t)k * FC + row) * 3 + tap];
    } else {
        int64_t r = idx - OFF_WTC;
        const int e = r & 7; r >>= 3;
        const int l = r & 63; r >>= 6;
        const int ks = (int)(r % 32), g = (int)(r / 32);
        const int row = 32 * g + (l & 31), k = 16 * ks + 8 * (l >> 5) + e;
        v = wcond[(int64_t)k * FH + row];
    }
    img[idx] = f2bf(v);
}

// =====================================================================================================================
// forward
// =====================================================================================================================
// acc[NRB][NCB] += A * B with a caller-supplied A address: aoff(ks, rb) = byte offset of the 1 KiB fragment block
template <int NRB, int NCB, int PFD = PF, typename AF, typename BF>
__device__ __forceinline__ void gemm_bf16_a(f32x16 (&acc)[NRB][NCB], rsrc_t img, unsigned lane16, int nks, const unsigned char *lds,
                                            AF aoff, BF bfrag) {
    u32x4_t A[PFD][NRB];
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) A[p][rb] = buf_load_u4(img, lane16, aoff(min(p, nks - 1), rb));
    for (int kb = 0; kb < nks; kb += PFD) {
#pragma unroll
        for (int p = 0; p < PFD; ++p) {
            const int ks = kb + p;  // nks is a multiple of PFD
            u32x4_t Bv[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) Bv[cb] = *reinterpret_cast<const u32x4_t *>(lds + bfrag(ks, cb));
            __builtin_amdgcn_sched_barrier(0);  // pinned order, see gemm_bf16
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = mma16(A[p][rb], Bv[cb], acc[rb][cb]);
            __builtin_amdgcn_s_setprio(0);
            const int kn = min(ks + PFD, nks - 1);  // tail: harmless re-load of the last k-step
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) A[p][rb] = buf_load_u4(img, lane16, aoff(kn, rb));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// TRAIN: also store the pre-gate y and the gated z in bf16 (operands of the backward).
// NT: frames per tile = 128: 8 waves, one 32-row gate block + its filter block per wave, one block per CU (120 KB of LDS).
template <bool TRAIN, int NT>
__global__ void __launch_bounds__(NT * 4, 1) diffnet_layer_fwd_bf16_kernel(SetDiffnetLayerBf16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NW = NT / 16;   // waves per block
    constexpr int RBW = 8 / NW;   // 32-row gate blocks per wave (and as many filter blocks)
    constexpr int NCB = NT / 32;  // 32-frame column blocks
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, t0 = blockIdx.x * NT, T = a.T, d = a.dil;
    const int XROWS = NT + 2 * d;
    unsigned char *xs = lds;                   // [XROWS][XR]   x + d, row j <-> frame t0 - d + j   (z overlays it: row j <-> t0 + j)
    unsigned char *cs = lds + XROWS * XR;      // [NT][CR]      cond,  row j <-> frame t0 + j
    float *dsh = reinterpret_cast<float *>(lds + XROWS * XR + NT * CR);  // [256] per-utterance step offsets
    float *bsh = dsh + FC;                                                // [512] b_dil + b_cond | [512] b_out
    const unsigned T4 = 4u * (unsigned)T, T2 = 2u * (unsigned)T;
    const rsrc_t rx = make_rsrc(a.x_in + (int64_t)b * FC * T), rxo = make_rsrc(a.x_out + (int64_t)b * FC * T);
    const rsrc_t rsk = make_rsrc(a.skip + (int64_t)b * FC * T), rcd = make_rsrc(a.cond + (int64_t)b * FH * T);
    const rsrc_t rd = make_rsrc(a.dstep + (int64_t)b * a.d_bs);
    const rsrc_t ry = make_rsrc(TRAIN ? a.y16 + (int64_t)b * 2 * FC * T : (uint16_t *)a.skip);
    const rsrc_t rz = make_rsrc(TRAIN ? a.z16 + (int64_t)b * FC * T : (uint16_t *)a.skip);
    const rsrc_t rbd = make_rsrc(a.b_dil), rbc = make_rsrc(a.b_cond), rbo = make_rsrc(a.b_out);
    const unsigned short *img = reinterpret_cast<const unsigned short *>(a.img);
    const rsrc_t rw1 = make_rsrc(img), rw2 = make_rsrc(img + OFF_W2B);
    const unsigned lane16 = 16u * (unsigned)lane;
    // accumulator index rb = g * RBW + q: g = 0 gate / residual rows, 1 filter / skip rows; q-th 32-row block of the wave,
    // i.e. block vw = RBW * w + q of the packed images ("virtual wave" of the 8-wave layout)
    auto row0 = [&](int rb) { return ((rb / RBW) ? FC : 0) + 32 * (RBW * w + (rb % RBW)); };
    auto aoff1 = [&](int ks, int rb) { return (unsigned)((((RBW * w + (rb % RBW)) * KS1 + ks) * 2 + rb / RBW) * 1024); };
    auto aoff2 = [&](int ks, int rb) { return (unsigned)((((RBW * w + (rb % RBW)) * KS2 + ks) * 2 + rb / RBW) * 1024); };

    BF16_PHASE(0)
    // ---- stage the tiles: thread (frame row f, channel group cg).  ALL loads of the main pass (64 x + 48 cond per
    //      thread; no accumulator is live yet) are issued before the first one is consumed: one memory round trip instead
    //      of one per 32-channel batch; the per-utterance step offsets d[256] go through LDS (one load per channel per
    //      block instead of one per element).  Loads are unconditional on clamped addresses, selects after.
    //      (Measured and dropped, round 3: 16-byte loads -- a lane fetching 4 consecutive frames of a channel, 24 - 40 loads per
    //      thread instead of 112 - 176, 4 rows x 16 bytes to LDS per 8 loads: the kernel went from 61.7 to 75.7 us; the x tile starts
    //      at frame t0 - d, so every one of those loads is misaligned, and the LDS writes of a wave land 4 rows apart.)
    {
        const int f = tid % NT, cg = __builtin_amdgcn_readfirstlane(tid / NT);  // cg 0..3
        if (tid < FC) dsh[tid] = buf_load(rd, 0u, (unsigned)tid * 4u * (unsigned)a.d_cs);
        // biases through LDS too (round 6): the accumulator starts read them as 16-byte LDS reads instead of 96 dependent 4-byte loads per lane
        bsh[tid] = buf_load(rbd, 4u * (unsigned)tid, 0u) + buf_load(rbc, 4u * (unsigned)tid, 0u);
        bsh[2 * FC + tid] = buf_load(rbo, 4u * (unsigned)tid, 0u);
        const int t = t0 - d + f;           // x row j = f
        const bool tvx = t >= 0 && t < T;
        const unsigned vox = 4u * (unsigned)min(max(t, 0), T - 1);
        const int tcn = t0 + f;             // cond row f
        const bool tvc = tcn < T;
        const unsigned voc = 4u * (unsigned)min(tcn, T - 1);
        float vx[64], vc[48];
#pragma unroll
        for (int k = 0; k < 64; ++k) vx[k] = buf_load(rx, vox, (unsigned)(64 * cg + k) * T4);
#pragma unroll
        for (int k = 0; k < 48; ++k) vc[k] = buf_load(rcd, voc, (unsigned)(48 * cg + k) * T4);
        __syncthreads();  // dsh
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dsh + 64 * cg + 8 * q);
            const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dsh + 64 * cg + 8 * q + 4);
            u32x4_t u;
            u[0] = tvx ? pack2(vx[8 * q + 0] + d0[0], vx[8 * q + 1] + d0[1]) : 0u;
            u[1] = tvx ? pack2(vx[8 * q + 2] + d0[2], vx[8 * q + 3] + d0[3]) : 0u;
            u[2] = tvx ? pack2(vx[8 * q + 4] + d1[0], vx[8 * q + 5] + d1[1]) : 0u;
            u[3] = tvx ? pack2(vx[8 * q + 6] + d1[2], vx[8 * q + 7] + d1[3]) : 0u;
            *reinterpret_cast<u32x4_t *>(xs + f * XR + (64 * cg + 8 * q) * 2) = u;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            u32x4_t u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = tvc ? pack2(vc[8 * q + 2 * e], vc[8 * q + 2 * e + 1]) : 0u;
            *reinterpret_cast<u32x4_t *>(cs + f * CR + (48 * cg + 8 * q) * 2) = u;
        }
        // halo rows j = NT .. NT + 2d - 1: only the first 2d lanes of each channel group have one
        if (f < 2 * d) {
            const int j = NT + f, th = t0 - d + j;
            const bool tvh = th >= 0 && th < T;
            const unsigned voh = 4u * (unsigned)min(max(th, 0), T - 1);
            float vh[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) vh[k] = buf_load(rx, voh, (unsigned)(64 * cg + k) * T4);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32x4_t u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    u[e] = tvh ? pack2(vh[8 * q + 2 * e] + dsh[64 * cg + 8 * q + 2 * e], vh[8 * q + 2 * e + 1] + dsh[64 * cg + 8 * q + 2 * e + 1]) : 0u;
                *reinterpret_cast<u32x4_t *>(xs + j * XR + (64 * cg + 8 * q) * 2) = u;
            }
        }
    }
    // ---- accumulators start at the biases (b_dil + b_cond): row of register r = row0(rb) + urow(r) + 4 half
    f32x16 acc[2 * RBW][NCB];
    const unsigned lb = 16u * (unsigned)half;  // byte offset of the lane's 4-row group
#pragma unroll
    for (int rb = 0; rb < 2 * RBW; ++rb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bsh + row0(rb) + 8 * g4 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb][4 * g4 + e] = bv[e];
        }
    __syncthreads();
    BF16_PHASE(1)

    // ---- GEMM 1: y = [Wcond | Wdil tap 0 | tap 1 | tap 2] x [cond ; x+d shifted]
    gemm_bf16_a<2 * RBW, NCB>(acc, rw1, lane16, KS_C, lds, aoff1, [&](int ks, int cb) {
        return (unsigned)(XROWS * XR + (cb * 32 + l31) * CR + (ks * 16 + half * 8) * 2);
    });
    gemm_bf16_a<2 * RBW, NCB>(acc, rw1, lane16, 3 * KS_T, lds, [&](int ks, int rb) { return aoff1(KS_C + ks, rb); },
                              [&](int ks, int cb) {
        const int tap = ks >> 4, c0 = (ks & 15) * 16;
        return (unsigned)((cb * 32 + l31 + tap * d) * XR + (c0 + half * 8) * 2);
    });

    BF16_PHASE(2)
    // ---- gate (lane-local: rb < RBW gate rows, rb + RBW the matching filter rows); save y and z in bf16; z tile over the x tile
    bool tv[NCB];
    unsigned vo4[NCB], vo2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int t = t0 + cb * 32 + l31;
        tv[cb] = t < T;
        vo4[cb] = 4u * (unsigned)(4 * half * T + min(t, T - 1));
        vo2[cb] = 8u * (unsigned)(half * T + min(t, T - 1));  // quad-interleaved bf16: channel quad + half, frame t
    }
    // residual rows of x for GEMM 2's accumulator start: issued here, consumed after the gate (hidden under it)
    float xres[RBW][NCB][16];
#pragma unroll
    for (int q = 0; q < RBW; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) xres[q][cb][r] = buf_load(rx, vo4[cb], (unsigned)(row0(q) + urow(r)) * T4);
    __syncthreads();  // every wave is done reading the x tile
#pragma unroll
    for (int q = 0; q < RBW; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {  // registers 4 g4 .. 4 g4 + 3 = channels row0(q) + 8 g4 + 4 half + (0 .. 3)
                float yg[4], yf[4], z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yg[e] = acc[q][cb][4 * g4 + e]; yf[e] = acc[RBW + q][cb][4 * g4 + e];
                    z[e] = tv[cb] ? fsig(yg[e]) * ftanh(yf[e]) : 0.0f;
                }
                u32x2_t zq;
                zq[0] = pack2(z[0], z[1]); zq[1] = pack2(z[2], z[3]);
                if constexpr (TRAIN) {
                    if (tv[cb]) {
                        const unsigned so = (unsigned)(row0(q) + 8 * g4) * T2;
                        u32x2_t gq, fq;
                        gq[0] = pack2(yg[0], yg[1]); gq[1] = pack2(yg[2], yg[3]);
                        fq[0] = pack2(yf[0], yf[1]); fq[1] = pack2(yf[2], yf[3]);
                        buf_store_q4(gq, ry, vo2[cb], so);
                        buf_store_q4(fq, ry, vo2[cb], so + (unsigned)FC * T2);
                        buf_store_q4(zq, rz, vo2[cb], so);
                    }
                }
                *reinterpret_cast<u32x2_t *>(xs + (cb * 32 + l31) * XR + (row0(q) + 8 * g4 + 4 * half) * 2) = zq;
            }
    // ---- accumulators of GEMM 2: residual rows start at x + b_out, skip rows at b_out (the running skip sum is added in
    //      the epilogue, after the x' stores are in flight)
    const bool first = a.first != 0;
#pragma unroll
    for (int rb = 0; rb < 2 * RBW; ++rb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bsh + 2 * FC + row0(rb) + 8 * g4 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rb][cb][4 * g4 + e] = rb < RBW ? bv[e] + xres[rb % RBW][cb][4 * g4 + e] : bv[e];
        }
    __syncthreads();
    BF16_PHASE(3)

    // ---- GEMM 2: o = Wout z
    gemm_bf16_a<2 * RBW, NCB>(acc, rw2, lane16, KS2, lds, aoff2, [&](int ks, int cb) {
        return (unsigned)((cb * 32 + l31) * XR + (ks * 16 + half * 8) * 2);
    });

    BF16_PHASE(4)
    // ---- epilogue: x' stores first (they need nothing), then the running skip sum is fetched and updated
#pragma unroll
    for (int q = 0; q < RBW; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (tv[cb]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) buf_store(acc[q][cb][r] * RSQRT2, rxo, vo4[cb], (unsigned)(row0(q) + urow(r)) * T4);
            }
        }
    float sk[RBW][NCB][16];
#pragma unroll
    for (int q = 0; q < RBW; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sk[q][cb][r] = buf_load(rsk, vo4[cb], (unsigned)(row0(q) + urow(r)) * T4);
#pragma unroll
    for (int q = 0; q < RBW; ++q)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (tv[cb]) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    buf_store(first ? acc[RBW + q][cb][r] : acc[RBW + q][cb][r] + sk[q][cb][r], rsk, vo4[cb],
                              (unsigned)(row0(q) + urow(r)) * T4);
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BF16_PHASE(5)
}


// =====================================================================================================================
// Several residual layers per launch (inference).  One launch per layer streams x (1024 B per frame), the conditioner (768),
// x' (1024) and the running skip sum (2048, read-modify-write) through HBM for ~17 us of MFMA work per tile, and the HBM phases
// of all blocks coincide: 55 us per layer, 28 % of the HBM peak.  In the kernels below a block keeps its tile ON CHIP for NL consecutive
// layers: x is loaded once, the conditioner tile once (every layer projects the SAME conditioner), the layer output goes straight back
// into the LDS operand tile.  Price: a tile of NT computed frames yields NT - 2 H valid ones, H = sum of the dilations of layers
// 1 .. NL - 1 (frames whose receptive field left the tile are recomputed by the neighbour).
//   compute frames of the tile: ts + j, j < NT, ts = tile * NV - H;  stored: the NV frames from ts + H on
//   xs row j of layer m <-> frame ts - d_m + j   (d_m halo rows on either side; beyond the loaded / computed frames: garbage that
//   only ever reaches frames outside the stored range)
// (Round 3's shapes -- <128, 8> with the fp32 x' and the skip sum through the block's private memory, <64, 4> as two blocks per CU --
// were measured slower and are gone; history: DESIGN.md section 3.5.)
// =====================================================================================================================
struct LayersArgs {
    SetDiffnetLayersBf16Args a;
    int hh, nv;  // halo H and valid frames per tile
};

// =====================================================================================================================
// Register-resident fused layer groups (round 4; the default shape of set_diffnet_layers_fwd_bf16): 64-frame tiles on 8 waves,
// the wave's rows of the fp32 x' and of the skip sum in registers for the whole group (as the round-3 <64, 8> shape), and
//   * the gated z has its OWN LDS tile: a layer has two barriers (x tile complete -> GEMM 1; z tile complete -> GEMM 2) instead of
//     four -- a wave that leaves GEMM 1 goes straight into its gate math (VALU) while its SIMD partner is still issuing MFMAs, and
//     its epilogue runs under the partner's GEMM 2;
//   * barriers wait for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): the A-fragment loads of the NEXT GEMM are issued
//     BEFORE the barrier that opens it, so the L2 round trip of the first k-steps is spent waiting for the slowest wave anyway;
//   * GEMM 1 is ONE 60-k-step stream (12 conditioner + 3 x 16 tap k-steps; the ring of PFD k-steps never drains between the
//     conditioner and the taps);
//   * the biases of every fused layer are staged in LDS once per launch (they were 96 dependent 4-byte L2 loads per wave and layer);
//   * SKEW (template parameter, not instantiated in the shipped library): a static priority for waves 0-3 (wave w and w + 4 share a
//     SIMD) instead of the per-k-step toggles.  Measured both ways round (priority to the older / to the younger half): the favoured
//     half leaves GEMM 1 after 54 % of a layer and waits at the barrier, the other after 69 %; the layer takes the same time.
// Same arithmetic per frame as diffnet_layer_fwd_bf16_kernel (same images, k order, accumulator start, rounding points): bit-identical.
// =====================================================================================================================
// acc[2][2] += A B over NKS k-steps in groups of 4; the ring A[4 D][2] holds k-steps 0 .. 4 D - 1 on entry (gemm_reg_prefetch) and is
// refilled 4 D k-steps ahead (D = 1, 2 groups).  The schedule is pinned, one sched_barrier per k-step:
//     ds_read B(k + 1)  |  4 MFMAs of k-step k  |  buffer_load A(k + 4 D) into the registers those MFMAs just read
// Left to itself the scheduler (a) fully unrolled: sank every A load to one MFMA pair in front of its use and spilled the hoisted SGPR
// offsets of 120 loads; (b) as a loop: moved all refills of a group to the END of the group and waited for all of them (vmcnt(0))
// at the top of the next -- an L2 round trip per 16 MFMAs.  The group loop is not unrolled (ring registers are loop-carried).
// bgrp(kb) returns the LDS byte addresses of the lane's B fragments of k-step kb for the two column blocks; k-step kb + p is 32 p
// bytes further (a group of 4 never straddles the conditioner / tap segments: 12 and 16 are multiples of 4).
template <int NKS, int D, bool TOGGLE, typename BG>
__device__ __forceinline__ void gemm_reg_run(f32x16 (&acc)[2][2], u32x4_t (&A)[4 * D][2], rsrc_t img, unsigned lane16, unsigned abase,
                                             const unsigned char *lds, BG bgrp) {
    // abase: byte offset of the wave's first fragment block (k-step 0, gate rows); k-step ks, row block rb at abase + (2 ks + rb) KiB
    constexpr int NG = NKS / 4;
    static_assert(NKS % 4 == 0 && NG >= 2 * D && (D == 1 || D == 2), "gemm_reg_run");
    u32x4_t Bq[2][2];  // B fragments of the current / the next k-step (parity of the k-step), [cb]
    {
        unsigned b0, b1;
        bgrp(0, b0, b1);
        Bq[0][0] = *reinterpret_cast<const u32x4_t *>(lds + b0);
        Bq[0][1] = *reinterpret_cast<const u32x4_t *>(lds + b1);
    }
    auto group = [&](int kb, int slot, bool refill, bool more) {  // slot / refill / more: compile-time at every call site
        unsigned b0, b1, n0 = 0, n1 = 0;
        bgrp(kb, b0, b1);
        if (more) bgrp(kb + 4, n0, n1);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p < 3) {
                Bq[(p + 1) & 1][0] = *reinterpret_cast<const u32x4_t *>(lds + b0 + 32 * (p + 1));
                Bq[(p + 1) & 1][1] = *reinterpret_cast<const u32x4_t *>(lds + b1 + 32 * (p + 1));
            } else if (more) {
                Bq[0][0] = *reinterpret_cast<const u32x4_t *>(lds + n0);
                Bq[0][1] = *reinterpret_cast<const u32x4_t *>(lds + n1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (TOGGLE) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = mma16(A[4 * slot + p][rb], Bq[p & 1][cb], acc[rb][cb]);
            if (TOGGLE) __builtin_amdgcn_s_setprio(0);
            if (refill) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    A[4 * slot + p][rb] = buf_load_u4(img, lane16, abase + (unsigned)((2 * (kb + p + 4 * D) + rb) * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (D == 1) {
#pragma unroll 1
        for (int g = 0; g < NG - 1; ++g) group(4 * g, 0, true, true);
        group(4 * (NG - 1), 0, false, false);
    } else {
        constexpr int NR = NG - 2;          // groups that refill their slot
        constexpr int NP = NR / 2;          // ... in pairs
#pragma unroll 1
        for (int g = 0; g < 2 * NP; g += 2) { group(4 * g, 0, true, true); group(4 * g + 4, 1, true, true); }
        if constexpr (NR % 2) {             // (NG odd) one more refilling group: slot 0, then slots 1, 0 drain
            group(4 * (NG - 3), 0, true, true); group(4 * (NG - 2), 1, false, true); group(4 * (NG - 1), 0, false, false);
        } else {
            group(4 * (NG - 2), 0, false, true); group(4 * (NG - 1), 1, false, false);
        }
    }
}
template <int PFD>
__device__ __forceinline__ void gemm_reg_prefetch(u32x4_t (&A)[PFD][2], rsrc_t img, unsigned lane16, unsigned abase) {
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) A[p][rb] = buf_load_u4(img, lane16, abase + (unsigned)((2 * p + rb) * 1024));
}
// pin a computed value where it is written in the source: an empty volatile asm that "modifies" the register.  Without it LLVM sinks a
// pure computation to its first use -- e.g. the gate of pass 0, used only after the barrier behind pass 1, moved below pass 1 and kept
// the 64 accumulators of pass 0 alive through it (92 spilled registers), and the packing of the next x tile moved behind its barrier
__device__ __forceinline__ void pin(unsigned &v) { asm volatile("" : "+v"(v)); }
// LDS-only barrier: the waves exchange nothing but LDS tiles, and the A-fragment loads in flight must stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool SKEW, int PF1>
__global__ void __launch_bounds__(512, 1) diffnet_layers_reg_bf16_kernel(LayersArgs la) {
    const SetDiffnetLayersBf16Args &a = la.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = 64, NCB = 2, NTH = 512, NCG = NTH / NT, CPX = FC / NCG, CPC = FH / NCG;
    constexpr int PF2 = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, T = a.T;
    const int tv0 = blockIdx.x * la.nv, ts = tv0 - la.hh;  // first stored / first computed frame
    const int tv1 = min(tv0 + la.nv, T);                    // end of the stored range
    int dmax = 1;
    for (int m = 0; m < a.nl; ++m) dmax = max(dmax, 1 << ((a.l0 + m) % a.dilation_cycle_length));
    const int XROWS_MAX = NT + 2 * dmax;
    // LDS: x operand tile [XROWS_MAX][XR] | conditioner tile [NT][CR] | z tile [NT][XR] | step offsets [nl][256] | biases [nl][2][512]
    const unsigned XS = 0u, CS = (unsigned)(XROWS_MAX * XR), ZS = CS + NT * CR, DS = ZS + NT * XR;
    unsigned char *xs = lds + XS, *cs = lds + CS, *zs = lds + ZS;
    float *dsh = reinterpret_cast<float *>(lds + DS);
    float *bsh = dsh + a.nl * FC;  // [m][0]: b_dil + b_cond, [m][1]: b_out
    const unsigned T4 = 4u * (unsigned)T;
    const rsrc_t rx = make_rsrc(a.x_in + (int64_t)b * FC * T), rxo = make_rsrc(a.x_out + (int64_t)b * FC * T);
    const rsrc_t rsk = make_rsrc(a.skip + (int64_t)b * FC * T), rcd = make_rsrc(a.cond + (int64_t)b * FH * T);
    const unsigned lane16 = 16u * (unsigned)lane;
    const int row_g = 32 * w;                  // the wave's gate / residual rows [row_g, row_g + 32); filter / skip rows + 256
    const unsigned ab1 = (unsigned)(w * KS1 * 2 * 1024), ab2 = (unsigned)(w * KS2 * 2 * 1024);

    // ---- stage: step offsets and biases of every fused layer, x + d_0 (halo d of layer 0), the conditioner tile
    {
        for (int i = tid; i < a.nl * FC; i += NTH) {
            const int m = i >> 8, c = i & 255;
            dsh[i] = a.dstep[(int64_t)(a.l0 + m) * a.d_ls + (int64_t)b * a.d_bs + (int64_t)c * a.d_cs];
        }
        for (int i = tid; i < a.nl * 512; i += NTH) {
            const int m = i >> 9, r = i & 511;
            bsh[m * 1024 + r] = a.b_dil[i] + a.b_cond[i];
            bsh[m * 1024 + 512 + r] = a.b_out[i];
        }
        const int d = 1 << (a.l0 % a.dilation_cycle_length);
        const int f = tid % NT, cg = __builtin_amdgcn_readfirstlane(tid / NT);  // cg 0 .. 7
        const int t = ts - d + f;
        const bool tvx = t >= 0 && t < T;
        const unsigned vox = 4u * (unsigned)min(max(t, 0), T - 1);
        const int tcn = ts + f;
        const bool tvc = tcn >= 0 && tcn < T;
        const unsigned voc = 4u * (unsigned)min(max(tcn, 0), T - 1);
        float vx[CPX], vc[CPC];
#pragma unroll
        for (int k = 0; k < CPX; ++k) vx[k] = buf_load(rx, vox, (unsigned)(CPX * cg + k) * T4);
#pragma unroll
        for (int k = 0; k < CPC; ++k) vc[k] = buf_load(rcd, voc, (unsigned)(CPC * cg + k) * T4);
        // rows beyond layer 0's tile (layers with a larger dilation address up to XROWS_MAX rows): finite filler
        for (int i = tid; i < (XROWS_MAX - (NT + 2 * d)) * (XR / 16); i += NTH)
            *reinterpret_cast<u32x4_t *>(xs + (NT + 2 * d) * XR + i * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        __syncthreads();  // dsh
#pragma unroll
        for (int q = 0; q < CPX / 8; ++q) {
            const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dsh + CPX * cg + 8 * q);
            const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dsh + CPX * cg + 8 * q + 4);
            u32x4_t u;
            u[0] = tvx ? pack2(vx[8 * q + 0] + d0[0], vx[8 * q + 1] + d0[1]) : 0u;
            u[1] = tvx ? pack2(vx[8 * q + 2] + d0[2], vx[8 * q + 3] + d0[3]) : 0u;
            u[2] = tvx ? pack2(vx[8 * q + 4] + d1[0], vx[8 * q + 5] + d1[1]) : 0u;
            u[3] = tvx ? pack2(vx[8 * q + 6] + d1[2], vx[8 * q + 7] + d1[3]) : 0u;
            *reinterpret_cast<u32x4_t *>(xs + f * XR + (CPX * cg + 8 * q) * 2) = u;
        }
#pragma unroll
        for (int q = 0; q < CPC / 8; ++q) {
            u32x4_t u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = tvc ? pack2(vc[8 * q + 2 * e], vc[8 * q + 2 * e + 1]) : 0u;
            *reinterpret_cast<u32x4_t *>(cs + f * CR + (CPC * cg + 8 * q) * 2) = u;
        }
        if (f < 2 * d) {
            const int j = NT + f, th = ts - d + j;
            const bool tvh = th >= 0 && th < T;
            const unsigned voh = 4u * (unsigned)min(max(th, 0), T - 1);
            float vh[CPX];
#pragma unroll
            for (int k = 0; k < CPX; ++k) vh[k] = buf_load(rx, voh, (unsigned)(CPX * cg + k) * T4);
#pragma unroll
            for (int q = 0; q < CPX / 8; ++q) {
                u32x4_t u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    u[e] = tvh ? pack2(vh[8 * q + 2 * e] + dsh[CPX * cg + 8 * q + 2 * e], vh[8 * q + 2 * e + 1] + dsh[CPX * cg + 8 * q + 2 * e + 1]) : 0u;
                *reinterpret_cast<u32x4_t *>(xs + j * XR + (CPX * cg + 8 * q) * 2) = u;
            }
        }
    }
    // per column block: frame of this lane, whether it lies inside the utterance, whether it is stored
    // (mT: all-ones / zero word ANDed onto the packed bf16 pairs -- a `frame inside T ? f(y) : 0` select compiled to one exec-mask
    //  branch per ELEMENT around the transcendental chain, 32 serialized chains per wave and gate)
    bool st[NCB];
    unsigned vo4[NCB], mT[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int t = ts + cb * 32 + l31;
        mT[cb] = (t >= 0 && t < T) ? 0xffffffffu : 0u;
        st[cb] = t >= tv0 && t < tv1;
        vo4[cb] = 4u * (unsigned)(4 * half * T + min(max(t, 0), T - 1));
    }
    // the wave's rows of x (fp32, the residual chain) and of the skip sum, in registers for the whole group
    float xkeep[NCB][16], skacc[NCB][16];
    {
        const bool first0 = a.first != 0;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xkeep[cb][r] = buf_load(rx, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
                skacc[cb][r] = first0 ? 0.0f : buf_load(rsk, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
            }
    }
    // per-lane LDS byte offsets of the B fragments (frame row l31, k half) and of the lane's 4-row groups in the bias vectors
    const unsigned bc0 = CS + (unsigned)(l31 * CR + half * 16);   // conditioner tile
    const unsigned bx0 = XS + (unsigned)(l31 * XR + half * 16);   // x tile, tap 0
    const unsigned bz0 = ZS + (unsigned)(l31 * XR + half * 16);   // z tile
    const unsigned bb0 = DS + (unsigned)(a.nl * FC * 4) + (unsigned)((row_g + 4 * half) * 4);

    // debug (tools/bf16_layers_probe.py): s_memtime ticks of the phases of the layers m >= 1 of block (1, 1), summed in scalar registers
    // and written once at the end by wave 0 (buf[0..4], count in buf[7]) and wave 4 (buf[8..12]): no memory operation inside the
    // layer loop (a flat access there made the wait-count pass drain the A ring, vmcnt(0), at the top of every k-step group)
    const bool probe = g_bf16_phase_buf && blockIdx.x == 1 && blockIdx.y == 1;
    uint64_t tprev = 0, tph[5] = {0, 0, 0, 0, 0};
#define LR_PHASE(i)                                                   \
    if (probe && m >= 1) {                                            \
        const uint64_t tn = __builtin_amdgcn_s_memtime();             \
        tph[i] += tn - tprev;                                         \
        tprev = tn;                                                   \
    }
    typedef unsigned lr_u32x2 __attribute__((ext_vector_type(2)));
    // SKEW: a static priority instead of the per-k-step toggles -- waves 0-3 win every arbitration against their SIMD partners 4-7
    if (SKEW && w < 4) __builtin_amdgcn_s_setprio(1);
    u32x4_t A1[PF1][2], A2[PF2][2];
    {
        const rsrc_t rw1 = make_rsrc(reinterpret_cast<const unsigned short *>(a.img));
        gemm_reg_prefetch<PF1>(A1, rw1, lane16, ab1);
    }
    for (int m = 0; m < a.nl; ++m) {
        if (probe) tprev = __builtin_amdgcn_s_memtime();
        const int l = a.l0 + m, d = 1 << (l % a.dilation_cycle_length);
        const bool last = m == a.nl - 1;
        const unsigned short *img = reinterpret_cast<const unsigned short *>(a.img) + (int64_t)m * N_IMG;
        const rsrc_t rw1 = make_rsrc(img), rw2 = make_rsrc(img + OFF_W2B);
        const unsigned bbm = bb0 + (unsigned)(m * 4096);
        f32x16 acc[2][NCB];
        // accumulators start at b_dil + b_cond: register r <-> row (rb ? 256 : 0) + row_g + urow(r) + 4 half
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + bbm + (unsigned)((rb * FC + 8 * g4) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb][4 * g4 + e] = bv[e];
            }
        lds_barrier();  // the x tile of this layer (staged above / written by the previous layer's epilogue) is complete
        LR_PHASE(0)

        // ---- GEMM 1: y = [Wcond | Wdil tap 0 | tap 1 | tap 2] x [cond ; x + d shifted]
        {
            const unsigned dx = (unsigned)(d * XR);
            auto bf1 = [&](int kb, unsigned &b0, unsigned &b1) {
                // branch-free (a select on the uniform kb compiled to branches inside the k loop, which made the wait-count
                // pass give up on the ring: vmcnt(0) at the top of every group)
                const unsigned mc = (unsigned)((kb - KS_C) >> 31);  // all ones for the conditioner k-steps
                const unsigned kt = (unsigned)(kb - KS_C) & 63u, tap = kt >> 4, c0 = (kt & 15u) * 32u;
                b0 = (mc & (bc0 + (unsigned)kb * 32u)) | (~mc & (bx0 + tap * dx + c0));
                b1 = b0 + ((mc & (32u * CR)) | (~mc & (32u * XR)));
            };
            gemm_reg_run<KS1, PF1 / 4, !SKEW>(acc, A1, rw1, lane16, ab1, lds, bf1);
        }
        gemm_reg_prefetch<PF2>(A2, rw2, lane16, ab2);  // GEMM 2's first fragments travel under the gate
        LR_PHASE(1)

        // ---- gate -> z tile (own LDS tile: no wave has to wait for the others to leave GEMM 1)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {  // registers 4 g4 .. 4 g4 + 3 are 4 consecutive channels: one 8-byte write
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = fsig(acc[0][cb][4 * g4 + e]) * ftanh(acc[1][cb][4 * g4 + e]);
                lr_u32x2 u;
                u[0] = pack2(z[0], z[1]) & mT[cb]; u[1] = pack2(z[2], z[3]) & mT[cb];
                *reinterpret_cast<lr_u32x2 *>(zs + (cb * 32 + l31) * XR + (row_g + 8 * g4 + 4 * half) * 2) = u;
            }
        // accumulators of GEMM 2: residual rows start at b_out + x, skip rows at b_out
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + bbm + (unsigned)((512 + rb * FC + 8 * g4) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb][4 * g4 + e] = rb == 0 ? bv[e] + xkeep[cb][4 * g4 + e] : bv[e];
            }
        lds_barrier();  // the z tile is complete (and every wave has left GEMM 1: the x tile may be overwritten)
        LR_PHASE(2)

        // ---- GEMM 2: o = Wout z  (z tile row j <-> frame ts + j)
        {
            auto bf2 = [&](int kb, unsigned &b0, unsigned &b1) { b0 = bz0 + (unsigned)kb * 32u; b1 = b0 + 32u * XR; };
            gemm_reg_run<KS2, PF2 / 4, !SKEW>(acc, A2, rw2, lane16, ab2, lds, bf2);
        }
        if (!last) gemm_reg_prefetch<PF1>(A1, make_rsrc(img + N_IMG), lane16, ab1);  // the next layer's first fragments
        LR_PHASE(3)

        // ---- epilogue: x' = (x + o_res) / sqrt 2 and the skip sum stay in registers; bf16(x' + d_next) -> the x tile
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xkeep[cb][r] = acc[0][cb][r] * RSQRT2;
                skacc[cb][r] = (a.first != 0 && m == 0) ? acc[1][cb][r] : acc[1][cb][r] + skacc[cb][r];
            }
        if (!last) {
            const int dn = 1 << ((l + 1) % a.dilation_cycle_length);
            const float *dnx = dsh + (m + 1) * FC;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ch = row_g + 8 * g4 + 4 * half;
                    const f32x4 dv = *reinterpret_cast<const f32x4 *>(dnx + ch);
                    lr_u32x2 u;
                    u[0] = pack2(xkeep[cb][4 * g4] + dv[0], xkeep[cb][4 * g4 + 1] + dv[1]) & mT[cb];
                    u[1] = pack2(xkeep[cb][4 * g4 + 2] + dv[2], xkeep[cb][4 * g4 + 3] + dv[3]) & mT[cb];
                    *reinterpret_cast<lr_u32x2 *>(xs + (dn + cb * 32 + l31) * XR + ch * 2) = u;
                }
        } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (st[cb]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        buf_store(xkeep[cb][r], rxo, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
                        buf_store(skacc[cb][r], rsk, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
                    }
                }
            }
        }
        LR_PHASE(4)
    }
#undef LR_PHASE
    if (probe && (tid == 0 || tid == 256)) {
        uint64_t *pb = g_bf16_phase_buf + (tid ? 8 : 0);
        for (int i = 0; i < 5; ++i) pb[i] += tph[i];
        pb[7] += (uint64_t)(a.nl - 1);
    }
}

// =====================================================================================================================
// 128-frame register-resident fused layer groups (round 4; the shape for batches that fill the chip).
//
// Why: at 64 frames per block every weight fragment (1 KiB from L2) feeds 2 MFMAs per wave -- 16 x 1 KiB loads per CU and k-step = 256
// cycles of the 64 B/clk L1 path against 256 cycles of MFMA issue per SIMD: the weight stream co-limits the matrix pipe, and a deeper
// ring / fewer barriers did not move the time (profiles/r04_bf16_ab.log).  Here a block owns 128 frames and GEMM 1 runs in TWO PASSES
// over the wave's rows: pass h multiplies ONE 32-row A block [16 gate rows ; the matching 16 filter rows] (gathered from the same
// packed image: lanes 0-15 of each k half read gate-row fragments, lanes 16-31 filter-row fragments) by 4 column blocks -- 4 MFMAs
// per 1 KiB of weights, half the L2 bytes per MFMA, 64 accumulator registers per pass.  The gate stays lane-local (gate row i in
// register r, filter row i in register r + 8).  128 frames also make B = 32, T = 800 exactly ONE round: 8 tiles per utterance = 256
// blocks on 256 CUs for any halo <= 14 (64-frame tiles: 480 blocks = 1.9 rounds).
//   registers: x' rows 64 (they ARE GEMM 2's residual accumulators) + pass accumulators 64 + packed z of pass 0 (16) + A ring 16 +
//   B fragments 16; the skip sum does not fit next to them: between the layers of a group the block's skip rows live in a private
//   buffer in ACCUMULATOR order (16 coalesced 16-byte loads + 16 stores per wave and layer; the skip tensor itself is touched by the
//   first and the last layer of the group only), and GEMM 2 runs in two passes too -- skip rows first, with the old rows requested
//   before the barrier in front of it and added right behind it (the round trip hides under the pass), residual rows second.  (Measured on the way: reading the
//   row-major skip tensor before GEMM 2 and storing after it -- 64 + 64 dword accesses and the L2 round trip in front of a barrier:
//   50 us per layer, 18 % of it there; adding the rows at the L2 with buffer_atomic_add_f32: 56 us, the atomics of 256 blocks
//   serialise and GEMM 2's fragment loads queue behind them.)
//   LDS: x tile (z overlays it), conditioner tile, step offsets of the group, biases double-buffered per layer: 4 barriers per layer, but
//   all VALU work (gate, x' epilogue, packing) is done BEFORE the barrier that frees the tile it is written to; between those barrier
//   pairs a wave only issues its 16 / 32 LDS writes.
// Arithmetic per frame as diffnet_layer_fwd_bf16_kernel (same images, k order, accumulator start, rounding points, skip = old + (bias +
// products)): x and the skip sum are bit-identical.
// =====================================================================================================================
// acc[NRB][4] += A B over NKS k-steps in groups of 4 with the pinned schedule of gemm_reg_run; ring A[4 D][NRB].
// aoff(ks, rb): wave-uniform byte offset of the fragment block; avo: per-lane byte offset inside it.
#ifndef SET_T128_EXP
#define SET_T128_EXP 0
#endif
template <int NRB, int NKS, int D, bool TOGGLE, bool BDB, typename AO, typename BG>
__device__ __forceinline__ void gemm_t128_run(f32x16 (&acc)[NRB][4], u32x4_t (&A)[4 * D][NRB], rsrc_t img, unsigned avo, AO aoff,
                                              const unsigned char *lds, BG bgrp) {
    // BDB: the B fragments of k-step k + 1 are read under the MFMAs of k-step k (16 more registers); else each k-step reads its own
    constexpr int NG = NKS / 4;
    static_assert(NKS % 4 == 0 && NG >= 2 * D && (D == 1 || D == 2), "gemm_t128_run");
    u32x4_t Bq[BDB ? 2 : 1][4];
    if (BDB) {
        unsigned b0, bs;
        bgrp(0, b0, bs);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) Bq[0][cb] = *reinterpret_cast<const u32x4_t *>(lds + b0 + cb * bs);
    }
    auto group = [&](int kb, int slot, bool refill, bool more) {
        unsigned b0, bs, n0 = 0, ns = 0;
        bgrp(kb, b0, bs);
        if (BDB && more) bgrp(kb + 4, n0, ns);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#if (SET_T128_EXP & 2)  // measurement build (tools/build_exp.sh): no B-fragment reads after the first k-step (results are wrong)
            if (kb == 0 && p == 0) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) Bq[0][cb] = *reinterpret_cast<const u32x4_t *>(lds + b0 + cb * bs);
                if (BDB) { for (int cb = 0; cb < 4; ++cb) Bq[1][cb] = Bq[0][cb]; }
            }
#else
            if (!BDB) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) Bq[0][cb] = *reinterpret_cast<const u32x4_t *>(lds + b0 + cb * bs + 32 * p);
            } else if (p < 3) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) Bq[(p + 1) & 1][cb] = *reinterpret_cast<const u32x4_t *>(lds + b0 + cb * bs + 32 * (p + 1));
            } else if (more) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) Bq[0][cb] = *reinterpret_cast<const u32x4_t *>(lds + n0 + cb * ns);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (TOGGLE) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = mma16(A[4 * slot + p][rb], Bq[BDB ? (p & 1) : 0][cb], acc[rb][cb]);
            if (TOGGLE) __builtin_amdgcn_s_setprio(0);
#if !(SET_T128_EXP & 1)  // measurement build: bit 0 = no weight-fragment refills (results are wrong)
            if (refill) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) A[4 * slot + p][rb] = buf_load_u4(img, avo, aoff(kb + p + 4 * D, rb));
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (D == 1) {
#pragma unroll 1
        for (int g = 0; g < NG - 1; ++g) group(4 * g, 0, true, true);
        group(4 * (NG - 1), 0, false, false);
    } else {
        constexpr int NR = NG - 2, NP = NR / 2;
#pragma unroll 1
        for (int g = 0; g < 2 * NP; g += 2) { group(4 * g, 0, true, true); group(4 * g + 4, 1, true, true); }
        if constexpr (NR % 2) {
            group(4 * (NG - 3), 0, true, true); group(4 * (NG - 2), 1, false, true); group(4 * (NG - 1), 0, false, false);
        } else {
            group(4 * (NG - 2), 0, false, true); group(4 * (NG - 1), 1, false, false);
        }
    }
}
template <int NRB, int PFD, typename AO>
__device__ __forceinline__ void gemm_t128_prefetch(u32x4_t (&A)[PFD][NRB], rsrc_t img, unsigned avo, AO aoff) {
#pragma unroll
    for (int p = 0; p < PFD; ++p)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) A[p][rb] = buf_load_u4(img, avo, aoff(p, rb));
}

template <bool SKEW, int NSKR>
__global__ void __launch_bounds__(512, 1) diffnet_layers_t128_bf16_kernel(LayersArgs la) {
    const SetDiffnetLayersBf16Args &a = la.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int NT = 128, NCB = 4, NTH = 512, NCG = NTH / NT, CPX = FC / NCG, CPC = FH / NCG;
    constexpr int D1 = 1, D2 = 1;  // A ring depth in groups of 4 k-steps: GEMM 1 passes 4 k-steps x 1 fragment, GEMM 2 4 k-steps x 2
    constexpr bool BDB1 = false, BDB2 = false;  // B fragments single-buffered (the SIMD partner covers the LDS latency; the registers go to the ring / the old skip rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, T = a.T;
    const int tv0 = blockIdx.x * la.nv, ts = tv0 - la.hh;  // first stored / first computed frame
    const int tv1 = min(tv0 + la.nv, T);                    // end of the stored range
    int dmax = 1;
    for (int m = 0; m < a.nl; ++m) dmax = max(dmax, 1 << ((a.l0 + m) % a.dilation_cycle_length));
    const int XROWS_MAX = NT + 2 * dmax;
    // LDS: x operand tile [XROWS_MAX][XR] (z overlays rows 0 .. 127) | conditioner tile [NT][CR] | step offsets [nl][256] | biases [2][2][512]
    const unsigned XS = 0u, CS = (unsigned)(XROWS_MAX * XR), DS = CS + NT * CR, BS = DS + (unsigned)(a.nl * FC * 4);
    unsigned char *xs = lds + XS, *cs = lds + CS;
    float *dsh = reinterpret_cast<float *>(lds + DS);
    float *bsh = reinterpret_cast<float *>(lds + BS);  // [parity of m][0: b_dil + b_cond, 1: b_out][512]
    const unsigned T4 = 4u * (unsigned)T;
    const rsrc_t rx = make_rsrc(a.x_in + (int64_t)b * FC * T), rxo = make_rsrc(a.x_out + (int64_t)b * FC * T);
    const rsrc_t rsk = make_rsrc(a.skip + (int64_t)b * FC * T), rcd = make_rsrc(a.cond + (int64_t)b * FH * T);
    const unsigned lane16 = 16u * (unsigned)lane;
    const int row_g = 32 * w;  // the wave's gate / residual rows [row_g, row_g + 32); filter / skip rows + 256
    // the block's private copy of its skip rows between the layers of the group, in accumulator order: [wave][cb][g4][lane][4 floats]
    // (one coalesced 16-byte access per lane and 4 registers; the skip tensor itself is row-major: 4-byte accesses)
    const rsrc_t rps = make_rsrc(a.scratch + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (FC * NT) + (int64_t)w * (NCB * 4 * 64 * 4));
    // GEMM 1 pass h: lane l of the A block holds row i = l & 31: i < 16 gate row row_g + 16 h + i, else filter row row_g + 16 h + i - 16;
    // in the packed image w1b[w][ks][rb][lane'][8] that fragment is rb = i >> 4, lane' = 16 h + (i & 15) + 32 (l >> 5)
    const unsigned avo1 = (unsigned)(((l31 >> 4) * 64 + (l31 & 15) + 32 * half) * 16);
    auto aoff1h = [&](int h) { return [=](int ks, int) { return (unsigned)((w * KS1 + ks) * 2048 + h * 256); }; };
    auto aoff2r = [&](int rbsel) { return [=](int ks, int) { return (unsigned)(((w * KS2 + ks) * 2 + rbsel) * 1024); }; };  // GEMM 2: rb 0 residual rows, 1 skip rows

    // ---- stage: step offsets of every fused layer, biases of layer 0, x + d_0 (halo d of layer 0), the conditioner tile
    {
        for (int i = tid; i < a.nl * FC; i += NTH) {
            const int m = i >> 8, c = i & 255;
            dsh[i] = a.dstep[(int64_t)(a.l0 + m) * a.d_ls + (int64_t)b * a.d_bs + (int64_t)c * a.d_cs];
        }
        bsh[tid] = a.b_dil[tid] + a.b_cond[tid];
        bsh[512 + tid] = a.b_out[tid];
        const int d = 1 << (a.l0 % a.dilation_cycle_length);
        const int f = tid % NT, cg = __builtin_amdgcn_readfirstlane(tid / NT);  // cg 0 .. 3
        const int t = ts - d + f;
        const bool tvx = t >= 0 && t < T;
        const unsigned vox = 4u * (unsigned)min(max(t, 0), T - 1);
        const int tcn = ts + f;
        const bool tvc = tcn >= 0 && tcn < T;
        const unsigned voc = 4u * (unsigned)min(max(tcn, 0), T - 1);
        float vx[CPX], vc[CPC];
#pragma unroll
        for (int k = 0; k < CPX; ++k) vx[k] = buf_load(rx, vox, (unsigned)(CPX * cg + k) * T4);
#pragma unroll
        for (int k = 0; k < CPC; ++k) vc[k] = buf_load(rcd, voc, (unsigned)(CPC * cg + k) * T4);
        for (int i = tid; i < (XROWS_MAX - (NT + 2 * d)) * (XR / 16); i += NTH)
            *reinterpret_cast<u32x4_t *>(xs + (NT + 2 * d) * XR + i * 16) = (u32x4_t){0u, 0u, 0u, 0u};
        __syncthreads();  // dsh
#pragma unroll
        for (int q = 0; q < CPX / 8; ++q) {
            const f32x4 d0 = *reinterpret_cast<const f32x4 *>(dsh + CPX * cg + 8 * q);
            const f32x4 d1 = *reinterpret_cast<const f32x4 *>(dsh + CPX * cg + 8 * q + 4);
            u32x4_t u;
            u[0] = tvx ? pack2(vx[8 * q + 0] + d0[0], vx[8 * q + 1] + d0[1]) : 0u;
            u[1] = tvx ? pack2(vx[8 * q + 2] + d0[2], vx[8 * q + 3] + d0[3]) : 0u;
            u[2] = tvx ? pack2(vx[8 * q + 4] + d1[0], vx[8 * q + 5] + d1[1]) : 0u;
            u[3] = tvx ? pack2(vx[8 * q + 6] + d1[2], vx[8 * q + 7] + d1[3]) : 0u;
            *reinterpret_cast<u32x4_t *>(xs + f * XR + (CPX * cg + 8 * q) * 2) = u;
        }
#pragma unroll
        for (int q = 0; q < CPC / 8; ++q) {
            u32x4_t u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = tvc ? pack2(vc[8 * q + 2 * e], vc[8 * q + 2 * e + 1]) : 0u;
            *reinterpret_cast<u32x4_t *>(cs + f * CR + (CPC * cg + 8 * q) * 2) = u;
        }
        if (f < 2 * d) {
            const int j = NT + f, th = ts - d + j;
            const bool tvh = th >= 0 && th < T;
            const unsigned voh = 4u * (unsigned)min(max(th, 0), T - 1);
            float vh[CPX];
#pragma unroll
            for (int k = 0; k < CPX; ++k) vh[k] = buf_load(rx, voh, (unsigned)(CPX * cg + k) * T4);
#pragma unroll
            for (int q = 0; q < CPX / 8; ++q) {
                u32x4_t u;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    u[e] = tvh ? pack2(vh[8 * q + 2 * e] + dsh[CPX * cg + 8 * q + 2 * e], vh[8 * q + 2 * e + 1] + dsh[CPX * cg + 8 * q + 2 * e + 1]) : 0u;
                *reinterpret_cast<u32x4_t *>(xs + j * XR + (CPX * cg + 8 * q) * 2) = u;
            }
        }
    }
    // per column block: frame of this lane; mT: all-ones inside the utterance (ANDed onto packed bf16 pairs); st: stored by this block
    bool st[NCB];
    unsigned vo4[NCB], mT[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int t = ts + cb * 32 + l31;
        mT[cb] = (t >= 0 && t < T) ? 0xffffffffu : 0u;
        st[cb] = t >= tv0 && t < tv1;
        vo4[cb] = 4u * (unsigned)(4 * half * T + min(max(t, 0), T - 1));
    }
    f32x16 accx[1][NCB];  // the wave's rows of x (fp32, the residual chain): GEMM 2's residual accumulators, live across the layers
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accx[0][cb][r] = buf_load(rx, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
    const unsigned bc0 = CS + (unsigned)(l31 * CR + half * 16);   // conditioner tile
    const unsigned bx0 = XS + (unsigned)(l31 * XR + half * 16);   // x tile, tap 0 (and the z tile)
    // byte offsets of the lane's 4-row groups in a bias vector: GEMM 1 block [16 gate ; 16 filter] rows, GEMM 2 32-row blocks
    const unsigned bb1 = BS + (unsigned)((row_g + 4 * half) * 4), bb2 = BS + 2048u + (unsigned)((row_g + 4 * half) * 4);

    const bool probe = g_bf16_phase_buf && blockIdx.x == 1 && blockIdx.y == 1;
    uint64_t tprev = 0, tph[6] = {0, 0, 0, 0, 0, 0};
#define LT_PHASE(i)                                                   \
    if (probe && m >= 1) {                                            \
        const uint64_t tn = __builtin_amdgcn_s_memtime();             \
        tph[i] += tn - tprev;                                         \
        tprev = tn;                                                   \
    }
    typedef unsigned lt_u32x2 __attribute__((ext_vector_type(2)));
    if (SKEW && w < 4) __builtin_amdgcn_s_setprio(1);
    // old skip rows per column block: blocks < NSKR live in registers for the whole group, the others travel through the private copy per layer
    f32x16 sko[NCB];
    u32x4_t A1[4 * D1][1], A2[4 * D2][1];
    gemm_t128_prefetch<1, 4 * D1>(A1, make_rsrc(reinterpret_cast<const unsigned short *>(a.img)), avo1, aoff1h(0));
    for (int m = 0; m < a.nl; ++m) {
        if (probe) tprev = __builtin_amdgcn_s_memtime();
        const int l = a.l0 + m, d = 1 << (l % a.dilation_cycle_length);
        const bool last = m == a.nl - 1;
        const unsigned short *img = reinterpret_cast<const unsigned short *>(a.img) + (int64_t)m * N_IMG;
        const rsrc_t rw1 = make_rsrc(img), rw2 = make_rsrc(img + OFF_W2B);
        const unsigned bpm = (unsigned)((m & 1) * 4096);
        // the next layer's biases: loaded now, written to the other LDS buffer after the next barrier pair
        float nb1 = 0.0f, nb2 = 0.0f;
        if (!last) { nb1 = a.b_dil[(m + 1) * 512 + tid] + a.b_cond[(m + 1) * 512 + tid]; nb2 = a.b_out[(m + 1) * 512 + tid]; }
        const unsigned dx = (unsigned)(d * XR);
        auto bf1 = [&](int kb, unsigned &b0, unsigned &bs) {
            const unsigned mc = (unsigned)((kb - KS_C) >> 31);  // all ones for the conditioner k-steps
            const unsigned kt = (unsigned)(kb - KS_C) & 63u, tap = kt >> 4, c0 = (kt & 15u) * 32u;
            b0 = (mc & (bc0 + (unsigned)kb * 32u)) | (~mc & (bx0 + tap * dx + c0));
            bs = (mc & (32u * CR)) | (~mc & (32u * XR));
        };
        unsigned zp[2][NCB][2][2];  // packed gated z of both passes: [h][cb][g2][2 words = 4 channels]
        lds_barrier();  // B1: the x tile of this layer (staged above / written by the previous layer's epilogue) is complete
        LT_PHASE(0)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // ---- GEMM 1, pass h: rows [16 gate ; 16 filter] x 128 frames, accumulators start at b_dil + b_cond
            f32x16 acc[1][NCB];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + bb1 + bpm + (unsigned)(((g4 >> 1) * FC + 16 * h + 8 * (g4 & 1)) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[0][cb][4 * g4 + e] = bv[e];
            }
            gemm_t128_run<1, KS1, D1, !SKEW, BDB1>(acc, A1, rw1, avo1, aoff1h(h), lds, bf1);
            if (h == 0) gemm_t128_prefetch<1, 4 * D1>(A1, rw1, avo1, aoff1h(1));  // pass 1's first fragments travel under the gate of pass 0
            // ---- gate of the pass (lane-local: gate row in register r < 8, its filter row in register r + 8), packed, kept in registers
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    float z[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = fsig(acc[0][cb][4 * g2 + e]) * ftanh(acc[0][cb][8 + 4 * g2 + e]);
                    zp[h][cb][g2][0] = pack2(z[0], z[1]) & mT[cb];
                    zp[h][cb][g2][1] = pack2(z[2], z[3]) & mT[cb];
                    pin(zp[h][cb][g2][0]); pin(zp[h][cb][g2][1]);
                    __builtin_amdgcn_sched_barrier(0);  // 4 chains in flight are enough to fill the VALU; 32 at once cost 23 spilled registers
                }
            __builtin_amdgcn_sched_barrier(0);  // (phase fences: the scheduler otherwise hoists the next phase's loads over this one's live values)
        }
        // GEMM 2 runs in two passes as well: skip rows first, residual rows second.  The old skip rows (from the skip tensor for the first
        // layer of the group unless it is the first of the network, from the private copy afterwards) and the first weight fragments are
        // requested here: they travel while this wave waits for the others and through the skip pass
        gemm_t128_prefetch<1, 4 * D2>(A2, rw2, lane16, aoff2r(1));
        const bool sk_zero = (SET_T128_EXP & 4) ? true : (a.first != 0 && m == 0);  // measurement build bit 2: no skip traffic between the layers
        if (m == 0 || (SET_T128_EXP & 4)) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sko[cb][r] = sk_zero ? 0.0f : buf_load(rsk, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
        } else {
#pragma unroll
            for (int cb = NSKR; cb < NCB; ++cb) {
                if (st[cb]) {  // the frames this block stores: the others never leave the tile
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 v = (SET_T128_EXP & 8) ? buf_load4(rps, lane16, (unsigned)((cb * 4 + g4) * 1024)) : buf_load4_stream(rps, lane16, (unsigned)((cb * 4 + g4) * 1024));
#pragma unroll
                        for (int e = 0; e < 4; ++e) sko[cb][4 * g4 + e] = v[e];
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        LT_PHASE(1)
        lds_barrier();  // B2: every wave has left GEMM 1: the z tile goes over the x tile
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    lt_u32x2 u;
                    u[0] = zp[h][cb][g2][0]; u[1] = zp[h][cb][g2][1];
                    *reinterpret_cast<lt_u32x2 *>(xs + (cb * 32 + l31) * XR + (row_g + 16 * h + 8 * g2 + 4 * half) * 2) = u;
                }
        if (!last) { bsh[((m + 1) & 1) * 1024 + tid] = nb1; bsh[((m + 1) & 1) * 1024 + 512 + tid] = nb2; }
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();  // B3: the z tile is complete
        LT_PHASE(2)

        auto bf2 = [&](int kb, unsigned &b0, unsigned &bs) { b0 = bx0 + (unsigned)kb * 32u; bs = 32u * XR; };
        // ---- GEMM 2, skip pass: o_skip = Wout[skip rows] z + b_out; new skip rows = o_skip + old ones -> private copy / skip tensor
        {
            f32x16 acc[1][NCB];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + bb2 + bpm + (unsigned)((FC + 8 * g4) * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[0][cb][4 * g4 + e] = bv[e];
            }
            gemm_t128_run<1, KS2, D2, !SKEW, BDB2>(acc, A2, rw2, lane16, aoff2r(1), lds, bf2);
            gemm_t128_prefetch<1, 4 * D2>(A2, rw2, lane16, aoff2r(0));
            if (!last) {
#pragma unroll
                for (int cb = 0; cb < NSKR; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sko[cb][r] = acc[0][cb][r] + sko[cb][r];
                if (!(SET_T128_EXP & 4)) {
#pragma unroll
                    for (int cb = NSKR; cb < NCB; ++cb) {
                        if (st[cb]) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = acc[0][cb][4 * g4 + e] + sko[cb][4 * g4 + e];
                                if (SET_T128_EXP & 8) buf_store4(v, rps, lane16, (unsigned)((cb * 4 + g4) * 1024)); else buf_store4_stream(v, rps, lane16, (unsigned)((cb * 4 + g4) * 1024));
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) {
                    if (st[cb]) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) buf_store(acc[0][cb][r] + sko[cb][r], rsk, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- GEMM 2, residual pass: accumulators = the x rows + b_out
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(lds + bb2 + bpm + (unsigned)((8 * g4) * 4));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) accx[0][cb][4 * g4 + e] += bv[e];
        }
        gemm_t128_run<1, KS2, D2, !SKEW, BDB2>(accx, A2, rw2, lane16, aoff2r(0), lds, bf2);
        LT_PHASE(3)

        // ---- epilogue: x' = (x + o_res) / sqrt 2 stays in accx; the next layer's first weight fragments; bf16(x' + d_next) packed in
        //      registers and written to the x tile once every wave has left GEMM 2
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[0][cb][r] *= RSQRT2;
        if (!last) {
            gemm_t128_prefetch<1, 4 * D1>(A1, make_rsrc(img + N_IMG), avo1, aoff1h(0));  // the next layer's first fragments
            const int dn = 1 << ((l + 1) % a.dilation_cycle_length);
            const float *dnx = dsh + (m + 1) * FC;
            unsigned xp[NCB][4][2];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 dv = *reinterpret_cast<const f32x4 *>(dnx + row_g + 8 * g4 + 4 * half);
                    xp[cb][g4][0] = pack2(accx[0][cb][4 * g4] + dv[0], accx[0][cb][4 * g4 + 1] + dv[1]) & mT[cb];
                    xp[cb][g4][1] = pack2(accx[0][cb][4 * g4 + 2] + dv[2], accx[0][cb][4 * g4 + 3] + dv[3]) & mT[cb];
                    pin(xp[cb][g4][0]); pin(xp[cb][g4][1]);
                }
            lds_barrier();  // B4: every wave has left GEMM 2: the next x tile goes over the z tile
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    lt_u32x2 u;
                    u[0] = xp[cb][g4][0]; u[1] = xp[cb][g4][1];
                    *reinterpret_cast<lt_u32x2 *>(xs + (dn + cb * 32 + l31) * XR + (row_g + 8 * g4 + 4 * half) * 2) = u;
                }
        } else {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (st[cb]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) buf_store(accx[0][cb][r], rxo, vo4[cb], (unsigned)(row_g + urow(r)) * T4);
                }
            }
        }
        LT_PHASE(4)
    }
#undef LT_PHASE
    if (probe && (tid == 0 || tid == 256)) {
        uint64_t *pb = g_bf16_phase_buf + (tid ? 8 : 0);
        for (int i = 0; i < 5; ++i) pb[i] += tph[i];
        pb[7] += (uint64_t)(a.nl - 1);
    }
}

// =====================================================================================================================
// backward.  Tile = 128 frames [ts, ts + 128), ts = tile * (128 - 2 dil) - dil: dz / dy are computed for all 128 frames,
// dx / dcond / the stored dy, d_o and the bias partial sums for the central 128 - 2 dil (the halo frames are recomputed by
// the neighbouring tiles).  LDS: one [128 + 2 dil][512] bf16 tile (tile row j lives in LDS row j + dil, so the +-dil
// row shifts of the transposed taps stay inside the allocation): first d_o, then dy over it.
// =====================================================================================================================
__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes of a half-wave (xor offsets < 32)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// NT = frames per tile: 128 is what ships (one block per CU, 133 KB of LDS).  Measured in round 5 and not instantiated (profiles/r05_layer_bwd_probe.log):
// NT = 64 as two blocks per CU (68 KB of LDS each, 120 registers, no spills) runs the same 86 - 88 us per launch -- the kernel is not
// waiting for its own phases; builds without its 2-byte memory instructions (the bf16 d_o / dy stores, the y loads: 384 per lane) run 65 us,
// and even that moves the training step by 0.15 ms only.
template <int NT>
__global__ void __launch_bounds__(512, NT == 128 ? 1 : 2) diffnet_layer_bwd_bf16_kernel(SetDiffnetLayerBf16BwdArgs a) {
    constexpr int NCB = NT / 32;       // 32-frame column blocks
    constexpr int NCG = 512 / NT;      // channel groups of the staging pass (thread = frame x channel group)
    constexpr int CPG = 2 * FC / NCG;  // channels of d_o per staging thread
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, T = a.T, d = a.dil;
    const int NTc = NT - 2 * d;
    const int ts = blockIdx.x * NTc - d;
    const int part_row = b * gridDim.x + blockIdx.x;
    const unsigned T4 = 4u * (unsigned)T, T2 = 2u * (unsigned)T;
    const bool has_dxo = a.dx_out != nullptr;
    const rsrc_t rdxo = make_rsrc(has_dxo ? a.dx_out + (int64_t)b * FC * T : a.dskip + (int64_t)b * FC * T);
    const rsrc_t rdsk = make_rsrc(a.dskip + (int64_t)b * FC * T);
    const rsrc_t ry = make_rsrc(a.y16 + (int64_t)b * 2 * FC * T);
    const rsrc_t rdx = make_rsrc(a.dx + (int64_t)b * FC * T);
    const rsrc_t rdy = make_rsrc(a.dy16 + (int64_t)b * 2 * FC * T), rdo = make_rsrc(a.do16 + (int64_t)b * 2 * FC * T);
    const rsrc_t rdc = make_rsrc(a.dcond + (int64_t)b * FH * T);
    const unsigned short *img = reinterpret_cast<const unsigned short *>(a.img);
    const rsrc_t rwt2 = make_rsrc(img + OFF_WT2 + (int64_t)w * 32 * 512);
    const rsrc_t rwt1 = make_rsrc(img + OFF_WT1 + (int64_t)w * 96 * 512);
    const rsrc_t rwtc = make_rsrc(img + OFF_WTC + (int64_t)min(w, 5) * 32 * 512);
    const unsigned lane16 = 16u * (unsigned)lane;

    // ---- stage d_o = [dx_out / sqrt2 ; dskip] as [frame][512] bf16 (+ its bf16 copy in HBM for the weight gradient)
    {
        const int f = tid % NT, cg = __builtin_amdgcn_readfirstlane(tid / NT);
        const int t = ts + f;
        const bool tv = t >= 0 && t < T;
        const bool central = tv && f >= d && f < NT - d;
        const unsigned tc = (unsigned)min(max(t, 0), T - 1);
        const bool from_dx = cg < NCG / 2;  // the first 256 channels of d_o come from dx_out / sqrt2, the other 256 from dskip
        const rsrc_t rs = from_dx ? rdxo : rdsk;
        const float sc = from_dx ? (has_dxo ? RSQRT2 : 0.0f) : 1.0f;
        const int cs0 = CPG * (cg % (NCG / 2));  // channel inside the source tensor
        // all loads of the thread in flight at once (nothing else is live yet): one memory round trip
        float v[CPG];
#pragma unroll
        for (int k = 0; k < CPG; ++k) v[k] = buf_load(rs, 4u * tc, (unsigned)(cs0 + k) * T4) * sc;
#pragma unroll
        for (int q = 0; q < CPG / 8; ++q) {
            u32x4_t u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = tv ? pack2(v[8 * q + 2 * e], v[8 * q + 2 * e + 1]) : 0u;
            *reinterpret_cast<u32x4_t *>(lds + (f + d) * DR + (CPG * cg + 8 * q) * 2) = u;
        }
        if (central) {  // (quad-interleaved: channels CPG cg + 4 j .. + 3 of frame tc as one 8-byte store)
#pragma unroll
            for (int j = 0; j < CPG / 4; ++j) {
                u32x2_t u;
                u[0] = pack2(v[4 * j], v[4 * j + 1]); u[1] = pack2(v[4 * j + 2], v[4 * j + 3]);
                buf_store_q4(u, rdo, 8u * tc, (unsigned)(CPG * cg + 4 * j) * T2);
            }
        }
    }
    __syncthreads();
    // ---- bias gradient of the output projection: column sums of the central rows of the d_o tile (thread = channel)
    {
        float s = 0.0f;
        for (int j = d; j < NT - d; ++j) s += bf2f(*reinterpret_cast<const unsigned short *>(lds + (j + d) * DR + tid * 2));
        a.part_dbo[(int64_t)part_row * 2 * FC + tid] = s;  // rows outside [0, T) were staged as zeros
    }

    // ---- GEMM A: dz[256 x 128] = Wout^T d_o
    f32x16 dz[1][NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dz[0][cb] = (f32x16){0};
    gemm_bf16<1, NCB>(dz, rwt2, lane16, 0, 32, lds, [&](int ks, int cb) {
        return (unsigned)((cb * 32 + l31 + d) * DR + (ks * 16 + half * 8) * 2);
    });

    // ---- gate derivative (lane-local), dy over the tile, dy to HBM, bias partial sums
    bool cen[NCB];
    unsigned vo4[NCB], vo2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int j = cb * 32 + l31, t = ts + j;
        cen[cb] = t >= 0 && t < T && j >= d && j < NT - d;
        const int tc = min(max(t, 0), T - 1);
        vo4[cb] = 4u * (unsigned)(4 * half * T + tc);
        vo2[cb] = 8u * (unsigned)(half * T + tc);  // quad-interleaved bf16
    }
    __syncthreads();  // every wave is done reading the d_o tile
    {
        float sg[16], sf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sg[r] = sf[r] = 0.0f;
        // the saved pre-gate values of all four column blocks are fetched up front (one round trip, not four)
        u32x2_t yg[NCB][4], yf[NCB][4];  // [column block][register group of 4 = channel quad]
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const unsigned so = (unsigned)(32 * w + 8 * g4) * T2;
                yg[cb][g4] = buf_load_q4(ry, vo2[cb], so);
                yf[cb][g4] = buf_load_q4(ry, vo2[cb], so + (unsigned)FC * T2);
            }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                unsigned short bg[4], bfv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    const unsigned wg = yg[cb][g4][e >> 1], wf = yf[cb][g4][e >> 1];
                    const unsigned short hg = (unsigned short)((e & 1) ? (wg >> 16) : (wg & 0xffffu)), hf = (unsigned short)((e & 1) ? (wf >> 16) : (wf & 0xffffu));
                    const float s = fsig(bf2f(hg)), th = ftanh(bf2f(hf));
                    const float g = dz[0][cb][r];  // exactly 0 on frames outside [0, T): d_o was staged as zeros there
                    const float dg = g * th * s * (1.0f - s), df = g * s * (1.0f - th * th);
                    bg[e] = f2bf(dg); bfv[e] = f2bf(df);
                    if (cen[cb]) {
                        sg[r] += bf2f(bg[e]);
                        sf[r] += bf2f(bfv[e]);
                    }
                }
                u32x2_t qg, qf;
                qg[0] = (unsigned)bg[0] | ((unsigned)bg[1] << 16); qg[1] = (unsigned)bg[2] | ((unsigned)bg[3] << 16);
                qf[0] = (unsigned)bfv[0] | ((unsigned)bfv[1] << 16); qf[1] = (unsigned)bfv[2] | ((unsigned)bfv[3] << 16);
                unsigned char *row = lds + (cb * 32 + l31 + d) * DR + (32 * w + 8 * g4 + 4 * half) * 2;
                *reinterpret_cast<u32x2_t *>(row) = qg;
                *reinterpret_cast<u32x2_t *>(row + FC * 2) = qf;
                if (cen[cb]) {
                    const unsigned so = (unsigned)(32 * w + 8 * g4) * T2;
                    buf_store_q4(qg, rdy, vo2[cb], so);
                    buf_store_q4(qf, rdy, vo2[cb], so + (unsigned)FC * T2);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a0 = half_sum(sg[r]), a1 = half_sum(sf[r]);
            if (l31 == 0) {
                const int c = 32 * w + urow(r) + 4 * half;
                a.part_dby[(int64_t)part_row * 2 * FC + c] = a0;
                a.part_dby[(int64_t)part_row * 2 * FC + FC + c] = a1;
            }
        }
    }
    __syncthreads();

    // ---- GEMM B: dxd[256 x 128] = sum_tap Wdil[tap]^T dy shifted by (1 - tap) dil ;  GEMM C (waves 0..5): dcond = Wcond^T dy
    f32x16 dxd[1][NCB], dcn[1][NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) { dxd[0][cb] = (f32x16){0}; dcn[0][cb] = (f32x16){0}; }
    gemm_bf16<1, NCB>(dxd, rwt1, lane16, 0, 96, lds, [&](int ks, int cb) {
        const int tap = ks >> 5, c0 = (ks & 31) * 16;
        return (unsigned)((cb * 32 + l31 + (2 - tap) * d) * DR + (c0 + half * 8) * 2);
    });
    if (w < 6)
        gemm_bf16<1, NCB>(dcn, rwtc, lane16, 0, 32, lds, [&](int ks, int cb) {
            return (unsigned)((cb * 32 + l31 + d) * DR + (ks * 16 + half * 8) * 2);
        });

    // ---- epilogue: dx = dx_out / sqrt2 + dxd ; step-embedding partial sums ; dcond (+)= dcn
    {
        float sd[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) sd[r] = 0.0f;
        // The read-modify-write batches are software-pipelined: the loads of column block cb + 1 are issued BEFORE the stores of cb.
        // vmcnt retires loads and stores in issue order, so a load issued after a batch of stores cannot be waited for without
        // waiting for those stores (round 3's `load 16 -> store 16` per column block paid a load AND a store round trip, 8 times).
        float rv[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[0][r] = buf_load(rdxo, vo4[0], (unsigned)(32 * w + urow(r)) * T4);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (cb + 1 < NCB) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[(cb + 1) & 1][r] = buf_load(rdxo, vo4[cb + 1], (unsigned)(32 * w + urow(r)) * T4);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (cen[cb]) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = dxd[0][cb][r];
                    sd[r] += v;
                    buf_store(has_dxo ? v + rv[cb & 1][r] * RSQRT2 : v, rdx, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a0 = half_sum(sd[r]);
            if (l31 == 0) a.part_dd[(int64_t)part_row * FC + 32 * w + urow(r) + 4 * half] = a0;
        }
        if (w < 6) {
            float pv[2][16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[0][r] = buf_load(rdc, vo4[0], (unsigned)(32 * w + urow(r)) * T4);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                if (cb + 1 < NCB) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[(cb + 1) & 1][r] = buf_load(rdc, vo4[cb + 1], (unsigned)(32 * w + urow(r)) * T4);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (cen[cb]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        buf_store(a.dcond_first ? dcn[0][cb][r] : pv[cb & 1][r] + dcn[0][cb][r], rdc, vo4[cb], (unsigned)(32 * w + urow(r)) * T4);
                }
            }
        }
    }
}

// out[g][j] (+)= scale * sum_r part[(g * rows + r) * cols + j].  Block = 64 columns x ROWS_RG row groups, one fixed association
// (rows_sum.h): deterministic.
__global__ void __launch_bounds__(64 * ROWS_RG) partial_rows_sum_kernel(const float *part, float *out, int rows, int cols,
                                                                        int accumulate, float scale) {
    __shared__ float red[ROWS_RG][64];
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tl, g = blockIdx.y;
    float s = j < cols ? rows_sum_chains(part + (int64_t)g * rows * cols + j, cols, rg, rows) : 0.0f;
    s = rows_sum_groups(s, red, rg, tl) * scale;
    if (rg == 0 && j < cols) {
        float *o = out + (int64_t)g * cols + j;
        *o = accumulate ? *o + s : s;
    }
}

// all ordered partial sums of one layer's backward in ONE launch (block = 64 columns x ROWS_RG row groups, fixed association):
//   columns [0,512): db_out += sum over all rows of part_dbo ; [512,1024): db_dil, db_cond += ... of part_dby ;
//   [1024,1280): dd[b][c] = sum over the tiles of utterance b of part_dd   (blockIdx.y = b there)
// blockIdx.z = layer q of a stack swept with one tile count: partials of layer q at q * (their per-layer size), targets at q * their
// element strides (s_out / s_dil / s_cond between the layers' bias gradients, dd_ls between the layers' step-offset columns)
__global__ void __launch_bounds__(64 * ROWS_RG) layer_bwd_reduce_kernel(const float *pdbo, const float *pdby, const float *pdd, int B,
                                                                        int tiles, float *db_out, float *db_dil, float *db_cond,
                                                                        float *dd, int64_t dd_bs, int64_t s_out, int64_t s_dil,
                                                                        int64_t s_cond, int64_t dd_ls) {
    __shared__ float red[ROWS_RG][64];
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int cbk = blockIdx.x;  // 0..7 dbo, 8..15 dby, 16..19 dd
    const int64_t q = blockIdx.z;
    pdbo += q * B * tiles * 2 * FC; pdby += q * B * tiles * 2 * FC; pdd += q * B * tiles * FC;
    db_out += q * s_out; db_dil += q * s_dil; db_cond += q * s_cond; dd += q * dd_ls;
    if (cbk < 16) {
        if (blockIdx.y != 0) return;
        const float *part = cbk < 8 ? pdbo : pdby;
        const int j = (cbk & 7) * 64 + tl;
        const float s = rows_sum_groups(rows_sum_chains(part + j, 2 * FC, rg, B * tiles), red, rg, tl);
        if (rg == 0) {
            if (cbk < 8) db_out[j] += s;
            else { db_dil[j] += s; db_cond[j] += s; }
        }
    } else {
        const int b = blockIdx.y, j = (cbk - 16) * 64 + tl;
        const float s = rows_sum_groups(rows_sum_chains(pdd + (int64_t)b * tiles * FC + j, FC, rg, tiles), red, rg, tl);
        if (rg == 0) dd[(int64_t)b * dd_bs + j] = s;
    }
}

}  // namespace

extern "C" int64_t set_diffnet_layer_bf16_image_size(void) { return N_IMG; }

extern "C" int set_pack_diffnet_layer_bf16(const float *wdil, const float *wcond, const float *wout, void *img, void *stream) {
    SET_REQUIRE(wdil && wcond && wout && img, "set_pack_diffnet_layer_bf16");
    hipLaunchKernelGGL(pack_layer_bf16_kernel, dim3(set_blocks(N_IMG, 256)), dim3(256), 0, (hipStream_t)stream, wdil, wcond, wout,
                       reinterpret_cast<unsigned short *>(img), (int64_t)0, (int64_t)0, (int64_t)0);
    return set_check_launch("set_pack_diffnet_layer_bf16");
}

extern "C" int set_pack_diffnet_layers_bf16(const float *wdil, const float *wcond, const float *wout, int64_t sdil, int64_t scond,
                                            int64_t sout, void *img, int32_t L, void *stream) {
    SET_REQUIRE(wdil && wcond && wout && img && L >= 1 && L <= 65535, "set_pack_diffnet_layers_bf16");
    hipLaunchKernelGGL(pack_layer_bf16_kernel, dim3(set_blocks(N_IMG, 256), L), dim3(256), 0, (hipStream_t)stream, wdil, wcond, wout,
                       reinterpret_cast<unsigned short *>(img), sdil, scond, sout);
    return set_check_launch("set_pack_diffnet_layers_bf16");
}

extern "C" int64_t set_sizeof_diffnet_layer_bf16_args(void) { return (int64_t)sizeof(SetDiffnetLayerBf16Args); }

extern "C" int set_diffnet_layer_fwd_bf16(const SetDiffnetLayerBf16Args *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffnet_layer_fwd_bf16");
    const SetDiffnetLayerBf16Args &a = *args;
    SET_REQUIRE(a.x_in && a.x_out && a.skip && a.cond && a.dstep && a.img && a.b_dil && a.b_cond && a.b_out &&
                    ((a.y16 != nullptr) == (a.z16 != nullptr)), "set_diffnet_layer_fwd_bf16");
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.dil >= 1 && a.dil <= 8, "set_diffnet_layer_fwd_bf16");
    SET_REQUIRE((int64_t)2 * FC * a.T * 4 < ((int64_t)1 << 31), "set_diffnet_layer_fwd_bf16 (T too large)");
    // tile: 128 frames, 8 waves, one block per CU (round 3's 64-frame / two-blocks-per-CU experiment spilled registers, measured
    // slower, and is gone)
    constexpr int tile = 128;
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_layer_fwd_bf16_kernel<true, 128>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "layer fwd bf16 attr");
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_layer_fwd_bf16_kernel<false, 128>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "layer fwd bf16 attr");
        attr_set = true;
    }
    const size_t ldsz = (size_t)(tile + 2 * a.dil) * XR + (size_t)tile * CR + 5 * FC * sizeof(float);  // tiles, step offsets, the two bias vectors
    dim3 grid((a.T + tile - 1) / tile, a.B);
    hipStream_t st = (hipStream_t)stream;
    if (a.y16) hipLaunchKernelGGL((diffnet_layer_fwd_bf16_kernel<true, 128>), grid, dim3(512), ldsz, st, a);
    else hipLaunchKernelGGL((diffnet_layer_fwd_bf16_kernel<false, 128>), grid, dim3(512), ldsz, st, a);
    return set_check_launch("set_diffnet_layer_fwd_bf16");
}

static int layers_halo(int l0, int nl, int dcl) {
    int h = 0;
    for (int m = 1; m < nl; ++m) h += 1 << ((l0 + m) % dcl);
    return h;
}
// test switch, read at every launch: SET_AMD_BF16_FUSE_TILE = 64 | 128 forces the tile width (default: by occupancy, layers_pick_tile)
static int layers_tile_env() { const char *e = getenv("SET_AMD_BF16_FUSE_TILE"); const int t = e ? atoi(e) : 0; return (t == 64 || t == 128) ? t : 0; }

extern "C" int64_t set_sizeof_diffnet_layers_bf16_args(void) { return (int64_t)sizeof(SetDiffnetLayersBf16Args); }

static int layers_cu_count() {
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t pr; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
    return n_cu;
}
// tile width for a group: 128 frames (half the weight bytes per MFMA, 4 barriers per layer) when that still fills 3/4 of the chip,
// else 64 (twice the blocks); a halo that leaves fewer than 32 of 64 frames: 128
static int layers_pick_tile(int B, int T, int hh) {
    int tile = layers_tile_env();
    if (!tile) {
        tile = 64;
        if (128 - 2 * hh >= 32) {
            const int64_t tiles128 = (int64_t)B * ((T + (128 - 2 * hh) - 1) / (128 - 2 * hh));
            if (tiles128 * 4 >= (int64_t)layers_cu_count() * 3) tile = 128;
        }
    }
    if (tile - 2 * hh < 32) tile = 128;
    return tile;
}
// layers per launch for a stack of L layers (the bf16 reverse loop asks): 10 where 128-frame tiles fill the chip (B = 32, T = 800: 8 tiles
// of 100 stored frames per utterance = 256 blocks), else 5 (64-frame tiles: 56 of 64 computed frames stored)
extern "C" int32_t set_diffnet_layers_bf16_plan(int32_t B, int32_t T, int32_t L, int32_t dcl) {
    if (B < 1 || T < 1 || L < 1 || dcl < 1 || dcl > 2) return 1;
    const int n10 = L < 10 ? L : 10;
    const int h10 = layers_halo(0, n10, dcl);
    if (128 - 2 * h10 >= 32 && layers_pick_tile(B, T, h10) == 128) return n10;
    return L < 5 ? L : 5;
}

// balanced tiles: the same number of tiles per utterance as width - 2 H stored frames would give, equal stored widths
static int layers_balanced_nv(int T, int tile, int hh) { const int nv = tile - 2 * hh, nt = (T + nv - 1) / nv; return (T + nt - 1) / nt; }

extern "C" int64_t set_diffnet_layers_bf16_scratch_floats(int32_t B, int32_t T, int32_t l0, int32_t nl, int32_t dcl) {
    // 128-frame tiles: a block's private copy of its skip rows (256 x 128 fp32) between the layers of a group; an upper bound over the
    // first layers l0 of the groups of a network (so that one buffer serves every group); 64-frame tiles keep everything in registers
    if (B < 1 || T < 1 || nl < 1 || dcl < 1 || l0 < 0) return 0;
    int hmax = 0;
    for (int s0 = 0; s0 < dcl; ++s0) { const int h = layers_halo(s0, nl, dcl); hmax = h > hmax ? h : hmax; }
    if (128 - 2 * hmax < 32) return 64;
    const int nv = layers_balanced_nv(T, 128, hmax);
    return (int64_t)B * ((T + nv - 1) / nv) * FC * 128;
}

extern "C" int set_diffnet_layers_fwd_bf16(const SetDiffnetLayersBf16Args *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffnet_layers_fwd_bf16");
    LayersArgs la;
    la.a = *args;
    const SetDiffnetLayersBf16Args &a = la.a;
    SET_REQUIRE(a.x_in && a.x_out && a.x_in != a.x_out && a.skip && a.cond && a.dstep && a.img && a.b_dil && a.b_cond && a.b_out && a.scratch,
                "set_diffnet_layers_fwd_bf16");
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.nl >= 1 && a.nl <= 16 && a.l0 >= 0 && a.dilation_cycle_length >= 1 && a.dilation_cycle_length <= 4,
                "set_diffnet_layers_fwd_bf16");
    SET_REQUIRE((int64_t)2 * FC * a.T * 4 < ((int64_t)1 << 31), "set_diffnet_layers_fwd_bf16 (T too large)");
    la.hh = layers_halo(a.l0, a.nl, a.dilation_cycle_length);
    const int tile = layers_pick_tile(a.B, a.T, la.hh);
    la.nv = tile - 2 * la.hh;
    if (la.nv < 32) return set_fail(SET_E_UNSUPPORTED, "set_diffnet_layers_fwd_bf16", "halo of the fused layers leaves fewer than 32 valid frames per tile");
    la.nv = layers_balanced_nv(a.T, tile, la.hh);
    if (tile == 128) SET_REQUIRE(a.scratch_floats >= (int64_t)a.B * ((a.T + la.nv - 1) / la.nv) * FC * 128, "set_diffnet_layers_fwd_bf16 (scratch too small)");
    int dmax = 1;
    for (int m = 0; m < a.nl; ++m) { const int d = 1 << ((a.l0 + m) % a.dilation_cycle_length); dmax = d > dmax ? d : dmax; }
    // 64: x tile | conditioner tile | z tile | step offsets | biases of the group;  128: x tile (z over it) | conditioner | step offsets | 2 bias buffers
    const size_t ldsz = tile == 64 ? (size_t)(64 + 2 * dmax) * XR + (size_t)64 * CR + (size_t)64 * XR + (size_t)a.nl * (FC + 1024) * sizeof(float)
                                   : (size_t)(128 + 2 * dmax) * XR + (size_t)128 * CR + (size_t)a.nl * FC * sizeof(float) + 2 * 1024 * sizeof(float);
    if (ldsz > 160 * 1024) return set_fail(SET_E_UNSUPPORTED, "set_diffnet_layers_fwd_bf16", "tiles do not fit LDS");
    static bool attr_set = false;
    if (!attr_set) {
        const void *ks[] = {reinterpret_cast<const void *>(diffnet_layers_reg_bf16_kernel<false, 4>),
                            reinterpret_cast<const void *>(diffnet_layers_t128_bf16_kernel<false, 1>)};
        for (const void *k : ks) SET_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "layers bf16 attr");
        attr_set = true;
    }
    dim3 grid((a.T + la.nv - 1) / la.nv, a.B);
    // Shipped instantiations only.  Measured in round 4 and left out (B = 32, T = 800, sustained us per layer; profiles/r04_bf16_ab.log):
    // 128-frame shape with 0 / 1 / 2 / 3 / 4 column blocks of the skip rows in registers 40.7 / 39.4 / 39.6 / 39.8 / 41.8 (21 / 45 / 67
    // spilled registers from 2 on); 64-frame shape with a static priority for one half of the waves instead of the k-step toggles and / or
    // a ring of 8 k-steps: within 0.5 % of the shipped one.
    if (tile == 128) hipLaunchKernelGGL((diffnet_layers_t128_bf16_kernel<false, 1>), grid, dim3(512), ldsz, (hipStream_t)stream, la);
    else hipLaunchKernelGGL((diffnet_layers_reg_bf16_kernel<false, 4>), grid, dim3(512), ldsz, (hipStream_t)stream, la);
    return set_check_launch("set_diffnet_layers_fwd_bf16");
}

extern "C" int64_t set_sizeof_diffnet_layer_bf16_bwd_args(void) { return (int64_t)sizeof(SetDiffnetLayerBf16BwdArgs); }

static int layer_bwd_tile() { return FNT; }
extern "C" int32_t set_diffnet_layer_bwd_bf16_tiles(int32_t T, int32_t dil) {
    const int nt = layer_bwd_tile();
    return (T + (nt - 2 * dil) - 1) / (nt - 2 * dil);
}

extern "C" int set_diffnet_layer_bwd_bf16(const SetDiffnetLayerBf16BwdArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffnet_layer_bwd_bf16");
    const SetDiffnetLayerBf16BwdArgs &a = *args;
    SET_REQUIRE(a.dskip && a.y16 && a.img && a.dx && a.dy16 && a.do16 && a.dcond && a.part_dbo && a.part_dby && a.part_dd,
                "set_diffnet_layer_bwd_bf16");
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.dil >= 1 && a.dil <= 8, "set_diffnet_layer_bwd_bf16");
    SET_REQUIRE((int64_t)2 * FC * a.T * 4 < ((int64_t)1 << 31), "set_diffnet_layer_bwd_bf16 (T too large)");
    const int nt = layer_bwd_tile();
    SET_REQUIRE(nt - 2 * a.dil >= 32, "set_diffnet_layer_bwd_bf16 (dilation too large for the tile)");
    const size_t ldsz = (size_t)(nt + 2 * a.dil) * DR;
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_layer_bwd_bf16_kernel<128>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), "layer bwd bf16 attr");
        attr_set = true;
    }
    dim3 grid(set_diffnet_layer_bwd_bf16_tiles(a.T, a.dil), a.B);
    hipLaunchKernelGGL(diffnet_layer_bwd_bf16_kernel<128>, grid, dim3(512), ldsz, (hipStream_t)stream, a);
    return set_check_launch("set_diffnet_layer_bwd_bf16");
}

extern "C" int set_partial_rows_sum(const float *part, float *out, int32_t groups, int32_t rows, int32_t cols, int32_t accumulate,
                                    float scale, void *stream) {
    SET_REQUIRE(part && out && groups > 0 && rows > 0 && cols > 0, "set_partial_rows_sum");
    hipLaunchKernelGGL(partial_rows_sum_kernel, dim3((cols + 63) / 64, groups), dim3(64 * ROWS_RG), 0, (hipStream_t)stream, part, out, rows,
                       cols, accumulate, scale);
    return set_check_launch("set_partial_rows_sum");
}

extern "C" int set_diffnet_layer_bwd_reduce(const float *part_dbo, const float *part_dby, const float *part_dd, int32_t B,
                                            int32_t tiles, float *db_out, float *db_dil, float *db_cond, float *dd, int64_t dd_bs,
                                            void *stream) {
    SET_REQUIRE(part_dbo && part_dby && part_dd && db_out && db_dil && db_cond && dd && B > 0 && tiles > 0,
                "set_diffnet_layer_bwd_reduce");
    hipLaunchKernelGGL(layer_bwd_reduce_kernel, dim3(20, B), dim3(64 * ROWS_RG), 0, (hipStream_t)stream, part_dbo, part_dby, part_dd, B,
                       tiles, db_out, db_dil, db_cond, dd, dd_bs, (int64_t)0, (int64_t)0, (int64_t)0, (int64_t)0);
    return set_check_launch("set_diffnet_layer_bwd_reduce");
}

extern "C" int set_diffnet_layers_bwd_reduce(const float *part_dbo, const float *part_dby, const float *part_dd, int32_t B, int32_t tiles,
                                             int32_t L, float *db_out, int64_t s_out, float *db_dil, int64_t s_dil, float *db_cond,
                                             int64_t s_cond, float *dd, int64_t dd_bs, int64_t dd_ls, void *stream) {
    SET_REQUIRE(part_dbo && part_dby && part_dd && db_out && db_dil && db_cond && dd && B > 0 && tiles > 0 && L >= 1 && L <= 65535,
                "set_diffnet_layers_bwd_reduce");
    hipLaunchKernelGGL(layer_bwd_reduce_kernel, dim3(20, B, L), dim3(64 * ROWS_RG), 0, (hipStream_t)stream, part_dbo, part_dby, part_dd, B,
                       tiles, db_out, db_dil, db_cond, dd, dd_bs, s_out, s_dil, s_cond, dd_ls);
    return set_check_launch("set_diffnet_layers_bwd_reduce");
}

// debug hook (tools/bf16_phase_probe.py): block (1, 1), thread 0 of the layer kernels stores s_memtime at its phase
// boundaries into buf[0..7]; NULL switches it off.  Not part of the product path.
extern "C" int set_debug_bf16_phase_buffer(uint64_t *buf) {
    SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bf16_phase_buf), &buf, sizeof(buf)), "set_debug_bf16_phase_buffer");
    return SET_OK;
}
