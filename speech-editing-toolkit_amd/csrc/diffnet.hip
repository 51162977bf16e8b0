// DiffNet (FluentSpeech denoiser) kernels and the reverse-diffusion loop for gfx950.
//
// Hot kernel: diffnet_layer_kernel -- ONE launch per residual layer (diffnet.py:60-81), fusing
//   x+d  ->  k=3 dilated conv (implicit GEMM, 512x768)  -> +bias +hoisted conditioner projection
//        ->  sigmoid*tanh gate  ->  1x1 conv (GEMM 512x256)  ->  residual/sqrt(2) + skip accumulate.
// fp32 end to end on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain): the parity bar is |dmel| < 1e-4
// against the fp32 reference, which a bf16 path cannot meet.
//
// Work decomposition (64-wide waves, 256 CUs):
//   grid  = (ceil(T/64), B); block = 256 threads = 4 waves; block tile = all 512 rows x 64 frames.
//   wave w owns gate rows [64w,64w+64) and the matching filter rows [256+64w, 256+64w+64) so the
//   gate is lane-local in the accumulator layout; same split for residual/skip rows of GEMM 2.
//   8 accumulators (4 row blocks x 2 col blocks) of 32x32 per wave = 128 VGPRs -> 2 blocks per CU.
//   B operand (activations): LDS tile xs[256][64 + 2*dil], lanes read 32 consecutive frames ->
//   conflict free; the k=3 taps are column shifts of the same tile (im2col is free).
//   A operand (weights): pre-packed in fragment order so one k-step of a wave is a single 1 KiB
//   coalesced global_load_dwordx4 (L2 resident: 2 MiB per layer); prefetched 4 k-steps ahead.
//   z (gate output) overwrites the xs tile in LDS; x for the residual is re-read from L2.
#include <stdlib.h>

#include "common.h"
#include "boundary_x2.h"

namespace {

constexpr int DC = 256;       // residual_channels this kernel is specialised for
constexpr int NT = 64;        // frames per block tile
constexpr int KS1 = 3 * DC / 2;  // 384 k-steps (K=768) of GEMM 1
constexpr int KS2 = DC / 2;      // 128 k-steps (K=256) of GEMM 2

// Accumulator register r of the 32x32 block holds row  urow(r) + 4*(lane>>5):  the first term is wave-uniform, so
// every global access below is  (uniform row pointer, SGPR) + (one per-lane 32-bit offset, VGPR)  -- no per-row
// 64-bit address registers (those spilled and serialised the epilogue stores behind vmcnt(0) reloads).
__device__ __forceinline__ int urow16(int r) { return (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ int layer_row(int w, int rb, int i) {
    // rb 0,1 -> first half rows (gate / residual); rb 2,3 -> second half (filter / skip)
    return (rb < 2 ? 64 * w + 32 * rb : DC + 64 * w + 32 * (rb - 2)) + i;
}

// fast gate math: v_exp_f32 / v_rcp_f32 (<= 1-2 ulp each); absolute error ~1e-7, far inside the 1e-4 mel bar
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// One (layer, 32*NCB-frame tile) unit of work; shared by the per-layer kernel (NCB = 2) and the persistent stack
// kernel (NCB = 1 or 2).
struct LayerTile {
    const float *xin;      // x_in  + b*256*T
    float *xout;           // x_out + b*256*T
    float *skp;            // skip  + b*256*T
    const float *cpb;      // condproj slab of this layer + b*cp_bs
    const float *dstep;    // d[c] = dstep[c*d_cs]  (already offset by b*d_bs)
    int64_t d_cs;
    const float *w1p, *b_dil, *w2p, *b_out;
    int T, t0, dil, first;
    uint64_t *dbg;
    // Winograd kernel only: tile geometry over frame offsets q = -1 .. 64 from the tile start.  Frames q >= o belong
    // to the NEXT utterance (cross-utterance tiling of the concatenated frame axis; o = 1 << 20 when the tile lies
    // inside one utterance): slab pointers above are those of the first utterance, x_bs4 / cp_bs4 the byte strides to
    // the next one.  nvalid = frames of the tile that exist, halo_l / halo_r = frames q = -1 / q = 64 exist (and, for
    // q = -1, belong to the same utterance as q = 0).
    int o, nvalid, halo_l, halo_r;
    unsigned x_bs4, cp_bs4;
    float *sy, *sz;  // optional: slabs [512][T] / [256][T] receiving the pre-gate values and the gated activations
};

// pair index p (0..31) -> frame offset of its first output: blocks of 2d frames, (q, q + d) paired inside a block
__device__ __forceinline__ int wn_pair_q(int p, int d) { return 2 * d * (p / d) + (p % d); }

// frame offset q (already clamped into the valid range) -> byte offset inside the first utterance's row
__device__ __forceinline__ unsigned wn_voff(const LayerTile &a, int q, unsigned slab4) {
    const bool side = q >= a.o;
    return 4u * (unsigned)(a.t0 + q - (side ? a.T : 0)) + (side ? slab4 : 0u);
}

// AGENT: tile outputs are stored agent-scope write-through (persistent kernel: the publish then needs no release fence)
template <int NCB, int GS, bool AGENT = false>
__device__ __forceinline__ void layer_tile(const LayerTile &a, float *smem) {
    constexpr int NTt = 32 * NCB;           // frames per tile;  GS = k-steps per operand group
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int t0 = a.t0;
    const int dil = a.dil;
    const int XW = NTt + 2 * dil;  // tile width incl. halo
    const int T = a.T;
    const float *xin = a.xin;
    uint64_t *dbg = a.dbg;
#define PHASE_STAMP(i) \
    if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime();
    PHASE_STAMP(0)

    // NB every global load below is UNCONDITIONAL on a clamped (always in-bounds) address and the validity
    // select happens afterwards: a `cond ? load : 0` makes hipcc branch around each load and drain vmcnt(0)
    // per element (128 serialized round trips per lane).
    // All global traffic goes through raw-buffer instructions: SGPR descriptor of the slab + SGPR row offset + ONE
    // per-lane byte offset shared by every row (hipcc otherwise builds a 64-bit VGPR address per load / store).
    unsigned lo[NCB];  // per-lane byte offsets (clamped) relative to a wave-uniform row
    bool tv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        lo[cb] = 4u * (unsigned)(4 * half * T + min(t0 + 32 * cb + l31, T - 1));
        tv[cb] = t0 + 32 * cb + l31 < T;
    }
    const unsigned lb = 16u * (unsigned)half;
    const unsigned T4 = 4u * (unsigned)T;
    const rsrc_t rxin = make_rsrc(xin), rskp = make_rsrc(a.skp), rxout = make_rsrc(a.xout), rcp = make_rsrc(a.cpb),
                 rbd = make_rsrc(a.b_dil), rbo = make_rsrc(a.b_out);

    // ---- phase 0a: accumulators of GEMM 1 start at  b_dil + condproj  (so the gate needs no loads later);
    //      all loads are independent and issued together.
    f32x16 acc[4][NCB];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ur = (unsigned)layer_row(w, rb, urow16(r));  // wave-uniform
            const float bias = buf_load(rbd, lb, 4u * ur);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[rb][cb][r] = bias + buf_load(rcp, lo[cb], ur * T4);
        }
    // ---- phase 0b: stage the (x + d) tile: wave w owns channels [64w, 64w+64); one row = one coalesced
    //      64-lane load (+ a halo load when the row is wider than 64); 16 rows in flight.  Zero outside [0,T):
    //      the conv's zero padding applies to x + d (diffnet.py:71,74).
    {
        const int tA = t0 - dil + lane;       // columns 0..63
        const int tB = t0 - dil + 64 + lane;  // columns 64..XW-1
        const bool vA = tA >= 0 && tA < T;
        const bool vB = tB >= 0 && tB < T;
        const unsigned cA = 4u * (unsigned)min(max(tA, 0), T - 1), cB = 4u * (unsigned)min(max(tB, 0), T - 1);
        const bool laneA = lane < XW;
        const bool laneB = 64 + lane < XW;
        for (int r0 = 0; r0 < 64; r0 += 16) {
            float xa[16], xb[16], dd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 64 * w + r0 + u;
                xa[u] = buf_load(rxin, cA, (unsigned)c * T4);
                if constexpr (NCB == 2) xb[u] = buf_load(rxin, cB, (unsigned)c * T4);
                dd[u] = a.dstep[(int64_t)c * a.d_cs];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 64 * w + r0 + u;
                if (laneA) smem[c * XW + lane] = vA ? xa[u] + dd[u] : 0.0f;
                if constexpr (NCB == 2) {
                    if (laneB) smem[c * XW + 64 + lane] = vB ? xb[u] + dd[u] : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    PHASE_STAMP(1)

    // ---- phase 1: GEMM 1  y[512 x NTt] += Wdil[512 x 768] * im2col(xs): k-step ks -> tap ks/128, channels 2*(ks%128)+{0,1}
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w1p) + (int64_t)w * KS1 * 64 + lane;
        const float *bp = smem + half * XW + l31;
        const int rstep = 2 * XW;  // LDS floats between consecutive k-steps (2 channels)
        constexpr int GPT = 128 / GS;  // groups per tap
        gemm_groups<4, NCB, GS>(acc, wp, bp, rstep, KS1 / GS, [&](int g) {
            wp += GS * 64;
            bp += GS * rstep;
            if ((g % GPT) == GPT - 1) bp += dil - 128 * rstep;  // next tap: back to channel 0, shift by dil columns
        });
    }
    PHASE_STAMP(2)

    // ---- phase 2: gate, lane-local (acc[rb] pairs with acc[rb+2]); bias + conditioner are already inside -----
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float z = fast_sigmoid(acc[rb][cb][r]) * fast_tanh(acc[rb + 2][cb][r]);
                acc[rb][cb][r] = tv[cb] ? z : 0.0f;
            }
    PHASE_STAMP(3)
    __syncthreads();  // every wave is done reading xs
    // z tile zs[256][NTt] overlays the xs tile
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 64 * w + 32 * rb + mfma32_row(r, lane);
                smem[c * NTt + cb * 32 + l31] = acc[rb][cb][r];
            }
    // ---- accumulators of GEMM 2 start at  x_in + b_out  (residual rows) /  skip + b_out  (skip rows): the
    //      epilogue is then store-only and these loads fly while the other waves finish their z stores.
    __builtin_amdgcn_sched_barrier(0);  // z registers are dead from here: keep the init loads below the z stores
    const bool first = a.first != 0;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ur = (unsigned)layer_row(w, rb, urow16(r));  // wave-uniform
            const float bias = buf_load(rbo, lb, 4u * ur);
            // residual rows read x_in, skip rows read the running skip sum (ignored by a select when first)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float sv = rb < 2 ? buf_load(rxin, lo[cb], ur * T4) : buf_load(rskp, lo[cb], (ur - DC) * T4);
                acc[rb][cb][r] = (rb >= 2 && first) ? bias : bias + sv;
            }
        }
        if (rb == 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    PHASE_STAMP(4)

    // ---- phase 3: GEMM 2  o[512 x NTt] += Wout[512 x 256] * zs ------------------------------------------------
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w2p) + (int64_t)w * KS2 * 64 + lane;
        const float *bp = smem + half * NTt + l31;
        constexpr int rstep = 2 * NTt;
        gemm_groups<4, NCB, GS>(acc, wp, bp, rstep, KS2 / GS, [&](int) {
            wp += GS * 64;
            bp += GS * rstep;
        });
    }
    PHASE_STAMP(5)

    // ---- phase 4: store-only epilogue: x_out = (x + o_res) * 2^-1/2 ; skip = skip + o_skip ----------------------
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (tv[cb]) {  // one exec-mask region per column block, not one per store (valid frame: the clamp was a no-op)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned ur = (unsigned)layer_row(w, rb, urow16(r));  // wave-uniform
                    const float v = acc[rb][cb][r] * (rb < 2 ? 0.70710678118654752440f : 1.0f);
                    const unsigned so = rb < 2 ? ur * T4 : (ur - DC) * T4;
                    if constexpr (AGENT) buf_store_agent(v, rb < 2 ? rxout : rskp, lo[cb], so);
                    else buf_store(v, rb < 2 ? rxout : rskp, lo[cb], so);
                }
        }
    }
    PHASE_STAMP(6)
#undef PHASE_STAMP
}

__global__ void __launch_bounds__(256, 2) diffnet_layer_kernel(SetDiffnetLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.y;
    LayerTile lt;
    lt.xin = a.x_in + (int64_t)b * DC * a.T;
    lt.xout = a.x_out + (int64_t)b * DC * a.T;
    lt.skp = a.skip + (int64_t)b * DC * a.T;
    lt.cpb = a.condproj + (int64_t)b * a.cp_bs;
    lt.dstep = a.dstep + (int64_t)b * a.d_bs;
    lt.d_cs = a.d_cs;
    lt.w1p = a.w1p; lt.b_dil = a.b_dil; lt.w2p = a.w2p; lt.b_out = a.b_out;
    lt.T = a.T; lt.t0 = blockIdx.x * NT; lt.dil = a.dil; lt.first = a.first;
    lt.dbg = a.dbg_clock ? a.dbg_clock + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
    layer_tile<2, 4>(lt, smem);
}

// ---- Winograd F(2,3) variant of the layer tile (dilation 1) -----------------------------------------------------
// y[n] = g0 x[n-1] + g1 x[n] + g2 x[n+1].  For an output pair (y0, y1) with inputs d0..d3 = x[n-1..n+2]:
//   m1 = (d0 - d2) g0          m2 = (d1 + d2) (g0 + g1 + g2)/2
//   m3 = (d2 - d1) (g0 - g1 + g2)/2      m4 = (d1 - d3) g2          y0 = m1 + m2 + m3,  y1 = m2 - m3 - m4
// i.e. four 512x256 GEMMs over 32 pair-columns instead of one 512x768 GEMM over 64 frames: 2/3 of the MACs of the
// k=3 conv, 3/4 of the layer.  512 threads = 8 waves (2 per SIMD, so each hides the other's operand loads), one
// block per CU, 64-frame tiles.  Wave w owns gate rows [32w,32w+32) and filter rows 256+[32w,32w+32): 2 x 4 planes
// x 16 = 128 accumulator VGPRs.  The input transform is computed on the fly from the RAW (x + d) tile in LDS (two
// ds_read_b64 + 4 VALU per k-step); the filter transform is folded into the packed weights; bias + conditioner
// projection are folded into the M1 / M4 accumulator init.  GEMM 2 and the epilogue are as in layer_tile<2>.
constexpr int WN_NT = 64;
constexpr int WN_MAXD = 8;             // largest dilation (dilation_cycle_length <= 4)
constexpr int WN_XW = WN_NT + 2 * WN_MAXD;  // 64 frames + a halo of d on each side (d = 1: + the 2-column boundary gap)
constexpr int WN_KS = DC / 2;  // 128 k-steps (2 channels each) for every GEMM here
constexpr int WN_GS = 2;       // k-steps per operand group of GEMM 1

// The phases are split so the persistent kernel can order them around its dependency wait:
//   wino_init   - accumulator init from bias + conditioner projection (does NOT depend on the previous layer)
//   wino_main   - stage x, GEMM 1, gate, GEMM 2, epilogue
__device__ __forceinline__ void wino_init(const LayerTile &a, f32x16 (&m)[2][4]) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int T = a.T;
    // output pair j = l31 -> frame offsets q_e ("even" slot) and q_e + d: the F(2,3) pairing runs over frames that are
    // d apart (d = 1: 2j, 2j + 1); clamped to the last existing frame
    const int qe0 = wn_pair_q(l31, a.dil);
    const int qe = min(qe0, a.nvalid - 1), qo = min(qe0 + a.dil, a.nvalid - 1);
    const unsigned rowh = 16u * (unsigned)half * (unsigned)T;  // + 4*half rows
    const unsigned loe = rowh + wn_voff(a, qe, a.cp_bs4), loo = rowh + wn_voff(a, qo, a.cp_bs4);
    const unsigned lb = 16u * (unsigned)half;
    const rsrc_t rcp = make_rsrc(a.cpb), rb = make_rsrc(a.b_dil);
    // M1 <- b + cp(even), M4 <- -(b + cp(odd)), M2 = M3 = 0
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
        m[rf][1] = (f32x16){0};
        m[rf][2] = (f32x16){0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ur = (unsigned)(rf * DC + 32 * w + urow16(r));  // wave-uniform
            const float bias = buf_load(rb, lb, 4u * ur);
            m[rf][0][r] = bias + buf_load(rcp, loe, 4u * ur * (unsigned)T);
            m[rf][3][r] = -(bias + buf_load(rcp, loo, 4u * ur * (unsigned)T));
        }
    }
}

// xs = smem[0 .. 256*68), zs = smem + WN_ZS_OFF (own region: no barrier between the last xs read and the zs write)
constexpr int WN_ZS_OFF = DC * WN_XW;

template <bool UNIT_DIL>  // UNIT_DIL: every layer has dilation 1 (compile-time, keeps the shipped configuration's inner loop lean)
__device__ __forceinline__ void wino_main(const LayerTile &a, f32x16 (&m)[2][4], float *smem, uint64_t *ph, int *s_task,
                                          int claimed) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
#ifdef SET_WINO_PHASES
#define WPH(k) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - ph[9]; ph[9] = t_; }
#else
#define WPH(k)
#endif
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    const int half = lane >> 5, l31 = lane & 31;
    const int t0 = a.t0, T = a.T;
    const unsigned T4 = 4u * (unsigned)T;
    float *zs = smem + WN_ZS_OFF;
    constexpr int XW = UNIT_DIL ? WN_NT + 4 : WN_XW;  // LDS row of the x tile (68 keeps the dilation-1 layout)
    const int d = UNIT_DIL ? 1 : a.dil, qe = UNIT_DIL ? 2 * l31 : wn_pair_q(l31, d);
    const bool tve = qe < a.nvalid, tvo = qe + d < a.nvalid;
    const rsrc_t rxin = make_rsrc(a.xin), rskp = make_rsrc(a.skp), rxout = make_rsrc(a.xout);

    // ---- stage the raw (x + d) tile xs[256][XW]; wave w owns channels [32w, 32w+32).  Frame offset q sits in column
    //      q + d (+ 2 when it belongs to the next utterance, d = 1 only): the two columns in between stay zero, so the
    //      last pair of one utterance and the first pair of the next both see the conv's zero padding.
    {
        const int qA = lane;                                         // columns of frames 0..63
        const int qB = lane < d ? lane - d : 64 + lane - d;          // lanes 0..2d-1: the halo frames -d..-1, 64..64+d-1
        const bool isB = lane < 2 * d;
        // halo frames exist iff they lie in the same utterance (or, right halo in concatenated mode, in the batch)
        const bool vA = qA < a.nvalid;
        const bool vB = isB && (qB < 0 ? (d == 1 ? a.halo_l != 0 : a.t0 + qB >= 0)
                                       : (d == 1 ? a.halo_r != 0 : a.t0 + qB < a.T));
        const unsigned cA = wn_voff(a, min(qA, a.nvalid - 1), a.x_bs4);
        const unsigned cB = vB ? wn_voff(a, qB, a.x_bs4) : cA;  // (invalid: any valid address, value unused)
        const int colA = qA + d + (qA >= a.o ? 2 : 0), colB = qB + d + (qB >= a.o ? 2 : 0);
        const bool gap = a.o <= 64;                     // an utterance boundary lies in (or at the end of) this tile
        for (int r0 = 0; r0 < 32; r0 += 16) {
            float xa[16], xb[16], dd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 32 * w + r0 + u;
                xa[u] = buf_load(rxin, cA, (unsigned)c * T4);
                xb[u] = buf_load(rxin, cB, (unsigned)c * T4);
                dd[u] = a.dstep[(int64_t)c * a.d_cs];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 32 * w + r0 + u;
                smem[c * XW + colA] = vA ? xa[u] + dd[u] : 0.0f;
                if (isB) smem[c * XW + colB] = vB ? xb[u] + dd[u] : 0.0f;
                if (gap && lane >= 32 && lane < 34) smem[c * XW + a.o + lane - 31] = 0.0f;  // columns o+1, o+2
            }
        }
    }
    __syncthreads();
    WPH(1)

    // ---- GEMM 1 (Winograd): m[rf][p] += G_p[rows][c] * D_p[c][pair],  k-step = channels (2ks, 2ks+1)
    {
        const rsrc_t rw = make_rsrc(a.w1p);
        const unsigned wv = 32u * (unsigned)lane;                // 8 floats per lane per k-step
        unsigned ws = (unsigned)w * (WN_KS * 64 * 8 * 4);        // wave-uniform byte offset of the current group
        // inputs of pair (q_e, q_e + d): frames q_e - d, q_e, q_e + d, q_e + 2d = columns c0, c0 + d, c0 + 2d, c0 + 3d
        const float *bpf = smem + half * XW + qe + (qe >= a.o ? 2 : 0);
        const f32x2 *bp = reinterpret_cast<const f32x2 *>(bpf);  // d = 1: two aligned 8-byte reads
        struct Ops { f32x4 A[WN_GS][2]; f32x2 D[WN_GS][2]; };
        Ops P, Q;
        auto load_step = [&](Ops &o, int u) {
            o.A[u][0] = buf_load4(rw, wv, ws + (unsigned)u * 2048u);
            o.A[u][1] = buf_load4(rw, wv + 16u, ws + (unsigned)u * 2048u);
            if (UNIT_DIL) {
                o.D[u][0] = bp[u * XW];      // (d0, d1): row stride 2*XW floats = XW float2
                o.D[u][1] = bp[u * XW + 1];  // (d2, d3)
            } else {
                const float *q = bpf + u * 2 * XW;
                o.D[u][0] = (f32x2){q[0], q[d]};
                o.D[u][1] = (f32x2){q[2 * d], q[3 * d]};
            }
        };
        auto mma_step = [&](const Ops &o, int u) {
            const float d0 = o.D[u][0][0], d1 = o.D[u][0][1], d2 = o.D[u][1][0], d3 = o.D[u][1][1];
            const float D1 = d0 - d2, D2 = d1 + d2, D3 = d2 - d1, D4 = d1 - d3;
            __builtin_amdgcn_s_setprio(1);
            m[0][0] = mfma32(o.A[u][0][0], D1, m[0][0]);
            m[0][1] = mfma32(o.A[u][0][1], D2, m[0][1]);
            m[0][2] = mfma32(o.A[u][0][2], D3, m[0][2]);
            m[0][3] = mfma32(o.A[u][0][3], D4, m[0][3]);
            m[1][0] = mfma32(o.A[u][1][0], D1, m[1][0]);
            m[1][1] = mfma32(o.A[u][1][1], D2, m[1][1]);
            m[1][2] = mfma32(o.A[u][1][2], D3, m[1][2]);
            m[1][3] = mfma32(o.A[u][1][3], D4, m[1][3]);
            __builtin_amdgcn_s_setprio(0);
        };
        constexpr int NG = WN_KS / WN_GS;
#pragma unroll
        for (int u = 0; u < WN_GS; ++u) load_step(P, u);
        for (int g = 0; g < NG; g += 2) {
            ws += WN_GS * 2048u;
            bp += WN_GS * XW;
            bpf += WN_GS * 2 * XW;
#pragma unroll
            for (int u = 0; u < WN_GS; ++u) {
                load_step(Q, u);
                __builtin_amdgcn_sched_barrier(0);
                mma_step(P, u);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g + 2 < NG) { ws += WN_GS * 2048u; bp += WN_GS * XW; bpf += WN_GS * 2 * XW; }
#pragma unroll
            for (int u = 0; u < WN_GS; ++u) {
                load_step(P, u);
                __builtin_amdgcn_sched_barrier(0);
                mma_step(Q, u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    WPH(2)
    // ---- output transform + gate (lane-local): even frame y0 = M1 + M2 + M3, odd frame y1 = M2 - M3 - M4
    {
        const bool save = a.sy != nullptr;  // training forward: keep y (gate | filter rows) and z for the backward pass
        const rsrc_t rsy = make_rsrc(save ? a.sy : a.xin), rsz = make_rsrc(save ? a.sz : a.xin);
        const unsigned rowh_s = 16u * (unsigned)half * (unsigned)T;
        const unsigned se = rowh_s + wn_voff(a, min(qe, a.nvalid - 1), 0u), so_ = rowh_s + wn_voff(a, min(qe + d, a.nvalid - 1), 0u);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g0 = m[0][0][r] + m[0][1][r] + m[0][2][r], g1 = m[0][1][r] - m[0][2][r] - m[0][3][r];
            const float f0 = m[1][0][r] + m[1][1][r] + m[1][2][r], f1 = m[1][1][r] - m[1][2][r] - m[1][3][r];
            const float z0 = tve ? fast_sigmoid(g0) * fast_tanh(f0) : 0.0f;
            const float z1 = tvo ? fast_sigmoid(g1) * fast_tanh(f1) : 0.0f;
            const int c = 32 * w + mfma32_row(r, lane);
            if (UNIT_DIL) {
                *reinterpret_cast<f32x2 *>(zs + c * WN_NT + qe) = (f32x2){z0, z1};
            } else {
                zs[c * WN_NT + qe] = z0;
                zs[c * WN_NT + qe + d] = z1;
            }
            if (save) {  // (uniform branch; the saved tensors are per utterance, so only the per-utterance tiling gets here)
                const unsigned ur = (unsigned)(32 * w + urow16(r));
                if (tve) {
                    buf_store(g0, rsy, se, ur * T4);
                    buf_store(f0, rsy, se, (ur + DC) * T4);
                    buf_store(z0, rsz, se, ur * T4);
                }
                if (tvo) {
                    buf_store(g1, rsy, so_, ur * T4);
                    buf_store(f1, rsy, so_, (ur + DC) * T4);
                    buf_store(z1, rsz, so_, ur * T4);
                }
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- put the residual / running-skip tiles in flight (m is dead now); they are consumed after GEMM 2
    const unsigned rowh = 16u * (unsigned)half * (unsigned)T;  // + 4*half rows
    const unsigned lo0 = rowh + wn_voff(a, min(l31, a.nvalid - 1), a.x_bs4), lo1 = rowh + wn_voff(a, min(32 + l31, a.nvalid - 1), a.x_bs4);
    const bool tv0 = l31 < a.nvalid, tv1 = 32 + l31 < a.nvalid;
    const bool first = a.first != 0;
    f32x16 prev[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ur = (unsigned)(32 * w + urow16(r));  // wave-uniform row of x (rb 0) / skip (rb 1)
            prev[rb][0][r] = buf_load(rb == 0 ? rxin : rskp, lo0, ur * T4);
            prev[rb][1][r] = buf_load(rb == 0 ? rxin : rskp, lo1, ur * T4);
        }
    __builtin_amdgcn_sched_barrier(0);

    // ---- GEMM 2: o[512 x 64] = Wout * zs; wave w owns residual rows [32w, +32) and skip rows 256 + [32w, +32)
    f32x16 acc[2][2];
    {
        const rsrc_t rbo = make_rsrc(a.b_out);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bias = buf_load(rbo, 16u * (unsigned)half, 4u * (unsigned)(rb * DC + 32 * w + urow16(r)));
                acc[rb][0][r] = bias;
                acc[rb][1][r] = bias;
            }
    }
    if (tid == 0) s_task[0] = claimed;  // next task of this block, read by everyone after the epilogue
    __syncthreads();
    WPH(3)
    {
        const AVec<2>::type *wp = reinterpret_cast<const AVec<2>::type *>(a.w2p) + (int64_t)w * WN_KS * 64 + lane;
        const float *bp = zs + half * WN_NT + l31;
        constexpr int rstep = 2 * WN_NT;
        gemm_groups<2, 2, 8>(acc, wp, bp, rstep, WN_KS / 8, [&](int) {
            wp += 8 * 64;
            bp += 8 * rstep;
        });
    }
    WPH(4)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (cb == 0 ? tv0 : tv1) {
            const unsigned so = cb == 0 ? lo0 : lo1;  // valid frame: the clamp was a no-op
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned ur = (unsigned)(32 * w + urow16(r));
                    const float v = (rb == 1 && first) ? acc[rb][cb][r] : acc[rb][cb][r] + prev[rb][cb][r];
                    // agent-scope write-through (sc1): once vmcnt drains, the tile is visible to every XCD, so the
                    // publish needs no L2 write-back (a release fence = buffer_wbl2 of the whole XCD L2: -5 %)
#ifdef SET_WINO_FENCED
                    buf_store(v * (rb == 0 ? 0.70710678118654752440f : 1.0f), rb == 0 ? rxout : rskp, so, ur * T4);
#else
                    buf_store_agent(v * (rb == 0 ? 0.70710678118654752440f : 1.0f), rb == 0 ? rxout : rskp, so, ur * T4);
#endif
                }
        }
    }
    WPH(5)
#undef WPH
}

// ---- persistent layer stack: (layer, tile) task queue + per-tile epoch flags -----------------------------------
// Inter-workgroup hand-off follows cdna_hip_programming.md Guideline 16: producer = every wave drains vmcnt,
// __syncthreads, ONE lane agent-scope release fence + asm vmcnt(0) + relaxed agent flag store; consumer = ONE lane
// polls relaxed, ONE agent-scope acquire, __syncthreads, then plain loads.  Every spin is bounded.
constexpr unsigned STACK_SPIN_LIMIT = 1u << 22;  // x s_sleep(8) ~ 1 s

__device__ __forceinline__ int ld_agent(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NCB, int GS, int WPS>
__global__ void __launch_bounds__(256, WPS) diffnet_stack_kernel(SetDiffnetStackArgs a, int tiles_per_utt, int ntiles,
                                                                int ntasks, int task_slot) {
    constexpr int NTt = 32 * NCB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *s_task = reinterpret_cast<int *>(smem + task_slot);  // inside the ONE dynamic LDS array
    int *counter = a.sync_ws, *abort_flag = a.sync_ws + 1, *done = a.sync_ws + 4;
    const int tid = threadIdx.x;
    uint64_t wait_ticks = 0, fence_ticks = 0;  // diagnostics (lane 0 only): summed into sync_ws[2], sync_ws[3]
    for (;;) {
        __syncthreads();  // LDS (tile + task slot) of the previous task is free
        if (tid == 0) {
            const uint64_t tw0 = __builtin_amdgcn_s_memtime();
            int n = atomicAdd(counter, 1);
            if (n < ntasks && n >= ntiles) {  // layer >= 1: wait for the three producer tiles of layer l-1
                const int l = n / ntiles, i = n - l * ntiles, j = i % tiles_per_utt;
                unsigned spins = 0;
                const int *f0 = done + i, *fl = done + (j > 0 ? i - 1 : i), *fr = done + (j < tiles_per_utt - 1 ? i + 1 : i);
                for (;;) {
                    const int v0 = ld_agent(f0), v1 = ld_agent(fl), v2 = ld_agent(fr);  // three independent loads
                    if (min(v0, min(v1, v2)) >= l) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > STACK_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
                        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.err_flag) __hip_atomic_store(a.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        n = ntasks;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            wait_ticks += __builtin_amdgcn_s_memtime() - tw0;
            *s_task = n;
        }
        __syncthreads();
        const int n = __builtin_amdgcn_readfirstlane(*s_task);
        if (n >= ntasks) {
            if (tid == 0) {  // units of 1024 ticks
                atomicAdd(a.sync_ws + 2, (int)(wait_ticks >> 10));
                atomicAdd(a.sync_ws + 3, (int)(fence_ticks >> 10));
            }
            break;
        }
        const int l = n / ntiles, i = n - l * ntiles;
        const int b = i / tiles_per_utt, j = i - b * tiles_per_utt;
        LayerTile lt;
        const float *xi = (l & 1) ? a.xb : a.xa;
        float *xo = (l & 1) ? a.xa : a.xb;
        lt.xin = xi + (int64_t)b * DC * a.T;
        lt.xout = xo + (int64_t)b * DC * a.T;
        lt.skp = a.skip + (int64_t)b * DC * a.T;
        lt.cpb = a.condproj + (int64_t)l * a.cp_ls + (int64_t)b * a.cp_bs;
        lt.dstep = a.dstep + (int64_t)l * a.d_ls + (int64_t)b * a.d_bs;
        lt.d_cs = a.d_cs;
        lt.w1p = a.w1p_all + (int64_t)l * (512 * 768);
        lt.w2p = a.w2p_all + (int64_t)l * (512 * 256);
        lt.b_dil = a.b_dil_all + (int64_t)l * 512;
        lt.b_out = a.b_out_all + (int64_t)l * 512;
        lt.T = a.T; lt.t0 = j * NTt; lt.dil = 1 << (l % a.dilation_cycle_length); lt.first = (l == 0);
        lt.dbg = nullptr;
        layer_tile<NCB, GS, true>(lt, smem);
        // publish tile i of layer l (outputs were stored agent-scope write-through: vmcnt drain = visible to every XCD)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
        __syncthreads();
        if (tid == 0) {
            const uint64_t tf0 = __builtin_amdgcn_s_memtime();
            __hip_atomic_store(done + i, l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fence_ticks += __builtin_amdgcn_s_memtime() - tf0;
        }
    }
}

// ---- small batches: every 32-frame tile split over FOUR co-operating blocks ---------------------------------------
// With a handful of utterances a layer has fewer tiles than the chip has CUs (B = 1, T = 800: 25 tiles on 256 CUs) and
// the time of a layer is the time of ONE tile on ONE CU: 4 row blocks x 512 MFMAs per SIMD.  Here block (tile i, part h)
// does the work of wave h of layer_tile<1>, one 32-row block per wave (wave j < 2: gate rows 64h + 32j.., then residual
// rows; j >= 2: the matching filter rows, then skip rows), i.e. a quarter of the MFMA chain per SIMD on four times as
// many CUs.  Blocks are bound to their (tile, part) for all L layers (grid = 4 x tiles, all co-resident) and meet twice
// per layer through agent-scope counters:
//   zcnt[i]  += 1 once the part's gated rows z[64h .. 64h+63] are stored (agent-scope write-through) to the tile's slot
//               of z_ws; GEMM 2 of every part reads all 256 rows after zcnt[i] reached 4 (l + 1);
//   ready[i] += 1 once the part's rows of x' are stored; layer l + 1 of tiles i-1, i, i+1 starts at ready >= 4 (l + 1).
// Every output element sees the same accumulation chain as in layer_tile (same initial value, same k order), the gate
// is the same product of the same two values (the tanh factor crosses from the filter wave to the gate wave through
// LDS): results are bit-identical to the direct kernels.  Weight images: one float per lane per k-step and 32-row block
// ([16][KS][64], set_diffnet_stack's w1s_all / w2s_all).
constexpr int SP_XW = 32 + 2 * WN_MAXD;      // widest x tile (d = 8)
constexpr int SP_LDS_FLOATS = DC * SP_XW + 64 * 32 + 4;
constexpr unsigned SP_SPIN_LIMIT = 1u << 20;  // polls of ~1 us each (two agent-scope loads + s_sleep(1)): ~1-2 s

// GEMM of the row-split kernel: ONE 32x32 accumulator per wave, so one MFMA (64 cycles) per k-step and nothing else in
// the wave to hide operand latency behind.  The weight images are cold in L2 at every layer (2 MiB per layer, read once per
// XCD), i.e. A comes from the memory side at ~1.5-2 us: A is prefetched 64 k-steps ahead (two sets of 64 registers,
// ping-pong; the first set is loaded by split_preload BEFORE the inter-block wait that precedes the GEMM), B (LDS) 4 ahead.
//   A k-step ks: wp[ks * 64];  B k-step ks: *bq, bq += rstep per k-step, += tap_jump after every 128th (next conv tap)
__device__ __forceinline__ void split_preload(float (&A)[64], const float *wp) {
#pragma unroll
    for (int u = 0; u < 64; ++u) A[u] = wp[u * 64];
}

template <int KS>
__device__ __forceinline__ void split_gemm(f32x16 &acc, const float *wp, float (&Ap)[64], const float *bq, int rstep,
                                           int tap_jump) {
    static_assert(KS % 128 == 0, "two 64-k-step sets per iteration");
    float Aq[64], Bv[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        Bv[u] = *bq;
        bq += rstep;
    }
    for (int g = 0; g < KS / 64; g += 2) {
        const float *wq = wp + (int64_t)(g + 1) * 64 * 64;
#pragma unroll
        for (int u = 0; u < 64; ++u) {  // k-steps 64 g + u (g even: no tap boundary among the B prefetches of this half)
            Aq[u] = wq[u * 64];
            Bv[(u + 4) & 7] = *bq;
            bq += rstep;
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(Ap[u], Bv[u & 7], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float *wn = wp + (int64_t)(g + 2 < KS / 64 ? g + 2 : g) * 64 * 64;  // last pair: harmless re-load
#pragma unroll
        for (int u = 0; u < 64; ++u) {  // k-steps 64 (g + 1) + u; the prefetch of k-step 64 (g + 2) starts the next tap
            Ap[u] = wn[u * 64];
            Bv[(u + 4) & 7] = *bq;  // (past the last k-step: a harmless read inside the tile)
            bq += rstep;
            if (u == 59) bq += tap_jump;
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(Aq[u], Bv[u & 7], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// debug: lane 0 of block (tile 1, part 1) adds the s_memtime ticks of its phases, summed over the layers, to buf[0..7]
__device__ uint64_t *g_split_phase_buf = nullptr;

// lane 0 of the block: wait until all three counters reach `want`; false = gave up (spin limit or another block aborted)
__device__ __forceinline__ bool split_wait(const int *f0, const int *f1, const int *f2, int want, int *abort_flag,
                                           int *err_flag) {
    unsigned spins = 0;
    for (;;) {
        const int v0 = ld_agent(f0), v1 = ld_agent(f1), v2 = ld_agent(f2);
        if (min(v0, min(v1, v2)) >= want) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SP_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (err_flag) __hip_atomic_store(err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

__global__ void __launch_bounds__(256, 2) diffnet_stack_split_kernel(SetDiffnetStackArgs a, int tiles_per_utt, int ntiles,
                                                                    int fault_tile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // x tile [256][32 + 2d]; the z tile [256][32] overlays it
    float *gs = smem + DC * SP_XW;                                // [64][32] tanh(filter rows) of this part
    int *s_ok = reinterpret_cast<int *>(gs + 64 * 32);
    int *abort_flag = a.sync_ws + 1, *ready = a.sync_ws + 4, *zcnt = a.sync_ws + 4 + ntiles;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int b = i / tiles_per_utt, jt = i - b * tiles_per_utt;
    const int T = a.T, t0 = jt * 32;
    const int il = jt > 0 ? i - 1 : i, ir = jt < tiles_per_utt - 1 ? i + 1 : i;
    const unsigned T4 = 4u * (unsigned)T;
    const unsigned lo = 4u * (unsigned)(4 * half * T + min(t0 + l31, T - 1));  // per-lane byte offset, clamped
    const bool tv = t0 + l31 < T;
    const unsigned lb = 16u * (unsigned)half;
    const rsrc_t rz = make_rsrc(a.z_ws + (int64_t)i * (DC * 32));
    const rsrc_t rskp = make_rsrc(a.skip + (int64_t)b * DC * T);
    const int vr = 4 * h + j;  // 32-row block of the images
    uint64_t *dbg = (blockIdx.x == 5 && tid == 0) ? g_split_phase_buf : nullptr;
    uint64_t tprev = dbg ? __builtin_amdgcn_s_memtime() : 0;
#define SP_PHASE(p)                                           \
    if (dbg) {                                                \
        const uint64_t tn = __builtin_amdgcn_s_memtime();     \
        dbg[p] += tn - tprev;                                 \
        tprev = tn;                                           \
    }
    for (int l = 0; l < a.L; ++l) {
        const int dil = 1 << (l % a.dilation_cycle_length), XW = 32 + 2 * dil;
        const rsrc_t rxin = make_rsrc(((l & 1) ? a.xb : a.xa) + (int64_t)b * DC * T);
        const rsrc_t rxout = make_rsrc(((l & 1) ? a.xa : a.xb) + (int64_t)b * DC * T);
        const rsrc_t rcp = make_rsrc(a.condproj + (int64_t)l * a.cp_ls + (int64_t)b * a.cp_bs);
        const rsrc_t rbd = make_rsrc(a.b_dil_all + (int64_t)l * 512), rbo = make_rsrc(a.b_out_all + (int64_t)l * 512);
        const float *dstep = a.dstep + (int64_t)l * a.d_ls + (int64_t)b * a.d_bs;
        // ---- GEMM 1 accumulator = b_dil + conditioner projection: independent of the previous layer, issued before the wait
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned ur = (unsigned)layer_row(h, j, urow16(r));  // wave-uniform
            acc[0][0][r] = buf_load(rbd, lb, 4u * ur) + buf_load(rcp, lo, ur * T4);
        }
        float Ap[64];  // first 64 k-steps of the wave's weight rows (cold in L2): in flight during the wait and the staging
        const float *wp1 = a.w1s_all + (int64_t)l * (512 * 768) + (int64_t)vr * KS1 * 64 + lane;
        split_preload(Ap, wp1);
        if (tid == 0) *s_ok = (l == 0 || split_wait(ready + i, ready + il, ready + ir, 4 * l, abort_flag, a.err_flag)) ? 1 : 0;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*s_ok) == 0) return;
        SP_PHASE(0)
        // ---- stage x + d (wave j: channels 64j .. 64j+63; zero outside [0, T): the conv pads x + d).  All 64 row loads
        //      of the wave are in flight at once: one memory round trip
        {
            const int tA = t0 - dil + lane;
            const bool vA = tA >= 0 && tA < T, laneA = lane < XW;
            const unsigned cA = 4u * (unsigned)min(max(tA, 0), T - 1);
            float xa[64];
            // step offsets d[c] of the wave's 64 channels: ONE gather (lane u <-> channel 64j + u, stride d_cs: a table
            // column per diffusion step), broadcast per row with v_readlane -- 64 dependent scalar loads would serialise
            const float dv = dstep[(int64_t)(64 * j + lane) * a.d_cs];
#pragma unroll
            for (int u = 0; u < 64; ++u) xa[u] = buf_load(rxin, cA, (unsigned)(64 * j + u) * T4);
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv), u));
                if (laneA) smem[(64 * j + u) * XW + lane] = vA ? xa[u] + dd : 0.0f;
            }
        }
        __syncthreads();
        SP_PHASE(1)
        // ---- GEMM 1: one 32-row block of  y = Wdil (*) (x + d)
        split_gemm<KS1>(acc[0][0], wp1, Ap, smem + half * XW + l31, 2 * XW, dil - 128 * 2 * XW);
        SP_PHASE(2)
        // ---- gate: the filter waves hand tanh(y_f) to the gate waves through LDS
        if (j >= 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gs[(32 * (j - 2) + mfma32_row(r, lane)) * 32 + l31] = fast_tanh(acc[0][0][r]);
        }
        __syncthreads();  // gs complete; every wave is done reading the x tile
        if (j < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fast_sigmoid(acc[0][0][r]) * gs[(32 * j + mfma32_row(r, lane)) * 32 + l31];
                buf_store_agent(tv ? z : 0.0f, rz, 4u * (unsigned)(4 * half * 32 + l31),
                                4u * 32u * (unsigned)(64 * h + 32 * j + urow16(r)));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the z rows of this wave are visible to every XCD
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(zcnt + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SP_PHASE(3)
        // ---- GEMM 2 accumulator = b_out + x (residual rows) / + running skip sum (skip rows): these rows were written
        //      by this very wave one layer ago; the loads fly while lane 0 waits for the other parts' z rows
        {
            // rows 64h + 32(j & 1) + .. of x (gate waves) or of the skip sum (filter waves): one descriptor, no per-row branch
            const rsrc_t rsv = make_rsrc(j < 2 ? ((l & 1) ? a.xb : a.xa) + (int64_t)b * DC * T : a.skip + (int64_t)b * DC * T);
            float bias[16], sv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bias[r] = buf_load(rbo, lb, 4u * (unsigned)layer_row(h, j, urow16(r)));
                sv[r] = buf_load(rsv, lo, (unsigned)(64 * h + 32 * (j & 1) + urow16(r)) * T4);
            }
            if (j >= 2 && l == 0) {  // first layer: the skip sum starts here (whatever the buffer held is ignored)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = bias[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][0][r] = bias[r] + sv[r];
            }
        }
        const float *wp2 = a.w2s_all + (int64_t)l * (512 * 256) + (int64_t)vr * KS2 * 64 + lane;
        split_preload(Ap, wp2);
        if (tid == 0) *s_ok = split_wait(zcnt + i, zcnt + i, zcnt + i, 4 * (l + 1), abort_flag, a.err_flag) ? 1 : 0;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(*s_ok) == 0) return;
        SP_PHASE(4)
        // ---- the whole z tile [256][32] (one contiguous 32 KiB slot) -> LDS, same layout
        {
            f32x4 zv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) zv[k] = buf_load4(rz, 16u * (unsigned)tid, 4096u * (unsigned)k);
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<f32x4 *>(smem + 4 * tid + 1024 * k) = zv[k];
        }
        __syncthreads();
        SP_PHASE(5)
        // ---- GEMM 2: one 32-row block of  o = Wout z
        split_gemm<KS2>(acc[0][0], wp2, Ap, smem + half * 32 + l31, 64, 0);
        SP_PHASE(6)
        // ---- epilogue: x' = (x + o_res) / sqrt 2 (agent scope: the neighbours' next layer reads it), skip += o_skip
        if (tv && j < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store_agent(acc[0][0][r] * 0.70710678118654752440f, rxout, lo, (unsigned)layer_row(h, j, urow16(r)) * T4);
        } else if (tv) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store(acc[0][0][r], rskp, lo, (unsigned)(layer_row(h, j, urow16(r)) - DC) * T4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every store of the block has completed; the LDS tile is free for the next layer
        if (tid == 0 && !(l == 0 && i == fault_tile && h == 0))
            __hip_atomic_fetch_add(ready + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SP_PHASE(7)
    }
#undef SP_PHASE
}

// Winograd flavour of the persistent kernel: 512 threads, one block per CU, 64-frame tiles.  Same queue and flags
// as above, but each block claims its NEXT task while the current one is in GEMM 2, and issues the next task's
// producer-independent loads (bias + conditioner projection -> accumulator init) BEFORE it waits for the producer
// tiles, so that wait, the store drain and the release fence of the previous task overlap with them.  Claiming
// ahead is deadlock-free: a block finishes its claims in claim order, and a claim only ever waits on earlier ones.
template <bool UNIT_DIL>  // separate kernels: the dilation-1 one carries none of the dilated-layer code or registers
__global__ void __launch_bounds__(512, 2) diffnet_stack_wino_kernel(SetDiffnetStackArgs a, int tiles_per_utt, int ntiles,
                                                                    int ntasks, int concat, int fault_tile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *s_task = reinterpret_cast<int *>(smem + WN_ZS_OFF + DC * WN_NT);
    int *counter = a.sync_ws, *abort_flag = a.sync_ws + 1, *done = a.sync_ws + 4;
    const int tid = threadIdx.x;
    uint64_t wait_ticks = 0, fence_ticks = 0;
    if (tid == 0) s_task[0] = atomicAdd(counter, 1);
    __syncthreads();
    int n = __builtin_amdgcn_readfirstlane(s_task[0]);
    int l = 0, i = 0, j = 0;
    LayerTile lt;
    auto decode = [&](int task) {
        l = task / ntiles;
        i = task - l * ntiles;
        int b;
        if (concat) {  // tile i covers frames [64 i, 64 i + 64) of the concatenated axis; tiles_per_utt == ntiles, j == i
            const int64_t f0 = (int64_t)WN_NT * i, F = (int64_t)a.B * a.T;
            b = (int)(f0 / a.T);
            j = i;
            lt.t0 = (int)(f0 - (int64_t)b * a.T);
            lt.o = (a.T - lt.t0 <= WN_NT && b + 1 < a.B) ? a.T - lt.t0 : (1 << 20);
            lt.nvalid = (int)min((int64_t)WN_NT, (b + 1 < a.B ? F : (int64_t)(b + 1) * a.T) - f0);
            lt.halo_l = lt.t0 > 0;
            lt.halo_r = f0 + WN_NT < F;
        } else {
            b = i / tiles_per_utt;
            j = i - b * tiles_per_utt;
            lt.t0 = j * WN_NT;
            lt.o = 1 << 20;
            lt.nvalid = min(WN_NT, a.T - lt.t0);
            lt.halo_l = lt.t0 > 0;
            lt.halo_r = lt.t0 + WN_NT < a.T;
        }
        lt.x_bs4 = 4u * (unsigned)DC * (unsigned)a.T;
        lt.cp_bs4 = 4u * (unsigned)a.cp_bs;
        const int64_t slab = (int64_t)a.B * DC * a.T;
        const float *xi = a.x_all ? a.x_all + l * slab : ((l & 1) ? a.xb : a.xa);
        float *xo = a.x_all ? a.x_all + (l + 1) * slab : ((l & 1) ? a.xa : a.xb);
        lt.xin = xi + (int64_t)b * DC * a.T;
        lt.xout = xo + (int64_t)b * DC * a.T;
        lt.skp = a.skip + (int64_t)b * DC * a.T;
        lt.sy = a.save_y ? a.save_y + 2 * l * slab + (int64_t)b * 2 * DC * a.T : nullptr;
        lt.sz = a.save_z ? a.save_z + l * slab + (int64_t)b * DC * a.T : nullptr;
        lt.cpb = a.condproj + (int64_t)l * a.cp_ls + (int64_t)b * a.cp_bs;
        lt.dstep = a.dstep + (int64_t)l * a.d_ls + (int64_t)b * a.d_bs;
        lt.d_cs = a.d_cs;
        lt.w1p = a.w1w_all + (int64_t)l * (512 * 256 * 4);
        lt.w2p = a.w2w_all + (int64_t)l * (512 * 256);
        lt.b_dil = a.b_dil_all + (int64_t)l * 512;
        lt.b_out = a.b_out_all + (int64_t)l * 512;
        lt.T = a.T; lt.dil = UNIT_DIL ? 1 : 1 << (l % a.dilation_cycle_length); lt.first = (l == 0);
        lt.dbg = nullptr;
    };
#ifdef SET_WINO_PHASES
    // [0] init loads + vmcnt drain, [6] barrier, [8] flag/dep-wait/claim + barrier, [1] stage, [2] gemm1, [3] gate.., [4] gemm2, [5] epilogue
    uint64_t ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    ph[9] = __builtin_amdgcn_s_memtime();
#else
    uint64_t *ph = nullptr;
#endif
    int i_done = -1, l_done = 0;  // finished but not yet published tile of this block
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave
#ifdef SET_WINO_PHASES
        { const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[0] += t_ - ph[9]; ph[9] = t_; }
#endif
        __syncthreads();
#ifdef SET_WINO_PHASES
        { const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[6] += t_ - ph[9]; ph[9] = t_; }
#endif
        if (tid == 0 && i_done >= 0) {
            const uint64_t tf0 = __builtin_amdgcn_s_memtime();
#ifdef SET_WINO_FENCED
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            // (fault_tile >= 0: test hook, SET_AMD_FAULT_TILE -- that tile of layer 0 is never published, so its
            // consumers must run into the spin limit and the launch must report it)
            if (!(l_done == 0 && i_done == fault_tile))
                __hip_atomic_store(done + i_done, l_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            fence_ticks += __builtin_amdgcn_s_memtime() - tf0;
        }
    };
    while (n < ntasks) {
        decode(n);
        f32x16 m[2][4];
        wino_init(lt, m);  // issued before the previous tile's store drain / publish and before the dependency wait
        __builtin_amdgcn_sched_barrier(0);
        publish();
        int claimed = 0;
        if (tid == 0) {
            int ok_all = 1;
            if (l > 0) {
                const uint64_t tw0 = __builtin_amdgcn_s_memtime();
                unsigned spins = 0;
                const int *f0 = done + i, *fl = done + (j > 0 ? i - 1 : i), *fr = done + (j < tiles_per_utt - 1 ? i + 1 : i);
                for (;;) {
                    const int v0 = ld_agent(f0), v1 = ld_agent(fl), v2 = ld_agent(fr);  // three independent loads
                    if (min(v0, min(v1, v2)) >= l) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > STACK_SPIN_LIMIT || ld_agent(abort_flag) != 0) {
                        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (a.err_flag) __hip_atomic_store(a.err_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok_all = 0;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                wait_ticks += __builtin_amdgcn_s_memtime() - tw0;
            }
            s_task[1] = ok_all;
            claimed = atomicAdd(counter, 1);  // claim the next task; the result is only needed after GEMM 1
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(s_task[1]) == 0) {
            i_done = -1;
            break;
        }
#ifdef SET_WINO_PHASES
        { const uint64_t t_ = __builtin_amdgcn_s_memtime(); ph[8] += t_ - ph[9]; ph[9] = t_; }
#endif
        wino_main<UNIT_DIL>(lt, m, smem, ph, s_task, claimed);
        i_done = i;
        l_done = l;
        n = __builtin_amdgcn_readfirstlane(s_task[0]);  // written by thread 0 before the barrier in front of GEMM 2
    }
    publish();
#ifdef SET_WINO_PHASES
    if (tid == 0)
        for (int k = 0; k < 9; ++k) atomicAdd(a.sync_ws + 4 + ntiles + k, (int)(ph[k] >> 10));
#endif
    if (tid == 0) {  // units of 1024 ticks
        atomicAdd(a.sync_ws + 2, (int)(wait_ticks >> 10));
        atomicAdd(a.sync_ws + 3, (int)(fence_ticks >> 10));
    }
}

// blockIdx.y = layer q of a stack: weights at w_dil + q wd_ls / w_out + q wo_ls (element strides), images at q * (their per-layer size)
__global__ void __launch_bounds__(256) pack_diffnet_layer_kernel(const float *w_dil, const float *w_out, float *w1p,
                                                                 float *w2p, int64_t wd_ls = 0, int64_t wo_ls = 0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n1 = (int64_t)4 * KS1 * 64 * 4, n2 = (int64_t)4 * KS2 * 64 * 4;
    w_dil += (int64_t)blockIdx.y * wd_ls;
    w_out += (int64_t)blockIdx.y * wo_ls;
    w1p += (int64_t)blockIdx.y * n1;
    w2p += (int64_t)blockIdx.y * n2;
    if (idx < n1) {
        const int rb = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const int ks = (int)((idx >> 8) % KS1), w = (int)((idx >> 8) / KS1);
        const int tap = ks >> 7, cp = ks & 127;
        const int row = layer_row(w, rb, lane & 31), c = 2 * cp + (lane >> 5);
        w1p[idx] = w_dil[((int64_t)row * DC + c) * 3 + tap];
    } else if (idx < n1 + n2) {
        const int64_t j = idx - n1;
        const int rb = (int)(j & 3), lane = (int)((j >> 2) & 63);
        const int ks = (int)((j >> 8) % KS2), w = (int)((j >> 8) / KS2);
        const int row = layer_row(w, rb, lane & 31), c = 2 * ks + (lane >> 5);
        w2p[j] = w_out[(int64_t)row * DC + c];
    }
}

// Winograd images: w1w[w][ks][lane][rf*4 + p] = G_p(Wdil[rf*256 + 32w + (lane&31)][2ks + (lane>>5)][0..2]),
//                  w2w[w][ks][lane][rb]       = Wout[rb*256 + 32w + (lane&31)][2ks + (lane>>5)]          (8 waves)
__global__ void __launch_bounds__(256) pack_diffnet_wino_kernel(const float *w_dil, const float *w_out, float *w1w,
                                                                float *w2w, int64_t wd_ls = 0, int64_t wo_ls = 0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n1 = (int64_t)8 * WN_KS * 64 * 8, n2 = (int64_t)8 * WN_KS * 64 * 2;
    w_dil += (int64_t)blockIdx.y * wd_ls;  // (blockIdx.y = layer, as in pack_diffnet_layer_kernel)
    w_out += (int64_t)blockIdx.y * wo_ls;
    w1w += (int64_t)blockIdx.y * n1;
    w2w += (int64_t)blockIdx.y * n2;
    if (idx < n1) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int ks = (int)((idx >> 9) % WN_KS), w = (int)((idx >> 9) / WN_KS);
        const int rf = e >> 2, p = e & 3;
        const int row = rf * DC + 32 * w + (lane & 31), c = 2 * ks + (lane >> 5);
        const float *g = w_dil + ((int64_t)row * DC + c) * 3;
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        const double v = p == 0 ? g0 : (p == 1 ? 0.5 * (g0 + g1 + g2) : (p == 2 ? 0.5 * (g0 - g1 + g2) : g2));
        w1w[idx] = (float)v;
    } else if (idx < n1 + n2) {
        const int64_t j = idx - n1;
        const int rb = (int)(j & 1), lane = (int)((j >> 1) & 63);
        const int ks = (int)((j >> 7) % WN_KS), w = (int)((j >> 7) / WN_KS);
        const int row = rb * DC + 32 * w + (lane & 31), c = 2 * ks + (lane >> 5);
        w2w[j] = w_out[(int64_t)row * DC + c];
    }
}

}  // namespace

extern "C" int64_t set_diffnet_w1w_size(void) { return (int64_t)512 * 256 * 4; }
extern "C" int set_pack_diffnet_layer_wino(const float *w_dil, const float *w_out, float *w1w, float *w2w, void *stream) {
    SET_REQUIRE(w_dil && w_out && w1w && w2w, "set_pack_diffnet_layer_wino");
    const int64_t total = set_diffnet_w1w_size() + (int64_t)512 * 256;
    hipLaunchKernelGGL(pack_diffnet_wino_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream, w_dil,
                       w_out, w1w, w2w);
    return set_check_launch("set_pack_diffnet_layer_wino");
}

extern "C" int64_t set_diffnet_w1p_size(void) { return (int64_t)512 * 768; }
extern "C" int64_t set_diffnet_w2p_size(void) { return (int64_t)512 * 256; }

// every layer of a stack in one launch per image family (the training forward re-packs all of them after each optimizer step: 2 x L
// launches of ~5 us otherwise).  Layer q's weights at w_dil + q wd_ls / w_out + q wo_ls (the flat optimizer's layout: one stride between
// the layers' tensors); w1p / w2p are [L][set_diffnet_w1p_size()] / [L][set_diffnet_w2p_size()], w1w / w2w (optional, both or neither) the
// Winograd images [L][set_diffnet_w1w_size()] / [L][512 * 256].
extern "C" int set_pack_diffnet_layers(const float *w_dil, const float *w_out, int64_t wd_ls, int64_t wo_ls, float *w1p, float *w2p,
                                       float *w1w, float *w2w, int32_t L, void *stream) {
    SET_REQUIRE(w_dil && w_out && w1p && w2p && L > 0 && L <= 65535 && (w1w == nullptr) == (w2w == nullptr), "set_pack_diffnet_layers");
    SET_REQUIRE((int64_t)8 * WN_KS * 64 * 8 == set_diffnet_w1w_size() && (int64_t)8 * WN_KS * 64 * 2 == (int64_t)512 * 256 &&
                    (int64_t)4 * KS1 * 64 * 4 == set_diffnet_w1p_size() && (int64_t)4 * KS2 * 64 * 4 == set_diffnet_w2p_size(),
                "set_pack_diffnet_layers(image sizes)");
    const int64_t total = set_diffnet_w1p_size() + set_diffnet_w2p_size();
    hipLaunchKernelGGL(pack_diffnet_layer_kernel, dim3(set_blocks(total, 256), L), dim3(256), 0, (hipStream_t)stream, w_dil, w_out, w1p,
                       w2p, wd_ls, wo_ls);
    int rc = set_check_launch("set_pack_diffnet_layers");
    if (rc != SET_OK || w1w == nullptr) return rc;
    const int64_t totw = set_diffnet_w1w_size() + (int64_t)512 * 256;
    hipLaunchKernelGGL(pack_diffnet_wino_kernel, dim3(set_blocks(totw, 256), L), dim3(256), 0, (hipStream_t)stream, w_dil, w_out, w1w,
                       w2w, wd_ls, wo_ls);
    return set_check_launch("set_pack_diffnet_layers(wino)");
}
extern "C" int set_pack_diffnet_layer(const float *w_dil, const float *w_out, float *w1p, float *w2p, void *stream) {
    SET_REQUIRE(w_dil && w_out && w1p && w2p, "set_pack_diffnet_layer");
    const int64_t total = set_diffnet_w1p_size() + set_diffnet_w2p_size();
    hipLaunchKernelGGL(pack_diffnet_layer_kernel, dim3(set_blocks(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_dil, w_out, w1p, w2p);
    return set_check_launch("set_pack_diffnet_layer");
}

extern "C" int set_diffnet_layer(const SetDiffnetLayerArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffnet_layer");
    const SetDiffnetLayerArgs &a = *args;
    SET_REQUIRE(a.x_in && a.condproj && a.dstep && a.w1p && a.b_dil && a.w2p && a.b_out && a.x_out && a.skip,
                "set_diffnet_layer");
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.dil >= 1, "set_diffnet_layer");
    SET_REQUIRE(a.x_in != a.x_out, "set_diffnet_layer(x_in must not alias x_out)");
    if (a.dil > 8) return set_fail(SET_E_UNSUPPORTED, "set_diffnet_layer", "dilation > 8 (LDS tile > 80 KiB)");
    const size_t lds = (size_t)DC * (NT + 2 * a.dil) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_layer_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                "set_diffnet_layer(attr)");
        attr_set = true;
    }
    dim3 grid((a.T + NT - 1) / NT, a.B);
    hipLaunchKernelGGL(diffnet_layer_kernel, grid, dim3(256), lds, (hipStream_t)stream, a);
    return set_check_launch("set_diffnet_layer");
}

extern "C" int set_debug_split_phase_buffer(uint64_t *buf) {
    SET_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_split_phase_buf), &buf, sizeof(buf)), "set_debug_split_phase_buffer");
    return SET_OK;
}

extern "C" int64_t set_sizeof_diffnet_stack_args(void) { return (int64_t)sizeof(SetDiffnetStackArgs); }

// 0 = direct kernel, 64-frame tiles; 1 = direct kernel, 32-frame tiles; 2 = Winograd F(2,3) kernel (64-frame tiles,
// 8-wave blocks, needs its packed images, dilation_cycle_length <= 4 and at least ~0.68 tiles per CU to be worth it);
// 3 = row-split kernel for small batches (4 blocks per 32-frame tile, needs its images and the z workspace);
// 4 / 5 = split-operand kernel (fp32 = 3 bf16 pieces, six bf16 MFMAs per product / 2 fp16 pieces, three; csrc/diffnet_x3.hip)
static int stack_variant(int B, int T, int dcl, bool have_wino, bool have_split, int x3_mode, int n_cu) {
    const bool have_x3 = x3_mode == 2 || x3_mode == 3;
    const int64_t tiles64 = (int64_t)B * ((T + 63) / 64);
    // row-split kernel: 4 blocks per 32-frame tile, all co-resident (2 per CU); SET_AMD_SPLIT=0 disables, =2 forces it
    // (when it fits); an explicit SET_AMD_WINO choice also rules it out
    // (measured at T = 800: one utterance 75 ms per 100 steps, two 79 ms; from three utterances on two blocks would share a CU
    // and the split-operand kernel, ~123 ms whatever the batch up to B = 16, is the faster one)
    const int64_t split_blocks = 4 * (int64_t)B * ((T + 31) / 32);
    // co-residency: two blocks per CU for the fp32-pipe kernel, ONE for the two-piece fp16 one (its A ring takes the whole register
    // file of a SIMD lane group: launch bounds (256, 1)) unless SET_AMD_SPLIT_F32 pins the fp32-pipe kernel
    const bool split_one_per_cu = have_x3 && !(getenv("SET_AMD_SPLIT_F32") && atoi(getenv("SET_AMD_SPLIT_F32")) != 0);
    const bool split_fits = have_split && dcl <= 4 && split_blocks <= (split_one_per_cu ? 1 : 2) * (int64_t)n_cu;
    const bool split_pays = split_blocks <= (int64_t)(have_x3 ? 1 : 2) * n_cu;
    int split_env = 1;
    if (const char *e = getenv("SET_AMD_SPLIT")) split_env = atoi(e);
    if (split_fits && (split_env == 2 || (split_env == 1 && !getenv("SET_AMD_WINO") && split_pays))) return 3;
    // split-operand kernel: every batch the row-split kernel does not take (a task is 61 us against 77 us for a 32-frame
    // task of the direct fp32 kernel, so it wins even when the chip is far from full; tiny inputs stay on the fp32 kernels);
    // SET_AMD_X3=0 disables, =2 forces it at any size; an explicit SET_AMD_WINO choice also rules it out
    int x3_env = 1;
    if (const char *e = getenv("SET_AMD_X3")) x3_env = atoi(e);
    if (have_x3 && dcl <= 4 && (x3_env == 2 || (x3_env == 1 && !getenv("SET_AMD_WINO") && tiles64 >= 8)))
        return x3_mode == 3 ? 4 : 5;
    int ncb = tiles64 < 3 * n_cu ? 1 : 2;
    if (const char *e = getenv("SET_AMD_STACK_NCB")) ncb = atoi(e) == 2 ? 2 : 1;
    const bool wino_ok = have_wino && (1 << (dcl - 1)) <= WN_MAXD;
    bool wino = wino_ok && 25 * tiles64 >= 17 * n_cu;  // measured crossover vs the direct 32-frame kernel: ~0.68 tiles per CU
    if (const char *e = getenv("SET_AMD_WINO")) wino = wino_ok && (atoi(e) == 2 || (wino && atoi(e) != 0));  // 2 = force
    return wino ? 2 : (ncb == 1 ? 1 : 0);
}
extern "C" int set_diffnet_stack_variant(int B, int T, int dilation_cycle_length, int images) {
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    return stack_variant(B, T, dilation_cycle_length, (images & 1) != 0, (images & 2) != 0,
                         (images & 4) ? 3 : ((images & 8) ? 2 : 0), n_cu);
}

int set_launch_diffnet_stack_x3(const SetDiffnetStackArgs &a, int n_cu, int fault_tile, hipStream_t s);  // csrc/diffnet_x3.hip
int set_x3_winograd_selected(int x3_mode, int B, int T, int dilation_cycle_length, int n_cu);                // csrc/diffnet_x3.hip (0 / 1 / 2)
extern "C" int set_diffnet_stack_x3_winograd(int B, int T, int dilation_cycle_length, int images) {
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const int x3_mode = (images & 4) ? 3 : ((images & 8) ? 2 : 0);
    if (stack_variant(B, T, dilation_cycle_length, (images & 1) != 0, (images & 2) != 0, x3_mode, n_cu) != 5) return 0;
    return set_x3_winograd_selected(x3_mode, B, T, dilation_cycle_length, n_cu);
}
int set_launch_diffnet_stack_split_x2(const SetDiffnetStackArgs &a, int fault_tile, hipStream_t s);      // csrc/diffnet_x3.hip

extern "C" int set_diffnet_stack(const SetDiffnetStackArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffnet_stack");
    const SetDiffnetStackArgs &a = *args;
    SET_REQUIRE(a.xa && a.xb && a.skip && a.condproj && a.dstep && a.w1p_all && a.w2p_all && a.b_dil_all &&
                    a.b_out_all && a.sync_ws,
                "set_diffnet_stack");
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.L > 0 && a.dilation_cycle_length >= 1 && a.dilation_cycle_length <= 4,
                "set_diffnet_stack");
    hipStream_t s = (hipStream_t)stream;
    static int n_cu = 0;
    static bool attr_set = false;
    if (!attr_set) {
        const void *fns[5] = {reinterpret_cast<const void *>(diffnet_stack_kernel<1, 8, 2>),
                              reinterpret_cast<const void *>(diffnet_stack_kernel<2, 4, 2>),
                              reinterpret_cast<const void *>(diffnet_stack_wino_kernel<true>),
                              reinterpret_cast<const void *>(diffnet_stack_wino_kernel<false>),
                              reinterpret_cast<const void *>(diffnet_stack_split_kernel)};
        for (const void *f : fns)
            SET_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024),
                    "set_diffnet_stack(attr)");
        int dev = 0;
        SET_HIP(hipGetDevice(&dev), "set_diffnet_stack");
        SET_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev), "set_diffnet_stack");
        attr_set = true;
    }
    // Tile width: a task (l, i) needs tiles i-1..i+1 of layer l-1, so at most `tiles per layer` tasks are ever
    // runnable.  Workers (2 per CU) must stay BELOW that or the youngest ones only wait (measured: 36 % wait time
    // with 512 workers on 416 64-frame tiles).  Use 64-frame tiles when a layer has >= 1.5x the workers, else
    // 32-frame tiles (B=32, T=800: 800 tiles, no tail waste); the grid is capped at 0.8x the tile count.
    const bool plain = !a.x_all && !a.save_y && !a.save_z;
    const int variant = stack_variant(a.B, a.T, a.dilation_cycle_length, a.w1w_all && a.w2w_all,
                                      a.w1s_all && a.w2s_all && a.z_ws && plain, (a.wx3_all && plain) ? a.x3_mode : 0, n_cu);
    int fault_tile = -1;  // test hook: never publish this tile of layer 0 (exercises the time-out / error path)
    if (const char *e = getenv("SET_AMD_FAULT_TILE")) fault_tile = atoi(e);
    if (variant >= 4) return set_launch_diffnet_stack_x3(a, n_cu, fault_tile, s);
    if (variant == 3 && a.wx3_all && a.x3_mode == 2 && !(getenv("SET_AMD_SPLIT_F32") && atoi(getenv("SET_AMD_SPLIT_F32")) != 0))
        return set_launch_diffnet_stack_split_x2(a, fault_tile, s);  // the same scheme on the two-piece fp16 operands
    if (variant == 3) {
        const int tiles = (a.T + 31) / 32, nt = a.B * tiles;
        SET_HIP(set_zero_async(a.sync_ws, (size_t)(4 + 2 * nt) * sizeof(int32_t), s), "set_diffnet_stack(memset)");
        hipLaunchKernelGGL(diffnet_stack_split_kernel, dim3(4 * nt), dim3(256), (size_t)SP_LDS_FLOATS * sizeof(float), s, a,
                           tiles, nt, fault_tile);
        return set_check_launch("set_diffnet_stack");
    }
    const bool wino = variant == 2;
    const int ncb = variant == 1 ? 1 : 2;
    const int ntt = 32 * ncb;
    int tiles_per_utt = (a.T + ntt - 1) / ntt;
    int ntiles = a.B * tiles_per_utt;
    // Winograd kernel: tile the CONCATENATED frame axis when an utterance boundary can only fall between output pairs
    // (T even) and at most once per tile (T >= 64): B*T/64 tiles instead of B*ceil(T/64) (T = 800: 400 vs 416).
    bool concat = wino && a.dilation_cycle_length == 1 && a.T % 2 == 0 && a.T >= WN_NT &&
                  (int64_t)a.cp_bs * 4 < (1ll << 31) && a.d_bs == 0 && !a.save_y && !a.save_z;
    if (concat) {
        ntiles = (int)(((int64_t)a.B * a.T + WN_NT - 1) / WN_NT);
        tiles_per_utt = ntiles;  // one chain of tiles: neighbours are i-1 / i+1 everywhere
    }
    const int64_t ntasks64 = (int64_t)ntiles * a.L;
    SET_REQUIRE(ntasks64 < (1ll << 30), "set_diffnet_stack(task count)");
    const int max_dil = 1 << (a.dilation_cycle_length - 1);
    const int task_slot = DC * (ntt + 2 * max_dil);  // float index of the task word
    const size_t lds = (size_t)(task_slot + 4) * sizeof(float);
    SET_HIP(set_zero_async(a.sync_ws, (size_t)(4 + ntiles + (wino ? 12 : 0)) * sizeof(int32_t), s),
            "set_diffnet_stack(memset)");
    const int wps = wino ? 1 : 2;  // resident blocks per CU (three 168-VGPR blocks per CU measured slower: see DESIGN.md)
    int grid = wps * n_cu;
    if (grid > ntiles * 4 / 5) grid = ntiles * 4 / 5;
    if (grid < n_cu) grid = n_cu < ntiles ? n_cu : ntiles;
    if (const char *e = getenv("SET_AMD_STACK_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
    if ((int64_t)grid > ntasks64) grid = (int)ntasks64;
    if (grid < 1) grid = 1;
    if (wino) {
        if (a.dilation_cycle_length == 1)
            hipLaunchKernelGGL(diffnet_stack_wino_kernel<true>, dim3(grid), dim3(512),
                               (size_t)(WN_ZS_OFF + DC * WN_NT + 4) * sizeof(float), s, a, tiles_per_utt, ntiles,
                               (int)ntasks64, concat ? 1 : 0, fault_tile);
        else
            hipLaunchKernelGGL(diffnet_stack_wino_kernel<false>, dim3(grid), dim3(512),
                               (size_t)(WN_ZS_OFF + DC * WN_NT + 4) * sizeof(float), s, a, tiles_per_utt, ntiles,
                               (int)ntasks64, 0, fault_tile);
    } else if (ncb == 1)
        hipLaunchKernelGGL((diffnet_stack_kernel<1, 8, 2>), dim3(grid), dim3(256), lds, s, a, tiles_per_utt, ntiles,
                           (int)ntasks64, task_slot);
    else
        hipLaunchKernelGGL((diffnet_stack_kernel<2, 4, 2>), dim3(grid), dim3(256), lds, s, a, tiles_per_utt, ntiles,
                           (int)ntasks64, task_slot);
    return set_check_launch("set_diffnet_stack");
}

// ----------------------------------------------------------------------------------------------------------
// unfused pieces (any residual_channels; also the device-side cross-check of the fused kernel)
// ----------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) gate_kernel(const float *y, float *z, int B, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int64_t ct = i % ((int64_t)C * T);
    const int64_t b = i / ((int64_t)C * T);
    const float *yb = y + b * 2 * C * T;
    z[i] = dev_sigmoid(yb[ct]) * tanhf(yb[(int64_t)C * T + ct]);
}
__global__ void __launch_bounds__(256) res_skip_kernel(const float *x_in, const float *o, float *x_out, float *skip,
                                                       int B, int C, int T, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int64_t ct = i % ((int64_t)C * T);
    const int64_t b = i / ((int64_t)C * T);
    const float *ob = o + b * 2 * C * T;
    x_out[i] = (x_in[i] + ob[ct]) / 1.41421356237309504880f;
    const float s = ob[(int64_t)C * T + ct];
    skip[i] = first ? s : skip[i] + s;
}
__global__ void __launch_bounds__(256) sinusoid_kernel(const float *t, float *out, int dim, int n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)dim * n) return;
    const int j = (int)(i / n), k = (int)(i % n);
    const int half = dim / 2;
    const int jj = j < half ? j : j - half;
    // emb = exp(arange(half) * -(ln(1e4)/(half-1)))   (diffnet.py:42-43, all fp32 tensor ops)
    const float e = (float)(9.210340371976184 / (double)(half - 1));  // python float -> fp32 scalar
    const float freq = expf((float)jj * -e);
    const float ang = t[k] * freq;
    out[i] = j < half ? sinf(ang) : cosf(ang);
}

// (Philox4x32-10 + Box-Muller: csrc/boundary_x2.h, shared with the whole-loop kernel of csrc/diffnet_x3.hip)

// seed_delta (set_rng_seed_delta, may be NULL): a device word ADDED to the seed argument -- a captured graph carries the seed of the
// step it was captured at; the replay of step k stores (seed_k - seed_captured) there and draws exactly the eager step's numbers
__global__ void __launch_bounds__(256) randn_kernel(float *out, int64_t n, uint64_t seed, uint64_t offset, const uint64_t *seed_delta) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index
    if (q * 4 >= n) return;
    if (seed_delta) seed += *seed_delta;
    float z[4];
    randn4(seed, offset + (uint64_t)q, z);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (q * 4 + k < n) out[q * 4 + k] = z[k];
}

// x_prev = c1*x0 + c2*x_t + nonzero*exp(0.5*logvar)*eps      (spec_denoiser.py:86-101)
__global__ void __launch_bounds__(256) posterior_kernel(const float *x0, const float *x_t, const float *eps,
                                                        const float *coef4, int64_t coef_bs, float *x_prev,
                                                        int64_t per_batch, int64_t n, uint64_t seed,
                                                        uint64_t offset, const uint64_t *seed_delta) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
    if (seed_delta) seed += *seed_delta;
    float z[4];
    if (!eps) randn4(seed, offset + (uint64_t)q, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = q * 4 + k;
        if (i >= n) break;
        const float *cf = coef4 + (i / per_batch) * coef_bs;
        const float mean = cf[0] * x0[i] + cf[1] * x_t[i];
        const float e = eps ? eps[i] : z[k];
        x_prev[i] = mean + cf[3] * expf(0.5f * cf[2]) * e;
    }
}

__global__ void __launch_bounds__(256) q_sample_kernel(const float *x_start, const float *eps, const float *ab2,
                                                       const float *nonpad, float *x_t, int B, int M, int T) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * M * T) return;
    const int t = (int)(i % T);
    const int b = (int)(i / ((int64_t)M * T));
    float v = ab2[2 * b] * x_start[i] + ab2[2 * b + 1] * eps[i];
    if (nonpad) v *= nonpad[(int64_t)b * T + t];
    x_t[i] = v;
}

// ---- MFMA layout self test ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) selftest_mfma_kernel(float *max_err) {
    constexpr int K = 8;
    const int lane = threadIdx.x;
    auto Af = [](int i, int k) { return 0.25f * (float)((i * 7 + k * 3) % 11) - 1.0f; };
    auto Bf = [](int k, int j) { return 0.125f * (float)((k * 5 + j * 13) % 17) - 0.75f; };
    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += 2) {
        const int k = k0 + (lane >> 5);
        acc = mfma32(Af(lane & 31, k), Bf(k, lane & 31), acc);
    }
    float err = 0.0f;
    for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, lane), col = lane & 31;
        float ref = 0.0f;
        for (int k = 0; k < K; ++k) ref = fmaf(Af(row, k), Bf(k, col), ref);
        err = fmaxf(err, fabsf(ref - acc[r]));
    }
    for (int off = 32; off > 0; off >>= 1) err = fmaxf(err, __shfl_xor(err, off));
    if (lane == 0) *max_err = err;
}
}  // namespace

extern "C" int set_gate(const float *y, float *z, int32_t B, int32_t C, int32_t T, void *stream) {
    SET_REQUIRE(y && z && B > 0 && C > 0 && T > 0, "set_gate");
    hipLaunchKernelGGL(gate_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0, (hipStream_t)stream, y, z,
                       B, C, T);
    return set_check_launch("set_gate");
}
extern "C" int set_res_skip(const float *x_in, const float *o, float *x_out, float *skip, int32_t B, int32_t C,
                            int32_t T, int32_t first, void *stream) {
    SET_REQUIRE(x_in && o && x_out && skip && B > 0 && C > 0 && T > 0, "set_res_skip");
    hipLaunchKernelGGL(res_skip_kernel, dim3(set_blocks((int64_t)B * C * T, 256)), dim3(256), 0, (hipStream_t)stream,
                       x_in, o, x_out, skip, B, C, T, first);
    return set_check_launch("set_res_skip");
}
extern "C" int set_sinusoid_embed(const float *t, float *out, int32_t dim, int32_t n, void *stream) {
    SET_REQUIRE(t && out && dim >= 4 && (dim % 2) == 0 && n > 0, "set_sinusoid_embed");
    hipLaunchKernelGGL(sinusoid_kernel, dim3(set_blocks((int64_t)dim * n, 256)), dim3(256), 0, (hipStream_t)stream, t,
                       out, dim, n);
    return set_check_launch("set_sinusoid_embed");
}
static const uint64_t *g_seed_delta = nullptr;
const uint64_t *set_seed_delta_ptr() { return g_seed_delta; }
extern "C" int set_rng_seed_delta(const uint64_t *dev_word) {
    g_seed_delta = dev_word;
    return SET_OK;
}

extern "C" int set_randn(float *out, int64_t n, uint64_t seed, uint64_t offset, void *stream) {
    SET_REQUIRE(out && n > 0, "set_randn");
    hipLaunchKernelGGL(randn_kernel, dim3(set_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                       seed, offset, g_seed_delta);
    return set_check_launch("set_randn");
}
extern "C" int set_posterior_step(const float *x0, const float *x_t, const float *eps, const float *coef4,
                                  int64_t coef_bs, float *x_prev, int32_t B, int64_t per_batch, uint64_t seed,
                                  uint64_t offset, void *stream) {
    SET_REQUIRE(x0 && x_t && coef4 && x_prev && B > 0 && per_batch > 0, "set_posterior_step");
    const int64_t n = (int64_t)B * per_batch;
    hipLaunchKernelGGL(posterior_kernel, dim3(set_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, x0,
                       x_t, eps, coef4, coef_bs, x_prev, per_batch, n, seed, offset, g_seed_delta);
    return set_check_launch("set_posterior_step");
}
extern "C" int set_q_sample(const float *x_start, const float *eps, const float *ab2, const float *nonpad, float *x_t,
                            int32_t B, int32_t M, int32_t T, void *stream) {
    SET_REQUIRE(x_start && eps && ab2 && x_t && B > 0 && M > 0 && T > 0, "set_q_sample");
    hipLaunchKernelGGL(q_sample_kernel, dim3(set_blocks((int64_t)B * M * T, 256)), dim3(256), 0, (hipStream_t)stream,
                       x_start, eps, ab2, nonpad, x_t, B, M, T);
    return set_check_launch("set_q_sample");
}
extern "C" int set_selftest_mfma(float *max_err_host, void *stream) {
    SET_REQUIRE(max_err_host != nullptr, "set_selftest_mfma");
    float *d = nullptr;
    SET_HIP(hipMalloc(&d, sizeof(float)), "set_selftest_mfma");
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
    int rc = set_check_launch("set_selftest_mfma");
    if (rc == SET_OK) {
        hipError_t e = hipMemcpyAsync(max_err_host, d, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) rc = set_fail(SET_E_LAUNCH, "set_selftest_mfma", hipGetErrorString(e));
    }
    (void)hipFree(d);
    return rc;
}

// ----------------------------------------------------------------------------------------------------------
// the reverse loop: enqueue steps x (in-proj, L fused layers, skip-proj, out-proj, posterior)
// ----------------------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------------------
// Step boundary: everything between the layer stack of step k and the layer stack of step k+1 in ONE launch:
//   h   = ReLU(W_skip * (skip / sqrt(L)) + b_skip)            (diffnet.py:128-130)
//   x0  = W_out * h + b_out                                    (diffnet.py:131)
//   x'  = c1 x0 + c2 x_t + nonzero * exp(logvar/2) * eps       (spec_denoiser.py:86-101, eps explicit or Philox)
//   xin = ReLU(W_in * x' + b_in)                               (diffnet.py:118-120, input of the next step)
// One block = one utterance x 64 frames, 4 waves, everything stays in LDS/registers between the three GEMMs.  The
// four separate launches this replaces were latency-bound (31 + 75 + 51 + 14 us at B=32, T=800).  Weights are the
// ordinary packed conv images (set_pack_conv_weight); arithmetic order (prologue divide, bias after the sum, Philox
// quad = 4 consecutive frames of one row) equals the unfused kernels, so results are bit-identical to them.
// Needs T % 4 == 0 (quad alignment), 256 residual channels, M <= 96 mel bins.
// ----------------------------------------------------------------------------------------------------------
struct BoundaryArgs {
    const float *skip;      // [B][256][T]
    float *x;               // [B][M][T]  in: x_t, out: x_{t-1}
    const float *eps;       // [B][M][T] or NULL
    const float *coef4;     // {c1, c2, logvar, nonzero} of this step (device)
    const float *w_skip_p, *b_skip, *w_outp_p, *b_outp, *w_in_p, *b_in;
    float *xin_next;        // [B][256][T] or NULL (last step)
    float inv_div;          // unused (division by sqrt(L) is done exactly as the conv prologue does: x / p)
    float div;
    uint64_t seed, quad_offset;
    int T, M, MP;           // MP = M rounded up to 16 (rows of the x' tile in LDS, K of the head GEMM)
};
constexpr int BD_LD = 64;

__global__ void __launch_bounds__(256, 2) diffnet_boundary_kernel(BoundaryArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [256][64]: skip tile -> h tile -> x' tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, t0 = blockIdx.x * 64, T = a.T, M = a.M;
    // ---- phase 1: skip tile / sqrt(L) -> LDS (wave w: rows 64w .. 64w+63, lanes along t; unconditional clamped loads)
    {
        const rsrc_t rs = make_rsrc(a.skip + (int64_t)b * DC * T);
        const unsigned vo = 4u * (unsigned)min(t0 + lane, T - 1);
        const bool tv = t0 + lane < T;
        for (int r0 = 0; r0 < 64; r0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = buf_load(rs, vo, 4u * (unsigned)(64 * w + r0 + u) * (unsigned)T);
#pragma unroll
            for (int u = 0; u < 16; ++u) smem[(64 * w + r0 + u) * BD_LD + lane] = tv ? v[u] / a.div : 0.0f;
        }
    }
    __syncthreads();
    // ---- phase 2: h = ReLU(W_skip * s + b): wave w owns rows [64w, 64w+64) = row blocks 2w, 2w+1
    f32x16 acc[2][1][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc[i][0][0] = (f32x16){0};
        acc[i][0][1] = (f32x16){0};
        const float *wp = a.w_skip_p + (int64_t)(2 * w + i) * (DC / 2) * 64 + lane;
        const float *bp = smem + half * BD_LD + l31;
        gemm_groups<1, 2, 8>(acc[i], wp, bp, 2 * BD_LD, (DC / 2) / 8, [&](int) {
            wp += 8 * 64;
            bp += 8 * 2 * BD_LD;
        });
    }
    __syncthreads();  // every wave is done reading the skip tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * (2 * w + i) + mfma32_row(r, lane);
            const float bias = a.b_skip[row];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) smem[row * BD_LD + 32 * cb + l31] = fmaxf(acc[i][0][cb][r] + bias, 0.0f);
        }
    __syncthreads();
    // ---- phase 3: x0 = W_out * h + b: row blocks 0..ceil(M/32)-1 on waves 0..2
    const int rbn = (M + 31) / 32;
    f32x16 xo[1][2];
    xo[0][0] = (f32x16){0};
    xo[0][1] = (f32x16){0};
    if (w < rbn) {
        const float *wp = a.w_outp_p + (int64_t)w * (DC / 2) * 64 + lane;
        const float *bp = smem + half * BD_LD + l31;
        gemm_groups<1, 2, 8>(xo, wp, bp, 2 * BD_LD, (DC / 2) / 8, [&](int) {
            wp += 8 * 64;
            bp += 8 * 2 * BD_LD;
        });
    }
    __syncthreads();  // h tile consumed
    if (w < rbn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * w + mfma32_row(r, lane);
            const float bias = a.b_outp[min(row, M - 1)];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) smem[row * BD_LD + 32 * cb + l31] = row < M ? xo[0][cb][r] + bias : 0.0f;
        }
    }
    __syncthreads();
    // ---- phase 4: posterior update on quads of 4 consecutive frames (T % 4 == 0: a quad never straddles rows)
    {
        const float c1 = a.coef4[0], c2 = a.coef4[1], sig = a.coef4[3] * expf(0.5f * a.coef4[2]);
        float *xb = a.x + (int64_t)b * M * T;
        const float *eb = a.eps ? a.eps + (int64_t)b * M * T : nullptr;
        for (int qi = tid; qi < a.MP * 16; qi += 256) {
            const int m = qi >> 4, tq = qi & 15, t = t0 + 4 * tq;
            float *cell = smem + m * BD_LD + 4 * tq;
            if (m < M && t < T) {
                const int64_t i = (int64_t)m * T + t;
                const f32x4 xt = *reinterpret_cast<const f32x4 *>(xb + i);
                float z[4];
                if (eb) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(eb + i);
                    z[0] = e4[0]; z[1] = e4[1]; z[2] = e4[2]; z[3] = e4[3];
                } else {
                    randn4(a.seed, a.quad_offset + (uint64_t)(((int64_t)b * M * T + i) >> 2), z);
                }
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float mean = c1 * cell[k] + c2 * xt[k];
                    o[k] = mean + sig * z[k];
                }
                *reinterpret_cast<f32x4 *>(xb + i) = o;
                *reinterpret_cast<f32x4 *>(cell) = o;
            } else {
                *reinterpret_cast<f32x4 *>(cell) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};  // K padding rows / frames >= T
            }
        }
    }
    if (!a.xin_next) return;
    __syncthreads();
    // ---- phase 5: next step's input projection xin = ReLU(W_in * x' + b_in), K = MP
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc[i][0][0] = (f32x16){0};
        acc[i][0][1] = (f32x16){0};
        const float *wp = a.w_in_p + (int64_t)(2 * w + i) * (a.MP / 2) * 64 + lane;
        const float *bp = smem + half * BD_LD + l31;
        gemm_groups<1, 2, 4>(acc[i], wp, bp, 2 * BD_LD, (a.MP / 2) / 4, [&](int) {
            wp += 4 * 64;
            bp += 4 * 2 * BD_LD;
        });
    }
    const rsrc_t ro = make_rsrc(a.xin_next + (int64_t)b * DC * T);
    // all 32 bias values first: a bias load placed between the stores cannot be moved across them (b_in may alias
    // xin_next as far as the compiler knows), which serialises one L2 round trip per store
    float bin[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bin[i][r] = (a.b_in + 32 * (2 * w + i) + urow16(r))[4 * half];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (t0 + 32 * cb + l31 < T) {
            const unsigned so = 4u * (unsigned)(4 * half * T + t0 + 32 * cb + l31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ur = 32 * (2 * w + i) + urow16(r);  // wave-uniform; + 4*half rows in the lane offset
                    buf_store(fmaxf(acc[i][0][cb][r] + bin[i][r], 0.0f), ro, so, 4u * (unsigned)ur * (unsigned)T);
                }
        }
    }
}

// ---- the same step boundary on the two-piece fp16 operands (csrc/diffnet_x3.hip, csrc/conv_x2.hip: fp32 operands as two
// fp16 pieces, three fp16 MFMAs per product, fp32 accumulate).  The fp32 kernel above is bound by its three small GEMMs on
// the fp32 MFMA pipe (23 us of pipe time per 64-frame tile); here they take a fifth of that.  Same tile, same five phases;
// the operand tiles live in LDS as [piece][frame][channel] fp16 (rows padded by 16 B), the weights come from the images of
// set_pack_conv_weight_x2 (A-fragment order, straight from global memory), x0 / x' pass through an fp32 tile for the
// posterior update exactly as above.  Used by the reverse loop whenever the layer stack runs on two-piece fp16 operands.
// (operand types, bx_split / bx_mma / bx_gemm and the tile constants: csrc/boundary_x2.h)
struct BoundaryX2Args {
    BoundaryArgs g;
    const unsigned short *w_skip_x2, *w_outp_x2, *w_in_x2;
    int32_t *err_flag;
};

__global__ void __launch_bounds__(256, 2) diffnet_boundary_x2_kernel(BoundaryX2Args ax) {
    const BoundaryArgs &a = ax.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char bl[];  // [2][64][BX_XR]: s -> h pieces; x0 / x' (fp32) and x' pieces overlay
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, t0 = blockIdx.x * 64, T = a.T, M = a.M;
    const unsigned lane16 = 16u * (unsigned)lane;
    const unsigned T4 = 4u * (unsigned)T;
    float amax = 0.0f;
    // ---- phase 1: skip tile / sqrt(L), split -> LDS [piece][frame][256]: thread (frame f, 64 channels cg)
    {
        const int f = lane, cg = w;
        const rsrc_t rs = make_rsrc(a.skip + (int64_t)b * DC * T);
        const unsigned vo = 4u * (unsigned)min(t0 + f, T - 1);
        const bool tv = t0 + f < T;
        for (int c0 = 0; c0 < 64; c0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = buf_load(rs, vo, (unsigned)(64 * cg + c0 + u) * T4);
#pragma unroll
            for (int q8 = 0; q8 < 2; ++q8) {
                bx_u32x4 u0, u1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned short p0[2], p1[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float sv = v[8 * q8 + 2 * e + k] / a.div;
                        const float x = tv ? sv : 0.0f;
                        amax = fmaxf(amax, fabsf(x));
                        bx_split(x, p0[k], p1[k]);
                    }
                    u0[e] = (unsigned)p0[0] | ((unsigned)p0[1] << 16);
                    u1[e] = (unsigned)p1[0] | ((unsigned)p1[1] << 16);
                }
                *reinterpret_cast<bx_u32x4 *>(bl + f * BX_XR + (64 * cg + c0 + 8 * q8) * 2) = u0;
                *reinterpret_cast<bx_u32x4 *>(bl + BX_PIECE + f * BX_XR + (64 * cg + c0 + 8 * q8) * 2) = u1;
            }
        }
    }
    __syncthreads();
    auto bfrag256 = [&](int ks, int cb) { return (unsigned)((cb * 32 + l31) * BX_XR + (ks * 16 + half * 8) * 2); };
    // ---- phase 2: h = ReLU(W_skip s + b): wave w owns rows [64w, 64w+64)
    {
        const rsrc_t rw = make_rsrc(ax.w_skip_x2);
        const float inv = reinterpret_cast<const float *>(ax.w_skip_x2 + (DC / 32) * (DC / 16) * 1024)[1];
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[i][cb] = (f32x16){0};
        bx_gemm<2>(acc, rw, lane16, 2 * w, DC / 16, DC / 16, bl, BX_PIECE, bfrag256);
        __syncthreads();  // every wave is done reading the s tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    unsigned short p0[4], p1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 32 * (2 * w + i) + 8 * g + 4 * half + e;
                        const float h = fmaxf(acc[i][cb][4 * g + e] * inv + a.b_skip[row], 0.0f);
                        amax = fmaxf(amax, h);
                        bx_split(h, p0[e], p1[e]);
                    }
                    const unsigned off = (unsigned)((cb * 32 + l31) * BX_XR + (32 * (2 * w + i) + 8 * g + 4 * half) * 2);
                    bx_u32x2 u;
                    u[0] = (unsigned)p0[0] | ((unsigned)p0[1] << 16); u[1] = (unsigned)p0[2] | ((unsigned)p0[3] << 16);
                    *reinterpret_cast<bx_u32x2 *>(bl + off) = u;
                    u[0] = (unsigned)p1[0] | ((unsigned)p1[1] << 16); u[1] = (unsigned)p1[2] | ((unsigned)p1[3] << 16);
                    *reinterpret_cast<bx_u32x2 *>(bl + BX_PIECE + off) = u;
                }
    }
    __syncthreads();
    // ---- phase 3: x0 = W_out h + b: row blocks 0 .. ceil(M/32)-1 on waves 0..2; x0 -> fp32 tile xs[96][64] (over piece 0)
    float *xs = reinterpret_cast<float *>(bl);
    {
        const int rbn = (M + 31) / 32;
        f32x16 xo[1][2];
        xo[0][0] = (f32x16){0};
        xo[0][1] = (f32x16){0};
        const float inv = reinterpret_cast<const float *>(ax.w_outp_x2 + ((M + 31) / 32) * (DC / 16) * 1024)[1];
        if (w < rbn) bx_gemm<1>(xo, make_rsrc(ax.w_outp_x2), lane16, w, DC / 16, DC / 16, bl, BX_PIECE, bfrag256);
        __syncthreads();  // h tile consumed
        if (w < 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * w + mfma32_row(r, lane);
                const float bias = a.b_outp[min(row, M - 1)];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) xs[row * 64 + 32 * cb + l31] = (w < rbn && row < M) ? xo[0][cb][r] * inv + bias : 0.0f;
            }
        }
    }
    __syncthreads();
    // ---- phase 4: posterior update on quads of 4 consecutive frames (T % 4 == 0), as in the fp32 kernel
    {
        const float c1 = a.coef4[0], c2 = a.coef4[1], sig = a.coef4[3] * expf(0.5f * a.coef4[2]);
        float *xb = a.x + (int64_t)b * M * T;
        const float *eb = a.eps ? a.eps + (int64_t)b * M * T : nullptr;
        for (int qi = tid; qi < 96 * 16; qi += 256) {
            const int m = qi >> 4, tq = qi & 15, t = t0 + 4 * tq;
            float *cell = xs + m * 64 + 4 * tq;
            if (m < M && t < T) {
                const int64_t i = (int64_t)m * T + t;
                const f32x4 xt = *reinterpret_cast<const f32x4 *>(xb + i);
                float z[4];
                if (eb) {
                    const f32x4 e4 = *reinterpret_cast<const f32x4 *>(eb + i);
                    z[0] = e4[0]; z[1] = e4[1]; z[2] = e4[2]; z[3] = e4[3];
                } else {
                    randn4(a.seed, a.quad_offset + (uint64_t)(((int64_t)b * M * T + i) >> 2), z);
                }
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float mean = c1 * cell[k] + c2 * xt[k];
                    o[k] = mean + sig * z[k];
                }
                *reinterpret_cast<f32x4 *>(xb + i) = o;
                *reinterpret_cast<f32x4 *>(cell) = o;
            } else {
                *reinterpret_cast<f32x4 *>(cell) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};  // K padding rows / frames >= T
            }
        }
    }
    if (!a.xin_next) {
        if (!(amax < 32768.0f) && ax.err_flag) __hip_atomic_store(ax.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __syncthreads();
    // ---- x' (fp32 [96][64]) -> two fp16 pieces [frame][96] in the piece-1 region: thread (frame f, 24 channels cg)
    unsigned char *xp = bl + BX_PIECE;
    constexpr unsigned XP_PIECE = 64 * BX_PR;
    {
        const int f = lane, cg = w;
#pragma unroll
        for (int q8 = 0; q8 < 3; ++q8) {
            bx_u32x4 u0, u1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned short p0[2], p1[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float x = xs[(24 * cg + 8 * q8 + 2 * e + k) * 64 + f];
                    amax = fmaxf(amax, fabsf(x));
                    bx_split(x, p0[k], p1[k]);
                }
                u0[e] = (unsigned)p0[0] | ((unsigned)p0[1] << 16);
                u1[e] = (unsigned)p1[0] | ((unsigned)p1[1] << 16);
            }
            *reinterpret_cast<bx_u32x4 *>(xp + f * BX_PR + (24 * cg + 8 * q8) * 2) = u0;
            *reinterpret_cast<bx_u32x4 *>(xp + XP_PIECE + f * BX_PR + (24 * cg + 8 * q8) * 2) = u1;
        }
    }
    if (!(amax < 32768.0f) && ax.err_flag) __hip_atomic_store(ax.err_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // ---- phase 5: next step's input projection xin = ReLU(W_in x' + b_in), K = M rounded up to 32 (zero padded)
    {
        const int ngin = ((M + 31) / 32) * 2;  // 16-channel groups of the image (Cin = M rounded up to 32, zero padded)
        const float inv = reinterpret_cast<const float *>(ax.w_in_x2 + (DC / 32) * ngin * 1024)[1];
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[i][cb] = (f32x16){0};
        bx_gemm<2>(acc, make_rsrc(ax.w_in_x2), lane16, 2 * w, ngin, ngin, xp, XP_PIECE,
                   [&](int ks, int cb) { return (unsigned)((cb * 32 + l31) * BX_PR + (ks * 16 + half * 8) * 2); });
        const rsrc_t ro = make_rsrc(a.xin_next + (int64_t)b * DC * T);
        float bin[2][16];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bin[i][r] = (a.b_in + 32 * (2 * w + i) + urow16(r))[4 * half];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (t0 + 32 * cb + l31 < T) {
                const unsigned so = 4u * (unsigned)(4 * half * T + t0 + 32 * cb + l31);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ur = 32 * (2 * w + i) + urow16(r);
                        buf_store(fmaxf(acc[i][cb][r] * inv + bin[i][r], 0.0f), ro, so, 4u * (unsigned)ur * (unsigned)T);
                    }
            }
        }
    }
}

static bool boundary_fusable(const SetDiffLoopArgs &a) {
    if (const char *e = getenv("SET_AMD_FUSED_BOUNDARY"))
        if (atoi(e) == 0) return false;
    return a.T % 4 == 0 && a.M <= 96 && a.M >= 2 && ((a.M + 15) / 16 * 16 / 2) % 8 == 0;
}

static int launch_boundary(const SetDiffLoopArgs &a, int Bg, const float *skip, float *x, const float *eps, int sid,
                           uint64_t quad_offset, float *xin_next, bool x2, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_boundary_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "boundary(attr)");
        SET_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(diffnet_boundary_x2_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "boundary(attr)");
        attr_set = true;
    }
    BoundaryArgs g = {};
    g.skip = skip; g.x = x; g.eps = eps; g.coef4 = a.coef4 + 4 * sid;
    g.w_skip_p = a.w_skip_p; g.b_skip = a.b_skip; g.w_outp_p = a.w_outp_p; g.b_outp = a.b_outp;
    g.w_in_p = a.w_in_p; g.b_in = a.b_in; g.xin_next = xin_next;
    g.div = sqrtf((float)a.L); g.seed = a.seed; g.quad_offset = quad_offset;
    g.T = a.T; g.M = a.M; g.MP = (a.M + 15) / 16 * 16;
    if (x2) {
        BoundaryX2Args gx = {};
        gx.g = g;
        gx.w_skip_x2 = reinterpret_cast<const unsigned short *>(a.w_skip_x2);
        gx.w_outp_x2 = reinterpret_cast<const unsigned short *>(a.w_outp_x2);
        gx.w_in_x2 = reinterpret_cast<const unsigned short *>(a.w_in_x2);
        gx.err_flag = a.err_flag;
        hipLaunchKernelGGL(diffnet_boundary_x2_kernel, dim3((a.T + 63) / 64, Bg), dim3(256), (size_t)2 * BX_PIECE, s, gx);
    } else {
        hipLaunchKernelGGL(diffnet_boundary_kernel, dim3((a.T + 63) / 64, Bg), dim3(256), (size_t)DC * BD_LD * sizeof(float), s, g);
    }
    return set_check_launch("set_diffusion_loop(boundary)");
}

static SetConv1dArgs conv1x1_args(const float *in, const float *wp, const float *bias, float *out, int B, int Cin,
                                  int Cout, int T) {
    SetConv1dArgs c = {};
    c.in = in; c.w = wp; c.bias = bias; c.out = out;
    c.in_bs = (int64_t)Cin * T; c.in_cs = T; c.out_bs = (int64_t)Cout * T; c.out_cs = T;
    c.B = B; c.Cin = Cin; c.Cout = Cout; c.K = 1; c.dil = 1; c.pad = 0;
    c.T_in = T; c.T_iter = T; c.T_out = T; c.out_stride = 1; c.out_off = 0;
    c.alpha = 1.0f; c.impl = SET_IMPL_MFMA;
    return c;
}

// auxiliary streams for utterance groups (created once, never destroyed)
static hipStream_t g_aux_streams[8] = {nullptr};
static int aux_stream(int i, hipStream_t *out) {
    if (!g_aux_streams[i]) SET_HIP(hipStreamCreateWithFlags(&g_aux_streams[i], hipStreamNonBlocking), "aux stream");
    *out = g_aux_streams[i];
    return SET_OK;
}

// enqueue the chain of one utterance group [b0, b0+Bg) on stream s
static int diffusion_chain(const SetDiffLoopArgs &a, int g, int b0, int Bg, hipStream_t s, hipEvent_t *ev) {
    const int T = a.T, M = a.M, L = a.L;
    const int64_t per_batch = (int64_t)M * T;
    float *x = a.x + (int64_t)b0 * per_batch;
    float *ws_x0 = a.ws_x0 + (int64_t)b0 * DC * T, *ws_x1 = a.ws_x1 + (int64_t)b0 * DC * T;
    float *ws_skip = a.ws_skip + (int64_t)b0 * DC * T, *ws_h = a.ws_h + (int64_t)b0 * DC * T;
    float *ws_x0pred = a.ws_x0pred + (int64_t)b0 * per_batch;
    const float *condproj = a.condproj ? a.condproj + (int64_t)b0 * L * 512 * T : nullptr;
    const int tiles_per_utt = (T + NT - 1) / NT;
    int32_t *sync_ws = a.sync_ws ? a.sync_ws + 16 * (int64_t)g + 2 * (int64_t)b0 * ((T + 31) / 32) : nullptr;  // per-group slice
    const uint64_t quads_before = (uint64_t)((int64_t)b0 * per_batch / 4);
    const uint64_t quads_total = (uint64_t)(((int64_t)a.B * per_batch + 3) / 4);
    int rc = SET_OK;
    const bool fused_boundary = boundary_fusable(a);
    const bool bf16_loop = a.img16_all != nullptr;
    // the step boundary on two-piece fp16 operands whenever the layer stack runs on them (same splitting, same range guard)
    static int n_cu_chain = 0;
    if (!n_cu_chain) {
        int dev = 0;
        SET_HIP(hipGetDevice(&dev), "set_diffusion_loop");
        SET_HIP(hipDeviceGetAttribute(&n_cu_chain, hipDeviceAttributeMultiprocessorCount, dev), "set_diffusion_loop");
    }
    bool boundary_x2 = false;
    if (a.persistent && !bf16_loop && fused_boundary && a.w_skip_x2 && a.w_outp_x2 && a.w_in_x2 && a.wx3_all && a.x3_mode == 2 &&
        a.M <= 96) {
        const int v = stack_variant(Bg, T, a.dilation_cycle_length, a.w1w_all && a.w2w_all, a.w1s_all && a.w2s_all && a.z_ws,
                                    a.x3_mode, n_cu_chain);
        boundary_x2 = v == 5 || (v == 3 && !(getenv("SET_AMD_SPLIT_F32") && atoi(getenv("SET_AMD_SPLIT_F32")) != 0));
    }
    // the bf16-operand loop takes the split-operand boundary whenever its images are given (round 4: 85 -> 37 us per step at B = 32,
    // T = 800; it is the fp32-equivalent one, and it raises the same range word, which the caller must read)
    if (bf16_loop && fused_boundary && a.w_skip_x2 && a.w_outp_x2 && a.w_in_x2 && a.M <= 96) boundary_x2 = true;
    if (const char *e = getenv("SET_AMD_BOUNDARY_X2")) boundary_x2 = boundary_x2 && atoi(e) != 0;
    for (int k = 0; k < a.steps && rc == SET_OK; ++k) {
        const int sid = a.steps - 1 - k;  // diffusion step id t = steps-1 .. 0 (spec_denoiser.py:181)
        // input projection + ReLU (diffnet.py:118-120); with the fused boundary it is part of the previous step's
        // boundary launch
        if (!fused_boundary || k == 0) {
            SetConv1dArgs cin = conv1x1_args(x, a.w_in_p, a.b_in, ws_x0, Bg, M, DC, T);
            cin.act = SET_ACT_RELU;
            rc = set_conv1d(&cin, s);
            if (rc != SET_OK) break;
        }
        float *cur = ws_x0, *nxt = ws_x1;
        if (ev) (void)hipEventRecord(ev[2 * k], s);
        if (a.persistent && !bf16_loop) {
            SetDiffnetStackArgs sa = {};
            sa.xa = ws_x0; sa.xb = ws_x1; sa.skip = ws_skip;
            sa.condproj = condproj; sa.cp_bs = (int64_t)L * 512 * T; sa.cp_ls = (int64_t)512 * T;
            sa.dstep = a.dstep + sid; sa.d_bs = 0; sa.d_cs = a.steps; sa.d_ls = (int64_t)DC * a.steps;
            sa.w1p_all = a.w1p_all; sa.w2p_all = a.w2p_all; sa.b_dil_all = a.b_dil_all; sa.b_out_all = a.b_out_all;
            sa.w1w_all = a.w1w_all; sa.w2w_all = a.w2w_all;
            sa.w1s_all = a.w1s_all; sa.w2s_all = a.w2s_all; sa.wx3_all = a.wx3_all; sa.x3_mode = a.x3_mode;
            sa.z_ws = a.z_ws ? a.z_ws + (int64_t)b0 * DC * 32 * ((T + 31) / 32) : nullptr;
            sa.err_flag = a.err_flag;
            sa.sync_ws = sync_ws;
            sa.B = Bg; sa.T = T; sa.L = L; sa.dilation_cycle_length = a.dilation_cycle_length;
            rc = set_diffnet_stack(&sa, s);
        }
        // opt-in bf16-operand layers, `fuse` per launch when the workspace is there (csrc/diffnet_bf16.hip: the tile stays on chip
        // between the layers of a group)
        int fuse = 1;
        if (bf16_loop && a.bf16_ws && a.dilation_cycle_length <= 2) {
            fuse = set_diffnet_layers_bf16_plan(Bg, T, L, a.dilation_cycle_length);  // 10 (128-frame tiles fill the chip) or 5
            if (const char *e = getenv("SET_AMD_BF16_FUSE")) fuse = atoi(e) < 1 ? 1 : (atoi(e) > 16 ? 16 : atoi(e));
            if (fuse > 1 && a.bf16_ws_floats < set_diffnet_layers_bf16_scratch_floats(a.B, T, 0, fuse, a.dilation_cycle_length)) fuse = 1;
        }
        for (int l = 0; l < L && rc == SET_OK && bf16_loop && fuse > 1; l += fuse) {
            SetDiffnetLayersBf16Args fa = {};
            fa.x_in = cur; fa.x_out = nxt; fa.skip = ws_skip;
            fa.cond = a.cond + (int64_t)b0 * 192 * T;
            fa.dstep = a.dstep + sid; fa.d_bs = 0; fa.d_cs = a.steps; fa.d_ls = (int64_t)DC * a.steps;
            fa.img = reinterpret_cast<const uint16_t *>(a.img16_all) + (int64_t)l * set_diffnet_layer_bf16_image_size();
            fa.b_dil = a.b_dil_all + (int64_t)l * 512; fa.b_cond = a.b_cond_all + (int64_t)l * 512; fa.b_out = a.b_out_all + (int64_t)l * 512;
            // utterance groups (n_groups > 1) run concurrently on their own streams and the 128-frame kernel indexes its private skip
            // rows by (blockIdx.y, blockIdx.x) of its own launch: every group gets its own slice (per-utterance floats do not depend on B)
            const int64_t per_utt = set_diffnet_layers_bf16_scratch_floats(1, T, 0, fuse, a.dilation_cycle_length);
            fa.scratch = a.bf16_ws + (int64_t)b0 * per_utt; fa.scratch_floats = (int64_t)Bg * per_utt;
            fa.B = Bg; fa.T = T; fa.l0 = l; fa.nl = L - l < fuse ? L - l : fuse; fa.dilation_cycle_length = a.dilation_cycle_length;
            fa.first = (l == 0);
            rc = set_diffnet_layers_fwd_bf16(&fa, s);
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        for (int l = 0; l < L && rc == SET_OK && bf16_loop && fuse == 1; ++l) {
            // one launch per layer: conditioner projection inside the layer GEMM
            SetDiffnetLayerBf16Args la = {};
            la.x_in = cur; la.x_out = nxt; la.skip = ws_skip;
            la.cond = a.cond + (int64_t)b0 * 192 * T;
            la.dstep = a.dstep + (int64_t)l * DC * a.steps + sid;
            la.d_bs = 0; la.d_cs = a.steps;
            la.img = reinterpret_cast<const uint16_t *>(a.img16_all) + (int64_t)l * set_diffnet_layer_bf16_image_size();
            la.b_dil = a.b_dil_all + (int64_t)l * 512; la.b_cond = a.b_cond_all + (int64_t)l * 512;
            la.b_out = a.b_out_all + (int64_t)l * 512;
            la.B = Bg; la.T = T; la.dil = 1 << (l % a.dilation_cycle_length); la.first = (l == 0);
            rc = set_diffnet_layer_fwd_bf16(&la, s);
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        for (int l = 0; l < L && rc == SET_OK && !a.persistent && !bf16_loop; ++l) {
            SetDiffnetLayerArgs la = {};
            la.x_in = cur; la.x_out = nxt; la.skip = ws_skip;
            la.condproj = condproj + (int64_t)l * 512 * T;
            la.cp_bs = (int64_t)L * 512 * T;
            la.dstep = a.dstep + (int64_t)l * DC * a.steps + sid;
            la.d_bs = 0; la.d_cs = a.steps;
            la.w1p = a.w1p_all + (int64_t)l * (512 * 768); la.b_dil = a.b_dil_all + (int64_t)l * 512;
            la.w2p = a.w2p_all + (int64_t)l * (512 * 256); la.b_out = a.b_out_all + (int64_t)l * 512;
            la.B = Bg; la.T = T; la.dil = 1 << (l % a.dilation_cycle_length); la.first = (l == 0);
            rc = set_diffnet_layer(&la, s);
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        if (ev) (void)hipEventRecord(ev[2 * k + 1], s);
        if (rc != SET_OK) break;
        const float *eps = a.noise ? a.noise + (int64_t)k * a.B * per_batch + (int64_t)b0 * per_batch : nullptr;
        if (fused_boundary) {
            // only the skip sum feeds the output head (diffnet.py:128); the next step's stack input buffer is ws_x0
            rc = launch_boundary(a, Bg, ws_skip, x, eps, sid, (uint64_t)(k + 1) * quads_total + quads_before,
                                 k + 1 < a.steps ? ws_x0 : nullptr, boundary_x2, s);
            continue;
        }
        // skip sum / sqrt(L) -> skip_projection -> ReLU -> output_projection (diffnet.py:128-131)
        SetConv1dArgs cs = conv1x1_args(ws_skip, a.w_skip_p, a.b_skip, ws_h, Bg, DC, DC, T);
        cs.pro = SET_PRO_DIV; cs.pro_param = sqrtf((float)L); cs.act = SET_ACT_RELU;
        rc = set_conv1d(&cs, s);
        if (rc != SET_OK) break;
        SetConv1dArgs co = conv1x1_args(ws_h, a.w_outp_p, a.b_outp, ws_x0pred, Bg, DC, M, T);
        rc = set_conv1d(&co, s);
        if (rc != SET_OK) break;
        // Philox counters are global element quads, so the noise does not depend on the grouping
        rc = set_posterior_step(ws_x0pred, x, eps, a.coef4 + 4 * sid, 0, x, Bg, per_batch, a.seed,
                                (uint64_t)(k + 1) * quads_total + quads_before, s);
    }
    return rc;
}

extern "C" int set_diffusion_loop(const SetDiffLoopArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_diffusion_loop");
    const SetDiffLoopArgs &a = *args;
    SET_REQUIRE(a.B > 0 && a.T > 0 && a.M > 0 && a.L > 0 && a.steps > 0 && a.dilation_cycle_length >= 1,
                "set_diffusion_loop");
    SET_REQUIRE(a.x && a.dstep && a.coef4 && a.w_in_p && a.b_in && a.b_dil_all && a.b_out_all && a.w_skip_p && a.b_skip &&
                    a.w_outp_p && a.b_outp, "set_diffusion_loop");
    if (a.img16_all) {
        SET_REQUIRE(a.cond && a.b_cond_all, "set_diffusion_loop(bf16 loop needs cond and b_cond_all)");
    } else {
        SET_REQUIRE(a.condproj && a.w1p_all && a.w2p_all, "set_diffusion_loop");
        SET_REQUIRE(!a.persistent || a.sync_ws, "set_diffusion_loop(persistent needs sync_ws)");
    }
    SET_REQUIRE(a.ws_x0 && a.ws_x1 && a.ws_skip && a.ws_h && a.ws_x0pred, "set_diffusion_loop");
    hipStream_t s = (hipStream_t)stream;
    const int64_t per_batch = (int64_t)a.M * a.T;
    int G = a.n_groups < 1 ? 1 : (a.n_groups > 8 ? 8 : a.n_groups);
    if (G > a.B) G = a.B;
    if (per_batch % 4 != 0) G = 1;  // group slices must start on a Philox quad boundary
    const bool timing = a.layer_span_ms != nullptr;
    hipEvent_t *ev = nullptr;
    if (timing) {
        ev = new hipEvent_t[(size_t)2 * a.steps * G];
        for (int i = 0; i < 2 * a.steps * G; ++i) SET_HIP(hipEventCreate(&ev[i]), "set_diffusion_loop(event)");
    }
    hipEvent_t loop_ev[2] = {nullptr, nullptr};
    if (a.loop_ms) {
        SET_HIP(hipEventCreate(&loop_ev[0]), "set_diffusion_loop(event)");
        SET_HIP(hipEventCreate(&loop_ev[1]), "set_diffusion_loop(event)");
        (void)hipEventRecord(loop_ev[0], s);
    }
    int rc = SET_OK;
    if (G == 1) {
        rc = diffusion_chain(a, 0, 0, a.B, s, ev);
    } else {
        hipEvent_t fork = nullptr, join[8] = {nullptr};
        SET_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming), "set_diffusion_loop(fork)");
        SET_HIP(hipEventRecord(fork, s), "set_diffusion_loop(fork)");
        for (int g = 0; g < G && rc == SET_OK; ++g) {
            const int b0 = (int)((int64_t)a.B * g / G), b1 = (int)((int64_t)a.B * (g + 1) / G);
            hipStream_t sg;
            rc = aux_stream(g, &sg);
            if (rc != SET_OK) break;
            SET_HIP(hipStreamWaitEvent(sg, fork, 0), "set_diffusion_loop(fork wait)");
            rc = diffusion_chain(a, g, b0, b1 - b0, sg, ev ? ev + (size_t)2 * a.steps * g : nullptr);
            SET_HIP(hipEventCreateWithFlags(&join[g], hipEventDisableTiming), "set_diffusion_loop(join)");
            SET_HIP(hipEventRecord(join[g], sg), "set_diffusion_loop(join)");
            SET_HIP(hipStreamWaitEvent(s, join[g], 0), "set_diffusion_loop(join wait)");
        }
        (void)hipEventDestroy(fork);
        for (int g = 0; g < G; ++g)
            if (join[g]) (void)hipEventDestroy(join[g]);
    }
    if (a.loop_ms) (void)hipEventRecord(loop_ev[1], s);
    if (timing || a.loop_ms) {
        if (rc == SET_OK) {
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) rc = set_fail(SET_E_LAUNCH, "set_diffusion_loop(sync)", hipGetErrorString(e));
        }
        if (rc == SET_OK && timing) {
            for (int k = 0; k < a.steps; ++k) {
                float acc_ms = 0.0f;
                for (int g = 0; g < G; ++g) {
                    float ms = 0.0f;
                    (void)hipEventElapsedTime(&ms, ev[(size_t)2 * a.steps * g + 2 * k], ev[(size_t)2 * a.steps * g + 2 * k + 1]);
                    acc_ms += ms;
                }
                a.layer_span_ms[k] = acc_ms / (float)G;
            }
        }
        if (rc == SET_OK && a.loop_ms) (void)hipEventElapsedTime(a.loop_ms, loop_ev[0], loop_ev[1]);
    }
    if (ev) {
        for (int i = 0; i < 2 * a.steps * G; ++i) (void)hipEventDestroy(ev[i]);
        delete[] ev;
    }
    if (loop_ev[0]) { (void)hipEventDestroy(loop_ev[0]); (void)hipEventDestroy(loop_ev[1]); }
    return rc;
}

extern "C" int64_t set_sizeof_conv1d_args(void) { return (int64_t)sizeof(SetConv1dArgs); }
extern "C" int64_t set_sizeof_diffnet_layer_args(void) { return (int64_t)sizeof(SetDiffnetLayerArgs); }
extern "C" int64_t set_sizeof_diff_loop_args(void) { return (int64_t)sizeof(SetDiffLoopArgs); }
