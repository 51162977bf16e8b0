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
#include "common.h"

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

__global__ void __launch_bounds__(256, 2) diffnet_layer_kernel(SetDiffnetLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * NT;
    const int dil = a.dil;
    const int XW = NT + 2 * dil;  // tile width incl. halo
    const int T = a.T;
    const float *xin = a.x_in + (int64_t)b * DC * T;
    uint64_t *dbg = a.dbg_clock ? a.dbg_clock + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define PHASE_STAMP(i) \
    if (dbg && tid == 0) dbg[i] = __builtin_amdgcn_s_memtime();
    PHASE_STAMP(0)

    // NB every global load below is UNCONDITIONAL on a clamped (always in-bounds) address and the validity
    // select happens afterwards: a `cond ? load : 0` makes hipcc branch around each load and drain vmcnt(0)
    // per element (128 serialized round trips per lane).
    const int tc0 = min(t0 + l31, T - 1), tc1 = min(t0 + 32 + l31, T - 1);
    const bool tv0 = t0 + l31 < T, tv1 = t0 + 32 + l31 < T;
    const unsigned lo0 = (unsigned)(4 * half * T + tc0), lo1 = (unsigned)(4 * half * T + tc1);  // per-lane offsets
    const unsigned lb = (unsigned)(4 * half);

    // ---- phase 0a: accumulators of GEMM 1 start at  b_dil + condproj  (so the gate needs no loads later);
    //      all 128 loads per lane are independent and issued together.
    f32x16 acc[4][2];
    {
        const float *cpb = a.condproj + (int64_t)b * a.cp_bs;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ur = layer_row(w, rb, urow16(r));  // wave-uniform
                const float bias = (a.b_dil + ur)[lb];
                const float *cr = cpb + (int64_t)ur * T;
                acc[rb][0][r] = bias + cr[lo0];
                acc[rb][1][r] = bias + cr[lo1];
            }
    }
    // ---- phase 0b: stage the (x + d) tile: wave w owns channels [64w, 64w+64); one row = one coalesced
    //      64-lane load (+ a halo load); 16 rows in flight.  Zero outside [0,T): the conv's zero padding
    //      applies to x + d (diffnet.py:71,74).
    {
        const int tA = t0 - dil + lane;       // columns 0..63
        const int tB = t0 - dil + 64 + lane;  // columns 64..XW-1 (lanes < 2*dil)
        const bool vA = tA >= 0 && tA < T;
        const bool vB = tB >= 0 && tB < T;
        const unsigned cA = (unsigned)min(max(tA, 0), T - 1), cB = (unsigned)min(max(tB, 0), T - 1);
        const bool haloLane = lane < 2 * dil;
        for (int r0 = 0; r0 < 64; r0 += 16) {
            float xa[16], xb[16], dd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 64 * w + r0 + u;
                const float *row = xin + (int64_t)c * T;
                xa[u] = row[cA];
                xb[u] = row[cB];
                dd[u] = a.dstep[(int64_t)b * a.d_bs + (int64_t)c * a.d_cs];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = 64 * w + r0 + u;
                smem[c * XW + lane] = vA ? xa[u] + dd[u] : 0.0f;
                if (haloLane) smem[c * XW + 64 + lane] = vB ? xb[u] + dd[u] : 0.0f;
            }
        }
    }
    __syncthreads();
    PHASE_STAMP(1)

    // ---- phase 1: GEMM 1  y[512 x 64] += Wdil[512 x 768] * im2col(xs) ------------------------------------
    // k-steps are processed in groups of 4 (32 MFMAs = 2048 cycles/SIMD).  The operands of the next group
    // (A: 4 x dwordx4 from L2, B: 4 x ds_read2) are issued before the current group's MFMAs and pinned there
    // with sched_barrier so the scheduler cannot sink them next to their use.
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w1p) + (int64_t)w * KS1 * 64 + lane;
        const float *bp = smem + half * XW + l31;
        const int rstep = 2 * XW;  // LDS floats between consecutive k-steps (2 channels)
        f32x4 A[4], nA[4];
        float Bv[4][2], nB[4][2];
        auto load_ops = [&](f32x4(&dA)[4], float(&dB)[4][2]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dA[u] = wp[u * 64];
                dB[u][0] = bp[u * rstep];
                dB[u][1] = bp[u * rstep + 32];
            }
        };
        auto compute = [&](const f32x4(&cA)[4], const float(&cB)[4][2]) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r][0] = mfma32(cA[u][r], cB[u][0], acc[r][0]);
                    acc[r][1] = mfma32(cA[u][r], cB[u][1], acc[r][1]);
                }
        };
        constexpr int NG = KS1 / 4;  // 96 groups; group g: tap = g / 32, channel pairs (g % 32) * 4 ..
        load_ops(A, Bv);
        for (int g = 0; g < NG - 1; ++g) {
            wp += 4 * 64;
            bp += 4 * rstep;
            if ((g & 31) == 31) bp += dil - 32 * 4 * rstep;  // next tap: back to channel 0, shift by dil columns
            load_ops(nA, nB);
            __builtin_amdgcn_sched_barrier(0);
            compute(A, Bv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                A[u] = nA[u];
                Bv[u][0] = nB[u][0];
                Bv[u][1] = nB[u][1];
            }
        }
        compute(A, Bv);
    }
    PHASE_STAMP(2)

    // ---- phase 2: gate, lane-local (acc[rb] pairs with acc[rb+2]); bias + conditioner are already inside -----
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z0 = fast_sigmoid(acc[rb][0][r]) * fast_tanh(acc[rb + 2][0][r]);
            const float z1 = fast_sigmoid(acc[rb][1][r]) * fast_tanh(acc[rb + 2][1][r]);
            acc[rb][0][r] = tv0 ? z0 : 0.0f;
            acc[rb][1][r] = tv1 ? z1 : 0.0f;
        }
    PHASE_STAMP(3)
    __syncthreads();  // every wave is done reading xs
    // z tile zs[256][64] overlays the xs tile
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 64 * w + 32 * rb + mfma32_row(r, lane);
                smem[c * NT + cb * 32 + l31] = acc[rb][cb][r];
            }
    // ---- accumulators of GEMM 2 start at  x_in + b_out  (residual rows) /  skip + b_out  (skip rows): the
    //      epilogue is then store-only and these 128 loads fly while the other waves finish their z stores.
    __builtin_amdgcn_sched_barrier(0);  // z registers are dead from here: keep the init loads below the z stores
    float *skp = a.skip + (int64_t)b * DC * T;
    const bool first = a.first != 0;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ur = layer_row(w, rb, urow16(r));  // wave-uniform
            const float bias = (a.b_out + ur)[lb];
            // residual rows read x_in, skip rows read the running skip sum (ignored by a select when first)
            const float *src = (rb < 2 ? xin + (int64_t)ur * T : skp + (int64_t)(ur - DC) * T);
            const float s0 = src[lo0], s1 = src[lo1];
            acc[rb][0][r] = (rb >= 2 && first) ? bias : bias + s0;
            acc[rb][1][r] = (rb >= 2 && first) ? bias : bias + s1;
        }
        if (rb == 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    PHASE_STAMP(4)

    // ---- phase 3: GEMM 2  o[512 x 64] += Wout[512 x 256] * zs ------------------------------------------------
    {
        const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w2p) + (int64_t)w * KS2 * 64 + lane;
        const float *bp = smem + half * NT + l31;
        constexpr int rstep = 2 * NT;
        f32x4 A[4], nA[4];
        float Bv[4][2], nB[4][2];
        auto load_ops = [&](f32x4(&dA)[4], float(&dB)[4][2]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                dA[u] = wp[u * 64];
                dB[u][0] = bp[u * rstep];
                dB[u][1] = bp[u * rstep + 32];
            }
        };
        auto compute = [&](const f32x4(&cA)[4], const float(&cB)[4][2]) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r][0] = mfma32(cA[u][r], cB[u][0], acc[r][0]);
                    acc[r][1] = mfma32(cA[u][r], cB[u][1], acc[r][1]);
                }
        };
        constexpr int NG = KS2 / 4;
        load_ops(A, Bv);
        for (int g = 0; g < NG - 1; ++g) {
            wp += 4 * 64;
            bp += 4 * rstep;
            load_ops(nA, nB);
            __builtin_amdgcn_sched_barrier(0);
            compute(A, Bv);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                A[u] = nA[u];
                Bv[u][0] = nB[u][0];
                Bv[u][1] = nB[u][1];
            }
        }
        compute(A, Bv);
    }
    PHASE_STAMP(5)

    // ---- phase 4: store-only epilogue: x_out = (x + o_res) * 2^-1/2 ; skip = skip + o_skip ----------------------
    float *xout = a.x_out + (int64_t)b * DC * T;
    const unsigned so0 = (unsigned)(4 * half * T + t0 + l31), so1 = so0 + 32u;  // unclamped store offsets
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (cb == 0 ? tv0 : tv1) {  // one exec-mask region per column block, not one per store
            const unsigned so = cb == 0 ? so0 : so1;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ur = layer_row(w, rb, urow16(r));  // wave-uniform
                    float *dst = (rb < 2 ? xout + (int64_t)ur * T : skp + (int64_t)(ur - DC) * T);
                    const float sc = rb < 2 ? 0.70710678118654752440f : 1.0f;
                    dst[so] = acc[rb][cb][r] * sc;
                }
        }
    }
    PHASE_STAMP(6)
#undef PHASE_STAMP
}

__global__ void __launch_bounds__(256) pack_diffnet_layer_kernel(const float *w_dil, const float *w_out, float *w1p,
                                                                 float *w2p) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n1 = (int64_t)4 * KS1 * 64 * 4, n2 = (int64_t)4 * KS2 * 64 * 4;
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

}  // namespace

extern "C" int64_t set_diffnet_w1p_size(void) { return (int64_t)512 * 768; }
extern "C" int64_t set_diffnet_w2p_size(void) { return (int64_t)512 * 256; }

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

// ---- Philox4x32-10 (counter based) + Box-Muller ------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void randn4(uint64_t seed, uint64_t ctr, float out[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // (0,1] uniforms, Box-Muller
    const float u0 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u1 = ((float)(c[1] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c[2] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u3 = ((float)(c[3] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.28318530717958647692f * u1, &s0, &c0);
    sincosf(6.28318530717958647692f * u3, &s1, &c1);
    out[0] = r0 * c0; out[1] = r0 * s0; out[2] = r1 * c1; out[3] = r1 * s1;
}

__global__ void __launch_bounds__(256) randn_kernel(float *out, int64_t n, uint64_t seed, uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // quad index
    if (q * 4 >= n) return;
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
                                                        uint64_t offset) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q * 4 >= n) return;
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
extern "C" int set_randn(float *out, int64_t n, uint64_t seed, uint64_t offset, void *stream) {
    SET_REQUIRE(out && n > 0, "set_randn");
    hipLaunchKernelGGL(randn_kernel, dim3(set_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                       seed, offset);
    return set_check_launch("set_randn");
}
extern "C" int set_posterior_step(const float *x0, const float *x_t, const float *eps, const float *coef4,
                                  int64_t coef_bs, float *x_prev, int32_t B, int64_t per_batch, uint64_t seed,
                                  uint64_t offset, void *stream) {
    SET_REQUIRE(x0 && x_t && coef4 && x_prev && B > 0 && per_batch > 0, "set_posterior_step");
    const int64_t n = (int64_t)B * per_batch;
    hipLaunchKernelGGL(posterior_kernel, dim3(set_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, x0,
                       x_t, eps, coef4, coef_bs, x_prev, per_batch, n, seed, offset);
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
static int diffusion_chain(const SetDiffLoopArgs &a, int b0, int Bg, hipStream_t s, hipEvent_t *ev) {
    const int T = a.T, M = a.M, L = a.L;
    const int64_t per_batch = (int64_t)M * T;
    float *x = a.x + (int64_t)b0 * per_batch;
    float *ws_x0 = a.ws_x0 + (int64_t)b0 * DC * T, *ws_x1 = a.ws_x1 + (int64_t)b0 * DC * T;
    float *ws_skip = a.ws_skip + (int64_t)b0 * DC * T, *ws_h = a.ws_h + (int64_t)b0 * DC * T;
    float *ws_x0pred = a.ws_x0pred + (int64_t)b0 * per_batch;
    const float *condproj = a.condproj + (int64_t)b0 * L * 512 * T;
    const uint64_t quads_before = (uint64_t)((int64_t)b0 * per_batch / 4);
    const uint64_t quads_total = (uint64_t)(((int64_t)a.B * per_batch + 3) / 4);
    int rc = SET_OK;
    for (int k = 0; k < a.steps && rc == SET_OK; ++k) {
        const int sid = a.steps - 1 - k;  // diffusion step id t = steps-1 .. 0 (spec_denoiser.py:181)
        // input projection + ReLU (diffnet.py:118-120)
        SetConv1dArgs cin = conv1x1_args(x, a.w_in_p, a.b_in, ws_x0, Bg, M, DC, T);
        cin.act = SET_ACT_RELU;
        rc = set_conv1d(&cin, s);
        if (rc != SET_OK) break;
        float *cur = ws_x0, *nxt = ws_x1;
        if (ev) (void)hipEventRecord(ev[2 * k], s);
        for (int l = 0; l < L && rc == SET_OK; ++l) {
            SetDiffnetLayerArgs la = {};
            la.x_in = cur; la.x_out = nxt; la.skip = ws_skip;
            la.condproj = condproj + (int64_t)l * 512 * T;
            la.cp_bs = (int64_t)L * 512 * T;
            la.dstep = a.dstep + (int64_t)l * DC * a.steps + sid;
            la.d_bs = 0; la.d_cs = a.steps;
            la.w1p = a.w1p[l]; la.b_dil = a.b_dil[l]; la.w2p = a.w2p[l]; la.b_out = a.b_out[l];
            la.B = Bg; la.T = T; la.dil = 1 << (l % a.dilation_cycle_length); la.first = (l == 0);
            rc = set_diffnet_layer(&la, s);
            float *tmp = cur; cur = nxt; nxt = tmp;
        }
        if (ev) (void)hipEventRecord(ev[2 * k + 1], s);
        if (rc != SET_OK) break;
        // skip sum / sqrt(L) -> skip_projection -> ReLU -> output_projection (diffnet.py:128-131)
        SetConv1dArgs cs = conv1x1_args(ws_skip, a.w_skip_p, a.b_skip, ws_h, Bg, DC, DC, T);
        cs.pro = SET_PRO_DIV; cs.pro_param = sqrtf((float)L); cs.act = SET_ACT_RELU;
        rc = set_conv1d(&cs, s);
        if (rc != SET_OK) break;
        SetConv1dArgs co = conv1x1_args(ws_h, a.w_outp_p, a.b_outp, ws_x0pred, Bg, DC, M, T);
        rc = set_conv1d(&co, s);
        if (rc != SET_OK) break;
        const float *eps = a.noise ? a.noise + (int64_t)k * a.B * per_batch + (int64_t)b0 * per_batch : nullptr;
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
    SET_REQUIRE(a.x && a.condproj && a.dstep && a.coef4 && a.w_in_p && a.b_in && a.w1p && a.w2p && a.b_dil &&
                    a.b_out && a.w_skip_p && a.b_skip && a.w_outp_p && a.b_outp,
                "set_diffusion_loop");
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
        rc = diffusion_chain(a, 0, a.B, s, ev);
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
            rc = diffusion_chain(a, b0, b1 - b0, sg, ev ? ev + (size_t)2 * a.steps * g : nullptr);
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
