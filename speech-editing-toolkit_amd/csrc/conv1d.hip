// Generic stride-1 conv1d with fused prologue/epilogue: a naive one-thread-per-output kernel
// (any shape; device-side cross-check) and an implicit-GEMM kernel on v_mfma_f32_32x32x2_f32.
// Replaces every F.conv1d / nn.Linear / nn.ConvTranspose1d (polyphase) call on the hot path -- see
// include/set_amd.h for the reference citations.
#include "common.h"

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
    *o = a.accumulate ? (*o + v) : v;
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
__global__ void __launch_bounds__(256) conv1d_mfma_kernel(SetConv1dArgs a, int lo, int halo, int CinP, int RBn) {
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

    for (int c0 = 0; c0 < CinP; c0 += KC) {
        __syncthreads();  // previous chunk fully consumed
        for (int i = tid; i < KC * W; i += 256) {
            const int ci = i / W, j = i - ci * W;
            const int c = c0 + ci;
            const int ti = t0 + lo + j;
            float v = 0.0f;
            if (c < a.Cin && ti >= 0 && ti < a.T_in) {
                v = inb[(int64_t)c * a.in_cs + ti];
                if (a.in_chan_add) v += a.in_chan_add[(int64_t)b * a.Cin + c];
                v = dev_pro(v, a.pro, a.pro_param);
            }
            smem[i] = v;
        }
        __syncthreads();
        if (rb_valid) {
            for (int tap = 0; tap < a.K; ++tap) {
                const int off = tap * a.dil - a.pad - lo;  // >= 0
                const float *ap = a.w + (((int64_t)rb * a.K + tap) * cp_total + (c0 >> 1)) * 64 + lane;
                const float *bp = smem + half * W + wn * 64 + l31 + off;
#pragma unroll
                for (int cp = 0; cp < KC / 2; ++cp) {
                    const float av = ap[cp * 64];
                    const float b0 = bp[(2 * cp) * W];
                    const float b1 = bp[(2 * cp) * W + 32];
                    acc0 = mfma32(av, b0, acc0);
                    acc1 = mfma32(av, b1, acc1);
                }
            }
        }
    }
    if (!rb_valid) return;
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
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

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

extern "C" int set_conv1d(const SetConv1dArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_conv1d");
    const SetConv1dArgs &a = *args;
    SET_REQUIRE(a.in && a.w && a.out, "set_conv1d");
    SET_REQUIRE(a.B > 0 && a.Cin > 0 && a.Cout > 0 && a.K > 0 && a.T_in > 0 && a.T_out > 0, "set_conv1d");
    SET_REQUIRE(a.out_stride >= 1, "set_conv1d");
    if (a.T_iter <= 0) return SET_OK;
    hipStream_t s = (hipStream_t)stream;
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
    const int CinP = round_up(a.Cin, KC);
    const int RBn = (a.Cout + 31) / 32;
    dim3 block(256);
    if (RBn >= 4) {
        constexpr int WM = 4, WN = 1;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float);
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN>), grid, block, lds, s, a, lo, halo, CinP, RBn);
    } else if (RBn >= 2) {
        constexpr int WM = 2, WN = 2;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float);
        hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN>), grid, block, lds, s, a, lo, halo, CinP, RBn);
    } else {
        constexpr int WM = 1, WN = 4;
        dim3 grid((a.T_iter + 64 * WN - 1) / (64 * WN), (RBn + WM - 1) / WM, a.B);
        const size_t lds = (size_t)KC * (64 * WN + halo) * sizeof(float);
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
