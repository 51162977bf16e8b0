// Fused multi-head attention for the CampNet rows (modules/speech_editing/commons/transformer.py:361-406: bmm -> masked fp32
// softmax -> bmm, and torch's multi_head_attention_forward for the self-attention layers), forward and backward, on the
// [B][C][T] activation layout of the library (a head is a slice of D channels; Q / K / V / O are addressed through strides and
// never transposed or copied).  The score matrix never exists in HBM: a block walks the keys in tiles of 32 with an online
// softmax (running max m and sum l per query), the backward recomputes P from the saved softmax statistics (row max m and sum l).
//
// Geometry (forward and the dQ pass): block = 4 waves, wave w owns 32 QUERIES; everything is computed TRANSPOSED,
//     S^T [key][query] = K Q^T,   O^T [d][query] = V^T P^T,
// so that a query is a COLUMN of every accumulator = one lane pair (l31, l31 + 32): max / sum / rescale of the online softmax
// are lane-local (one exchange with lane ^ 32 per tile), and the accumulator of S^T IS the B operand of the second GEMM:
// register r of lane (l31, half) holds key (r & 3) + 8 (r >> 2) + 4 half, and the k-order of a contraction is free as long as A
// and B agree, so V is read in that key order and P never moves between lanes.
// Two operand types (template): fp32 (v_mfma_f32_32x32x2_f32: the parity path, exact fp32 products) and bf16
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate and softmax: the training rows' bf16 arithmetic, BASELINE configs[4]).
// K and V tiles are staged by the whole block: K transposed to [key][d], V as [d][key]; the next tile's global loads are in
// flight under the current tile's MFMAs.  Every lane-dependent address part (frame, the half's channel offset) sits in the
// VECTOR offset of the buffer instructions: a lane-varying SCALAR offset makes the compiler wrap each access in a waterfall
// loop (one serialised round trip per distinct value) -- the first version of this file ran 3 - 6 x slower for exactly that.
// (Measured alternative, dropped: one wave per block reading its operands straight from global memory in fragment order -- no
// LDS, no barrier, but 3 - 5 x the vector-memory instructions and ~250 live registers: 323 vs 171 us for the T = 800 backward.)
#include <math.h>

#include "common.h"

typedef __bf16 af_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned af_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned af_u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int AF_KT = 32;  // keys per tile

__device__ __forceinline__ unsigned af_pk(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
__device__ __forceinline__ f32x16 af_mma16(af_u32x4 a, af_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(af_bf16x8, a), __builtin_bit_cast(af_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float af_xor32(float v) { return __shfl_xor(v, 32, 64); }
// key of accumulator register r of a lane in half `h` (32x32 MFMA output layout)
__device__ __forceinline__ int af_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LDS tiles of one key tile.  fp32: Ks[key][D + 4], Vs[d][32 + 4] floats; bf16: Ks[key][D] (+ 16 B), Vs[d][32] (+ 16 B)
template <int D, bool BF16> struct AfTile {
    static constexpr int KROW = BF16 ? D * 2 + 16 : (D + 4) * 4;  // bytes
    static constexpr int VROW = BF16 ? AF_KT * 2 + 16 : (AF_KT + 4) * 4;
    static constexpr int KBYTES = AF_KT * KROW, VBYTES = D * VROW;
};

// block-cooperative staging of tile kt of a [D][T] fp32 slice (row stride cs) into registers / from registers into LDS
template <int D, int NT>
__device__ __forceinline__ void af_issue(float (&reg)[D * AF_KT / NT], rsrc_t src, int cs, int t0, int Tk, int tid) {
    const int key = tid & 31, dg = tid >> 5;
    const unsigned vo = 4u * (unsigned)(min(t0 + key, Tk - 1) + dg * cs);
#pragma unroll
    for (int i = 0; i < D * AF_KT / NT; ++i) reg[i] = buf_load(src, vo, 4u * (unsigned)(((NT / 32) * i) * cs));
}
// transposed: LDS[key][d]
template <int D, int NT, bool BF16>
__device__ __forceinline__ void af_commit_kd(const float (&reg)[D * AF_KT / NT], unsigned char *lds, int t0, int Tk, int tid) {
    const int key = tid & 31, dg = tid >> 5;
    const bool ok = t0 + key < Tk;
#pragma unroll
    for (int i = 0; i < D * AF_KT / NT; ++i) {
        const int d = dg + (NT / 32) * i;
        const float v = ok ? reg[i] : 0.0f;
        if constexpr (BF16) *reinterpret_cast<__bf16 *>(lds + key * AfTile<D, true>::KROW + d * 2) = (__bf16)v;
        else *reinterpret_cast<float *>(lds + key * AfTile<D, false>::KROW + d * 4) = v;
    }
}
// as stored: LDS[d][key]
template <int D, int NT, bool BF16>
__device__ __forceinline__ void af_commit_dk(const float (&reg)[D * AF_KT / NT], unsigned char *lds, int t0, int Tk, int tid) {
    const int key = tid & 31, dg = tid >> 5;
    const bool ok = t0 + key < Tk;
#pragma unroll
    for (int i = 0; i < D * AF_KT / NT; ++i) {
        const int d = dg + (NT / 32) * i;
        const float v = ok ? reg[i] : 0.0f;
        if constexpr (BF16) *reinterpret_cast<__bf16 *>(lds + d * AfTile<D, true>::VROW + key * 2) = (__bf16)v;
        else *reinterpret_cast<float *>(lds + d * AfTile<D, false>::VROW + key * 4) = v;
    }
}

// B-operand fragments of a [D][T] slice for this lane's column t (clamped): X^T as the right-hand side of a GEMM whose
// contraction runs over d.  fp32: k-step j = 4 g + e <-> d = 8 g + 4 half + e (four consecutive d per 16-byte A read);
// bf16: k-step j <-> d = 16 j + 8 half + e.  `mul` scales the values (the attention's q scaling).
template <int D, bool BF16> struct AfFrag {
    float f[BF16 ? 1 : D / 2];
    af_u32x4 b[BF16 ? D / 16 : 1];
};
template <int D, bool BF16>
__device__ __forceinline__ void af_load_frag(AfFrag<D, BF16> &o, rsrc_t src, int cs, int t, int half, float mul) {
    const unsigned vo = 4u * (unsigned)(t + (BF16 ? 8 : 4) * half * cs);
    if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < D / 16; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = buf_load(src, vo, 4u * (unsigned)((16 * j + e) * cs)) * mul;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.b[j][e] = af_pk(v[2 * e], v[2 * e + 1]);
        }
    } else {
#pragma unroll
        for (int g = 0; g < D / 8; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) o.f[4 * g + e] = buf_load(src, vo, 4u * (unsigned)((8 * g + e) * cs)) * mul;
    }
}

// acc[key][col] += sum_d Ks[key][d] * frag[d][col]      (Ks = LDS tile [key][d])
template <int D, bool BF16>
__device__ __forceinline__ void af_gemm_kd(f32x16 &acc, const unsigned char *ks, const AfFrag<D, BF16> &fr, int l31, int half) {
    f32x16 a1 = (f32x16){0};  // two accumulation chains: consecutive MFMAs do not wait for each other
    if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < D / 16; ++j) {
            const af_u32x4 a = *reinterpret_cast<const af_u32x4 *>(ks + l31 * AfTile<D, true>::KROW + (16 * j + 8 * half) * 2);
            if (j & 1) a1 = af_mma16(a, fr.b[j], a1);
            else acc = af_mma16(a, fr.b[j], acc);
        }
    } else {
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ks + l31 * AfTile<D, false>::KROW + (8 * g + 4 * half) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e & 1) a1 = mfma32(a[e], fr.f[4 * g + e], a1);
                else acc = mfma32(a[e], fr.f[4 * g + e], acc);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += a1[r];
}
// acc[rb][d][col] += sum_key Vs[d][key] * p[key][col], p = an accumulator-layout register file (its key order)
template <int D, bool BF16>
__device__ __forceinline__ void af_gemm_dk(f32x16 (&acc)[D / 32], const unsigned char *vs, const f32x16 &p, int l31, int half) {
    if constexpr (BF16) {
        af_u32x4 pb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) pb[j][e] = af_pk(p[8 * j + 2 * e], p[8 * j + 2 * e + 1]);
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char *row = vs + (32 * rb + l31) * AfTile<D, true>::VROW;
                const af_u32x2 lo = *reinterpret_cast<const af_u32x2 *>(row + (16 * j + 4 * half) * 2);
                const af_u32x2 hi = *reinterpret_cast<const af_u32x2 *>(row + (16 * j + 8 + 4 * half) * 2);
                af_u32x4 a;
                a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
                acc[rb] = af_mma16(a, pb[j], acc[rb]);
            }
    } else {
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(vs + (32 * rb + l31) * AfTile<D, false>::VROW + (8 * g + 4 * half) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rb] = mfma32(a[e], p[4 * g + e], acc[rb]);
            }
    }
}

// masked scores of one tile from the raw accumulator: out-of-range keys -> -inf (never part of the softmax), padded keys ->
// `fill` (-inf: torch's multi_head_attention_forward; -1e8: transformer.py:381-386)
__device__ __forceinline__ void af_mask(f32x16 &s, const float *codes, int half, float fill) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float c = codes[af_row(r, half)];
        s[r] = c == 2.0f ? -INFINITY : (c == 1.0f ? fill : s[r]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: o = softmax(scale q k^T (+ mask)) v per (batch, head); lse[b][h][0][q] = m (row max), lse[b][h][1][q] = l (sum exp)
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool BF16>
__global__ void __launch_bounds__(256) attn_fwd_kernel(SetAttnArgs a) {
    typedef AfTile<D, BF16> TL;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[TL::KBYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[TL::VBYTES];
    __shared__ float codes[AF_KT];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + 32 * w + l31;
    const int qc = min(qi, a.Tq - 1);
    const rsrc_t rq = make_rsrc(a.q + (int64_t)b * a.q_bs + (int64_t)h * D * a.q_cs);
    const rsrc_t rk = make_rsrc(a.k + (int64_t)b * a.k_bs + (int64_t)h * D * a.k_cs);
    const rsrc_t rv = make_rsrc(a.v + (int64_t)b * a.v_bs + (int64_t)h * D * a.v_cs);
    const float *kpm = a.kpm ? a.kpm + (int64_t)b * a.Tk : nullptr;

    AfFrag<D, BF16> qf;
    af_load_frag<D, BF16>(qf, rq, a.q_cs, qc, half, a.scale);
    f32x16 oacc[D / 32];
#pragma unroll
    for (int rb = 0; rb < D / 32; ++rb) oacc[rb] = (f32x16){0};
    float m = -INFINITY, l = 0.0f;

    constexpr int NR = D * AF_KT / 256;
    float kreg[NR], vreg[NR];
    const int ntiles = (a.Tk + AF_KT - 1) / AF_KT;
    af_issue<D, 256>(kreg, rk, a.k_cs, 0, a.Tk, tid);
    af_issue<D, 256>(vreg, rv, a.v_cs, 0, a.Tk, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int t0 = kt * AF_KT;
        __syncthreads();  // the previous tile's MFMAs are done with the LDS tiles
        af_commit_kd<D, 256, BF16>(kreg, Ks, t0, a.Tk, tid);
        af_commit_dk<D, 256, BF16>(vreg, Vs, t0, a.Tk, tid);
        if (tid < AF_KT) codes[tid] = t0 + tid >= a.Tk ? 2.0f : ((kpm && kpm[t0 + tid] != 0.0f) ? 1.0f : 0.0f);
        __syncthreads();
        if (kt + 1 < ntiles) {  // next tile's loads fly under this tile's math
            af_issue<D, 256>(kreg, rk, a.k_cs, t0 + AF_KT, a.Tk, tid);
            af_issue<D, 256>(vreg, rv, a.v_cs, t0 + AF_KT, a.Tk, tid);
        }
        f32x16 s = (f32x16){0};
        af_gemm_kd<D, BF16>(s, Ks, qf, l31, half);
        af_mask(s, codes, half, a.fill);
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[r]);
        mt = fmaxf(mt, af_xor32(mt));
        const float mn = fmaxf(m, mt);
        const bool dead = mn == -INFINITY;  // nothing but -inf so far: keep the state neutral (exp(-inf + inf) would poison it)
        const float corr = dead ? 1.0f : __expf(m - mn);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = dead ? 0.0f : __expf(s[r] - mn);
            ps += s[r];
        }
        l = l * corr + ps;
        m = mn;
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[rb][r] *= corr;
        af_gemm_dk<D, BF16>(oacc, Vs, s, l31, half);
    }
    l += af_xor32(l);
    const float inv = 1.0f / l;  // l == 0 (every key masked with -inf): 0 * inf = NaN, as torch's softmax of an all -inf row
    if (qi < a.Tq) {
        const rsrc_t ro = make_rsrc(a.o + (int64_t)b * a.o_bs + (int64_t)h * D * a.o_cs);
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store(oacc[rb][r] * inv, ro, 4u * (unsigned)(qi + 4 * half * a.o_cs), 4u * (unsigned)((32 * rb + af_row(r, 0)) * a.o_cs));
        if (half == 0) {  // the softmax statistics, kept apart: m + log(l) would round log(l) away next to a -1e8 fill
            float *st = a.lse + ((int64_t)b * a.heads + h) * 2 * a.Tq;
            st[qi] = m;
            st[a.Tq + qi] = l;
        }
    }
}

// probabilities p[b][h][q][key] = exp(masked score - lse[q]) for callers that want them (the encoder-decoder attention map
// the model returns; transformer.py:396-410).  Plain fp32 FMAs: one thread per query, keys in an inner loop.  BF16: scale q and k are
// rounded to bf16 like the forward's MFMA operands, so that the scores are the ones whose row max / sum the forward stored (with fp32
// scores against bf16-operand statistics the rows did not sum to 1 and single entries exceeded 1).
template <int D, bool BF16>
__global__ void __launch_bounds__(256) attn_probs_kernel(SetAttnArgs a) {
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= a.Tq) return;
    const float *q = a.q + (int64_t)b * a.q_bs + (int64_t)h * D * a.q_cs + qi;
    const float *k = a.k + (int64_t)b * a.k_bs + (int64_t)h * D * a.k_cs;
    const float *kpm = a.kpm ? a.kpm + (int64_t)b * a.Tk : nullptr;
    float qv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { const float v = q[(int64_t)d * a.q_cs] * a.scale; qv[d] = BF16 ? (float)(__bf16)v : v; }
    const float *st = a.lse + ((int64_t)b * a.heads + h) * 2 * a.Tq;
    const float mx = st[qi], linv = 1.0f / st[a.Tq + qi];
    float *p = a.p + (((int64_t)b * a.heads + h) * a.Tq + qi) * a.Tk;
    for (int key = 0; key < a.Tk; ++key) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) { const float kv = k[(int64_t)d * a.k_cs + key]; s += qv[d] * (BF16 ? (float)(__bf16)kv : kv); }  // k: the same address in every lane (broadcast)
        if (kpm && kpm[key] != 0.0f) s = a.fill;
        p[key] = __expf(s - mx) * linv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, pass 1 (one wave = 32 queries, walks the keys): recompute P^T = exp(S^T - m) / l, dP^T = V dO^T,
// dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T.  delta[q] = sum_d dO[q][d] O[q][d] is computed here and stored for pass 2.
// K is needed in both orientations ([key][d] for S^T, [d][key] for dQ^T), V as [key][d].
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool BF16>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(SetAttnBwdArgs g) {
    const SetAttnArgs &a = g.fwd;
    typedef AfTile<D, BF16> TL;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[TL::KBYTES];   // [key][d]
    __shared__ __attribute__((aligned(16))) unsigned char Kt[TL::VBYTES];   // [d][key]
    __shared__ __attribute__((aligned(16))) unsigned char Vk[TL::KBYTES];   // [key][d]
    __shared__ float codes[AF_KT];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + 32 * w + l31;
    const int qc = min(qi, a.Tq - 1);
    const rsrc_t rq = make_rsrc(a.q + (int64_t)b * a.q_bs + (int64_t)h * D * a.q_cs);
    const rsrc_t rk = make_rsrc(a.k + (int64_t)b * a.k_bs + (int64_t)h * D * a.k_cs);
    const rsrc_t rv = make_rsrc(a.v + (int64_t)b * a.v_bs + (int64_t)h * D * a.v_cs);
    const rsrc_t rdo = make_rsrc(g.d_o + (int64_t)b * a.o_bs + (int64_t)h * D * a.o_cs);
    const rsrc_t ro = make_rsrc(a.o + (int64_t)b * a.o_bs + (int64_t)h * D * a.o_cs);
    const float *kpm = a.kpm ? a.kpm + (int64_t)b * a.Tk : nullptr;
    const int64_t row = ((int64_t)b * a.heads + h) * a.Tq + qc;
    const float *st = a.lse + ((int64_t)b * a.heads + h) * 2 * a.Tq;
    const float mx = st[qc], linv = 1.0f / st[a.Tq + qc];

    // delta = sum_d dO O over this lane's half of d, then the other half's share
    float delta = 0.0f;
    {
        const unsigned vo = 4u * (unsigned)qc;
        const unsigned voh = vo + 4u * (unsigned)(half * a.o_cs);
#pragma unroll
        for (int i = 0; i < D / 2; ++i) {
            const unsigned so = 4u * (unsigned)((2 * i) * a.o_cs);
            delta += buf_load(rdo, voh, so) * buf_load(ro, voh, so);
        }
        delta += af_xor32(delta);
        if (qi < a.Tq && half == 0) g.delta[row] = delta;
    }
    AfFrag<D, BF16> qf, dof;
    af_load_frag<D, BF16>(qf, rq, a.q_cs, qc, half, a.scale);
    af_load_frag<D, BF16>(dof, rdo, a.o_cs, qc, half, 1.0f);
    f32x16 dq[D / 32];
#pragma unroll
    for (int rb = 0; rb < D / 32; ++rb) dq[rb] = (f32x16){0};

    constexpr int NR = D * AF_KT / 256;
    float kreg[NR], vreg[NR];
    const int ntiles = (a.Tk + AF_KT - 1) / AF_KT;
    af_issue<D, 256>(kreg, rk, a.k_cs, 0, a.Tk, tid);
    af_issue<D, 256>(vreg, rv, a.v_cs, 0, a.Tk, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int t0 = kt * AF_KT;
        __syncthreads();
        af_commit_kd<D, 256, BF16>(kreg, Ks, t0, a.Tk, tid);
        af_commit_dk<D, 256, BF16>(kreg, Kt, t0, a.Tk, tid);
        af_commit_kd<D, 256, BF16>(vreg, Vk, t0, a.Tk, tid);
        if (tid < AF_KT) codes[tid] = t0 + tid >= a.Tk ? 2.0f : ((kpm && kpm[t0 + tid] != 0.0f) ? 1.0f : 0.0f);
        __syncthreads();
        if (kt + 1 < ntiles) {  // next tile's loads fly under this tile's math
            af_issue<D, 256>(kreg, rk, a.k_cs, t0 + AF_KT, a.Tk, tid);
            af_issue<D, 256>(vreg, rv, a.v_cs, t0 + AF_KT, a.Tk, tid);
        }
        f32x16 s = (f32x16){0}, dp = (f32x16){0};
        af_gemm_kd<D, BF16>(s, Ks, qf, l31, half);
        af_gemm_kd<D, BF16>(dp, Vk, dof, l31, half);
        af_mask(s, codes, half, a.fill);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(s[r] - mx) * linv;
            // masked keys: the score is the constant `fill`, its derivative with respect to q and k is zero
            const float c = codes[af_row(r, half)];
            s[r] = c != 0.0f ? 0.0f : p * (dp[r] - delta);
        }
        af_gemm_dk<D, BF16>(dq, Kt, s, l31, half);
    }
    if (qi < a.Tq) {
        const rsrc_t rdq = make_rsrc(g.dq + (int64_t)b * g.dq_bs + (int64_t)h * D * g.dq_cs);
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                buf_store(dq[rb][r] * a.scale, rdq, 4u * (unsigned)(qi + 4 * half * g.dq_cs), 4u * (unsigned)((32 * rb + af_row(r, 0)) * g.dq_cs));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, pass 2 (one wave = 32 KEYS, walks the queries): S [query][key] = Q K^T recomputed with the keys as COLUMNS,
// P = exp(S - m[q]) / l[q] (m, l, delta per ROW: read per register from LDS), dP = dO V^T, dS = P o (dP - delta),
// dV^T [d][key] += dO^T P,  dK^T [d][key] += scale Q^T dS  -- the accumulators of P and dS are again the B operands, their
// k-order now runs over queries.  No atomics: a key column belongs to one wave, queries are walked in order.
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool BF16>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(SetAttnBwdArgs g) {
    const SetAttnArgs &a = g.fwd;
    typedef AfTile<D, BF16> TL;
    __shared__ __attribute__((aligned(16))) unsigned char Qs[TL::KBYTES];    // [query][d]  (scaled)
    __shared__ __attribute__((aligned(16))) unsigned char Qt[TL::VBYTES];    // [d][query]  (scaled)
    __shared__ __attribute__((aligned(16))) unsigned char Os[TL::KBYTES];    // dO [query][d]
    __shared__ __attribute__((aligned(16))) unsigned char Ot[TL::VBYTES];    // dO [d][query]
    __shared__ float m_s[AF_KT], linv_s[AF_KT], delta_s[AF_KT];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, h = blockIdx.y;
    const int ki = blockIdx.x * 128 + 32 * w + l31;
    const int kc = min(ki, a.Tk - 1);
    const rsrc_t rq = make_rsrc(a.q + (int64_t)b * a.q_bs + (int64_t)h * D * a.q_cs);
    const rsrc_t rk = make_rsrc(a.k + (int64_t)b * a.k_bs + (int64_t)h * D * a.k_cs);
    const rsrc_t rv = make_rsrc(a.v + (int64_t)b * a.v_bs + (int64_t)h * D * a.v_cs);
    const rsrc_t rdo = make_rsrc(g.d_o + (int64_t)b * a.o_bs + (int64_t)h * D * a.o_cs);
    const bool kvalid = ki < a.Tk;
    const bool kmasked = a.kpm && a.kpm[(int64_t)b * a.Tk + kc] != 0.0f;
    const int64_t row0 = ((int64_t)b * a.heads + h) * a.Tq;
    const float *st = a.lse + ((int64_t)b * a.heads + h) * 2 * a.Tq;

    AfFrag<D, BF16> kf, vf;
    af_load_frag<D, BF16>(kf, rk, a.k_cs, kc, half, 1.0f);
    af_load_frag<D, BF16>(vf, rv, a.v_cs, kc, half, 1.0f);
    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int rb = 0; rb < D / 32; ++rb) dk[rb] = dv[rb] = (f32x16){0};

    constexpr int NR = D * AF_KT / 256;
    float qreg[NR], oreg[NR];
    const int ntiles = (a.Tq + AF_KT - 1) / AF_KT;
    af_issue<D, 256>(qreg, rq, a.q_cs, 0, a.Tq, tid);
    af_issue<D, 256>(oreg, rdo, a.o_cs, 0, a.Tq, tid);
    for (int qt = 0; qt < ntiles; ++qt) {
        const int t0 = qt * AF_KT;
#pragma unroll
        for (int i = 0; i < NR; ++i) qreg[i] *= a.scale;
        __syncthreads();
        af_commit_kd<D, 256, BF16>(qreg, Qs, t0, a.Tq, tid);
        af_commit_dk<D, 256, BF16>(qreg, Qt, t0, a.Tq, tid);
        af_commit_kd<D, 256, BF16>(oreg, Os, t0, a.Tq, tid);
        af_commit_dk<D, 256, BF16>(oreg, Ot, t0, a.Tq, tid);
        if (tid < AF_KT) {
            const bool ok = t0 + tid < a.Tq;
            m_s[tid] = ok ? st[t0 + tid] : INFINITY;  // rows beyond Tq: p = exp(s - inf) * 0 = 0
            linv_s[tid] = ok ? 1.0f / st[a.Tq + t0 + tid] : 0.0f;
            delta_s[tid] = ok ? g.delta[row0 + t0 + tid] : 0.0f;
        }
        __syncthreads();
        if (qt + 1 < ntiles) {  // next tile's loads fly under this tile's math
            af_issue<D, 256>(qreg, rq, a.q_cs, t0 + AF_KT, a.Tq, tid);
            af_issue<D, 256>(oreg, rdo, a.o_cs, t0 + AF_KT, a.Tq, tid);
        }
        f32x16 s = (f32x16){0}, dp = (f32x16){0};
        af_gemm_kd<D, BF16>(s, Qs, kf, l31, half);    // S [query][key]
        af_gemm_kd<D, BF16>(dp, Os, vf, l31, half);   // dP [query][key] = dO V^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = af_row(r, half);
            const float sc = !kvalid ? -INFINITY : (kmasked ? a.fill : s[r]);
            const float p = __expf(sc - m_s[qr]) * linv_s[qr];
            s[r] = p;
            dp[r] = kmasked ? 0.0f : p * (dp[r] - delta_s[qr]);  // dS (a padded key's score is the constant `fill`)
        }
        af_gemm_dk<D, BF16>(dv, Ot, s, l31, half);    // dV^T [d][key] += dO^T P
        af_gemm_dk<D, BF16>(dk, Qt, dp, l31, half);   // dK^T [d][key] += (scale Q)^T dS
    }
    if (kvalid) {
        const rsrc_t rdk = make_rsrc(g.dk + (int64_t)b * g.dk_bs + (int64_t)h * D * g.dk_cs);
        const rsrc_t rdv = make_rsrc(g.dv + (int64_t)b * g.dv_bs + (int64_t)h * D * g.dv_cs);
#pragma unroll
        for (int rb = 0; rb < D / 32; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned so_k = 4u * (unsigned)((32 * rb + af_row(r, 0)) * g.dk_cs);
                const unsigned so_v = 4u * (unsigned)((32 * rb + af_row(r, 0)) * g.dv_cs);
                buf_store(dk[rb][r], rdk, 4u * (unsigned)(ki + 4 * half * g.dk_cs), so_k);
                buf_store(dv[rb][r], rdv, 4u * (unsigned)(ki + 4 * half * g.dv_cs), so_v);
            }
    }
}

template <int D>
int attn_launch(const SetAttnArgs &a, hipStream_t s) {
    const dim3 grid((a.Tq + 127) / 128, a.heads, a.B);
    if (a.bf16) hipLaunchKernelGGL((attn_fwd_kernel<D, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<D, false>), grid, dim3(256), 0, s, a);
    if (a.p && a.bf16) hipLaunchKernelGGL((attn_probs_kernel<D, true>), dim3((a.Tq + 255) / 256, a.heads, a.B), dim3(256), 0, s, a);
    else if (a.p) hipLaunchKernelGGL((attn_probs_kernel<D, false>), dim3((a.Tq + 255) / 256, a.heads, a.B), dim3(256), 0, s, a);
    return set_check_launch("set_attention");
}
template <int D>
int attn_bwd_launch(const SetAttnBwdArgs &g, hipStream_t s) {
    const SetAttnArgs &a = g.fwd;
    const dim3 gq((a.Tq + 127) / 128, a.heads, a.B), gk((a.Tk + 127) / 128, a.heads, a.B);
    if (a.bf16) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D, true>), gq, dim3(256), 0, s, g);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, true>), gk, dim3(256), 0, s, g);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D, false>), gq, dim3(256), 0, s, g);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, false>), gk, dim3(256), 0, s, g);
    }
    return set_check_launch("set_attention_bwd");
}

int attn_check(const SetAttnArgs &a, const char *what) {
    SET_REQUIRE(a.q && a.k && a.v && a.o && a.lse && a.B > 0 && a.heads > 0 && a.Tq > 0 && a.Tk > 0, what);
    if (a.head_dim != 32 && a.head_dim != 64 && a.head_dim != 96)
        return set_fail(SET_E_UNSUPPORTED, what, "head_dim must be 32, 64 or 96");
    const int64_t lim = ((int64_t)1 << 31) / 4;
    if ((int64_t)a.heads * a.head_dim * a.q_cs >= lim || (int64_t)a.heads * a.head_dim * a.k_cs >= lim ||
        (int64_t)a.heads * a.head_dim * a.v_cs >= lim || (int64_t)a.heads * a.head_dim * a.o_cs >= lim)
        return set_fail(SET_E_UNSUPPORTED, what, "one batch slice exceeds 2 GiB");
    return SET_OK;
}

}  // namespace

extern "C" int64_t set_sizeof_attn_args(void) { return (int64_t)sizeof(SetAttnArgs); }
extern "C" int64_t set_sizeof_attn_bwd_args(void) { return (int64_t)sizeof(SetAttnBwdArgs); }

extern "C" int set_attention(const SetAttnArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_attention");
    if (int rc = attn_check(*args, "set_attention")) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (args->head_dim) {
        case 32: return attn_launch<32>(*args, s);
        case 64: return attn_launch<64>(*args, s);
        default: return attn_launch<96>(*args, s);
    }
}

extern "C" int set_attention_bwd(const SetAttnBwdArgs *args, void *stream) {
    SET_REQUIRE(args != nullptr, "set_attention_bwd");
    if (int rc = attn_check(args->fwd, "set_attention_bwd")) return rc;
    SET_REQUIRE(args->d_o && args->delta && args->dq && args->dk && args->dv, "set_attention_bwd");
    hipStream_t s = (hipStream_t)stream;
    switch (args->fwd.head_dim) {
        case 32: return attn_bwd_launch<32>(*args, s);
        case 64: return attn_bwd_launch<64>(*args, s);
        default: return attn_bwd_launch<96>(*args, s);
    }
}
