// Shared device helpers for libset_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "set_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error plumbing -------------------------------------------------------------
extern thread_local char g_set_err[512];

static inline int set_fail(int code, const char *what, const char *detail) {
    snprintf(g_set_err, sizeof(g_set_err), "%s: %s", what, detail ? detail : "");
    return code;
}
static inline int set_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_fail(SET_E_LAUNCH, what, hipGetErrorString(e));
    return SET_OK;
}
#define SET_REQUIRE(cond, what)                                        \
    do {                                                               \
        if (!(cond)) return set_fail(SET_E_INVALID, what, #cond);      \
    } while (0)
#define SET_HIP(call, what)                                                        \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) return set_fail(SET_E_LAUNCH, what, hipGetErrorString(e__)); \
    } while (0)

const uint64_t *set_seed_delta_ptr();  // the device word of set_rng_seed_delta (csrc/diffnet.hip), NULL when unset
// Clear `words` 32-bit words with a KERNEL.  Not hipMemsetAsync: as a memset node of a captured graph (training.GraphedTrainStep) a clear
// was not ordered against eager work enqueued between two replays (ROCm 7.2) -- buffers came out as uninitialised memory.
static __global__ void __launch_bounds__(256) set_zero_words_kernel(unsigned *p, int64_t words) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < words) p[i] = 0u;
}
static inline bool set_aligned16(const void *a, const void *b, const void *c) {  // 16-byte vector accesses are legal on all three
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}
static inline hipError_t set_zero_async(void *p, size_t bytes, hipStream_t s) {
    const int64_t words = (int64_t)((bytes + 3) / 4);
    hipLaunchKernelGGL(set_zero_words_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, reinterpret_cast<unsigned *>(p), words);
    return hipGetLastError();
}
static inline unsigned set_blocks(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---- activations (match torch CPU fp32 semantics) -----------------------------------------
__device__ __forceinline__ float dev_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float dev_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float dev_act(float v, int act, float p) {
    switch (act) {
        case SET_ACT_RELU: return v > 0.0f ? v : 0.0f;
        case SET_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        case SET_ACT_TANH: return tanhf(v);
        case SET_ACT_SOFTPLUS: return dev_softplus(v);
        case SET_ACT_MISH: return v * tanhf(dev_softplus(v));
        case SET_ACT_LRELU: return v > 0.0f ? v : v * p;
        default: return v;
    }
}
__device__ __forceinline__ float dev_pro(float v, int pro, float p) {
    switch (pro) {
        case SET_PRO_LRELU: return v > 0.0f ? v : v * p;
        case SET_PRO_DIV: return v / p;
        default: return v;
    }
}

// ---- MFMA f32 32x32x2 fragment maps (cdna_hip_programming.md section 3) --------------------
//   A operand: lane l holds A[i = l & 31][k = l >> 5]
//   B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D      : reg r of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- raw buffer addressing: wave-uniform base (SGPR descriptor) + per-lane byte offset (one VGPR shared by every
//      row) + wave-uniform byte offset (SGPR).  hipcc otherwise materialises a 64-bit VGPR address per load/store
//      (v_lshl_add_u64 + 2 VGPRs each), which is what pushes row-looped tile code into spills.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)0x7fffffff, (int)0x00020000);
}
// a per-lane offset at or beyond the descriptor's num_records (0x7fffffff): raw-buffer loads return 0 there, stores are dropped
constexpr unsigned BUF_OOB = 0x80000000u;
__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 buf_load4(rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void buf_store4(f32x4 v, rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}
// streaming 16-byte accesses of data ONE CU writes and reads once per pass (a block's private buffer): nt load (evict-first at the
// L2), sc1 store (written through and dropped from the L2) -- they must not push the weights every CU of the XCD re-reads out of the L2
__device__ __forceinline__ f32x4 buf_load4_stream(rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ void buf_store4_stream(f32x4 v, rsrc_t r, unsigned voff, unsigned soff) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 16);
}
// fp32 add at the L2 without return (buffer_atomic_add_f32): for read-modify-write of data this block owns exclusively
__device__ __forceinline__ void buf_atomic_add(float v, rsrc_t r, unsigned voff, unsigned soff) {
    (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, (int)voff, (int)soff, 0);
}
// agent-scope write-through store (sc1): complete (vmcnt) = visible to every XCD, no L2 write-back needed later
__device__ __forceinline__ void buf_store_agent(float v, rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 16);
}

// two fp32 values -> their two-piece fp16 splittings (a0 + a1 = a to 22 significand bits), each piece pair packed into one dword with the
// first value in the low half: v_cvt_pk_f16_f32, two back-conversions, v_pk_add_f32, v_cvt_pk_f16_f32 -- five instructions where two
// scalar splittings and their packing take ten.  Same roundings (to nearest even, twice) as `h0 = (f16)a; h1 = (f16)(a - (float)h0)`.
__device__ __forceinline__ void split2_f16(float a, float b, unsigned &p0, unsigned &p1) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {a, b};
    const h2_t h0 = __builtin_convertvector(v, h2_t);
    const h2_t h1 = __builtin_convertvector(v - __builtin_convertvector(h0, f2_t), h2_t);
    p0 = __builtin_bit_cast(unsigned, h0);
    p1 = __builtin_bit_cast(unsigned, h1);
}

// ---- shared fp32 MFMA GEMM inner loop -----------------------------------------------------------------------
// acc[RB][NCB] (32x32 blocks) += A * B over `ng` groups of GS k-steps (2 k values each).
//   A operand: packed image, one vector of RB floats per lane per k-step:  wp[u * 64] for k-step u of the group
//   B operand: LDS, bp[u * rstep + 32 * cb]
// Two operand sets ping-pong (no register copies): while the MFMAs of one group run, the loads of the next group are
// issued one k-step at a time, each pinned in front of its k-step's MFMAs with sched_barrier (otherwise the
// scheduler sinks them next to their use).  `advance(g)` moves wp / bp from group g to group g+1.  ng must be even.
template <int RB> struct AVec;
template <> struct AVec<1> { typedef float type; };
template <> struct AVec<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct AVec<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int RB>
__device__ __forceinline__ float avec_get(const typename AVec<RB>::type &v, int r) {
    if constexpr (RB == 1) return v; else return v[r];
}

template <int RB, int NCB, int GS>
struct KOps {
    typename AVec<RB>::type A[GS];
    float B[GS][NCB];
};

template <int RB, int NCB, int GS, typename Adv>
__device__ __forceinline__ void gemm_groups(f32x16 (&acc)[RB][NCB], const typename AVec<RB>::type *&wp, const float *&bp,
                                            int rstep, int ng, Adv advance) {
    KOps<RB, NCB, GS> P, Q;
    auto load_step = [&](KOps<RB, NCB, GS> &o, int u) {
        o.A[u] = wp[u * 64];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) o.B[u][cb] = bp[u * rstep + 32 * cb];
    };
    auto mma_step = [&](const KOps<RB, NCB, GS> &o, int u) {
        __builtin_amdgcn_s_setprio(1);  // co-resident blocks run out of phase: favour the wave that is in its MFMA burst (+3 %)
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[r][cb] = mfma32(avec_get<RB>(o.A[u], r), o.B[u][cb], acc[r][cb]);
        __builtin_amdgcn_s_setprio(0);
    };
#pragma unroll
    for (int u = 0; u < GS; ++u) load_step(P, u);
    for (int g = 0; g < ng; g += 2) {
        advance(g);  // -> group g+1 (always exists: ng is even)
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            load_step(Q, u);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(P, u);
#ifndef SET_NO_TRAILING_SB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (g + 2 < ng) advance(g + 1);  // last pair: re-load the final group (harmless, in bounds)
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            load_step(P, u);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(Q, u);
#ifndef SET_NO_TRAILING_SB
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
}
