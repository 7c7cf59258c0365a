// wr_common.h — shared device helpers and host-side error plumbing for libwhisprrec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "whisprrec_hip.h"

namespace wr {

// ------------------------------------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int32_t fail_hip(hipError_t e, const char *what);

#define WR_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::wr::set_error(__VA_ARGS__);    \
            return (code);                   \
        }                                    \
    } while (0)

#define WR_HIP(call)                                              \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return ::wr::fail_hip(e__, #call); \
    } while (0)

#define WR_LAUNCH_CHECK(name)                                          \
    do {                                                               \
        hipError_t e__ = hipGetLastError();                            \
        if (e__ != hipSuccess) return ::wr::fail_hip(e__, "launch " name); \
    } while (0)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

constexpr float kGamma = 1e-10f;  // BPRLoss(gamma=1e-10), reference src/utils/loss.py:33
constexpr int kBlock = 256;       // 4 waves per workgroup

// ------------------------------------------------------------------------------------------------ teams
// A "team" is T consecutive lanes (T in {1,2,4,8,16}) of one 16-lane DPP row that together hold one
// embedding row: lane l of the team holds float4 chunks l, l+T, ... (NV chunks).  D = 4*T*NV when the row
// is full; a row with D < 4*T*NV masks the tail chunks.  D=64 -> T=16, NV=1: one wave-instruction moves
// four 256-B rows (1 KiB, 16 B per lane).

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    // row_mask/bank_mask = 0xF (all), bound_ctrl = true: lanes outside EXEC read as 0 (never happens inside a team)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the T lanes of a team; every lane of the team gets the total.  DPP only (no LDS traffic):
// quad_perm [1,0,3,2] (0xB1), quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141), row_mirror (0x140).
template <int T>
__device__ __forceinline__ float team_sum(float v) {
    if constexpr (T >= 2) v += dpp_f<0xB1>(v);
    if constexpr (T >= 4) v += dpp_f<0x4E>(v);
    if constexpr (T >= 8) v += dpp_f<0x141>(v);
    if constexpr (T >= 16) v += dpp_f<0x140>(v);
    return v;
}

// Sum over the 64 lanes of a wave (result valid in every lane): team_sum<16> then two cross-row steps.
__device__ __forceinline__ float wave_sum(float v) {
    v = team_sum<16>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Deterministic block sum for kBlock threads; result valid in thread 0.  `scratch` >= 4 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float *scratch) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) r += scratch[w];
    }
    return r;
}

// Strided sum of n floats by the kBlock threads of a block (thread t takes t, t+kBlock, ...), with the first 16 loads of
// every thread issued together: one memory round trip instead of a chain of dependent ones.  Fixed order.
__device__ __forceinline__ float strided_partial_sum(const float *__restrict__ p, int n) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = threadIdx.x + j * kBlock;
        v[j] = (i < n) ? p[i] : 0.f;
    }
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) a += v[j];
    for (int i = threadIdx.x + 16 * kBlock; i < n; i += kBlock) a += p[i];
    return a;
}

template <int NV>
struct Row {
    float4 v[NV];
};

// chunk index of this lane's k-th float4, and whether it is inside the row
template <int T>
__device__ __forceinline__ int chunk_of(int lane, int k) { return lane + k * T; }

template <int T, int NV, bool FULL>
__device__ __forceinline__ Row<NV> load_row(const float *__restrict__ base, int64_t row, int D, int lane) {
    Row<NV> r;
    const float4 *p = reinterpret_cast<const float4 *>(base + row * (int64_t)D);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = chunk_of<T>(lane, k);
        if (FULL || c * 4 < D) r.v[k] = p[c];
        else r.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return r;
}

template <int T, int NV, bool FULL>
__device__ __forceinline__ void store_row(float *__restrict__ base, int64_t row, int D, int lane, const Row<NV> &r) {
    float4 *p = reinterpret_cast<float4 *>(base + row * (int64_t)D);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = chunk_of<T>(lane, k);
        if (FULL || c * 4 < D) p[c] = r.v[k];
    }
}

// 16-byte write-through store (`sc1`): the bytes go to memory at once instead of staying dirty in this XCD's L2 — the store
// form for rows that another workgroup of the SAME launch reads after a counter hand-off (bprmf_chain_step, wr_bpr.hip).
// The compiler does not count an asm store: the storing wave waits with its own s_waitcnt vmcnt(0) before it signals.
// s_nop 1: the data registers may be rewritten by the next instruction only after the store has read them.
typedef float wr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_wt(float4 *p, const float4 &v) {
    const wr_f4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

__device__ __forceinline__ void store_i32_wt(int *p, int v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int T, int NV, bool FULL>
__device__ __forceinline__ void store_row_wt(float *__restrict__ base, int64_t row, int D, int lane, const Row<NV> &r) {
    float4 *p = reinterpret_cast<float4 *>(base + row * (int64_t)D);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = chunk_of<T>(lane, k);
        if (FULL || c * 4 < D) store16_wt(p + c, r.v[k]);
    }
}

template <int NV>
__device__ __forceinline__ float dot_partial(const Row<NV> &a, const Row<NV> &b) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        s = fmaf(a.v[k].x, b.v[k].x, s);
        s = fmaf(a.v[k].y, b.v[k].y, s);
        s = fmaf(a.v[k].z, b.v[k].z, s);
        s = fmaf(a.v[k].w, b.v[k].w, s);
    }
    return s;
}

// BPR per-triplet loss term and coefficient from the two scores (reference src/utils/loss.py:38 and its
// autograd): term = -log(gamma + s), coef = -(s(1-s)/(gamma+s)) / B, s = sigmoid(pos-neg).
__device__ __forceinline__ void bpr_terms(float pos, float neg, float batch_f, float &term, float &coef) {
    // sigmoid, log and the two quotients on the hardware transcendental unit (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp):
    // every lane of a team evaluates this for its triplet, and the libm expf / logf + three IEEE divisions cost ~10x the
    // issue slots for a difference of 1e-7 relative (the parity bar is 1e-5).  exp overflow -> s = 0 -> coef = 0, as before.
    const float x = pos - neg;
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.44269504088896341f));
    const float gs = kGamma + s;
    term = -0.693147180559945309f * __builtin_amdgcn_logf(gs);        // v_log_f32 is log2
    coef = -(s * (1.0f - s) * __builtin_amdgcn_rcpf(gs)) * __builtin_amdgcn_rcpf(batch_f);
}

// Shape dispatch: picks <T, NV, FULL> for an embedding size D (multiple of 4, <= 1024).
//   D = 4,8,16,32 -> T = D/4, NV = 1;  D >= 64 -> T = 16, NV = ceil(D/64);  FULL iff D == 4*T*NV.
#define WR_DISPATCH_D(D, CALL)                                                      \
    do {                                                                            \
        if ((D) == 64) { CALL(16, 1, true); }                                       \
        else if ((D) == 128) { CALL(16, 2, true); }                                 \
        else if ((D) == 32) { CALL(8, 1, true); }                                   \
        else if ((D) == 256) { CALL(16, 4, true); }                                 \
        else if ((D) == 16) { CALL(4, 1, true); }                                   \
        else if ((D) == 8) { CALL(2, 1, true); }                                    \
        else if ((D) == 4) { CALL(1, 1, true); }                                    \
        else if ((D) < 64) { CALL(16, 1, false); }                                  \
        else if ((D) <= 128) { CALL(16, 2, false); }                                \
        else if ((D) <= 256) { CALL(16, 4, false); }                                \
        else if ((D) <= 512) { CALL(16, 8, false); }                                \
        else { CALL(16, 16, false); }                                               \
    } while (0)

static inline int32_t check_table(const void *tab, int64_t n_rows, int32_t D, const char *name) {
    WR_REQUIRE(tab != nullptr, WR_E_NULL, "%s is NULL", name);
    WR_REQUIRE(n_rows > 0 && n_rows < (int64_t(1) << 31), WR_E_SHAPE, "%s: n_rows=%lld out of range", name, (long long)n_rows);
    WR_REQUIRE(D >= 4 && D <= 1024 && D % 4 == 0, WR_E_SHAPE, "%s: D=%d must be a multiple of 4 in [4,1024]", name, D);
    WR_REQUIRE(aligned16(tab), WR_E_ALIGN, "%s is not 16-byte aligned", name);
    return WR_OK;
}

// wr_scatter.hip: scatter-add through a row plan (no sort of the positions); 0 words = not applicable
int64_t scatter_planned_words(int64_t n, int64_t n_rows);
int32_t scatter_add_planned_once(float *tab, int64_t n_rows, int32_t D, const int64_t *idx, const float *src, int64_t n,
                                 int64_t padding_idx, float alpha, int32_t *plan, hipStream_t stream);

// ------------------------------------------------------------------------------------------------ XCD placement
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one, each XCD has its own L2).  This bijective
// remap hands every XCD a CONTIGUOUS range of logical ids, so workgroups that read the same lines (one batch's index
// arrays in the plan kernels) fill one L2 instead of eight.  Speed only — never correctness.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------ optimizers
// One element of torch.optim.Adam's single-tensor step (amsgrad off) — shared by the dense pass (wr_rows.hip) and the
// lazy row replay (wr_lazy.hip), which must produce the same bits.  The two divisions and the square root of the
// reference formula use the hardware v_rcp_f32 / v_sqrt_f32 (1 ulp) and a host-side 1/sqrt(bias_correction2), and the
// multiply-adds are fused (as torch's own GPU kernels are free to): the lazy replay is ALU-bound on exactly this function,
// and IEEE-rounded fdiv/fsqrt expansions cost 2.3x as many issue slots for a difference far below the 1e-5 parity
// tolerance (the update term is off by <= 3 ulp, the weight by lr * that).
template <bool L2 = true>
__device__ __forceinline__ void adam_elem(float &w, float &m, float &v, float g, float l2, float b1, float b2, float eps,
                                          float step_size, float inv_bc2_sqrt) {
    if (L2 && l2 != 0.f) g = fmaf(l2, w, g);   // L2 = false: the caller knows l2 == 0
    m = fmaf(1.0f - b1, g - m, m);             // exp_avg.lerp_(grad, 1-beta1)
    v = fmaf(b2, v, (1.0f - b2) * g * g);      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_bc2_sqrt, eps);   // sqrt(v)/sqrt(bias_correction2) + eps
    w = fmaf(-step_size, m * __builtin_amdgcn_rcpf(denom), w);                 // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// The same element step with a zero gradient and l2 == 0 — what the lazy replay spends its time in: 5 VALU + sqrt + rcp.
// Bit-identical to adam_elem<false>(.., g = 0, ..): g - m = -m exactly, fma(b2, v, +0) = b2 * v for v >= +0.
__device__ __forceinline__ void adam_elem_zero_grad(float &w, float &m, float &v, float b1, float b2, float eps,
                                                    float step_size, float inv_bc2_sqrt) {
    m = fmaf(-(1.0f - b1), m, m);
    v = b2 * v;
    const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_bc2_sqrt, eps);
    w = fmaf(-step_size, m * __builtin_amdgcn_rcpf(denom), w);
}

// Host side of the same step: step_size = lr/(1-beta1^t), 1/sqrt(1-beta2^t), in double as torch computes the bias
// corrections for a python-number step, rounded to fp32 once.
static inline void adam_step_consts(int64_t t, float lr, float beta1, float beta2, float *step_size, float *inv_bc2_sqrt) {
    const double bc1 = 1.0 - pow((double)beta1, (double)t);
    const double bc2 = 1.0 - pow((double)beta2, (double)t);
    *step_size = t == 0 ? 0.f : (float)((double)lr / bc1);
    *inv_bc2_sqrt = t == 0 ? 1.f : (float)(1.0 / sqrt(bc2));
}

// torch.optim.SGD with weight_decay and a zero gradient: g' = 0 + l2*w ; w -= lr*g'
__device__ __forceinline__ float sgd_decay_elem(float w, float lr, float l2) { return w - lr * (l2 * w); }

}  // namespace wr
