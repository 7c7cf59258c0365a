// wr_bpr.hip — BPRMF forward and the fused batch-synchronous SGD step for gfx950 (MI355X).
//
// Reference path restated (paths relative to the reference root):
//   BPRMF.predict            src/models/general/BPRMF.py:69-80     3 row gathers, 2 row dots
//   BPRLoss.forward          src/utils/loss.py:37-39               -mean(log(1e-10 + sigmoid(pos-neg)))
//   loss.backward()          src/helpers/BaseRunner.py:198         3 scatter-adds of c_b-scaled rows
//   torch.optim.SGD.step()   src/helpers/BaseRunner.py:199         w -= lr (g + l2 w)
//
// Data layout in HBM: tables row-major fp32 [n_rows, D]; a team of T lanes holds one row as float4 per
// lane (D=64: 16 lanes x 16 B = one 256-B row; a wave-instruction moves four rows = 1 KiB).
//
// Step = two kernels over a sorted batch plan (wr_plan.hip):
//   user phase: position t of the user-sorted triplets; the first position of a run of equal users
//     ("head") owns U[u]: it reads U[u] once, loops over the run reading I[p], I[n], reduces the two dots
//     with DPP adds inside the 16-lane row, forms the loss term and coefficient c, accumulates
//     g_u += c (I[p]-I[n]) and finally rewrites U[u] in place.  Nobody else reads U[u] in this step, so
//     in-place is batch-synchronous.  An item row that occurs ONCE in the batch (plan flag, bit 31 of
//     tp/tn clear) is likewise read by this team only and is finished right here: I[p] -= lr*(+c U[u]),
//     I[n] -= lr*(-c U[u]).  Only for item rows with several occurrences is z_t = c U[u] stashed.
//   item phase: position q of the item-sorted occurrences; the head of a run of >= 2 occurrences owns
//     I[r]: it sums +-z of its occurrences and rewrites I[r] in place (all reads of I by the user phase
//     are complete at the kernel boundary).  Block 0 also folds the user phase's per-block loss partials into the loss.
// No float atomics anywhere: each row has one writer and a fixed summation order.
#include "wr_common.h"
#include <hip/hip_ext.h>

#include <algorithm>   // hipExtLaunchKernelGGL: stop events attached to a dispatch (timing hooks of launch_step)

#ifndef WR_USER_WAVES
#define WR_USER_WAVES 8      // waves per SIMD the headline instantiation of the user phase is held to (64 VGPRs)
#endif
#ifndef WR_ADAM_DBG
#define WR_ADAM_DBG 0        // timing-only variants of the folded Adam step (A/B builds): 1 no replay, 2 transposes but no replay loops
#endif
#ifndef WR_ADAM_WAVES
#define WR_ADAM_WAVES 5      // the same for the folded-Adam instantiation (MODE 4): 96 VGPRs + 8 spilled; A/B on MI355X, us per
                             // step at 1M x 1M x 64, B = 65,536: unconstrained (103 VGPRs, 4 waves) 107-111, 5 waves 104-106,
                             // 6 waves (35 spilled) 116
#endif

namespace wr {

constexpr int kHotRun = 32;     // an item row with more occurrences than this in one batch is "hot" (must match wr_plan.hip)
constexpr int kHotPiece = 256;  // occurrences summed by one workgroup
constexpr double kHotLossScale = 68719476736.0;  // 2^36: fixed-point scale of the hot pieces' loss terms

// ----------------------------------------------------------------------------------------------- forward only
template <int T, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void bpr_fwd_kernel(const float *__restrict__ U, const float *__restrict__ I, int D,
                                                          const int64_t *__restrict__ u, const int64_t *__restrict__ p,
                                                          const int64_t *__restrict__ n, int B, float *__restrict__ pos_out,
                                                          float *__restrict__ neg_out, float *__restrict__ coef_out,
                                                          float *__restrict__ partials, int64_t n_users, int64_t n_items) {
    __shared__ float scratch[kBlock / 64];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int b = blockIdx.x * TEAMS + threadIdx.x / T;
    float term_acc = 0.f;
    if (b < B) {
        // ids clamped into the tables: an id out of range is the caller's to report (BatchPlan.validate: IndexError, as
        // nn.Embedding raises); no kernel reads outside a table on the way there
        const int64_t ub = min(max(u[b], (int64_t)0), n_users - 1), pb = min(max(p[b], (int64_t)0), n_items - 1),
                      nb = min(max(n[b], (int64_t)0), n_items - 1);
        const Row<NV> ur = load_row<T, NV, FULL>(U, ub, D, lane);
        const Row<NV> pr = load_row<T, NV, FULL>(I, pb, D, lane);
        const Row<NV> nr = load_row<T, NV, FULL>(I, nb, D, lane);
        const float sp = team_sum<T>(dot_partial<NV>(ur, pr));
        const float sn = team_sum<T>(dot_partial<NV>(ur, nr));
        float term, coef;
        bpr_terms(sp, sn, (float)B, term, coef);
        if (lane == 0) {
            term_acc = term;
            if (pos_out) pos_out[b] = sp;
            if (neg_out) neg_out[b] = sn;
            if (coef_out) coef_out[b] = coef;
        }
    }
    const float s = block_sum(term_acc, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Sums `n` partials in a fixed order and writes out[0] = sum / denom.  One block.
__global__ __launch_bounds__(kBlock) void finish_loss_kernel(const float *__restrict__ partials, int n, float denom,
                                                              float *__restrict__ out) {
    __shared__ float scratch[kBlock / 64];
    const float a = strided_partial_sum(partials, n);
    const float s = block_sum(a, scratch);
    if (threadIdx.x == 0) out[0] = s / denom;
}

// ----------------------------------------------------------------------------------------------- MODE 3: Adam in place
// torch.optim.Adam applied where MODE 0 applies SGD: the team that finishes a row holds its weights and its complete
// gradient in registers, so it reads the row's two moments, takes the step (adam_elem: the dense kernel's arithmetic) and
// writes weights and moments back — no gradient table, no second pass over the rows.  The rows must be up to date through
// step t-1 (wr_adam_rows_lazy with grad = NULL before this launch); last[row] becomes t.
struct AdamArgs {
    float *mU, *vU, *mI, *vI;
    int *lastU, *lastI;
    float step_size, inv_bc2_sqrt, b1, b2, eps, l2;
    int t;
    const float *consts;   // MODE 4: per-step (step_size, 1/sqrt(bias_correction2)) of steps 0..t (wr_adam_consts)
};

// The moments a team keeps beside a weight row in MODE 4 (empty otherwise: no registers).
template <int NV, bool ON>
struct Moments {
    Row<NV> m, v;
};
template <int NV>
struct Moments<NV, false> {};

// MODE 4 — Adam with the catch-up folded into the row loads.  Rows are stored at the optimizer step last[row]; the team
// that loads a row replays the missed zero-gradient steps last[row]+1 .. t-1 on its register copy (adam_elem: the bits of the
// dense optimizer and of wr_adam_rows_lazy) before it uses the weights, and whoever finishes the row applies step t to that
// copy and writes weights and moments back once: 3 row reads + 3 row writes per touched row and step, where a separate
// catch-up pass (wr_adam_rows_lazy, then MODE 3) moves twice that.  A row with several occurrences in the batch is replayed
// by each of its readers (ALU only) and once more by its finisher in the item phase.
template <int T, int NV, bool FULL>
__device__ __forceinline__ void adam_load_caught_up(const float *__restrict__ W, const float *__restrict__ M,
                                                    const float *__restrict__ V, const int *__restrict__ last, int row, int D,
                                                    int lane, const AdamArgs &a, Row<NV> &w, Row<NV> &m, Row<NV> &v) {
    const int from = last[row];
    w = load_row<T, NV, FULL>(W, row, D, lane);
    m = load_row<T, NV, FULL>(M, row, D, lane);
    v = load_row<T, NV, FULL>(V, row, D, lane);
    const float2 *__restrict__ c2 = reinterpret_cast<const float2 *>(a.consts);
    auto replay = [&](float2 k) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            if (a.l2 != 0.f) {
                adam_elem<true>(w.v[q].x, m.v[q].x, v.v[q].x, 0.f, a.l2, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem<true>(w.v[q].y, m.v[q].y, v.v[q].y, 0.f, a.l2, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem<true>(w.v[q].z, m.v[q].z, v.v[q].z, 0.f, a.l2, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem<true>(w.v[q].w, m.v[q].w, v.v[q].w, 0.f, a.l2, a.b1, a.b2, a.eps, k.x, k.y);
            } else {
                adam_elem_zero_grad(w.v[q].x, m.v[q].x, v.v[q].x, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem_zero_grad(w.v[q].y, m.v[q].y, v.v[q].y, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem_zero_grad(w.v[q].z, m.v[q].z, v.v[q].z, a.b1, a.b2, a.eps, k.x, k.y);
                adam_elem_zero_grad(w.v[q].w, m.v[q].w, v.v[q].w, a.b1, a.b2, a.eps, k.x, k.y);
            }
        }
    };
    int s = from + 1;
    for (; s + 3 < a.t; s += 4) {   // four steps per trip: their constants are fetched together
        const float2 k0 = c2[s], k1 = c2[s + 1], k2 = c2[s + 2], k3 = c2[s + 3];
        replay(k0); replay(k1); replay(k2); replay(k3);
    }
    for (; s < a.t; ++s) replay(c2[s]);
}

// The same replay for the FOUR rows a wave's four 16-lane teams hold, without divergence.  Rows miss different numbers of
// steps (geometric: the longest of four is ~2.1x the mean), and in adam_load_caught_up every team of the wave pays for the
// longest.  Here the wave first transposes its data — lane (team a, position p) trades its float4 (elements 4p..4p+3 of row
// a) for element 4p+a of each of the four rows: two butterfly stages over lanes ^16 and ^32, four ds_bpermute per float4 —
// so that in every replay iteration ALL 64 lanes work on the same row: the loop over row k's missed steps is uniform
// (constants through scalar loads), the wave spends sum(gaps) single-element iterations instead of 4 * max(gaps), and the
// transpose back restores the team layout.  Every element goes through exactly the same operations in the same order:
// the same bits.  Must be called by all 64 lanes (teams without a row pass from = t - 1: nothing to replay).
#ifndef WR_TRANSPOSE_SHFL
#define WR_TRANSPOSE_SHFL 0     // 1: the butterfly through ds_bpermute (the form before the gfx950 lane swaps; A/B runs)
#endif
__device__ __forceinline__ void wave_transpose4(float4 &x, bool a0, bool a1) {
#if WR_TRANSPOSE_SHFL
    float s, r;
    s = a0 ? x.x : x.y; r = __shfl_xor(s, 16, 64); if (a0) x.x = r; else x.y = r;
    s = a0 ? x.z : x.w; r = __shfl_xor(s, 16, 64); if (a0) x.z = r; else x.w = r;
    s = a1 ? x.x : x.z; r = __shfl_xor(s, 32, 64); if (a1) x.x = r; else x.z = r;
    s = a1 ? x.y : x.w; r = __shfl_xor(s, 32, 64); if (a1) x.y = r; else x.w = r;
#else
    // gfx950 lane swaps, one VALU instruction per exchange and no LDS crossbar: v_permlane16_swap(a, b) swaps the odd
    // 16-lane rows of a with the even rows of b, v_permlane32_swap(a, b) the upper 32 lanes of a with the lower 32 of b —
    // exactly the two butterfly stages (scripts/exp/permlane_probe.hip prints what they do)
    (void)a0; (void)a1;
    auto swap16 = [](float &a, float &b) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
    auto swap32 = [](float &a, float &b) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
    };
    swap16(x.x, x.y); swap16(x.z, x.w);
    swap32(x.x, x.z); swap32(x.y, x.w);
#endif
}

template <int NV>
__device__ __forceinline__ void adam_replay_balanced(Row<NV> &w, Row<NV> &m, Row<NV> &v, int from, const AdamArgs &a) {
    const int l64 = (int)(threadIdx.x & 63u);
    const bool a0 = (l64 & 16) != 0, a1 = (l64 & 32) != 0;
    const int f0 = __builtin_amdgcn_readlane(from, 0), f1 = __builtin_amdgcn_readlane(from, 16),
              f2 = __builtin_amdgcn_readlane(from, 32), f3 = __builtin_amdgcn_readlane(from, 48);
    if (f0 + 1 >= a.t && f1 + 1 >= a.t && f2 + 1 >= a.t && f3 + 1 >= a.t) return;   // uniform: nothing missed anywhere
#if WR_ADAM_DBG & 1
    return;                                 // timing only: no replay at all (wrong tables; no index depends on the replay)
#endif
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        wave_transpose4(w.v[q], a0, a1);
        wave_transpose4(m.v[q], a0, a1);
        wave_transpose4(v.v[q], a0, a1);
    }
    const float2 *__restrict__ c2 = reinterpret_cast<const float2 *>(a.consts);
    auto run = [&](int fk, auto comp) {   // comp(float4&) -> the component that now holds row k's element
        auto replay = [&](float2 k) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (a.l2 != 0.f) adam_elem<true>(comp(w.v[q]), comp(m.v[q]), comp(v.v[q]), 0.f, a.l2, a.b1, a.b2, a.eps, k.x, k.y);
                else adam_elem_zero_grad(comp(w.v[q]), comp(m.v[q]), comp(v.v[q]), a.b1, a.b2, a.eps, k.x, k.y);
            }
        };
        int s = fk + 1;
        for (; s + 3 < a.t; s += 4) {
            const float2 k0 = c2[s], k1 = c2[s + 1], k2 = c2[s + 2], k3 = c2[s + 3];
            replay(k0); replay(k1); replay(k2); replay(k3);
        }
        for (; s < a.t; ++s) replay(c2[s]);
    };
#if !(WR_ADAM_DBG & 2)
    run(f0, [](float4 &x) -> float & { return x.x; });
    run(f1, [](float4 &x) -> float & { return x.y; });
    run(f2, [](float4 &x) -> float & { return x.z; });
    run(f3, [](float4 &x) -> float & { return x.w; });
#endif
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        wave_transpose4(w.v[q], a0, a1);
        wave_transpose4(m.v[q], a0, a1);
        wave_transpose4(v.v[q], a0, a1);
    }
}

// step t on a row whose caught-up weights and moments are in registers (WT: the three rows and the row's step stamp are
// stored write-through — the chained launch hands them to other workgroups of the same launch)
template <int T, int NV, bool FULL, bool WT = false>
__device__ __forceinline__ void adam_finish_row_regs(float *__restrict__ W, float *__restrict__ M, float *__restrict__ V,
                                                     int *__restrict__ last, int row, int D, int lane, Row<NV> w, Row<NV> m,
                                                     Row<NV> v, const Row<NV> &g, const AdamArgs &a) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        adam_elem(w.v[k].x, m.v[k].x, v.v[k].x, g.v[k].x, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].y, m.v[k].y, v.v[k].y, g.v[k].y, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].z, m.v[k].z, v.v[k].z, g.v[k].z, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].w, m.v[k].w, v.v[k].w, g.v[k].w, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
    }
    if constexpr (WT) {
        store_row_wt<T, NV, FULL>(W, row, D, lane, w);
        store_row_wt<T, NV, FULL>(M, row, D, lane, m);
        store_row_wt<T, NV, FULL>(V, row, D, lane, v);
        if (lane == 0) store_i32_wt(last + row, a.t);
    } else {
        store_row<T, NV, FULL>(W, row, D, lane, w);
        store_row<T, NV, FULL>(M, row, D, lane, m);
        store_row<T, NV, FULL>(V, row, D, lane, v);
        if (lane == 0) last[row] = a.t;
    }
}

template <int T, int NV, bool FULL>
__device__ __forceinline__ void adam_finish_row(float *__restrict__ W, float *__restrict__ M, float *__restrict__ V,
                                                int *__restrict__ last, int row, int D, int lane, const Row<NV> &w0,
                                                const Row<NV> &g, const AdamArgs &a) {
    Row<NV> m = load_row<T, NV, FULL>(M, row, D, lane), v = load_row<T, NV, FULL>(V, row, D, lane), w = w0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        adam_elem(w.v[k].x, m.v[k].x, v.v[k].x, g.v[k].x, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].y, m.v[k].y, v.v[k].y, g.v[k].y, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].z, m.v[k].z, v.v[k].z, g.v[k].z, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
        adam_elem(w.v[k].w, m.v[k].w, v.v[k].w, g.v[k].w, a.l2, a.b1, a.b2, a.eps, a.step_size, a.inv_bc2_sqrt);
    }
    store_row<T, NV, FULL>(W, row, D, lane, w);
    store_row<T, NV, FULL>(M, row, D, lane, m);
    store_row<T, NV, FULL>(V, row, D, lane, v);
    if (lane == 0) last[row] = a.t;
}

// MODE 5 / 6 — torch.optim.Adagrad / Adadelta where MODE 3 applies Adam (reference src/helpers/BaseRunner.py:34-37,120-124:
// the --optimizer flag is eval'ed into torch.optim.<name>(params, lr, weight_decay=l2); this path is for l2 = 0).  With a
// zero gradient neither optimizer moves a weight, so the TABLES are always current (no catch-up before gradients, no flush):
//   Adagrad   (MODE 5): state_sum += g*g; w -= lr * g / (sqrt(state_sum) + 1e-10) — a zero gradient changes nothing at all:
//             exactly sparse.  AdamArgs: mU / mI = state_sum tables, eps, step_size = lr.
//   Adadelta  (MODE 6): square_avg = rho*square_avg + (1-rho) g*g; std = sqrt(square_avg + eps);
//             delta = sqrt(acc_delta + eps) / std * g; acc_delta = rho*acc_delta + (1-rho) delta*delta; w -= lr*delta.
//             A zero gradient only multiplies both state rows by rho: the finisher replays the missed decays (t-1-last[row]
//             multiplications per element, the dense optimizer's bits) before it applies step t.  AdamArgs: mU / mI =
//             square_avg, vU / vI = acc_delta, lastU / lastI, b1 = rho, eps, step_size = lr, t.
// sqrt and the quotient on v_sqrt_f32 / v_rcp_f32 (1 ulp), like adam_elem.
template <int T, int NV, bool FULL, int MODE>
__device__ __forceinline__ void opt_finish_row(float *__restrict__ W, float *__restrict__ S1, float *__restrict__ S2,
                                               int *__restrict__ last, int row, int D, int lane, const Row<NV> &w0,
                                               const Row<NV> &g, const AdamArgs &a) {
    if constexpr (MODE == 3) {
        adam_finish_row<T, NV, FULL>(W, S1, S2, last, row, D, lane, w0, g, a);
    } else if constexpr (MODE == 5) {
        Row<NV> s = load_row<T, NV, FULL>(S1, row, D, lane), w = w0;
        auto elem = [&](float &ww, float &ss, float gg) {
            ss = fmaf(gg, gg, ss);
            ww = fmaf(-a.step_size, gg * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(ss) + a.eps), ww);
        };
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            elem(w.v[k].x, s.v[k].x, g.v[k].x); elem(w.v[k].y, s.v[k].y, g.v[k].y);
            elem(w.v[k].z, s.v[k].z, g.v[k].z); elem(w.v[k].w, s.v[k].w, g.v[k].w);
        }
        store_row<T, NV, FULL>(W, row, D, lane, w);
        store_row<T, NV, FULL>(S1, row, D, lane, s);
    } else {
        Row<NV> sq = load_row<T, NV, FULL>(S1, row, D, lane), ac = load_row<T, NV, FULL>(S2, row, D, lane), w = w0;
        const int missed = a.t - 1 - last[row];
        const float rho = a.b1;
        for (int j = 0; j < missed; ++j) {          // the decays of the steps this row had no gradient in
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                sq.v[k].x *= rho; sq.v[k].y *= rho; sq.v[k].z *= rho; sq.v[k].w *= rho;
                ac.v[k].x *= rho; ac.v[k].y *= rho; ac.v[k].z *= rho; ac.v[k].w *= rho;
            }
        }
        auto elem = [&](float &ww, float &s2, float &a2, float gg) {
            s2 = fmaf(1.0f - rho, gg * gg, rho * s2);
            const float std = __builtin_amdgcn_sqrtf(s2 + a.eps);
            const float delta = __builtin_amdgcn_sqrtf(a2 + a.eps) * __builtin_amdgcn_rcpf(std) * gg;
            a2 = fmaf(1.0f - rho, delta * delta, rho * a2);
            ww = fmaf(-a.step_size, delta, ww);
        };
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            elem(w.v[k].x, sq.v[k].x, ac.v[k].x, g.v[k].x); elem(w.v[k].y, sq.v[k].y, ac.v[k].y, g.v[k].y);
            elem(w.v[k].z, sq.v[k].z, ac.v[k].z, g.v[k].z); elem(w.v[k].w, sq.v[k].w, ac.v[k].w, g.v[k].w);
        }
        store_row<T, NV, FULL>(W, row, D, lane, w);
        store_row<T, NV, FULL>(S1, row, D, lane, sq);
        store_row<T, NV, FULL>(S2, row, D, lane, ac);
        if (lane == 0) last[row] = a.t;
    }
}

constexpr bool mode_has_state(int mode) { return mode == 3 || mode == 5 || mode == 6; }

// ----------------------------------------------------------------------------------------------- user phase
// MODE 0: SGD apply in place.  MODE 1: emit gradient rows + stamps, tables untouched.  MODE 3: Adam apply in place.
// MODE 2 (row-sharded step): user rows applied in place; item rows below ad.t (the shard's own rows) too; the item rows
// from ad.t on are rows received from their owners for this step: their gradients are emitted to gradI[row - ad.t] (the
// buffer of gradient rows sent back).
// Everything one triplet contributes, given the three rows in registers: loss term, coefficient, user-row gradient,
// single-occurrence item rows finished in place (or their gradient emitted), stash for multi-occurrence item rows.
template <int T, int NV, bool FULL, int MODE>
__device__ __forceinline__ void triplet_body(const Row<NV> &ur, const Row<NV> &pr, const Row<NV> &nr, int praw, int nraw, int t,
                                             float *I, float *__restrict__ gradI, float *__restrict__ Z,
                                             int *__restrict__ stampI, int step_id, int D, int lane, float lr, float l2,
                                             float denom, Row<NV> &g, float &terms, const AdamArgs &ad,
                                             const Moments<NV, MODE == 4> &pmv = Moments<NV, MODE == 4>{},
                                             const Moments<NV, MODE == 4> &nmv = Moments<NV, MODE == 4>{}) {
    const int p = praw & 0x7fffffff, n = nraw & 0x7fffffff;
    const bool p_shared = praw < 0, n_shared = nraw < 0;   // bit 31: the item row has other occurrences in this batch
    const float sp = team_sum<T>(dot_partial<NV>(ur, pr));
    const float sn = team_sum<T>(dot_partial<NV>(ur, nr));
    float term, c;
    bpr_terms(sp, sn, denom, term, c);
    terms += term;
    Row<NV> z;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        g.v[k].x = fmaf(c, pr.v[k].x - nr.v[k].x, g.v[k].x);
        g.v[k].y = fmaf(c, pr.v[k].y - nr.v[k].y, g.v[k].y);
        g.v[k].z = fmaf(c, pr.v[k].z - nr.v[k].z, g.v[k].z);
        g.v[k].w = fmaf(c, pr.v[k].w - nr.v[k].w, g.v[k].w);
        z.v[k] = make_float4(c * ur.v[k].x, c * ur.v[k].y, c * ur.v[k].z, c * ur.v[k].w);
    }
    // An item row that occurs once in the batch is read by this team only: finish it here
    // (gradient = +z for the positive, -z for the negative), no stash, no item-phase work.
    if constexpr (MODE == 4) {
        if (!p_shared) adam_finish_row_regs<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, p, D, lane, pr, pmv.m, pmv.v, z, ad);
        if (!n_shared) {
            Row<NV> zn;
#pragma unroll
            for (int k = 0; k < NV; ++k) zn.v[k] = make_float4(-z.v[k].x, -z.v[k].y, -z.v[k].z, -z.v[k].w);
            adam_finish_row_regs<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, n, D, lane, nr, nmv.m, nmv.v, zn, ad);
        }
        if (p_shared || n_shared) store_row<T, NV, FULL>(Z, t, D, lane, z);
        return;
    }
    if constexpr (mode_has_state(MODE)) {
        if (!p_shared) opt_finish_row<T, NV, FULL, MODE>(I, ad.mI, ad.vI, ad.lastI, p, D, lane, pr, z, ad);
        if (!n_shared) {
            Row<NV> zn;
#pragma unroll
            for (int k = 0; k < NV; ++k) zn.v[k] = make_float4(-z.v[k].x, -z.v[k].y, -z.v[k].z, -z.v[k].w);
            opt_finish_row<T, NV, FULL, MODE>(I, ad.mI, ad.vI, ad.lastI, n, D, lane, nr, zn, ad);
        }
        if (p_shared || n_shared) store_row<T, NV, FULL>(Z, t, D, lane, z);
        return;
    }
    // MODE 2 (row-sharded step): item rows below ad.t are the shard's OWN rows — updated in place like MODE 0; rows from ad.t
    // on were received from their owners: their gradient goes to gradI[row - ad.t] (ad.t = 0: every item row is a received one)
    if (!p_shared) {
        const bool in_place = MODE == 0 || (MODE == 2 && p < ad.t);
        Row<NV> w;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (in_place) {
                w.v[k].x = pr.v[k].x - lr * fmaf(l2, pr.v[k].x, z.v[k].x);
                w.v[k].y = pr.v[k].y - lr * fmaf(l2, pr.v[k].y, z.v[k].y);
                w.v[k].z = pr.v[k].z - lr * fmaf(l2, pr.v[k].z, z.v[k].z);
                w.v[k].w = pr.v[k].w - lr * fmaf(l2, pr.v[k].w, z.v[k].w);
            } else {
                w.v[k] = z.v[k];
            }
        }
        if (in_place) store_row<T, NV, FULL>(I, p, D, lane, w);
        else store_row<T, NV, FULL>(gradI, MODE == 2 ? p - ad.t : p, D, lane, w);
        if (stampI != nullptr && lane == 0) stampI[p] = step_id;
    }
    if (!n_shared) {
        const bool in_place = MODE == 0 || (MODE == 2 && n < ad.t);
        Row<NV> w;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (in_place) {
                w.v[k].x = nr.v[k].x - lr * fmaf(l2, nr.v[k].x, -z.v[k].x);
                w.v[k].y = nr.v[k].y - lr * fmaf(l2, nr.v[k].y, -z.v[k].y);
                w.v[k].z = nr.v[k].z - lr * fmaf(l2, nr.v[k].z, -z.v[k].z);
                w.v[k].w = nr.v[k].w - lr * fmaf(l2, nr.v[k].w, -z.v[k].w);
            } else {
                w.v[k] = make_float4(-z.v[k].x, -z.v[k].y, -z.v[k].z, -z.v[k].w);
            }
        }
        if (in_place) store_row<T, NV, FULL>(I, n, D, lane, w);
        else store_row<T, NV, FULL>(gradI, MODE == 2 ? n - ad.t : n, D, lane, w);
        if (stampI != nullptr && lane == 0) stampI[n] = step_id;
    }
    if (p_shared || n_shared) store_row<T, NV, FULL>(Z, t, D, lane, z);
}

// torch.optim.SGD on a finished user row: g' = g + l2 w ; w -= lr g'   (MODE 1: emit g instead)
template <int T, int NV, bool FULL, int MODE>
__device__ __forceinline__ void finish_user_row(float *__restrict__ U, float *__restrict__ gradU, int *__restrict__ stampU,
                                                int step_id, int u, int D, int lane, float lr, float l2, const Row<NV> &ur,
                                                const Row<NV> &g, const AdamArgs &ad) {
    if constexpr (mode_has_state(MODE)) {
        opt_finish_row<T, NV, FULL, MODE>(U, ad.mU, ad.vU, ad.lastU, u, D, lane, ur, g, ad);
        return;
    }
    if (MODE != 1) {
        Row<NV> w;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            w.v[k].x = ur.v[k].x - lr * fmaf(l2, ur.v[k].x, g.v[k].x);
            w.v[k].y = ur.v[k].y - lr * fmaf(l2, ur.v[k].y, g.v[k].y);
            w.v[k].z = ur.v[k].z - lr * fmaf(l2, ur.v[k].z, g.v[k].z);
            w.v[k].w = ur.v[k].w - lr * fmaf(l2, ur.v[k].w, g.v[k].w);
        }
        store_row<T, NV, FULL>(U, u, D, lane, w);
    } else {
        store_row<T, NV, FULL>(gradU, u, D, lane, g);
    }
    if (stampU != nullptr && lane == 0) stampU[u] = step_id;
}

// Each team works on SLOTS positions (t, t + seg, ...): the index loads of all slots are issued together, then the row
// loads of all slots, then the slots are finished one after the other.  (Launched with SLOTS = 1; see launch_step.)
// DEF selects which run heads a launch works on (the chained step launch, bprmf_chain_step):
//   0  every head (the ordinary step);
//   1  every head whose bit in `dmask` is clear — the runs that read no item row the PREVIOUS batch's item phase is still
//      rewriting, so this launch may run beside that item phase;
//   2  the heads listed in `dlist` (the deferred runs: positions in ascending order), one team each.
// `vblock` is the workgroup's index within the phase (blockIdx.x for the plain launches; the chained step launch carries
// several kinds of workgroups and numbers each kind itself); the workgroup's loss partial goes to *partial_out.
template <int T, int NV, bool FULL, int MODE, int SLOTS, bool SKIP_HOT, int DEF>
__device__ __forceinline__ void user_phase_block(float *__restrict__ U, float *I, int D, const int *__restrict__ tu,
                                                 const int *__restrict__ tp, const int *__restrict__ tn, int B, float lr,
                                                 float l2, float *__restrict__ Z, float *__restrict__ partial_out,
                                                 float *__restrict__ gradU, int *__restrict__ stampU,
                                                 float *__restrict__ gradI, int *__restrict__ stampI, int step_id,
                                                 float denom, const AdamArgs &ad, const int *__restrict__ dmask,
                                                 const int *__restrict__ dlist, int n_list,
                                                 const int *__restrict__ n_list_dev, int vblock, float *scratch) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int team = vblock * TEAMS + threadIdx.x / T;
    const int seg = (B + SLOTS - 1) / SLOTS;
    float term_acc = 0.f;

    // All six index loads of a position go out together, on clamped addresses and without a branch between them: the
    // obvious short-circuit tests (t == 0 || tu[t-1] != u, head && tu[t+kHotRun] == u, ...) compile to four memory round
    // trips in a row before the first row load is issued, and a workgroup's lifetime is what bounds the bytes in flight.
    int t0[SLOTS], uu[SLOTS], unext[SLOTS], praw0[SLOTS], nraw0[SLOTS];
    bool head[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        t0[s] = team + s * seg;
        head[s] = false;
        uu[s] = unext[s] = praw0[s] = nraw0[s] = 0;
        if constexpr (DEF == 2) {
            // a listed head: its position comes from the list, everything else as below (one more dependent load, on a
            // launch of a few dozen workgroups)
            // n_list_dev: the list's length lives in device memory (a captured launch is sized for the list's capacity)
            const int nl = n_list_dev != nullptr ? min(*n_list_dev, n_list) : n_list;
            if (team < nl) {
                const int t = dlist[team];
                t0[s] = t;
                const int u = tu[t], un = tu[min(t + 1, B - 1)];
                praw0[s] = tp[t];
                nraw0[s] = tn[t];
                uu[s] = u;
                unext[s] = (t + 1 < B) ? un : ~u;
                head[s] = true;
            }
        } else if (team < seg && t0[s] < B) {
            const int t = t0[s];
            const int u = tu[t], uprev = tu[max(t - 1, 0)], un = tu[min(t + 1, B - 1)];
            const int uhot = SKIP_HOT ? tu[min(t + kHotRun, B - 1)] : 0;
            const int dm = DEF == 1 ? dmask[t >> 5] : 0;
            praw0[s] = tp[t];
            nraw0[s] = tn[t];
            uu[s] = u;
            unext[s] = (t + 1 < B) ? un : ~u;
            head[s] = (t == 0) || (uprev != u);   // first position of a run of equal users
            // users with more than kHotRun triplets in the batch are cut into pieces by the plan (bprmf_user_hot_*)
            if (SKIP_HOT && t + kHotRun < B && uhot == u) head[s] = false;
            if (DEF == 1 && ((dm >> (t & 31)) & 1)) head[s] = false;   // a deferred run: the list launch does it
        }
    }
    Row<NV> ur[SLOTS], pr0[SLOTS], nr0[SLOTS];
    Moments<NV, MODE == 4> umv[SLOTS], pmv0[SLOTS], nmv0[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if constexpr (MODE == 4 && T == 16) {
            // all nine row loads of the position go out together, then the three replays, each balanced over the wave's four
            // teams (adam_replay_balanced: every lane takes part, teams without a head replay nothing)
            int fu = ad.t - 1, fp = fu, fn = fu;
            const Row<NV> zero{};
            ur[s] = pr0[s] = nr0[s] = zero;
            umv[s].m = umv[s].v = pmv0[s].m = pmv0[s].v = nmv0[s].m = nmv0[s].v = zero;
            if (head[s]) {
                const int p = praw0[s] & 0x7fffffff, n = nraw0[s] & 0x7fffffff;
                fu = ad.lastU[uu[s]];
                fp = ad.lastI[p];
                fn = ad.lastI[n];
                ur[s] = load_row<T, NV, FULL>(U, uu[s], D, lane);
                umv[s].m = load_row<T, NV, FULL>(ad.mU, uu[s], D, lane);
                umv[s].v = load_row<T, NV, FULL>(ad.vU, uu[s], D, lane);
                pr0[s] = load_row<T, NV, FULL>(I, p, D, lane);
                pmv0[s].m = load_row<T, NV, FULL>(ad.mI, p, D, lane);
                pmv0[s].v = load_row<T, NV, FULL>(ad.vI, p, D, lane);
                nr0[s] = load_row<T, NV, FULL>(I, n, D, lane);
                nmv0[s].m = load_row<T, NV, FULL>(ad.mI, n, D, lane);
                nmv0[s].v = load_row<T, NV, FULL>(ad.vI, n, D, lane);
            }
            adam_replay_balanced<NV>(ur[s], umv[s].m, umv[s].v, fu, ad);
            adam_replay_balanced<NV>(pr0[s], pmv0[s].m, pmv0[s].v, fp, ad);
            adam_replay_balanced<NV>(nr0[s], nmv0[s].m, nmv0[s].v, fn, ad);
        } else if (head[s]) {
            if constexpr (MODE == 4) {
                adam_load_caught_up<T, NV, FULL>(U, ad.mU, ad.vU, ad.lastU, uu[s], D, lane, ad, ur[s], umv[s].m, umv[s].v);
                adam_load_caught_up<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, praw0[s] & 0x7fffffff, D, lane, ad, pr0[s],
                                                 pmv0[s].m, pmv0[s].v);
                adam_load_caught_up<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, nraw0[s] & 0x7fffffff, D, lane, ad, nr0[s],
                                                 nmv0[s].m, nmv0[s].v);
            } else {
                ur[s] = load_row<T, NV, FULL>(U, uu[s], D, lane);
                pr0[s] = load_row<T, NV, FULL>(I, praw0[s] & 0x7fffffff, D, lane);
                nr0[s] = load_row<T, NV, FULL>(I, nraw0[s] & 0x7fffffff, D, lane);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (!head[s]) continue;
        const int u = uu[s];
        Row<NV> g;
#pragma unroll
        for (int k = 0; k < NV; ++k) g.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = t0[s];
        int praw = praw0[s], nraw = nraw0[s];
        Row<NV> pr = pr0[s], nr = nr0[s];
        Moments<NV, MODE == 4> pmv = pmv0[s], nmv = nmv0[s];
        bool more = unext[s] == u;   // known before the first body: the single-triplet user's row store waits on nothing
        for (;;) {
            triplet_body<T, NV, FULL, MODE>(ur[s], pr, nr, praw, nraw, t, I, gradI, Z, stampI, step_id, D, lane, lr, l2, denom, g,
                                            term_acc, ad, pmv, nmv);
            if (!more) break;
            ++t;   // next triplet of this user (nothing is kept live across the body: 8 waves per SIMD, no spill)
            praw = tp[t];
            nraw = tn[t];
            more = (t + 1 < B) && (tu[t + 1] == u);
            if constexpr (MODE == 4) {
                adam_load_caught_up<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, praw & 0x7fffffff, D, lane, ad, pr, pmv.m, pmv.v);
                adam_load_caught_up<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, nraw & 0x7fffffff, D, lane, ad, nr, nmv.m, nmv.v);
            } else {
                pr = load_row<T, NV, FULL>(I, praw & 0x7fffffff, D, lane);
                nr = load_row<T, NV, FULL>(I, nraw & 0x7fffffff, D, lane);
            }
        }
        if constexpr (MODE == 4)
            adam_finish_row_regs<T, NV, FULL>(U, ad.mU, ad.vU, ad.lastU, u, D, lane, ur[s], umv[s].m, umv[s].v, g, ad);
        else
            finish_user_row<T, NV, FULL, MODE>(U, gradU, stampU, step_id, u, D, lane, lr, l2, ur[s], g, ad);
    }
    if (lane != 0) term_acc = 0.f;  // every lane of a team holds the same terms: count them once
    const float sum = block_sum(term_acc, scratch);
    if (threadIdx.x == 0) *partial_out = sum;
}

template <int T, int NV, bool FULL, int MODE, int SLOTS, bool SKIP_HOT, int DEF = 0>
__global__ __launch_bounds__(kBlock, (NV == 1 && SLOTS == 1 && MODE == 0 && !SKIP_HOT) ? WR_USER_WAVES
                                     : (NV == 1 && SLOTS == 1 && MODE == 4 && !SKIP_HOT) ? WR_ADAM_WAVES : 1) void bprmf_user_phase(float *__restrict__ U, float *I, int D,
                                                            const int *__restrict__ tu, const int *__restrict__ tp,
                                                            const int *__restrict__ tn, int B, float lr, float l2,
                                                            float *__restrict__ Z, float *__restrict__ partials,
                                                            float *__restrict__ gradU, int *__restrict__ stampU,
                                                            float *__restrict__ gradI, int *__restrict__ stampI,
                                                            int step_id, float denom, AdamArgs ad,
                                                            const int *__restrict__ dmask,
                                                            const int *__restrict__ dlist, int n_list,
                                                            const int *__restrict__ n_list_dev) {
    __shared__ float scratch[kBlock / 64];
    user_phase_block<T, NV, FULL, MODE, SLOTS, SKIP_HOT, DEF>(U, I, D, tu, tp, tn, B, lr, l2, Z, partials + blockIdx.x, gradU,
                                                              stampU, gradI, stampI, step_id, denom, ad, dmask, dlist, n_list,
                                                              n_list_dev, (int)blockIdx.x, scratch);
}

// Hot users: a piece = up to kHotPiece consecutive triplets of ONE user.  One workgroup per piece: every team holds the
// user row, team j takes triplets j, j+TEAMS, ...; the TEAMS partial gradients are added in team order through LDS, the
// loss terms in wave order.  bprmf_user_hot_combine then adds the pieces in order and rewrites the user row once.
template <int T, int NV, bool FULL, int MODE>
__global__ __launch_bounds__(kBlock) void bprmf_user_hot_pieces(const float *__restrict__ U, float *I, int D,
                                                                 const int *__restrict__ tu, const int *__restrict__ tp,
                                                                 const int *__restrict__ tn, float lr, float l2,
                                                                 float *__restrict__ Z, float *__restrict__ /*unused*/,
                                                                 float *__restrict__ gradI, int *__restrict__ stampI, int step_id,
                                                                 float denom, const int *__restrict__ piece_q,
                                                                 const int *__restrict__ piece_len, float *__restrict__ hotPU,
                                                                 unsigned long long *__restrict__ hot_loss, AdamArgs ad) {
    extern __shared__ float rows[];  // [TEAMS][D]
    __shared__ float scratch[kBlock / 64];
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T, team = threadIdx.x / T;
    const int q0 = piece_q[blockIdx.x], len = piece_len[blockIdx.x];
    const int u = tu[q0];
    const Row<NV> ur = load_row<T, NV, FULL>(U, u, D, lane);
    Row<NV> g;
#pragma unroll
    for (int k = 0; k < NV; ++k) g.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    float terms = 0.f;
    for (int k = team; k < len; k += TEAMS) {
        const int t = q0 + k;
        const int praw = tp[t], nraw = tn[t];
        const Row<NV> pr = load_row<T, NV, FULL>(I, praw & 0x7fffffff, D, lane);
        const Row<NV> nr = load_row<T, NV, FULL>(I, nraw & 0x7fffffff, D, lane);
        triplet_body<T, NV, FULL, MODE>(ur, pr, nr, praw, nraw, t, I, gradI, Z, stampI, step_id, D, lane, lr, l2, denom, g, terms, ad);
    }
    store_row<T, NV, FULL>(rows, team, D, lane, g);
    __syncthreads();
    for (int c = threadIdx.x; c * 4 < D; c += kBlock) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < TEAMS; ++j) {
            const float4 v = reinterpret_cast<const float4 *>(rows + (int64_t)j * D)[c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4 *>(hotPU + (int64_t)blockIdx.x * D)[c] = a;
    }
    // The order of the pieces in the plan's list is not fixed (atomic append), so their loss terms are accumulated in
    // 64-bit fixed point (2^-36 resolution): integer addition is order-independent, the loss stays bitwise reproducible.
    const float sum = block_sum(lane == 0 ? terms : 0.f, scratch);
    if (threadIdx.x == 0) atomicAdd(hot_loss, (unsigned long long)(long long)llrint((double)sum * kHotLossScale));
}

template <int T, int NV, bool FULL, int MODE>
__global__ __launch_bounds__(kBlock) void bprmf_user_hot_combine(float *__restrict__ U, int D, const int *__restrict__ tu,
                                                                  const int *__restrict__ run_q, const int *__restrict__ run_first,
                                                                  const int *__restrict__ run_np, int n_runs,
                                                                  const float *__restrict__ hotPU, float lr, float l2,
                                                                  float *__restrict__ gradU, int *__restrict__ stampU, int step_id,
                                                                  AdamArgs ad) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int h = blockIdx.x * TEAMS + threadIdx.x / T;
    if (h >= n_runs) return;
    const int u = tu[run_q[h]];
    const int first = run_first[h], np = run_np[h];
    Row<NV> g = load_row<T, NV, FULL>(hotPU, first, D, lane);
    for (int k = 1; k < np; ++k) {
        const Row<NV> x = load_row<T, NV, FULL>(hotPU, first + k, D, lane);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            g.v[c].x += x.v[c].x; g.v[c].y += x.v[c].y; g.v[c].z += x.v[c].z; g.v[c].w += x.v[c].w;
        }
    }
    const Row<NV> ur = load_row<T, NV, FULL>(U, u, D, lane);
    finish_user_row<T, NV, FULL, MODE>(U, gradU, stampU, step_id, u, D, lane, lr, l2, ur, g, ad);
}

// ----------------------------------------------------------------------------------------------- item phase
#ifndef WR_ITEM_TILE
#define WR_ITEM_TILE 64
#endif
// Sorted occurrences per workgroup (its teams share the tile's runs).  A/B on MI355X (scripts/ab_step.py), steps only,
// tile 256 / 128 / 64 / 32: 1M x 1M tables 27.6-29.2 / 26.9-27.5 / 26.6-27.6 / 28.2 us; 125K x 62.5K (the per-GPU stratum
// of the 8-GPU schedule, almost every item row shared) 33.7 / 31.0 / 29.7 / 30.0 us: with 256 a team walks ~5 runs one
// after the other (two dependent round trips each), with 64 one or two.
constexpr int kItemTile = WR_ITEM_TILE;
static_assert(kItemTile <= kBlock && kItemTile >= 32, "item tile");
// A piece = up to kHotPiece consecutive occurrences of ONE item row.  One workgroup per piece: team j sums occurrences
// j, j+TEAMS, ... in that order, four stashed rows in flight at a time; the TEAMS partial rows are added in team order
// through LDS.  Runs as extra workgroups of the item-phase launch (the pieces touch no table row).
template <int T, int NV, bool FULL>
__device__ __forceinline__ void item_hot_piece(int piece, int D, const int *__restrict__ oc_src, const float *__restrict__ Z,
                                               const int *__restrict__ piece_q, const int *__restrict__ piece_len,
                                               float *__restrict__ hotP, float *__restrict__ rows) {
    constexpr int TEAMS = kBlock / T;
    constexpr int kFly = 4;
    const int lane = threadIdx.x % T, team = threadIdx.x / T;
    const int q0 = piece_q[piece], len = piece_len[piece];
    Row<NV> g;
#pragma unroll
    for (int k = 0; k < NV; ++k) g.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = team; k < len; k += kFly * TEAMS) {
        int src[kFly];
        Row<NV> z[kFly];
#pragma unroll
        for (int f = 0; f < kFly; ++f) src[f] = (k + f * TEAMS < len) ? oc_src[q0 + k + f * TEAMS] : -1;
#pragma unroll
        for (int f = 0; f < kFly; ++f)
            if (src[f] >= 0) z[f] = load_row<T, NV, FULL>(Z, src[f] >> 1, D, lane);
#pragma unroll
        for (int f = 0; f < kFly; ++f) {
            if (src[f] < 0) continue;
            const float sgn = (src[f] & 1) ? -1.0f : 1.0f;
#pragma unroll
            for (int c = 0; c < NV; ++c) {
                g.v[c].x = fmaf(sgn, z[f].v[c].x, g.v[c].x); g.v[c].y = fmaf(sgn, z[f].v[c].y, g.v[c].y);
                g.v[c].z = fmaf(sgn, z[f].v[c].z, g.v[c].z); g.v[c].w = fmaf(sgn, z[f].v[c].w, g.v[c].w);
            }
        }
    }
    store_row<T, NV, FULL>(rows, team, D, lane, g);
    __syncthreads();
    for (int c = threadIdx.x; c * 4 < D; c += kBlock) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < TEAMS; ++j) {
            const float4 v = reinterpret_cast<const float4 *>(rows + (int64_t)j * D)[c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4 *>(hotP + (int64_t)piece * D)[c] = a;
    }
}

// torch.optim.SGD on a finished item row (g' = g + l2 w ; w -= lr g'), or the gradient row itself for MODE != 0
// WT: the row is stored write-through (store_row_wt) — the chained step launch hands these rows to other workgroups of
// the same launch (bprmf_chain_step).
template <int T, int NV, bool FULL, int MODE, bool WT = false>
__device__ __forceinline__ void finish_item_row(float *__restrict__ I, float *__restrict__ gradI, int *__restrict__ stampI,
                                                int step_id, int r, int D, int lane, float lr, float l2, const Row<NV> &ir,
                                                const Row<NV> &g, const AdamArgs &ad) {
    if constexpr (mode_has_state(MODE)) {
        opt_finish_row<T, NV, FULL, MODE>(I, ad.mI, ad.vI, ad.lastI, r, D, lane, ir, g, ad);
        return;
    }
    if (MODE == 0 || (MODE == 2 && r < ad.t)) {      // MODE 2: the shard's own rows (below ad.t) are updated in place
        Row<NV> w;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            w.v[k].x = ir.v[k].x - lr * fmaf(l2, ir.v[k].x, g.v[k].x);
            w.v[k].y = ir.v[k].y - lr * fmaf(l2, ir.v[k].y, g.v[k].y);
            w.v[k].z = ir.v[k].z - lr * fmaf(l2, ir.v[k].z, g.v[k].z);
            w.v[k].w = ir.v[k].w - lr * fmaf(l2, ir.v[k].w, g.v[k].w);
        }
        if constexpr (WT) store_row_wt<T, NV, FULL>(I, r, D, lane, w);
        else store_row<T, NV, FULL>(I, r, D, lane, w);
    } else {
        store_row<T, NV, FULL>(gradI, MODE == 2 ? r - ad.t : r, D, lane, g);
    }
    if (stampI != nullptr && lane == 0) stampI[r] = step_id;
}

// One tile of the item phase (TILE sorted occurrences); `vblock` = the tile's index.
template <int T, int NV, bool FULL, int MODE, bool PIECES, bool WT, int TILE = kItemTile>
__device__ __forceinline__ void item_tile_block(float *__restrict__ I, int D, const int *__restrict__ oc_item,
                                                const int *__restrict__ oc_src, int B2, const float *__restrict__ Z, float lr,
                                                float l2, float *__restrict__ gradI, int *__restrict__ stampI, int step_id,
                                                const float *__restrict__ partials, int n_partials, float loss_denom,
                                                float *__restrict__ loss_out, const unsigned long long *__restrict__ hot_loss,
                                                const AdamArgs &ad, int vblock) {
    static_assert(TILE <= kBlock && TILE >= 32, "item tile");
    static_assert(!WT || MODE == 0 || MODE == 4, "write-through rows: plain SGD and the folded Adam step");
    __shared__ float scratch[kBlock / 64];
    __shared__ int heads[TILE];
    // entries staged beyond the tile: with a hot-run list every run that starts in the tile is in LDS up to the entry that
    // proves it hot; without one, eight (longer runs go on from the plan arrays)
    constexpr int kAhead = PIECES ? kHotRun + 1 : 8;
    __shared__ int item_tile[TILE + kAhead];
    __shared__ int src_tile[TILE + kAhead];
    __shared__ int n_heads;
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int tile0 = vblock * TILE;
    // 1) one THREAD per sorted occurrence: stage the tile's (item, source) pairs in LDS, find the heads of runs
    //    of >= 2 equal item rows and compact them (single-occurrence rows were finished by the user phase; the
    //    order of the list is irrelevant: every run is an independent row).  One global round trip.
    if (threadIdx.x == 0) n_heads = 0;
    for (int i = threadIdx.x; i < TILE + kAhead; i += kBlock) {   // the tile plus kAhead entries beyond it
        const int q = tile0 + i;
        item_tile[i] = (q < B2) ? oc_item[q] : -1;
        src_tile[i] = (q < B2) ? oc_src[q] : 0;
    }
    const int prev_item = (threadIdx.x == 0) ? ((tile0 > 0 && tile0 < B2) ? oc_item[tile0 - 1] : -1) : 0;
    __syncthreads();
    if (threadIdx.x < TILE) {
        const int r = item_tile[threadIdx.x];
        if (r >= 0) {
            const int before = (threadIdx.x == 0) ? prev_item : item_tile[threadIdx.x - 1];
            if (before != r && item_tile[threadIdx.x + 1] == r) heads[atomicAdd(&n_heads, 1)] = threadIdx.x;
        }
    }
    __syncthreads();
    const int nh = n_heads;
    // 2) one TEAM per run: the row and the first stashed contributions are requested together (second round
    //    trip); contributions are summed in sorted (fixed) order and the row is rewritten once.
    for (int h = threadIdx.x / T; h < nh; h += TEAMS) {
        const int j0 = heads[h];
        const int r = item_tile[j0];
        {
            Row<NV> ir;
            Moments<NV, MODE == 4> imv;
            if constexpr (MODE == 4) adam_load_caught_up<T, NV, FULL>(I, ad.mI, ad.vI, ad.lastI, r, D, lane, ad, ir, imv.m, imv.v);
            else ir = load_row<T, NV, FULL>(I, r, D, lane);
            const int s0 = src_tile[j0], s1 = src_tile[j0 + 1];
            const Row<NV> z0 = load_row<T, NV, FULL>(Z, s0 >> 1, D, lane);
            const Row<NV> z1 = load_row<T, NV, FULL>(Z, s1 >> 1, D, lane);
            if constexpr (PIECES) {
                // the plan lists this batch's hot runs (> kHotRun occurrences): they are skipped here, every other run is
                // inside the staged window; the run length is read off LDS while the first loads fly
                int m = 2;
                while (m <= kHotRun && item_tile[j0 + m] == r) ++m;
                if (m > kHotRun) continue;   // a hot run: its pieces and bprmf_item_hot_combine do the row
                Row<NV> g;
                const float g0 = (s0 & 1) ? -1.0f : 1.0f, g1 = (s1 & 1) ? -1.0f : 1.0f;  // d/dI[p] = +cU, d/dI[n] = -cU
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    g.v[k].x = fmaf(g1, z1.v[k].x, fmaf(g0, z0.v[k].x, 0.f));
                    g.v[k].y = fmaf(g1, z1.v[k].y, fmaf(g0, z0.v[k].y, 0.f));
                    g.v[k].z = fmaf(g1, z1.v[k].z, fmaf(g0, z0.v[k].z, 0.f));
                    g.v[k].w = fmaf(g1, z1.v[k].w, fmaf(g0, z0.v[k].w, 0.f));
                }
                // third and later occurrences: four stashed rows in flight at a time, added in sorted order
                for (int j = 2; j < m; j += 4) {
                    int src[4];
                    Row<NV> z[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f) src[f] = (j + f < m) ? src_tile[j0 + j + f] : -1;
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        if (src[f] >= 0) z[f] = load_row<T, NV, FULL>(Z, src[f] >> 1, D, lane);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        if (src[f] < 0) continue;
                        const float sgn = (src[f] & 1) ? -1.0f : 1.0f;
#pragma unroll
                        for (int k = 0; k < NV; ++k) {
                            g.v[k].x = fmaf(sgn, z[f].v[k].x, g.v[k].x);
                            g.v[k].y = fmaf(sgn, z[f].v[k].y, g.v[k].y);
                            g.v[k].z = fmaf(sgn, z[f].v[k].z, g.v[k].z);
                            g.v[k].w = fmaf(sgn, z[f].v[k].w, g.v[k].w);
                        }
                    }
                }
                if constexpr (MODE == 4)
                    adam_finish_row_regs<T, NV, FULL, WT>(I, ad.mI, ad.vI, ad.lastI, r, D, lane, ir, imv.m, imv.v, g, ad);
                else
                    finish_item_row<T, NV, FULL, MODE, WT>(I, gradI, stampI, step_id, r, D, lane, lr, l2, ir, g, ad);
            } else {
                // no hot-run list for this batch (none, or a plan built without one): walk the run to its end, whatever
                // its length — from LDS inside the staged window, from the plan arrays beyond it.  One stashed row at a
                // time: with short runs the four-in-flight walk of the other branch is slower (A/B, us/step: uniform 27.3
                // vs 26.8, 125K x 62.5K tables 30.5 vs 29.7); with power-law runs it wins (Zipf 39.7 vs 43.6)
                Row<NV> g;
                const float g0 = (s0 & 1) ? -1.0f : 1.0f, g1 = (s1 & 1) ? -1.0f : 1.0f;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    g.v[k].x = fmaf(g1, z1.v[k].x, fmaf(g0, z0.v[k].x, 0.f));
                    g.v[k].y = fmaf(g1, z1.v[k].y, fmaf(g0, z0.v[k].y, 0.f));
                    g.v[k].z = fmaf(g1, z1.v[k].z, fmaf(g0, z0.v[k].z, 0.f));
                    g.v[k].w = fmaf(g1, z1.v[k].w, fmaf(g0, z0.v[k].w, 0.f));
                }
                int j = j0 + 2;
                if constexpr (MODE == 1) {
                    // gradient mode (one small batch per call: LightGCN, the autograd path — popularity-skewed items, runs of
                    // tens of occurrences, nothing else on the GPU to hide a serial walk): four stashed rows in flight
                    for (;;) {
                        int src[4];
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const int q = tile0 + j + f;
                            const bool in_lds = j + f < TILE + kAhead;
                            src[f] = -1;
                            if (q < B2 && (f == 0 || src[f - 1] >= 0)) {
                                const int it = in_lds ? item_tile[j + f] : oc_item[q];
                                if (it == r) src[f] = in_lds ? src_tile[j + f] : oc_src[q];
                            }
                        }
                        Row<NV> z[4];
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            if (src[f] >= 0) z[f] = load_row<T, NV, FULL>(Z, src[f] >> 1, D, lane);
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            if (src[f] < 0) continue;
                            const float sgn = (src[f] & 1) ? -1.0f : 1.0f;
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                g.v[k].x = fmaf(sgn, z[f].v[k].x, g.v[k].x);
                                g.v[k].y = fmaf(sgn, z[f].v[k].y, g.v[k].y);
                                g.v[k].z = fmaf(sgn, z[f].v[k].z, g.v[k].z);
                                g.v[k].w = fmaf(sgn, z[f].v[k].w, g.v[k].w);
                            }
                        }
                        if (src[3] < 0) break;
                        j += 4;
                    }
                } else
                for (;;) {
                    const int q = tile0 + j;
                    if (q >= B2) break;
                    const bool in_lds = j < TILE + kAhead;
                    const int it = in_lds ? item_tile[j] : oc_item[q];
                    if (it != r) break;
                    const int src = in_lds ? src_tile[j] : oc_src[q];
                    const Row<NV> z = load_row<T, NV, FULL>(Z, src >> 1, D, lane);
                    const float sgn = (src & 1) ? -1.0f : 1.0f;
#pragma unroll
                    for (int k = 0; k < NV; ++k) {
                        g.v[k].x = fmaf(sgn, z.v[k].x, g.v[k].x);
                        g.v[k].y = fmaf(sgn, z.v[k].y, g.v[k].y);
                        g.v[k].z = fmaf(sgn, z.v[k].z, g.v[k].z);
                        g.v[k].w = fmaf(sgn, z.v[k].w, g.v[k].w);
                    }
                    ++j;
                }
                if constexpr (MODE == 4)
                    adam_finish_row_regs<T, NV, FULL, WT>(I, ad.mI, ad.vI, ad.lastI, r, D, lane, ir, imv.m, imv.v, g, ad);
                else
                    finish_item_row<T, NV, FULL, MODE, WT>(I, gradI, stampI, step_id, r, D, lane, lr, l2, ir, g, ad);
            }
        }
    }
    if (vblock == 0 && loss_out != nullptr) {  // uniform per block: fold the user phase's partials
        const float a = strided_partial_sum(partials, n_partials);
        float s = block_sum(a, scratch);
        if (threadIdx.x == 0) {
            if (hot_loss != nullptr) s += (float)((double)(long long)hot_loss[0] / kHotLossScale);
            loss_out[0] = s / loss_denom;
        }
    }
}

template <int T, int NV, bool FULL, int MODE, bool PIECES>
__global__ __launch_bounds__(kBlock) void bprmf_item_phase(float *__restrict__ I, int D, const int *__restrict__ oc_item,
                                                            const int *__restrict__ oc_src, int B2,
                                                            const float *__restrict__ Z, float lr, float l2,
                                                            float *__restrict__ gradI, int *__restrict__ stampI, int step_id,
                                                            const float *__restrict__ partials, int n_partials,
                                                            float loss_denom, float *__restrict__ loss_out, int skip_hot,
                                                            const unsigned long long *__restrict__ hot_loss, int n_tiles,
                                                            const int *__restrict__ piece_q, const int *__restrict__ piece_len,
                                                            float *__restrict__ hotP, AdamArgs ad) {
    if constexpr (PIECES) {                 // this instantiation carries hot pieces as extra workgroups (one each; they are
        extern __shared__ float piece_rows[];   // independent of the tiles' rows); the lean one keeps its 32 VGPRs
        if ((int)blockIdx.x >= n_tiles) {
            item_hot_piece<T, NV, FULL>((int)blockIdx.x - n_tiles, D, oc_src, Z, piece_q, piece_len, hotP, piece_rows);
            return;
        }
    }
    item_tile_block<T, NV, FULL, MODE, PIECES, false>(I, D, oc_item, oc_src, B2, Z, lr, l2, gradI, stampI, step_id, partials,
                                                     n_partials, loss_denom, loss_out, hot_loss, ad, (int)blockIdx.x);
}

// One team per hot row: adds its pieces in piece order and finishes the row like the item phase does.
template <int T, int NV, bool FULL, int MODE>
__global__ __launch_bounds__(kBlock) void bprmf_item_hot_combine(float *__restrict__ I, int D, const int *__restrict__ oc_item,
                                                                  const int *__restrict__ run_q, const int *__restrict__ run_first,
                                                                  const int *__restrict__ run_np, int n_runs,
                                                                  const float *__restrict__ hotP, float lr, float l2,
                                                                  float *__restrict__ gradI, int *__restrict__ stampI, int step_id,
                                                                  AdamArgs ad) {
    constexpr int TEAMS = kBlock / T;
    const int lane = threadIdx.x % T;
    const int h = blockIdx.x * TEAMS + threadIdx.x / T;
    if (h >= n_runs) return;
    const int r = oc_item[run_q[h]];
    const int first = run_first[h], np = run_np[h];
    Row<NV> g = load_row<T, NV, FULL>(hotP, first, D, lane);
    for (int k = 1; k < np; ++k) {
        const Row<NV> x = load_row<T, NV, FULL>(hotP, first + k, D, lane);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            g.v[c].x += x.v[c].x; g.v[c].y += x.v[c].y; g.v[c].z += x.v[c].z; g.v[c].w += x.v[c].w;
        }
    }
    if constexpr (mode_has_state(MODE)) {
        const Row<NV> ir = load_row<T, NV, FULL>(I, r, D, lane);
        opt_finish_row<T, NV, FULL, MODE>(I, ad.mI, ad.vI, ad.lastI, r, D, lane, ir, g, ad);
        return;
    }
    if (MODE == 0 || (MODE == 2 && r < ad.t)) {
        const Row<NV> ir = load_row<T, NV, FULL>(I, r, D, lane);
        Row<NV> w;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            w.v[c].x = ir.v[c].x - lr * fmaf(l2, ir.v[c].x, g.v[c].x);
            w.v[c].y = ir.v[c].y - lr * fmaf(l2, ir.v[c].y, g.v[c].y);
            w.v[c].z = ir.v[c].z - lr * fmaf(l2, ir.v[c].z, g.v[c].z);
            w.v[c].w = ir.v[c].w - lr * fmaf(l2, ir.v[c].w, g.v[c].w);
        }
        store_row<T, NV, FULL>(I, r, D, lane, w);
    } else {
        store_row<T, NV, FULL>(gradI, MODE == 2 ? r - ad.t : r, D, lane, g);
    }
    if (stampI != nullptr && lane == 0) stampI[r] = step_id;
}

// ----------------------------------------------------------------------------------------------- host side
static inline int teams_per_block(int D) {
    if (D >= 64) return kBlock / 16;
    if (D == 32) return kBlock / 8;
    if (D == 16) return kBlock / 4;
    if (D == 8) return kBlock / 2;
    if (D == 4) return kBlock / 1;
    return kBlock / 16;
}
static inline int64_t n_blocks_for(int64_t n_teams, int D) {
    const int tpb = teams_per_block(D);
    return (n_teams + tpb - 1) / tpb;
}

struct StepWs {
    float *Z;
    float *partials;
    float *hotP;   // item-side piece sums
    float *hotPU;  // user-side piece sums
    unsigned long long *hot_loss;  // fixed-point sum of the hot user pieces' loss terms
    int64_t n_partials;
};

// kind 0: item occurrences (2B positions per batch); kind 1: user positions (B per batch)
static inline int64_t hot_cap_pieces(int64_t B, int kind = 0) { const int64_t n = kind ? B : 2 * B; return n / kHotPiece + n / kHotRun + 8; }
static inline int64_t hot_cap_runs(int64_t B, int kind = 0) { const int64_t n = kind ? B : 2 * B; return n / kHotRun + 8; }

struct HotSide {  // hot runs of ONE batch on one side (device pointers already offset), counts from the host copy
    const int32_t *piece_q, *piece_len, *run_q, *run_first, *run_np;
    int n_pieces, n_runs;
};
struct HotBatch {
    HotSide item, user;
};

static inline HotBatch hot_of(const wr_hot_runs *hot, int64_t batch) {
    HotBatch h{{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}, {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}};
    if (hot == nullptr || hot->counts_host == nullptr) return h;
    const int32_t *c = hot->counts_host + 4 * batch;
    h.item = HotSide{hot->piece_q + batch * hot->cap_pieces, hot->piece_len + batch * hot->cap_pieces,
                     hot->run_q + batch * hot->cap_runs, hot->run_first + batch * hot->cap_runs,
                     hot->run_np + batch * hot->cap_runs, c[0], c[1]};
    h.user = HotSide{hot->u_piece_q + batch * hot->cap_u_pieces, hot->u_piece_len + batch * hot->cap_u_pieces,
                     hot->u_run_q + batch * hot->cap_u_runs, hot->u_run_first + batch * hot->cap_u_runs,
                     hot->u_run_np + batch * hot->cap_u_runs, c[2], c[3]};
    return h;
}

static inline int64_t step_ws_bytes(int64_t B, int32_t D) {
    // Z stash [B, D] + loss partials (one per user-phase block; bounded by B for the smallest team count)
    return align_up(B * (int64_t)D * 4, 256) + align_up((n_blocks_for(B, D) + hot_cap_pieces(B, 1)) * 4, 256) +
           align_up(hot_cap_pieces(B, 0) * (int64_t)D * 4, 256) + align_up(hot_cap_pieces(B, 1) * (int64_t)D * 4, 256) + 256;
}

static inline StepWs carve_step_ws(void *workspace, int64_t B, int32_t D) {
    StepWs w;
    w.Z = reinterpret_cast<float *>(workspace);
    w.partials = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + align_up(B * (int64_t)D * 4, 256));
    w.hotP = reinterpret_cast<float *>(reinterpret_cast<char *>(w.partials) +
                                       align_up((n_blocks_for(B, D) + hot_cap_pieces(B, 1)) * 4, 256));
    w.hotPU = reinterpret_cast<float *>(reinterpret_cast<char *>(w.hotP) + align_up(hot_cap_pieces(B, 0) * (int64_t)D * 4, 256));
    w.hot_loss = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(w.hotPU) +
                                                        align_up(hot_cap_pieces(B, 1) * (int64_t)D * 4, 256));
    w.n_partials = n_blocks_for(B, D);
    return w;
}

template <int MODE>
static int32_t launch_step(float *U, float *I, int32_t D, const int32_t *tu, const int32_t *tp, const int32_t *tn,
                           const int32_t *oc_item, const int32_t *oc_src, int64_t B, float lr, float l2, float *gradU,
                           float *gradI, int32_t *stamp_u, int32_t *stamp_i, int32_t step_id, float *loss_out,
                           void *workspace, hipStream_t stream, void *const *events = nullptr, float denom = 0.f,
                           HotBatch hot = HotBatch{{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0},
                                                   {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0}},
                           int64_t ws_batch = 0, AdamArgs ad = AdamArgs{}) {
    if (denom <= 0.f) denom = (float)B;  // single-device step: mean over this batch
    if (ws_batch <= 0) ws_batch = B;     // the workspace was sized for the plan's batch size (>= this batch)
    const StepWs w = carve_step_ws(workspace, ws_batch, D);
    const dim3 block(kBlock);
    const bool have_hot = hot.item.n_runs > 0 && hot.item.n_pieces > 0;
    const bool have_hot_u = hot.user.n_runs > 0 && hot.user.n_pieces > 0;
    WR_REQUIRE(MODE != 4 || (!have_hot && !have_hot_u), WR_E_RANGE,
               "the folded Adam step does not take batches with hot rows (use the catch-up pass + wr_bprmf_step_adam)");
    WR_REQUIRE(hot.item.n_pieces <= hot_cap_pieces(ws_batch, 0) && hot.item.n_runs <= hot_cap_runs(ws_batch, 0) &&
                   hot.user.n_pieces <= hot_cap_pieces(ws_batch, 1) && hot.user.n_runs <= hot_cap_runs(ws_batch, 1),
               WR_E_RANGE, "hot-run counts exceed their capacity");
    const size_t lds_rows = (size_t)teams_per_block(D) * D * 4;
    // SLOTS = 1: two positions per team (twice the loads in flight, one resident round of workgroups at B = 65,536) was
    // A/B-tested on MI355X and is ~4 % slower — the kernel is bound by the memory system's random 256-B row rate, not by
    // bytes in flight.
    const dim3 gridA((unsigned)n_blocks_for(B, D));
    const dim3 gridB((unsigned)((2 * B + kItemTile - 1) / kItemTile));  // item phase: one thread per occurrence
    // Timing hooks, four events per step, any of them NULL: [0] / [1] start of the first and stop of the last kernel of the
    // user phase, [2] / [3] the same for the item phase.  They are attached to the dispatches (hipExtLaunchKernelGGL: the
    // events carry the kernel's own start / end timestamps, the ones rocprofv3 reports); no marker packet is queued.
    auto ev = [&](int j) { return events ? reinterpret_cast<hipEvent_t>(events[j]) : (hipEvent_t) nullptr; };
    const hipEvent_t none = nullptr;
#define WR_LAUNCH(kernel_, grid_, lds_, start_, stop_, ...)                                                               \
    do {                                                                                                                  \
        if ((start_) != nullptr || (stop_) != nullptr)                                                                    \
            hipExtLaunchKernelGGL(kernel_, grid_, block, lds_, stream, (start_), (stop_), 0, __VA_ARGS__);                 \
        else                                                                                                              \
            hipLaunchKernelGGL(kernel_, grid_, block, lds_, stream, __VA_ARGS__);                                          \
    } while (0)
#define WR_CALL_USER(T_, NV_, FULL_)                                                                                      \
    do {                                                                                                                  \
        if (have_hot_u)                                                                                                   \
            WR_LAUNCH((bprmf_user_phase<T_, NV_, FULL_, MODE, 1, true>), gridA, 0, ev(0), none, U, I, D, tu, tp,         \
                      tn, (int)B, lr, l2, w.Z, w.partials, gradU, stamp_u, gradI, stamp_i, step_id, denom, ad,             \
                      (const int *)nullptr, (const int *)nullptr, 0, (const int *)nullptr);                               \
        else                                                                                                              \
            WR_LAUNCH((bprmf_user_phase<T_, NV_, FULL_, MODE, 1, false>), gridA, 0, ev(0), ev(1), U, I, D, tu,             \
                      tp, tn, (int)B, lr, l2, w.Z, w.partials, gradU, stamp_u, gradI, stamp_i, step_id, denom, ad,         \
                      (const int *)nullptr, (const int *)nullptr, 0, (const int *)nullptr);                               \
    } while (0)
    WR_DISPATCH_D(D, WR_CALL_USER);
#undef WR_CALL_USER
    WR_LAUNCH_CHECK("bprmf_user_phase");
    if (have_hot_u) {  // users with more than kHotRun triplets in this batch: many workgroups per user, fixed order
        WR_HIP(hipMemsetAsync(w.hot_loss, 0, 8, stream));
        const unsigned grun = (unsigned)n_blocks_for(hot.user.n_runs, D);
#define WR_CALL_HOTU(T_, NV_, FULL_)                                                                                      \
    do {                                                                                                                 \
        hipLaunchKernelGGL((bprmf_user_hot_pieces<T_, NV_, FULL_, MODE>), dim3((unsigned)hot.user.n_pieces), block,        \
                           lds_rows, stream, U, I, D, tu, tp, tn, lr, l2, w.Z, w.partials, gradI, stamp_i, step_id, denom,  \
                           hot.user.piece_q, hot.user.piece_len, w.hotPU, w.hot_loss, ad);                                \
        WR_LAUNCH((bprmf_user_hot_combine<T_, NV_, FULL_, MODE>), dim3(grun), 0, none, ev(1), U, D, tu,                    \
                  hot.user.run_q, hot.user.run_first, hot.user.run_np, hot.user.n_runs, w.hotPU, lr, l2, gradU,            \
                  stamp_u, step_id, ad);                                                                                 \
    } while (0)
        WR_DISPATCH_D(D, WR_CALL_HOTU);
#undef WR_CALL_HOTU
        WR_LAUNCH_CHECK("bprmf_user_hot_*");
    }
    // hot pieces ride along as extra workgroups of the same launch (they only read the stash and write hotP)
    const dim3 gridBP(gridB.x + (have_hot ? (unsigned)hot.item.n_pieces : 0u));
#define WR_CALL_ITEM(T_, NV_, FULL_)                                                                                   \
    do {                                                                                                               \
        if (have_hot)                                                                                                  \
            WR_LAUNCH((bprmf_item_phase<T_, NV_, FULL_, MODE, true>), gridBP, lds_rows, ev(2), none, I, D,         \
                      oc_item, oc_src, (int)(2 * B), w.Z, lr, l2, gradI, stamp_i, step_id, w.partials,                 \
                      (int)gridA.x, denom, loss_out, 1, have_hot_u ? w.hot_loss : nullptr, (int)gridB.x,               \
                      hot.item.piece_q, hot.item.piece_len, w.hotP, ad);                                               \
        else                                                                                                           \
            WR_LAUNCH((bprmf_item_phase<T_, NV_, FULL_, MODE, false>), gridB, 0, ev(2), ev(3), I, D,                  \
                      oc_item, oc_src, (int)(2 * B), w.Z, lr, l2, gradI, stamp_i, step_id, w.partials,                 \
                      (int)gridA.x, denom, loss_out, 0, have_hot_u ? w.hot_loss : nullptr, (int)gridB.x,               \
                      (const int *)nullptr, (const int *)nullptr, (float *)nullptr, ad);                               \
    } while (0)
    WR_DISPATCH_D(D, WR_CALL_ITEM);
#undef WR_CALL_ITEM
    WR_LAUNCH_CHECK("bprmf_item_phase");
    if (have_hot) {
        const unsigned grun = (unsigned)n_blocks_for(hot.item.n_runs, D);
#define WR_CALL_HOT(T_, NV_, FULL_)                                                                                      \
    do {                                                                                                                \
        WR_LAUNCH((bprmf_item_hot_combine<T_, NV_, FULL_, MODE>), dim3(grun), 0, none, ev(3), I, D, oc_item,             \
                  hot.item.run_q, hot.item.run_first, hot.item.run_np, hot.item.n_runs, w.hotP, lr, l2, gradI,          \
                  stamp_i, step_id, ad);                                                                                \
    } while (0)
        WR_DISPATCH_D(D, WR_CALL_HOT);
#undef WR_CALL_HOT
        WR_LAUNCH_CHECK("bprmf_item_hot_*");
    }
#undef WR_LAUNCH
    return WR_OK;
}

// ----------------------------------------------------------------------------------------------- chained step launch
// ONE launch per step: the item phase of step k-1 rides in the launch that carries the user phase of step k, so its ~5 us
// of latency (two dependent round trips over ~8 K rows) and the kernel boundary in front of it are hidden behind 20 us of
// user-phase traffic.  The launch has three kinds of workgroups, numbered by segments of blockIdx.x:
//     item tiles of batch k-1   rows with several occurrences in batch k-1 are summed and rewritten — WRITE-THROUGH
//                               (store_row_wt) — then every wave waits for its stores, the workgroup meets at a barrier and
//                               one lane adds 1 to the step's counter (agent scope);
//     user runs of batch k      all heads whose bit in the plan's deferred mask is clear (DEF = 1): they read and write no
//                               row an item tile of this launch touches (wr_overlap.hip: the plan marks every run of batch k
//                               that reads a row with several occurrences in batch k-1) — no ordering needed;
//     deferred runs of batch k  (~1.6 % of the runs at 1M x 1M, B = 65,536) at most kChainDefBlocks workgroups walk the
//                               plan's list of deferred heads (DEF = 2): one lane polls the counter (relaxed, agent scope)
//                               until every item tile has signalled, then ONE agent-scope acquire, s_waitcnt vmcnt(0), a
//                               workgroup barrier, and plain loads.
// Forward progress does not rest on dispatch order: only the deferred workgroups ever wait, there are at most
// min(kChainDefBlocks, CUs of the device / 2) of them (launch_chain_steps reads the device's CU count) — fewer than the CUs,
// each of which holds at least one workgroup — and nothing they wait for waits itself, so a free slot always goes to a
// workgroup that runs to completion.  The poll is bounded all the same (an
// exit every wave reaches): on expiry the sticky word `timeout` is set and the workgroup goes on; the host treats a
// non-zero word as a failed run (never observed).
// Rows are handed over in whole 128-B lines (the host takes this path only when D * 4 is a multiple of 128 and the tables
// are 128-B aligned), so no line holds bytes of a row an undeferred run touches beside bytes an item tile rewrites.
// Every table row still has exactly one writer per step and the same summation order: tables bit-identical to the
// two-launch step (tests/test_hip_chain.py).  The loss of a step is summed over more partials (one per deferred chunk): same
// terms, another grouping — equal to ~1 ulp, and a deterministic function of the plan.
// WR_CHAIN_DBG (timing experiments only, never in the shipped build; all variants stay inside the arrays — indices come
// from the plan, not from table values): 1 = no acquire fence, 2 = plain row stores, 4 = no wait for the item tiles,
// 8 = item tiles only signal, 16 = no deferred workgroups.
#ifndef WR_CHAIN_DBG
#define WR_CHAIN_DBG 0
#endif
constexpr int kChainDefBlocks = 128;
constexpr int kChainShards = 64;        // shards of a step's counter
constexpr int kChainShardStride = 16;   // words between shards (64 B)
constexpr int kChainStepWords = kChainShards * kChainShardStride;
constexpr unsigned kChainSpinLimit = 1u << 22;   // polls of ~0.25 us (s_sleep 8): about a second

// MODE 0: plain SGD.  MODE 4: the folded Adam step — the tiles apply step t-1 (`ad_prev`), the user runs step t (`ad`); a row
// handed over is its weights, both moments and its step stamp.
template <int T, int NV, bool FULL, int TILE, int MODE = 0>
__global__ __launch_bounds__(kBlock, NV == 1 ? (MODE == 4 ? WR_ADAM_WAVES : WR_USER_WAVES) : 1) void bprmf_chain_step(
    float *__restrict__ U, float *I, int D,
    // item tiles: batch k-1
    const int *__restrict__ oc_item, const int *__restrict__ oc_src, int B2_prev, const float *__restrict__ Z_prev,
    const float *__restrict__ partials_prev, int n_partials_prev, float denom_prev, float *__restrict__ loss_prev,
    int n_item_blocks, int item_at,
    // user runs: batch k
    const int *__restrict__ tu, const int *__restrict__ tp, const int *__restrict__ tn, int B, float lr,
    float *__restrict__ Z, float *__restrict__ partials, float denom, const int *__restrict__ dmask,
    const int *__restrict__ dlist, int n_def, int n_user_blocks, int def_at, int n_def_blocks,
    unsigned *__restrict__ done, unsigned *__restrict__ timeout, float l2 = 0.f, AdamArgs ad_prev = AdamArgs{},
    AdamArgs ad = AdamArgs{}) {
    __shared__ float scratch[kBlock / 64];
    constexpr int TEAMS = kBlock / T;
    // blockIdx.x -> kind: [user 0 .. item_at) [item tiles] [user item_at .. def_at) [deferred] [user def_at .. n_user_blocks)
    int b = (int)blockIdx.x;
    if (b >= item_at && b < item_at + n_item_blocks) {
        if (!(WR_CHAIN_DBG & 8))
        item_tile_block<T, NV, FULL, MODE, false, !(WR_CHAIN_DBG & 2), TILE>(I, D, oc_item, oc_src, B2_prev, Z_prev, lr, l2, nullptr, nullptr, 0,
                                                           partials_prev, n_partials_prev, denom_prev, loss_prev, nullptr, ad_prev,
                                                           b - item_at);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its write-through stores have left
        __syncthreads();
        // the counter is kept in kChainShards shards on lines of their own: atomic adds execute at the memory side, one
        // address takes one every 10-20 ns (2,048 tiles adding to ONE word held the launch for 22 us)
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(done + ((b - item_at) % kChainShards) * kChainShardStride, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (b >= item_at) b -= n_item_blocks;
    if (b >= def_at && b < def_at + n_def_blocks) {
        if (threadIdx.x < 64) {   // wave 0: lane j polls shard j until it holds the adds of all tiles j, j + 64, ...
            const int j = (int)threadIdx.x;
            const unsigned want = j < kChainShards ? (unsigned)((n_item_blocks - j + kChainShards - 1) / kChainShards) : 0u;
            const unsigned *shard = done + (j < kChainShards ? j : 0) * kChainShardStride;
            unsigned spins = 0;
            while (!(WR_CHAIN_DBG & 4)) {
                const bool ok = __hip_atomic_load(shard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
                if (++spins > kChainSpinLimit) {
                    if (j == 0) __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
#if !(WR_CHAIN_DBG & 1)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        __syncthreads();
        const int n_chunks = (n_def + TEAMS - 1) / TEAMS;
        for (int c = b - def_at; c < n_chunks; c += n_def_blocks) {
            user_phase_block<T, NV, FULL, MODE, 1, false, 2>(U, I, D, tu, tp, tn, B, lr, l2, Z, partials + n_user_blocks + c, nullptr,
                                                          nullptr, nullptr, nullptr, 0, denom, ad, nullptr, dlist, n_def, nullptr,
                                                          c, scratch);
            __syncthreads();   // the next chunk reuses the loss scratch
        }
        return;
    }
    if (b >= def_at) b -= n_def_blocks;
    user_phase_block<T, NV, FULL, MODE, 1, false, 1>(U, I, D, tu, tp, tn, B, lr, l2, Z, partials + b, nullptr, nullptr, nullptr,
                                                  nullptr, 0, denom, ad, dmask, nullptr, 0, nullptr, b, scratch);
}

// Where the item tiles and the deferred workgroups sit among the user workgroups (fractions of the user grid, in 1/16):
// WR_CHAIN_ITEM_AT = 0: the tiles are dispatched first, WR_CHAIN_DEF_AT = 8: the deferred workgroups after half the others.
#ifndef WR_CHAIN_ITEM_AT
#define WR_CHAIN_ITEM_AT 0
#endif
#ifndef WR_CHAIN_DEF_AT
#define WR_CHAIN_DEF_AT 8
#endif
#ifndef WR_CHAIN_TILE
#define WR_CHAIN_TILE 64
#endif

static inline bool chain_shape_ok(const float *U, const float *I, int32_t D) {
    return (D * 4) % 128 == 0 && (reinterpret_cast<uintptr_t>(U) & 127u) == 0 && (reinterpret_cast<uintptr_t>(I) & 127u) == 0;
}

template <int T, int NV, bool FULL, int MODE = 0>
static int32_t launch_chain_steps(float *U, float *I, int32_t D, const int32_t *tu, const int32_t *tp, const int32_t *tn,
                                  const int32_t *oc_item, const int32_t *oc_src, int64_t n_triplets, int64_t batch_size,
                                  int64_t first_batch, int64_t n_batches, float lr, float *loss_out, const int32_t *tdef,
                                  const int32_t *def_q, const int32_t *def_count_host, int64_t def_cap, int64_t def_limit,
                                  void *workspace, uint32_t *sync, int64_t sync_words, hipStream_t stream,
                                  void *const *events, int n_cu, float l2 = 0.f, AdamArgs ad_base = AdamArgs{},
                                  int64_t adam_step0 = 0) {
    const int64_t ws_one = step_ws_bytes(batch_size, D);
    // Forward progress: only the deferred workgroups ever wait, and nothing they wait for waits itself — so it is enough
    // that they never fill the device: at most half a workgroup per CU of THIS device (a CPX partition or a smaller agent
    // has fewer CUs than a whole MI355X's 256), never more than kChainDefBlocks.  With no room for even one, every step
    // takes the two-launch form.
    const int max_def_blocks = std::min(kChainDefBlocks, n_cu / 2);
    const StepWs w2[2] = {carve_step_ws(workspace, batch_size, D),
                          carve_step_ws(reinterpret_cast<char *>(workspace) + ws_one, batch_size, D)};
    const int64_t dwords = (batch_size + 31) / 32;
    constexpr int TEAMS = kBlock / T;
    // occurrences per item tile: 64 for plain SGD (128 / 256: 23.1 / 24.0 us against 23.2), 256 for the folded Adam step, whose
    // tiles carry three rows and a replay per run (64 / 128 / 256: 98.4 / 95.0 / 92.4 us per step)
    constexpr int kTile = MODE == 4 ? 256 : WR_CHAIN_TILE;
    const dim3 block(kBlock);
    // MODE 4: the optimizer step of batch k of this call is adam_step0 + k; its constants are those of wr_adam_consts
    auto ad_of = [&](int64_t k) {
        AdamArgs a = ad_base;
        if (MODE == 4) {
            a.t = (int)(adam_step0 + k);
            adam_step_consts(adam_step0 + k, lr, a.b1, a.b2, &a.step_size, &a.inv_bc2_sqrt);
        }
        return a;
    };
    // sync: one sharded counter per step of this call, zeroed here (the block starts the allocation and is a multiple of 16
    // bytes); the sticky timeout word is the first of the buffer's last four words
    const int64_t n_ctr = n_batches * kChainStepWords;
    WR_HIP(hipMemsetAsync(sync, 0, (size_t)n_ctr * 4, stream));
    uint32_t *timeout = sync + (sync_words - 4);
    auto ev = [&](int64_t k, int j) { return events ? reinterpret_cast<hipEvent_t>(events[4 * k + j]) : (hipEvent_t) nullptr; };
    // the item phase of the previous step, not yet launched: its arrays and what its loss needs
    struct Pending { bool on; int64_t off, Bk; int n_partials; int64_t k; } pend{false, 0, 0, 0, 0};
    auto launch_item = [&](const Pending &q, hipEvent_t e0, hipEvent_t e1) -> int32_t {
        const StepWs &w = w2[q.k & 1];
        const dim3 gridB((unsigned)((2 * q.Bk + kItemTile - 1) / kItemTile));
        const AdamArgs ad = ad_of(q.k);
        if (e0 != nullptr || e1 != nullptr)
            hipExtLaunchKernelGGL((bprmf_item_phase<T, NV, FULL, MODE, false>), gridB, block, 0, stream, e0, e1, 0, I, D,
                                  oc_item + 2 * q.off, oc_src + 2 * q.off, (int)(2 * q.Bk), w.Z, lr, l2, (float *)nullptr,
                                  (int *)nullptr, 0, w.partials, q.n_partials, (float)q.Bk, loss_out ? loss_out + q.k : nullptr, 0,
                                  (const unsigned long long *)nullptr, (int)gridB.x, (const int *)nullptr, (const int *)nullptr,
                                  (float *)nullptr, ad);
        else
            hipLaunchKernelGGL((bprmf_item_phase<T, NV, FULL, MODE, false>), gridB, block, 0, stream, I, D, oc_item + 2 * q.off,
                               oc_src + 2 * q.off, (int)(2 * q.Bk), w.Z, lr, l2, (float *)nullptr, (int *)nullptr, 0, w.partials,
                               q.n_partials, (float)q.Bk, loss_out ? loss_out + q.k : nullptr, 0,
                               (const unsigned long long *)nullptr, (int)gridB.x, (const int *)nullptr, (const int *)nullptr,
                               (float *)nullptr, ad);
        WR_LAUNCH_CHECK("bprmf_item_phase (chain)");
        return WR_OK;
    };
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k;
        const int64_t off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        const StepWs &w = w2[k & 1];
        const int nA = (int)n_blocks_for(Bk, D);
        const int n_def = (k == 0 || !pend.on) ? -1 : def_count_host[b];
        const bool chained = pend.on && n_def >= 0 && n_def <= def_cap && n_def <= def_limit && max_def_blocks >= 1;
        if (chained) {
            const int n_chunks = (WR_CHAIN_DBG & 16) ? 0 : (n_def + TEAMS - 1) / TEAMS;
            const int nD = n_chunks < max_def_blocks ? n_chunks : max_def_blocks;
            const int nI = (int)((2 * pend.Bk + kTile - 1) / kTile);
            const int item_at = (int)((int64_t)nA * WR_CHAIN_ITEM_AT / 16), def_at = (int)((int64_t)nA * WR_CHAIN_DEF_AT / 16);
            const StepWs &wp = w2[pend.k & 1];
            const dim3 grid((unsigned)(nA + nI + nD));
            hipEvent_t e0 = ev(k, 0), e1 = ev(k, 1);
#define WR_CHAIN_ARGS                                                                                                       \
    U, I, D, oc_item + 2 * pend.off, oc_src + 2 * pend.off, (int)(2 * pend.Bk), wp.Z, wp.partials, pend.n_partials,           \
        (float)pend.Bk, loss_out ? loss_out + pend.k : (float *)nullptr, nI, item_at, tu + off, tp + off, tn + off, (int)Bk, lr,  \
        w.Z, w.partials, (float)Bk, tdef + b * dwords, def_q + b * def_cap, n_def, nA, def_at, nD, sync + k * kChainStepWords,   \
        timeout, l2, ad_of(pend.k), ad_of(k)
            if (e0 != nullptr || e1 != nullptr)
                hipExtLaunchKernelGGL((bprmf_chain_step<T, NV, FULL, kTile, MODE>), grid, block, 0, stream, e0, e1, 0,
                                      WR_CHAIN_ARGS);
            else
                hipLaunchKernelGGL((bprmf_chain_step<T, NV, FULL, kTile, MODE>), grid, block, 0, stream, WR_CHAIN_ARGS);
#undef WR_CHAIN_ARGS
            WR_LAUNCH_CHECK("bprmf_chain_step");
            pend = Pending{true, off, Bk, nA + n_chunks, k};
        } else {
            if (pend.on) {
                const int32_t rc = launch_item(pend, ev(pend.k, 2), ev(pend.k, 3));
                if (rc != WR_OK) return rc;
            }
            hipEvent_t e0 = ev(k, 0), e1 = ev(k, 1);
            const AdamArgs ad = ad_of(k);
#define WR_PLAIN_ARGS                                                                                                       \
    U, I, D, tu + off, tp + off, tn + off, (int)Bk, lr, l2, w.Z, w.partials, (float *)nullptr, (int *)nullptr,               \
        (float *)nullptr, (int *)nullptr, 0, (float)Bk, ad, (const int *)nullptr, (const int *)nullptr, 0, (const int *)nullptr
            if (e0 != nullptr || e1 != nullptr)
                hipExtLaunchKernelGGL((bprmf_user_phase<T, NV, FULL, MODE, 1, false, 0>), dim3((unsigned)nA), block, 0, stream, e0, e1,
                                      0, WR_PLAIN_ARGS);
            else
                hipLaunchKernelGGL((bprmf_user_phase<T, NV, FULL, MODE, 1, false, 0>), dim3((unsigned)nA), block, 0, stream,
                                   WR_PLAIN_ARGS);
#undef WR_PLAIN_ARGS
            WR_LAUNCH_CHECK("bprmf_user_phase (chain, first)");
            pend = Pending{true, off, Bk, nA, k};
        }
    }
    if (pend.on) return launch_item(pend, ev(pend.k, 2), ev(pend.k, 3));
    return WR_OK;
}

static int32_t check_plan_args(const void *tu, const void *tp, const void *tn, const void *oc_item, const void *oc_src,
                               int64_t B) {
    WR_REQUIRE(tu && tp && tn && oc_item && oc_src, WR_E_NULL, "plan arrays must not be NULL");
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "batch size %lld out of range (1..2^29)", (long long)B);
    return WR_OK;
}

}  // namespace wr

using namespace wr;

extern "C" {

int64_t wr_bpr_fwd_workspace_bytes(int64_t B) { return align_up(((B + 15) / 16 + 1) * 4, 256); }

int32_t wr_bpr_fwd(const float *user_tab, int64_t n_users, const float *item_tab, int64_t n_items, int32_t D,
                   const int64_t *u, const int64_t *p, const int64_t *n, int64_t B, float *pos_score, float *neg_score,
                   float *coef, float *loss, void *workspace, int64_t workspace_bytes, void *stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    WR_REQUIRE(u && p && n, WR_E_NULL, "index arrays must not be NULL");
    WR_REQUIRE(B > 0 && B <= (int64_t(1) << 29), WR_E_SHAPE, "B=%lld out of range", (long long)B);
    WR_REQUIRE(loss != nullptr || pos_score || neg_score || coef, WR_E_NULL, "no output requested");
    const int64_t nblk = n_blocks_for(B, D);
    WR_REQUIRE(workspace != nullptr && workspace_bytes >= nblk * 4, WR_E_WORKSPACE,
               "wr_bpr_fwd: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)(nblk * 4));
    float *partials = reinterpret_cast<float *>(workspace);
#define WR_CALL_FWD(T_, NV_, FULL_)                                                                               \
    hipLaunchKernelGGL((bpr_fwd_kernel<T_, NV_, FULL_>), dim3((unsigned)nblk), dim3(kBlock), 0, stream, user_tab,  \
                       item_tab, D, u, p, n, (int)B, pos_score, neg_score, coef, partials, n_users, n_items)
    WR_DISPATCH_D(D, WR_CALL_FWD);
#undef WR_CALL_FWD
    WR_LAUNCH_CHECK("bpr_fwd_kernel");
    if (loss != nullptr) {
        hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(kBlock), 0, stream, partials, (int)nblk, (float)B, loss);
        WR_LAUNCH_CHECK("finish_loss_kernel");
    }
    return WR_OK;
}

int64_t wr_bprmf_step_workspace_bytes(int64_t B, int32_t D) { return step_ws_bytes(B, D); }

void wr_bprmf_hot_caps(int64_t batch_size, int32_t kind, int64_t *cap_pieces, int64_t *cap_runs) {
    if (cap_pieces) *cap_pieces = hot_cap_pieces(batch_size, kind ? 1 : 0);
    if (cap_runs) *cap_runs = hot_cap_runs(batch_size, kind ? 1 : 0);
}

int32_t wr_bprmf_step_sgd(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                          const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                          const int32_t *oc_src, int64_t B, float lr, float l2, int32_t *stamp_u, int32_t *stamp_i,
                          int32_t step_id, float *loss_out, const wr_hot_runs *hot, void *workspace,
                          int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, B)) != WR_OK) return rc;
    WR_REQUIRE(l2 == 0.0f || (stamp_u && stamp_i), WR_E_NULL, "l2 != 0 needs stamp_u/stamp_i for the dense decay pass");
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(B, D), WR_E_WORKSPACE,
               "wr_bprmf_step_sgd: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)step_ws_bytes(B, D));
    return launch_step<0>(user_tab, item_tab, D, tu, tp, tn, oc_item, oc_src, B, lr, l2, nullptr, nullptr, stamp_u,
                          stamp_i, step_id, loss_out, workspace, reinterpret_cast<hipStream_t>(stream_), nullptr, 0.f,
                          hot_of(hot, 0));
}

int32_t wr_bprmf_run_sgd(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                         const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                         const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                         int64_t n_batches, float lr, float *loss_out, void *const *phase_events, const wr_hot_runs *hot,
                         void *workspace, int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, batch_size)) != WR_OK) return rc;
    WR_REQUIRE(n_triplets > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_run_sgd: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)step_ws_bytes(batch_size, D));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k;
        const int64_t off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        rc = launch_step<0>(user_tab, item_tab, D, tu + off, tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk,
                            lr, 0.0f, nullptr, nullptr, nullptr, nullptr, 0, loss_out ? loss_out + k : nullptr,
                            workspace, stream, phase_events ? phase_events + 4 * k : nullptr, 0.f, hot_of(hot, b),
                            batch_size);
        if (rc != WR_OK) return rc;
    }
    return WR_OK;
}

int32_t wr_bprmf_grads(const float *user_tab, int64_t n_users, const float *item_tab, int64_t n_items, int32_t D,
                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                       const int32_t *oc_src, int64_t B, float *grad_u, float *grad_i, int32_t *stamp_u,
                       int32_t *stamp_i, int32_t step_id, float *loss_out, const wr_hot_runs *hot, void *workspace,
                       int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(grad_u, n_users, D, "grad_u")) != WR_OK) return rc;
    if ((rc = check_table(grad_i, n_items, D, "grad_i")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, B)) != WR_OK) return rc;
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(B, D), WR_E_WORKSPACE,
               "wr_bprmf_grads: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)step_ws_bytes(B, D));
    // MODE 1 never writes the tables; the const_cast only serves the shared kernel signature.
    return launch_step<1>(const_cast<float *>(user_tab), const_cast<float *>(item_tab), D, tu, tp, tn, oc_item, oc_src, B,
                          0.f, 0.f, grad_u, grad_i, stamp_u, stamp_i, step_id, loss_out, workspace,
                          reinterpret_cast<hipStream_t>(stream_), nullptr, 0.f, hot_of(hot, 0));
}

int32_t wr_bprmf_step_adam(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                           float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                           const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src, int64_t B,
                           int64_t adam_step, float lr, float l2, float beta1, float beta2, float eps, float *loss_out,
                           const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(m_u, n_users, D, "m_u")) != WR_OK) return rc;
    if ((rc = check_table(v_u, n_users, D, "v_u")) != WR_OK) return rc;
    if ((rc = check_table(m_i, n_items, D, "m_i")) != WR_OK) return rc;
    if ((rc = check_table(v_i, n_items, D, "v_i")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, B)) != WR_OK) return rc;
    WR_REQUIRE(last_u != nullptr && last_i != nullptr, WR_E_NULL, "last_u / last_i is NULL");
    WR_REQUIRE(adam_step >= 1 && adam_step < INT32_MAX, WR_E_RANGE, "adam_step must be >= 1");
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(B, D), WR_E_WORKSPACE,
               "wr_bprmf_step_adam: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)step_ws_bytes(B, D));
    AdamArgs ad{m_u, v_u, m_i, v_i, last_u, last_i, 0.f, 0.f, beta1, beta2, eps, l2, (int)adam_step, nullptr};
    adam_step_consts(adam_step, lr, beta1, beta2, &ad.step_size, &ad.inv_bc2_sqrt);
    return launch_step<3>(user_tab, item_tab, D, tu, tp, tn, oc_item, oc_src, B, lr, l2, nullptr, nullptr, nullptr, nullptr, 0,
                          loss_out, workspace, reinterpret_cast<hipStream_t>(stream_), nullptr, 0.f, hot_of(hot, 0), 0, ad);
}

int32_t wr_bprmf_step_adam_folded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                                  float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                                  const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                                  int64_t B, int64_t adam_step, float lr, const float *consts, int64_t n_consts, float l2,
                                  float beta1, float beta2, float eps, float *loss_out, void *workspace,
                                  int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(m_u, n_users, D, "m_u")) != WR_OK) return rc;
    if ((rc = check_table(v_u, n_users, D, "v_u")) != WR_OK) return rc;
    if ((rc = check_table(m_i, n_items, D, "m_i")) != WR_OK) return rc;
    if ((rc = check_table(v_i, n_items, D, "v_i")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, B)) != WR_OK) return rc;
    WR_REQUIRE(last_u != nullptr && last_i != nullptr && consts != nullptr, WR_E_NULL, "last_u / last_i / consts is NULL");
    WR_REQUIRE(adam_step >= 1 && adam_step < n_consts && adam_step < INT32_MAX, WR_E_RANGE,
               "adam_step %lld outside the consts table (%lld entries)", (long long)adam_step, (long long)n_consts);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(B, D), WR_E_WORKSPACE,
               "wr_bprmf_step_adam_folded: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)step_ws_bytes(B, D));
    AdamArgs ad{m_u, v_u, m_i, v_i, last_u, last_i, 0.f, 0.f, beta1, beta2, eps, l2, (int)adam_step, consts};
    adam_step_consts(adam_step, lr, beta1, beta2, &ad.step_size, &ad.inv_bc2_sqrt);   // = consts[2t], consts[2t+1]
    return launch_step<4>(user_tab, item_tab, D, tu, tp, tn, oc_item, oc_src, B, lr, l2, nullptr, nullptr, nullptr, nullptr, 0,
                          loss_out, workspace, reinterpret_cast<hipStream_t>(stream_), nullptr, 0.f, hot_of(nullptr, 0), 0, ad);
}

int32_t wr_bprmf_run_stateful(int32_t kind, float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                              float *s1_u, float *s2_u, float *s1_i, float *s2_i, int32_t *last_u, int32_t *last_i,
                              const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                              const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                              int64_t n_batches, int64_t step0, float lr, float rho, float eps, float *loss_out,
                              const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    WR_REQUIRE(kind == 1 || kind == 2, WR_E_RANGE, "kind must be 1 (Adagrad) or 2 (Adadelta)");
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(s1_u, n_users, D, "s1_u")) != WR_OK) return rc;
    if ((rc = check_table(s1_i, n_items, D, "s1_i")) != WR_OK) return rc;
    if (kind == 2) {
        if ((rc = check_table(s2_u, n_users, D, "s2_u")) != WR_OK) return rc;
        if ((rc = check_table(s2_i, n_items, D, "s2_i")) != WR_OK) return rc;
        WR_REQUIRE(last_u != nullptr && last_i != nullptr, WR_E_NULL, "Adadelta needs last_u / last_i");
    }
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, batch_size)) != WR_OK) return rc;
    WR_REQUIRE(n_triplets > 0 && first_batch >= 0 && n_batches >= 0 && step0 >= 1 && step0 + n_batches < INT32_MAX, WR_E_SHAPE,
               "bad batch / step range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_run_stateful: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)step_ws_bytes(batch_size, D));
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k;
        const int64_t off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        AdamArgs ad{s1_u, s2_u, s1_i, s2_i, last_u, last_i, lr, 0.f, rho, 0.f, eps, 0.f, (int)(step0 + k), nullptr};
        float *lo = loss_out ? loss_out + k : nullptr;
        if (kind == 1)
            rc = launch_step<5>(user_tab, item_tab, D, tu + off, tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk, lr,
                                0.f, nullptr, nullptr, nullptr, nullptr, 0, lo, workspace, stream, nullptr, 0.f, hot_of(hot, b),
                                batch_size, ad);
        else
            rc = launch_step<6>(user_tab, item_tab, D, tu + off, tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk, lr,
                                0.f, nullptr, nullptr, nullptr, nullptr, 0, lo, workspace, stream, nullptr, 0.f, hot_of(hot, b),
                                batch_size, ad);
        if (rc != WR_OK) return rc;
    }
    return WR_OK;
}

// wr_bprmf_run_stateful with a bounded lag for Adadelta's state rows (kind 2; Adagrad has nothing to replay): before every
// step a rotating window of ceil(rows / max_lag) consecutive rows of each table takes the decays it missed
// (wr_adadelta_decay_all on the sub-range) — the finisher of a row otherwise replays them itself, and with small batches on
// big tables the longest such replay among a batch's rows holds the launch (see wr_bprmf_run_adam_lazy_bounded).
int32_t wr_bprmf_run_stateful_bounded(int32_t kind, float *user_tab, int64_t n_users, float *item_tab, int64_t n_items,
                                      int32_t D, float *s1_u, float *s2_u, float *s1_i, float *s2_i, int32_t *last_u,
                                      int32_t *last_i, const int32_t *tu, const int32_t *tp, const int32_t *tn,
                                      const int32_t *oc_item, const int32_t *oc_src, int64_t n_triplets, int64_t batch_size,
                                      int64_t first_batch, int64_t n_batches, int64_t step0, float lr, float rho, float eps,
                                      float *loss_out, const wr_hot_runs *hot, int64_t max_lag, int64_t *sweep_pos,
                                      void *workspace, int64_t workspace_bytes, void *stream) {
    WR_REQUIRE(max_lag >= 1 && sweep_pos != nullptr && sweep_pos[0] >= 0 && sweep_pos[1] >= 0, WR_E_RANGE,
               "max_lag must be >= 1 and sweep_pos given");
    WR_REQUIRE(n_users > 0 && n_items > 0 && n_batches >= 0 && step0 >= 1, WR_E_SHAPE, "bad sizes");
    const int64_t rows[2] = {(n_users + max_lag - 1) / max_lag, (n_items + max_lag - 1) / max_lag};
    float *s1[2] = {s1_u, s1_i}, *s2[2] = {s2_u, s2_i};
    int32_t *lasts[2] = {last_u, last_i};
    const int64_t n_rows[2] = {n_users, n_items};
    for (int64_t k = 0; k < n_batches; ++k) {
        int32_t rc;
        const int64_t t = step0 + k;
        for (int side = 0; side < 2 && kind == 2 && t > 1; ++side) {
            int64_t lo = sweep_pos[side] % n_rows[side], left = rows[side] < n_rows[side] ? rows[side] : n_rows[side];
            while (left > 0) {
                const int64_t c = left < n_rows[side] - lo ? left : n_rows[side] - lo;
                if ((rc = wr_adadelta_decay_all(s1[side] + lo * (int64_t)D, s2[side] + lo * (int64_t)D, lasts[side] + lo, c, D,
                                                t - 1, rho, stream)) != WR_OK) return rc;
                lo = (lo + c) % n_rows[side];
                left -= c;
            }
            sweep_pos[side] = lo;
        }
        if ((rc = wr_bprmf_run_stateful(kind, user_tab, n_users, item_tab, n_items, D, s1_u, s2_u, s1_i, s2_i, last_u, last_i, tu,
                                        tp, tn, oc_item, oc_src, n_triplets, batch_size, first_batch + k, 1, t, lr, rho, eps,
                                        loss_out ? loss_out + k : nullptr, hot, workspace, workspace_bytes, stream)) != WR_OK)
            return rc;
    }
    return WR_OK;
}

int32_t wr_bprmf_shard_step(float *user_shard, int64_t n_user_rows, float *item_rows, int64_t n_rows, int64_t n_local_items,
                            int32_t D, const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                            const int32_t *oc_src, int64_t B, int64_t global_batch, float lr, float *grad_slots,
                            float *loss_partial, const wr_hot_runs *hot, void *workspace, int64_t workspace_bytes,
                            void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_shard, n_user_rows, D, "user_shard")) != WR_OK) return rc;
    if ((rc = check_table(item_rows, n_rows, D, "item_rows")) != WR_OK) return rc;
    WR_REQUIRE(n_local_items >= 0 && n_local_items <= n_rows, WR_E_SHAPE, "n_local_items %lld outside [0, %lld]",
               (long long)n_local_items, (long long)n_rows);
    WR_REQUIRE(grad_slots != nullptr && aligned16(grad_slots), WR_E_NULL, "grad_slots is NULL or not 16-byte aligned");
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, B)) != WR_OK) return rc;
    WR_REQUIRE(global_batch >= B, WR_E_SHAPE, "global_batch %lld < local batch %lld", (long long)global_batch, (long long)B);
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= step_ws_bytes(B, D), WR_E_WORKSPACE,
               "wr_bprmf_shard_step: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)step_ws_bytes(B, D));
    // MODE 2: rows below n_local_items are rewritten in place, the others only read (their gradients go to grad_slots)
    AdamArgs ad{};
    ad.t = (int)n_local_items;
    return launch_step<2>(user_shard, item_rows, D, tu, tp, tn, oc_item, oc_src, B, lr, 0.f, nullptr,
                          grad_slots, nullptr, nullptr, 0, loss_partial, workspace, reinterpret_cast<hipStream_t>(stream_),
                          nullptr, (float)global_batch, hot_of(hot, 0), 0, ad);
}

int32_t wr_bprmf_chain_supported(const float *user_tab, const float *item_tab, int32_t D) {
    return (user_tab && item_tab && D >= 4 && D <= 1024 && D % 4 == 0 && chain_shape_ok(user_tab, item_tab, D)) ? 1 : 0;
}

int64_t wr_bprmf_chain_sync_words(int64_t n_batches) { return n_batches < 0 ? WR_E_SHAPE : n_batches * kChainStepWords + 4; }

int32_t wr_bprmf_run_sgd_chain(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                               const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                               const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                               int64_t n_batches, float lr, float *loss_out, const int32_t *tdef, const int32_t *def_q,
                               const int32_t *def_count_host, int64_t def_cap, int64_t def_limit, void *const *events,
                               void *workspace, int64_t workspace_bytes, int32_t *sync, int64_t sync_words, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, batch_size)) != WR_OK) return rc;
    WR_REQUIRE(tdef && def_q && def_count_host && sync, WR_E_NULL, "chain marks / sync words must not be NULL");
    WR_REQUIRE(chain_shape_ok(user_tab, item_tab, D), WR_E_ALIGN,
               "wr_bprmf_run_sgd_chain: rows must be whole 128-B lines (D %% 32 == 0, tables 128-B aligned); D = %d", (int)D);
    WR_REQUIRE(n_triplets > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(def_cap > 0 && def_limit >= 0, WR_E_RANGE, "bad deferred-run capacity");
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= 2 * step_ws_bytes(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_run_sgd_chain: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)(2 * step_ws_bytes(batch_size, D)));
    WR_REQUIRE(aligned16(sync) && sync_words >= n_batches * kChainStepWords + 4, WR_E_WORKSPACE,
               "wr_bprmf_run_sgd_chain: %lld sync words < %lld", (long long)sync_words,
               (long long)(n_batches * kChainStepWords + 4));
    if (n_batches == 0) return WR_OK;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        WR_HIP(hipGetDevice(&dev));
        WR_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
#define WR_CALL_CHAIN(T_, NV_, FULL_)                                                                                      \
    return launch_chain_steps<T_, NV_, FULL_>(user_tab, item_tab, D, tu, tp, tn, oc_item, oc_src, n_triplets, batch_size,   \
                                              first_batch, n_batches, lr, loss_out, tdef, def_q, def_count_host, def_cap,  \
                                              def_limit, workspace, reinterpret_cast<uint32_t *>(sync), sync_words, stream, \
                                              events, n_cu)
    WR_DISPATCH_D(D, WR_CALL_CHAIN);
#undef WR_CALL_CHAIN
    return WR_OK;
}


int32_t wr_bprmf_run_adam_folded_chain(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                       float *m_u, float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i,
                                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                       const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                       int64_t n_batches, int64_t adam_step0, float lr, const float *consts, int64_t n_consts,
                                       float l2, float beta1, float beta2, float eps, float *loss_out, const int32_t *tdef,
                                       const int32_t *def_q, const int32_t *def_count_host, int64_t def_cap, int64_t def_limit,
                                       void *workspace, int64_t workspace_bytes, int32_t *sync, int64_t sync_words,
                                       void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    if ((rc = check_table(m_u, n_users, D, "m_u")) != WR_OK) return rc;
    if ((rc = check_table(v_u, n_users, D, "v_u")) != WR_OK) return rc;
    if ((rc = check_table(m_i, n_items, D, "m_i")) != WR_OK) return rc;
    if ((rc = check_table(v_i, n_items, D, "v_i")) != WR_OK) return rc;
    if ((rc = check_plan_args(tu, tp, tn, oc_item, oc_src, batch_size)) != WR_OK) return rc;
    WR_REQUIRE(last_u && last_i && consts && tdef && def_q && def_count_host && sync, WR_E_NULL,
               "last_u / last_i / consts / chain marks / sync words must not be NULL");
    WR_REQUIRE(chain_shape_ok(user_tab, item_tab, D) && chain_shape_ok(m_u, m_i, D) && chain_shape_ok(v_u, v_i, D), WR_E_ALIGN,
               "wr_bprmf_run_adam_folded_chain: rows must be whole 128-B lines (D %% 32 == 0, tables 128-B aligned); D = %d", (int)D);
    WR_REQUIRE(n_triplets > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(adam_step0 >= 1 && adam_step0 + n_batches <= n_consts && adam_step0 + n_batches < INT32_MAX, WR_E_RANGE,
               "adam steps [%lld,%lld) outside the consts table (%lld entries)", (long long)adam_step0,
               (long long)(adam_step0 + n_batches), (long long)n_consts);
    WR_REQUIRE(def_cap > 0 && def_limit >= 0, WR_E_RANGE, "bad deferred-run capacity");
    WR_REQUIRE(workspace && aligned16(workspace) && workspace_bytes >= 2 * step_ws_bytes(batch_size, D), WR_E_WORKSPACE,
               "wr_bprmf_run_adam_folded_chain: workspace %lld B < %lld B", (long long)workspace_bytes,
               (long long)(2 * step_ws_bytes(batch_size, D)));
    WR_REQUIRE(aligned16(sync) && sync_words >= n_batches * kChainStepWords + 4, WR_E_WORKSPACE,
               "wr_bprmf_run_adam_folded_chain: %lld sync words < %lld", (long long)sync_words,
               (long long)(n_batches * kChainStepWords + 4));
    if (n_batches == 0) return WR_OK;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        WR_HIP(hipGetDevice(&dev));
        WR_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const AdamArgs base{m_u, v_u, m_i, v_i, last_u, last_i, 0.f, 0.f, beta1, beta2, eps, l2, 0, consts};
#define WR_CALL_ACHAIN(T_, NV_, FULL_)                                                                                       \
    return launch_chain_steps<T_, NV_, FULL_, 4>(user_tab, item_tab, D, tu, tp, tn, oc_item, oc_src, n_triplets, batch_size,  \
                                                 first_batch, n_batches, lr, loss_out, tdef, def_q, def_count_host, def_cap,   \
                                                 def_limit, workspace, reinterpret_cast<uint32_t *>(sync), sync_words, stream, \
                                                 nullptr, n_cu, l2, base, adam_step0)
    WR_DISPATCH_D(D, WR_CALL_ACHAIN);
#undef WR_CALL_ACHAIN
    return WR_OK;
}


}  // extern "C"
