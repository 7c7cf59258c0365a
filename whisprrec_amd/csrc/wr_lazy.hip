// wr_lazy.hip — exact lazy evaluation of the dense optimizers of BaseRunner.py:120-124 on big tables.
//
// torch.optim.Adam / SGD(weight_decay) move EVERY row of the tables at every step (reference src/helpers/BaseRunner.py:199:
// a row without gradient still gets its moments decayed and its weights moved by the old momentum / shrunk by l2).  The
// dense kernels of wr_rows.hip do that literally: 7 table passes per step, ~0.7 ms at 1M x 1M x 64 whatever the batch size.
// A row's trajectory between two batches that contain it depends on nothing but the row itself, so it can be replayed
// later: every row carries the number of the last optimizer step applied to it, and
//   * before a batch's gradients are computed, the rows of THAT batch are brought up to step t-1 by replaying the missed
//     zero-gradient steps element by element (same arithmetic, same order, same per-step bias corrections: same bits);
//   * after the gradients are known, step t is applied to those rows only;
//   * before anything else reads the tables (evaluation, checkpoint), all rows are brought up to date in one pass.
// The replay costs ALU work (one adam_elem per element per missed step) but no memory traffic.
//
// One wave per row (lane-strided elements): rows of one wave replay the same number of steps, so there is no divergence.
#include "wr_common.h"

namespace wr {

constexpr int kWave = 64;

// R = rows a wave has in flight: their loads are issued together, the replays run one after the other (each at full lane
// use, no divergence).  4 for big batches (memory-level parallelism), 1 for small ones (more waves to spread the replays).
constexpr int kRowsPerWave = 4;
constexpr int64_t kSmallBatchKeys = 32768;

// Adam: bring each row from step last[row] to step `upto` (zero gradient), then, if grad != nullptr, apply step upto+1
// with the row's gradient.  consts[2s], consts[2s+1] = step_size and 1/sqrt(bias_correction2) of step s.
// rows[j] < 0: nothing to do for slot j.  All arguments are wave-uniform.
template <bool L2, int R>
__device__ __forceinline__ void adam_rows(float *__restrict__ w, float *__restrict__ m, float *__restrict__ v,
                                          int *__restrict__ last, const int64_t (&rows_in)[R], int D, int upto,
                                          const float *__restrict__ grad, const float *__restrict__ consts, float l2, float b1,
                                          float b2, float eps, int lane) {
    const float2 *__restrict__ c2 = reinterpret_cast<const float2 *>(consts);
    auto replay = [&](float &ww, float &mm, float &vv, float2 k) {   // one gradient-free step
        if (L2) adam_elem<true>(ww, mm, vv, 0.f, l2, b1, b2, eps, k.x, k.y);
        else adam_elem_zero_grad(ww, mm, vv, b1, b2, eps, k.x, k.y);
    };
    int64_t rows[R];
    int from[R];
#pragma unroll
    for (int j = 0; j < R; ++j) from[j] = rows_in[j] >= 0 ? last[rows_in[j]] : 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        from[j] = __builtin_amdgcn_readfirstlane(from[j]);
        rows[j] = (rows_in[j] >= 0 && (from[j] < upto || grad != nullptr)) ? rows_in[j] : -1;
    }
    for (int e = lane; e < D; e += kWave) {
        float ww[R], mm[R], vv[R], gg[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rows[j] < 0) continue;
            const int64_t at = rows[j] * (int64_t)D + e;
            ww[j] = w[at]; mm[j] = m[at]; vv[j] = v[at];
            gg[j] = grad != nullptr ? grad[at] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rows[j] < 0) continue;
            int s = from[j] + 1;
            for (; s + 3 <= upto; s += 4) {   // four steps per trip: their (wave-uniform) constants are fetched together
                const float2 k0 = c2[s], k1 = c2[s + 1], k2 = c2[s + 2], k3 = c2[s + 3];
                replay(ww[j], mm[j], vv[j], k0); replay(ww[j], mm[j], vv[j], k1);
                replay(ww[j], mm[j], vv[j], k2); replay(ww[j], mm[j], vv[j], k3);
            }
            for (; s <= upto; ++s) replay(ww[j], mm[j], vv[j], c2[s]);
            if (grad != nullptr) adam_elem<L2>(ww[j], mm[j], vv[j], gg[j], l2, b1, b2, eps, c2[upto + 1].x, c2[upto + 1].y);
            const int64_t at = rows[j] * (int64_t)D + e;
            w[at] = ww[j]; m[at] = mm[j]; v[at] = vv[j];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (rows[j] >= 0) last[rows[j]] = grad != nullptr ? upto + 1 : upto;
    }
}

// keys: row ids of one batch, equal ids adjacent (the plan's tu / oc_item arrays).  A wave takes kRowsPerWave consecutive
// keys; the first key of a run stands for the row.
template <int R>
__device__ __forceinline__ void head_rows(const int *__restrict__ keys, int64_t n_keys, int64_t n_rows, int64_t k0,
                                          int64_t (&rows)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int64_t k = k0 + j;
        int key = -1;
        if (k < n_keys) {
            key = keys[k];
            if (k > 0 && keys[k - 1] == key) key = -1;
            if (key >= n_rows) key = -1;   // plans are validated by the caller; never touch memory outside the table
        }
        rows[j] = __builtin_amdgcn_readfirstlane(key);
    }
}

template <bool L2, int R>
__global__ __launch_bounds__(kBlock) void adam_lazy_rows_kernel(float *__restrict__ w, float *__restrict__ m,
                                                                 float *__restrict__ v, int *__restrict__ last,
                                                                 int64_t n_rows, int D, const int *__restrict__ keys,
                                                                 int64_t n_keys, const float *__restrict__ grad, int upto,
                                                                 const float *__restrict__ consts, float l2, float b1,
                                                                 float b2, float eps) {
    const int64_t k0 = ((int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * R;
    if (k0 >= n_keys) return;
    int64_t rows[R];
    head_rows<R>(keys, n_keys, n_rows, k0, rows);
    adam_rows<L2, R>(w, m, v, last, rows, D, upto, grad, consts, l2, b1, b2, eps, threadIdx.x % kWave);
}

// R rows per wave and trip: kRowsPerWave for whole tables (memory parallelism), 1 for a window of a few thousand rows (the
// bounded-lag sweep: one wave per row keeps every SIMD busy)
template <bool L2, int R>
__global__ __launch_bounds__(kBlock) void adam_catchup_all_kernel(float *__restrict__ w, float *__restrict__ m,
                                                                   float *__restrict__ v, int *__restrict__ last,
                                                                   int64_t n_rows, int D, int upto,
                                                                   const float *__restrict__ consts, float l2, float b1,
                                                                   float b2, float eps) {
    const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave) * R;
    for (int64_t r0 = ((int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * R; r0 < n_rows; r0 += stride) {
        int64_t rows[R];
#pragma unroll
        for (int j = 0; j < R; ++j) rows[j] = r0 + j < n_rows ? r0 + j : -1;
        adam_rows<L2, R>(w, m, v, last, rows, D, upto, nullptr, consts, l2, b1, b2, eps, threadIdx.x % kWave);
    }
}

// SGD with weight decay: bring each row to step `upto` (zero gradient) and mark it as being at step `mark`.
template <int R>
__device__ __forceinline__ void sgd_rows(float *__restrict__ w, int *__restrict__ last, const int64_t (&rows_in)[R], int D,
                                         int upto, int mark, float lr, float l2, int lane) {
    int64_t rows[R];
    int from[R];
#pragma unroll
    for (int j = 0; j < R; ++j) from[j] = rows_in[j] >= 0 ? last[rows_in[j]] : 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        from[j] = __builtin_amdgcn_readfirstlane(from[j]);
        rows[j] = (rows_in[j] >= 0 && from[j] < mark) ? rows_in[j] : -1;
    }
    for (int e = lane; e < D; e += kWave) {
        float ww[R];
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (rows[j] >= 0 && from[j] < upto) ww[j] = w[rows[j] * (int64_t)D + e];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (rows[j] < 0 || from[j] >= upto) continue;
            for (int s = from[j]; s < upto; ++s) ww[j] = sgd_decay_elem(ww[j], lr, l2);
            w[rows[j] * (int64_t)D + e] = ww[j];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (rows[j] >= 0) last[rows[j]] = mark;
    }
}

template <int R>
__global__ __launch_bounds__(kBlock) void sgd_lazy_rows_kernel(float *__restrict__ w, int *__restrict__ last, int64_t n_rows,
                                                                int D, const int *__restrict__ keys, int64_t n_keys, int upto,
                                                                int mark, float lr, float l2) {
    const int64_t k0 = ((int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * R;
    if (k0 >= n_keys) return;
    int64_t rows[R];
    head_rows<R>(keys, n_keys, n_rows, k0, rows);
    sgd_rows<R>(w, last, rows, D, upto, mark, lr, l2, threadIdx.x % kWave);
}

template <int R>
__global__ __launch_bounds__(kBlock) void sgd_catchup_all_kernel(float *__restrict__ w, int *__restrict__ last, int64_t n_rows,
                                                                  int D, int upto, float lr, float l2) {
    const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave) * R;
    for (int64_t r0 = ((int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * R; r0 < n_rows; r0 += stride) {
        int64_t rows[R];
#pragma unroll
        for (int j = 0; j < R; ++j) rows[j] = r0 + j < n_rows ? r0 + j : -1;
        sgd_rows<R>(w, last, rows, D, upto, upto, lr, l2, threadIdx.x % kWave);
    }
}

// torch.optim.Adadelta with a zero gradient multiplies both state rows by rho (wr_bpr.hip, MODE 6: the finisher of a row
// replays the multiplications the row missed).  This kernel applies the missed ones of rows [0, n_rows) up to step `upto`
// and marks the rows — the bounded-lag window of wr_bprmf_run_stateful_bounded.  k multiplications are k multiplications,
// whoever applies them: same bits.
__global__ __launch_bounds__(kBlock) void adadelta_decay_all_kernel(float *__restrict__ sq, float *__restrict__ ac,
                                                                     int *__restrict__ last, int64_t n_rows, int D, int upto,
                                                                     float rho) {
    const int lane = threadIdx.x % kWave;
    const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave);
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave; r < n_rows; r += stride) {
        const int missed = upto - __builtin_amdgcn_readfirstlane(last[r]);
        if (missed <= 0) continue;
        for (int e = lane; e < D; e += kWave) {
            float a = sq[r * (int64_t)D + e], b = ac[r * (int64_t)D + e];
            for (int j = 0; j < missed; ++j) {
                a *= rho;
                b *= rho;
            }
            sq[r * (int64_t)D + e] = a;
            ac[r * (int64_t)D + e] = b;
        }
        if (lane == 0) last[r] = upto;
    }
}

static inline unsigned rows_grid(int64_t n, int R) {
    const int64_t per = kBlock / kWave * R;
    const int64_t g = (n + per - 1) / per;
    return (unsigned)(g < 1 ? 1 : g);
}
static inline unsigned all_grid(int64_t n_rows) {
    const int64_t per = kBlock / kWave * kRowsPerWave;
    const int64_t g = (n_rows + per - 1) / per, cap = 256 * 32;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace wr

using namespace wr;

extern "C" {

int32_t wr_adam_consts(int64_t first_step, int64_t n_steps, float lr, float beta1, float beta2, float *consts_host) {
    WR_REQUIRE(consts_host != nullptr && first_step >= 0 && n_steps >= 0, WR_E_NULL, "wr_adam_consts: bad arguments");
    for (int64_t s = first_step; s < first_step + n_steps; ++s)   // the expressions wr_adam_dense evaluates
        adam_step_consts(s, lr, beta1, beta2, &consts_host[2 * (s - first_step)], &consts_host[2 * (s - first_step) + 1]);
    return WR_OK;
}

int32_t wr_adam_rows_lazy(float *tab, float *exp_avg, float *exp_avg_sq, int32_t *last_step, int64_t n_rows, int32_t D,
                          const int32_t *keys, int64_t n_keys, const float *grad, int64_t adam_step, const float *consts,
                          int64_t n_consts, float l2, float beta1, float beta2, float eps, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg, n_rows, D, "exp_avg")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg_sq, n_rows, D, "exp_avg_sq")) != WR_OK) return rc;
    WR_REQUIRE(last_step != nullptr && consts != nullptr, WR_E_NULL, "last_step / consts is NULL");
    WR_REQUIRE(adam_step >= 1 && adam_step < n_consts && adam_step < INT32_MAX, WR_E_RANGE,
               "adam_step %lld outside the consts table (%lld entries)", (long long)adam_step, (long long)n_consts);
    WR_REQUIRE(n_keys >= 0 && (n_keys == 0 || keys != nullptr), WR_E_NULL, "keys is NULL");
    if (n_keys == 0) return WR_OK;
    const bool small = n_keys < kSmallBatchKeys;
    auto kern = small ? (l2 != 0.f ? adam_lazy_rows_kernel<true, 1> : adam_lazy_rows_kernel<false, 1>)
                      : (l2 != 0.f ? adam_lazy_rows_kernel<true, kRowsPerWave> : adam_lazy_rows_kernel<false, kRowsPerWave>);
    hipLaunchKernelGGL(kern, dim3(rows_grid(n_keys, small ? 1 : kRowsPerWave)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream_),
                       tab, exp_avg, exp_avg_sq, last_step, n_rows, D, keys, n_keys, grad, (int)(adam_step - 1), consts, l2,
                       beta1, beta2, eps);
    WR_LAUNCH_CHECK("adam_lazy_rows_kernel");
    return WR_OK;
}

int32_t wr_adam_catchup_all(float *tab, float *exp_avg, float *exp_avg_sq, int32_t *last_step, int64_t n_rows, int32_t D,
                            int64_t adam_step, const float *consts, int64_t n_consts, float l2, float beta1, float beta2,
                            float eps, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg, n_rows, D, "exp_avg")) != WR_OK) return rc;
    if ((rc = check_table(exp_avg_sq, n_rows, D, "exp_avg_sq")) != WR_OK) return rc;
    WR_REQUIRE(last_step != nullptr && consts != nullptr, WR_E_NULL, "last_step / consts is NULL");
    WR_REQUIRE(adam_step >= 0 && adam_step < n_consts && adam_step < INT32_MAX, WR_E_RANGE,
               "adam_step %lld outside the consts table (%lld entries)", (long long)adam_step, (long long)n_consts);
    const bool small = n_rows < kSmallBatchKeys;
    auto kern = small ? (l2 != 0.f ? adam_catchup_all_kernel<true, 1> : adam_catchup_all_kernel<false, 1>)
                      : (l2 != 0.f ? adam_catchup_all_kernel<true, kRowsPerWave> : adam_catchup_all_kernel<false, kRowsPerWave>);
    hipLaunchKernelGGL(kern, dim3(small ? rows_grid(n_rows, 1) : all_grid(n_rows)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream_), tab, exp_avg, exp_avg_sq, last_step, n_rows, D, (int)adam_step,
                       consts, l2, beta1, beta2, eps);
    WR_LAUNCH_CHECK("adam_catchup_all_kernel");
    return WR_OK;
}

int32_t wr_sgd_rows_lazy(float *tab, int32_t *last_step, int64_t n_rows, int32_t D, const int32_t *keys, int64_t n_keys,
                         int64_t step, float lr, float l2, void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    WR_REQUIRE(last_step != nullptr, WR_E_NULL, "last_step is NULL");
    WR_REQUIRE(step >= 1 && step < INT32_MAX, WR_E_RANGE, "step must be >= 1");
    WR_REQUIRE(n_keys >= 0 && (n_keys == 0 || keys != nullptr), WR_E_NULL, "keys is NULL");
    if (n_keys == 0) return WR_OK;
    const bool small = n_keys < kSmallBatchKeys;
    hipLaunchKernelGGL(small ? sgd_lazy_rows_kernel<1> : sgd_lazy_rows_kernel<kRowsPerWave>,
                       dim3(rows_grid(n_keys, small ? 1 : kRowsPerWave)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream_),
                       tab, last_step, n_rows, D, keys, n_keys, (int)(step - 1), (int)step, lr, l2);
    WR_LAUNCH_CHECK("sgd_lazy_rows_kernel");
    return WR_OK;
}

int32_t wr_sgd_catchup_all(float *tab, int32_t *last_step, int64_t n_rows, int32_t D, int64_t step, float lr, float l2,
                           void *stream_) {
    int32_t rc;
    if ((rc = check_table(tab, n_rows, D, "tab")) != WR_OK) return rc;
    WR_REQUIRE(last_step != nullptr, WR_E_NULL, "last_step is NULL");
    WR_REQUIRE(step >= 0 && step < INT32_MAX, WR_E_RANGE, "step must be >= 0");
    if (n_rows < kSmallBatchKeys)     // a window of rows (bounded lag): one wave per row
        hipLaunchKernelGGL(sgd_catchup_all_kernel<1>, dim3(rows_grid(n_rows, 1)), dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream_), tab, last_step, n_rows, D, (int)step, lr, l2);
    else
        hipLaunchKernelGGL(sgd_catchup_all_kernel<kRowsPerWave>, dim3(all_grid(n_rows)), dim3(kBlock), 0,
                           reinterpret_cast<hipStream_t>(stream_), tab, last_step, n_rows, D, (int)step, lr, l2);
    WR_LAUNCH_CHECK("sgd_catchup_all_kernel");
    return WR_OK;
}

static inline wr_hot_runs hot_at_batch(const wr_hot_runs *hot, int64_t b) {
    wr_hot_runs h = *hot;
    h.piece_q += b * hot->cap_pieces; h.piece_len += b * hot->cap_pieces;
    h.run_q += b * hot->cap_runs; h.run_first += b * hot->cap_runs; h.run_np += b * hot->cap_runs;
    h.u_piece_q += b * hot->cap_u_pieces; h.u_piece_len += b * hot->cap_u_pieces;
    h.u_run_q += b * hot->cap_u_runs; h.u_run_first += b * hot->cap_u_runs; h.u_run_np += b * hot->cap_u_runs;
    h.counts_host += 4 * b;
    return h;
}

// The per-batch sequence  catch-up rows -> fused step (gradients + Adam on the finished rows)  for n_batches consecutive
// batches of a plan, issued from native code: at the reference's default batch size (2,048) a Python loop around the calls
// costs about as much as the GPU work of a step.
int32_t wr_bprmf_run_adam_lazy(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                               float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                               const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                               int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches,
                               int64_t adam_step0, float lr, const float *consts, int64_t n_consts, float l2, float beta1,
                               float beta2, float eps, float *loss_out, const wr_hot_runs *hot, void *workspace,
                               int64_t workspace_bytes, void *stream) {
    WR_REQUIRE(n_triplets > 0 && batch_size > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(adam_step0 >= 1 && adam_step0 + n_batches <= n_consts, WR_E_RANGE,
               "adam steps [%lld,%lld) outside the consts table (%lld entries)", (long long)adam_step0,
               (long long)(adam_step0 + n_batches), (long long)n_consts);
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k, off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        const int64_t t = adam_step0 + k;
        int32_t rc;
        wr_hot_runs hb;
        if (hot != nullptr) hb = hot_at_batch(hot, b);
        if ((rc = wr_adam_rows_lazy(user_tab, m_u, v_u, last_u, n_users, D, tu + off, Bk, nullptr, t, consts, n_consts, l2, beta1,
                                    beta2, eps, stream)) != WR_OK) return rc;
        if ((rc = wr_adam_rows_lazy(item_tab, m_i, v_i, last_i, n_items, D, oc_item + 2 * off, 2 * Bk, nullptr, t, consts,
                                    n_consts, l2, beta1, beta2, eps, stream)) != WR_OK) return rc;
        if ((rc = wr_bprmf_step_adam(user_tab, n_users, item_tab, n_items, D, m_u, v_u, m_i, v_i, last_u, last_i, tu + off,
                                     tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk, t, lr, l2, beta1, beta2, eps,
                                     loss_out ? loss_out + k : nullptr, hot ? &hb : nullptr, workspace, workspace_bytes,
                                     stream)) != WR_OK) return rc;
    }
    return WR_OK;
}

// The same loop with a BOUNDED LAG.  Between two uses a row misses (rows / batch rows) steps on average, geometrically
// distributed: with small batches on big tables (the reference's defaults: Adam, B = 2,048; 1M-row tables -> ~490 on
// average) the longest replay among a batch's rows is ~8x the mean, and a row's replay is one wave's serial chain — the
// catch-up kernels then last as long as their unluckiest wave (steady state at 1M x 1M x 64, B = 2,048: 99 us per catch-up
// launch, 196 at worst; the ~20 us figures of short runs only hold while no row has been idle for long).  Here every
// step first sweeps ceil(rows / max_lag) consecutive rows of each table (a rotating window; wr_adam_catchup_all on the
// sub-range) up to step t-1: after max_lag steps the whole table has been visited, so no row ever lags more than max_lag
// steps, the batch rows' replays are short, and the bulk of the replay work — which exact dense-Adam semantics owe for every
// row and step anyway — runs as uniform-length replays over thousands of waves.  A replay is a replay: the same
// operations per row in the same order, so the tables stay bit-identical to the dense optimizer's.
// sweep_pos (host, in/out): [0] next user row of the window, [1] next item row.
static int32_t sweep_window(float *tab, float *m, float *v, int32_t *last, int64_t n_rows, int32_t D, int64_t rows, int64_t *pos,
                            int64_t upto, const float *consts, int64_t n_consts, float l2, float b1, float b2, float eps,
                            void *stream) {
    int64_t lo = *pos % n_rows, left = rows < n_rows ? rows : n_rows;
    while (left > 0) {
        const int64_t c = left < n_rows - lo ? left : n_rows - lo;
        const int32_t rc = wr_adam_catchup_all(tab + lo * (int64_t)D, m + lo * (int64_t)D, v + lo * (int64_t)D, last + lo, c, D,
                                               upto, consts, n_consts, l2, b1, b2, eps, stream);
        if (rc != WR_OK) return rc;
        lo = (lo + c) % n_rows;
        left -= c;
    }
    *pos = lo;
    return WR_OK;
}

int32_t wr_bprmf_run_adam_lazy_bounded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                       float *m_u, float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i,
                                       const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                       const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                       int64_t n_batches, int64_t adam_step0, float lr, const float *consts, int64_t n_consts,
                                       float l2, float beta1, float beta2, float eps, float *loss_out, const wr_hot_runs *hot,
                                       int64_t max_lag, int64_t *sweep_pos, void *workspace, int64_t workspace_bytes,
                                       void *stream) {
    WR_REQUIRE(n_triplets > 0 && batch_size > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    WR_REQUIRE(max_lag >= 1 && sweep_pos != nullptr, WR_E_RANGE, "max_lag must be >= 1 and sweep_pos given");
    WR_REQUIRE(n_users > 0 && n_items > 0 && sweep_pos[0] >= 0 && sweep_pos[1] >= 0, WR_E_SHAPE, "bad sweep position");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(adam_step0 >= 1 && adam_step0 + n_batches <= n_consts, WR_E_RANGE,
               "adam steps [%lld,%lld) outside the consts table (%lld entries)", (long long)adam_step0,
               (long long)(adam_step0 + n_batches), (long long)n_consts);
    const int64_t rows_u = (n_users + max_lag - 1) / max_lag, rows_i = (n_items + max_lag - 1) / max_lag;
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t t = adam_step0 + k;
        int32_t rc;
        if (t > 1) {   // rows of the window -> step t-1 (zero-gradient steps; nothing to do before the first step)
            if ((rc = sweep_window(user_tab, m_u, v_u, last_u, n_users, D, rows_u, &sweep_pos[0], t - 1, consts, n_consts, l2,
                                   beta1, beta2, eps, stream)) != WR_OK) return rc;
            if ((rc = sweep_window(item_tab, m_i, v_i, last_i, n_items, D, rows_i, &sweep_pos[1], t - 1, consts, n_consts, l2,
                                   beta1, beta2, eps, stream)) != WR_OK) return rc;
        }
        if ((rc = wr_bprmf_run_adam_lazy(user_tab, n_users, item_tab, n_items, D, m_u, v_u, m_i, v_i, last_u, last_i, tu, tp, tn,
                                         oc_item, oc_src, n_triplets, batch_size, first_batch + k, 1, t, lr, consts, n_consts,
                                         l2, beta1, beta2, eps, loss_out ? loss_out + k : nullptr, hot, workspace,
                                         workspace_bytes, stream)) != WR_OK) return rc;
    }
    return WR_OK;
}

// The same loop with the catch-up FOLDED into the step kernels' row loads (wr_bprmf_step_adam_folded, MODE 4 of wr_bpr.hip):
// one launch pair per step and 6 instead of 12 row transfers per touched row.  A batch with hot rows (its pieces / combine
// kernels read rows outside the two step kernels) takes the separate catch-up pass, as above.
int32_t wr_bprmf_run_adam_folded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, float *m_u,
                                 float *v_u, float *m_i, float *v_i, int32_t *last_u, int32_t *last_i, const int32_t *tu,
                                 const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                                 int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches,
                                 int64_t adam_step0, float lr, const float *consts, int64_t n_consts, float l2, float beta1,
                                 float beta2, float eps, float *loss_out, const wr_hot_runs *hot, void *workspace,
                                 int64_t workspace_bytes, void *stream) {
    WR_REQUIRE(n_triplets > 0 && batch_size > 0 && first_batch >= 0 && n_batches >= 0, WR_E_SHAPE, "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    WR_REQUIRE(adam_step0 >= 1 && adam_step0 + n_batches <= n_consts, WR_E_RANGE,
               "adam steps [%lld,%lld) outside the consts table (%lld entries)", (long long)adam_step0,
               (long long)(adam_step0 + n_batches), (long long)n_consts);
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k, off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        const int64_t t = adam_step0 + k;
        int32_t rc;
        const bool any_hot = hot != nullptr && hot->counts_host != nullptr &&
                             (hot->counts_host[4 * b] | hot->counts_host[4 * b + 1] | hot->counts_host[4 * b + 2] |
                              hot->counts_host[4 * b + 3]) != 0;
        if (!any_hot) {
            if ((rc = wr_bprmf_step_adam_folded(user_tab, n_users, item_tab, n_items, D, m_u, v_u, m_i, v_i, last_u, last_i,
                                                tu + off, tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk, t, lr,
                                                consts, n_consts, l2, beta1, beta2, eps, loss_out ? loss_out + k : nullptr,
                                                workspace, workspace_bytes, stream)) != WR_OK) return rc;
            continue;
        }
        wr_hot_runs hb = hot_at_batch(hot, b);
        if ((rc = wr_adam_rows_lazy(user_tab, m_u, v_u, last_u, n_users, D, tu + off, Bk, nullptr, t, consts, n_consts, l2, beta1,
                                    beta2, eps, stream)) != WR_OK) return rc;
        if ((rc = wr_adam_rows_lazy(item_tab, m_i, v_i, last_i, n_items, D, oc_item + 2 * off, 2 * Bk, nullptr, t, consts,
                                    n_consts, l2, beta1, beta2, eps, stream)) != WR_OK) return rc;
        if ((rc = wr_bprmf_step_adam(user_tab, n_users, item_tab, n_items, D, m_u, v_u, m_i, v_i, last_u, last_i, tu + off,
                                     tp + off, tn + off, oc_item + 2 * off, oc_src + 2 * off, Bk, t, lr, l2, beta1, beta2, eps,
                                     loss_out ? loss_out + k : nullptr, &hb, workspace, workspace_bytes, stream)) != WR_OK)
            return rc;
    }
    return WR_OK;
}

int32_t wr_bprmf_run_sgd_lazy(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D, int32_t *last_u,
                              int32_t *last_i, int32_t *stamp_u, int32_t *stamp_i, int32_t step_id0, const int32_t *tu,
                              const int32_t *tp, const int32_t *tn, const int32_t *oc_item, const int32_t *oc_src,
                              int64_t n_triplets, int64_t batch_size, int64_t first_batch, int64_t n_batches, int64_t step0,
                              float lr, float l2, float *loss_out, const wr_hot_runs *hot, void *workspace,
                              int64_t workspace_bytes, void *stream) {
    WR_REQUIRE(n_triplets > 0 && batch_size > 0 && first_batch >= 0 && n_batches >= 0 && step0 >= 1, WR_E_SHAPE,
               "bad batch range");
    const int64_t total_batches = (n_triplets + batch_size - 1) / batch_size;
    WR_REQUIRE(first_batch + n_batches <= total_batches, WR_E_SHAPE, "batches [%lld,%lld) exceed the plan's %lld",
               (long long)first_batch, (long long)(first_batch + n_batches), (long long)total_batches);
    for (int64_t k = 0; k < n_batches; ++k) {
        const int64_t b = first_batch + k, off = b * batch_size;
        const int64_t Bk = (off + batch_size <= n_triplets) ? batch_size : (n_triplets - off);
        int32_t rc;
        wr_hot_runs hb;
        if (hot != nullptr) hb = hot_at_batch(hot, b);
        if ((rc = wr_sgd_rows_lazy(user_tab, last_u, n_users, D, tu + off, Bk, step0 + k, lr, l2, stream)) != WR_OK) return rc;
        if ((rc = wr_sgd_rows_lazy(item_tab, last_i, n_items, D, oc_item + 2 * off, 2 * Bk, step0 + k, lr, l2, stream)) != WR_OK)
            return rc;
        if ((rc = wr_bprmf_step_sgd(user_tab, n_users, item_tab, n_items, D, tu + off, tp + off, tn + off, oc_item + 2 * off,
                                    oc_src + 2 * off, Bk, lr, l2, stamp_u, stamp_i, step_id0 + (int32_t)k,
                                    loss_out ? loss_out + k : nullptr, hot ? &hb : nullptr, workspace, workspace_bytes,
                                    stream)) != WR_OK) return rc;
    }
    return WR_OK;
}

// wr_bprmf_run_sgd_lazy with a bounded lag (see wr_bprmf_run_adam_lazy_bounded): before every step a rotating window of
// ceil(rows / max_lag) consecutive rows of each table takes the weight-decay steps it missed (wr_sgd_catchup_all on the
// sub-range).  sweep_pos (host, in/out): next user row, next item row.
int32_t wr_bprmf_run_sgd_lazy_bounded(float *user_tab, int64_t n_users, float *item_tab, int64_t n_items, int32_t D,
                                      int32_t *last_u, int32_t *last_i, int32_t *stamp_u, int32_t *stamp_i, int32_t step_id0,
                                      const int32_t *tu, const int32_t *tp, const int32_t *tn, const int32_t *oc_item,
                                      const int32_t *oc_src, int64_t n_triplets, int64_t batch_size, int64_t first_batch,
                                      int64_t n_batches, int64_t step0, float lr, float l2, float *loss_out,
                                      const wr_hot_runs *hot, int64_t max_lag, int64_t *sweep_pos, void *workspace,
                                      int64_t workspace_bytes, void *stream) {
    WR_REQUIRE(max_lag >= 1 && sweep_pos != nullptr && sweep_pos[0] >= 0 && sweep_pos[1] >= 0, WR_E_RANGE,
               "max_lag must be >= 1 and sweep_pos given");
    WR_REQUIRE(n_users > 0 && n_items > 0 && n_batches >= 0 && step0 >= 1, WR_E_SHAPE, "bad sizes");
    const int64_t rows[2] = {(n_users + max_lag - 1) / max_lag, (n_items + max_lag - 1) / max_lag};
    float *tabs[2] = {user_tab, item_tab};
    int32_t *lasts[2] = {last_u, last_i};
    const int64_t n_rows[2] = {n_users, n_items};
    for (int64_t k = 0; k < n_batches; ++k) {
        int32_t rc;
        const int64_t t = step0 + k;
        for (int side = 0; side < 2 && t > 1; ++side) {
            int64_t lo = sweep_pos[side] % n_rows[side], left = rows[side] < n_rows[side] ? rows[side] : n_rows[side];
            while (left > 0) {
                const int64_t c = left < n_rows[side] - lo ? left : n_rows[side] - lo;
                if ((rc = wr_sgd_catchup_all(tabs[side] + lo * (int64_t)D, lasts[side] + lo, c, D, t - 1, lr, l2, stream)) != WR_OK)
                    return rc;
                lo = (lo + c) % n_rows[side];
                left -= c;
            }
            sweep_pos[side] = lo;
        }
        if ((rc = wr_bprmf_run_sgd_lazy(user_tab, n_users, item_tab, n_items, D, last_u, last_i, stamp_u, stamp_i,
                                        step_id0 + (int32_t)k, tu, tp, tn, oc_item, oc_src, n_triplets, batch_size,
                                        first_batch + k, 1, t, lr, l2, loss_out ? loss_out + k : nullptr, hot, workspace,
                                        workspace_bytes, stream)) != WR_OK) return rc;
    }
    return WR_OK;
}

int32_t wr_adadelta_decay_all(float *square_avg, float *acc_delta, int32_t *last_step, int64_t n_rows, int32_t D, int64_t step,
                              float rho, void *stream_) {
    int32_t rc;
    if ((rc = check_table(square_avg, n_rows, D, "square_avg")) != WR_OK) return rc;
    if ((rc = check_table(acc_delta, n_rows, D, "acc_delta")) != WR_OK) return rc;
    WR_REQUIRE(last_step != nullptr, WR_E_NULL, "last_step is NULL");
    WR_REQUIRE(step >= 0 && step < INT32_MAX, WR_E_RANGE, "step must be >= 0");
    const int64_t g = (n_rows + kBlock / kWave - 1) / (kBlock / kWave), cap = 256 * 64;
    hipLaunchKernelGGL(adadelta_decay_all_kernel, dim3((unsigned)(g < 1 ? 1 : (g > cap ? cap : g))), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream_), square_avg, acc_delta, last_step, n_rows, D, (int)step, rho);
    WR_LAUNCH_CHECK("adadelta_decay_all_kernel");
    return WR_OK;
}

}  // extern "C"
