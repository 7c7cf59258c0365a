// wr_lightgcn.hip — LightGCN.predict + backward as ONE native call (reference src/models/general/LightGCN.py:134-175 and the
// autograd of it that src/helpers/BaseRunner.py:198 runs): the same kernels hip_ops / lightgcn.py issue one by one through
// ctypes — propagation (CSR products with the layer sum folded in), BPR + EmbLoss tail, per-batch plan, gradient kernels on
// the propagated rows, propagation of the gradient (the adjacency is symmetric), EmbLoss gradient — queued from native code.
// The eager step of the reference's own loop (INTEGRATION.md route A) was bound by ~25 Python / ctypes calls (0.43 ms at the
// ml-1m shape, 0.125 ms of which are the four products); this call takes their place.  Same kernels in the same order on
// the same buffers: the same bits as the call-by-call form (tests/test_lightgcn.py).
#include "wr_common.h"

using namespace wr;

extern "C" {

static inline int64_t lg_align(int64_t x) { return align_up(x, 256); }

// workspace layout (bytes): E0 | A | B | allE | gOut | gE | partials | plan (7 B int32 + err) | loss tail ws | step ws
int64_t wr_lightgcn_step_workspace_bytes(int64_t n_users, int64_t n_items, int32_t D, int64_t n_chunks, int64_t B) {
    if (n_users <= 0 || n_items <= 0 || D <= 0 || n_chunks < 0 || B <= 0) return WR_E_SHAPE;
    const int64_t nd = lg_align((n_users + n_items) * (int64_t)D * 4);
    return 5 * nd + lg_align(n_chunks * (int64_t)D * 4) + lg_align((7 * B + 4) * 4) + lg_align(wr_lightgcn_loss_workspace_bytes(B)) +
           lg_align(wr_bprmf_step_workspace_bytes(B, D)) + 256;
}

int32_t wr_lightgcn_step(const float *user_tab, const float *item_tab, int64_t n_users, int64_t n_items, int32_t D,
                         int64_t n_chunks, const int64_t *chunk_ptr, const int32_t *chunk_row, const int32_t *col,
                         const float *val, int32_t levels, int32_t n_layers, const int64_t *u, const int64_t *p, const int64_t *n,
                         int64_t B, float reg_weight, int32_t trusted_indices, float *loss, float *grad /* [n_users + n_items, D] */,
                         int32_t *err_flag, void *workspace, int64_t workspace_bytes, void *stream_) {
    int32_t rc;
    if ((rc = check_table(user_tab, n_users, D, "user_tab")) != WR_OK) return rc;
    if ((rc = check_table(item_tab, n_items, D, "item_tab")) != WR_OK) return rc;
    WR_REQUIRE(chunk_ptr && chunk_row && col && val && u && p && n && loss && grad, WR_E_NULL, "wr_lightgcn_step: NULL argument");
    WR_REQUIRE(n_layers >= 1 && (levels == 1 || levels == 2) && B > 0 && B <= wr_bprmf_plan_small_max_batch(), WR_E_RANGE,
               "wr_lightgcn_step: layers %d, levels %d, batch %lld (at most %lld)", (int)n_layers, (int)levels, (long long)B,
               (long long)wr_bprmf_plan_small_max_batch());
    WR_REQUIRE(trusted_indices || err_flag, WR_E_NULL, "err_flag is NULL");
    const int64_t need = wr_lightgcn_step_workspace_bytes(n_users, n_items, D, n_chunks, B);
    WR_REQUIRE(workspace && aligned16(workspace) && aligned16(grad) && workspace_bytes >= need, WR_E_WORKSPACE,
               "wr_lightgcn_step: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int64_t N = n_users + n_items, nd = lg_align(N * (int64_t)D * 4);
    char *w = reinterpret_cast<char *>(workspace);
    float *E0 = reinterpret_cast<float *>(w), *bufA = reinterpret_cast<float *>(w + nd), *bufB = reinterpret_cast<float *>(w + 2 * nd);
    float *allE = reinterpret_cast<float *>(w + 3 * nd), *gOut = reinterpret_cast<float *>(w + 4 * nd);
    char *at = w + 5 * nd;
    float *partials = reinterpret_cast<float *>(at);
    at += lg_align(n_chunks * (int64_t)D * 4);
    int32_t *plan = reinterpret_cast<int32_t *>(at);
    at += lg_align((7 * B + 4) * 4);
    void *loss_ws = at;
    const int64_t loss_ws_bytes = lg_align(wr_lightgcn_loss_workspace_bytes(B));
    at += loss_ws_bytes;
    void *step_ws = at;
    const int64_t step_ws_bytes_ = lg_align(wr_bprmf_step_workspace_bytes(B, D));
    float *sq3 = reinterpret_cast<float *>(at + step_ws_bytes_);
    int32_t *tu = plan, *tp = plan + B, *tn = plan + 2 * B, *oc_item = plan + 3 * B, *oc_src = plan + 5 * B;

    // E0 = cat(U0, I0) (LightGCN.py:131,135)
    WR_HIP(hipMemcpyAsync(E0, user_tab, (size_t)n_users * D * 4, hipMemcpyDeviceToDevice, stream));
    WR_HIP(hipMemcpyAsync(E0 + n_users * (int64_t)D, item_tab, (size_t)n_items * D * 4, hipMemcpyDeviceToDevice, stream));
    // mean over layers of A^l X (:138-143): the layer sum lives in the products (first: starts from its input; last: scales)
    auto propagate = [&](const float *X, float *out) -> int32_t {
        const float *cur = X;
        for (int l = 0; l < n_layers; ++l) {
            float *Y = (l & 1) ? bufB : bufA;
            const int32_t r = wr_spmm_csr_chunked_levels(N, n_chunks, chunk_ptr, chunk_row, col, val, cur, D, Y, out, partials, nullptr,
                                                         levels, l == 0 ? 1 : 0, l == n_layers - 1 ? 1.0f / (float)(n_layers + 1) : 1.0f,
                                                         stream_);
            if (r != WR_OK) return r;
            cur = Y;
        }
        return WR_OK;
    };
    if ((rc = propagate(E0, allE)) != WR_OK) return rc;
    // BPR on the propagated rows + reg_weight * EmbLoss on the ego rows (:150-175, loss.py:94-98)
    if ((rc = wr_lightgcn_loss(allE, allE + n_users * (int64_t)D, user_tab, item_tab, n_users, n_items, D, u, p, n, B, reg_weight, loss,
                               sq3, loss_ws, loss_ws_bytes, stream_)) != WR_OK) return rc;
    // backward: gradient w.r.t. the propagated tables (the BPRMF gradient kernels on one small batch) ...
    if ((rc = wr_bprmf_plan_build_small_i64(u, p, n, B, B, n_users, n_items, tu, tp, tn, nullptr, oc_item, oc_src,
                                            trusted_indices ? nullptr : err_flag, stream_)) != WR_OK) return rc;
    WR_HIP(hipMemsetAsync(gOut, 0, (size_t)N * D * 4, stream));
    if ((rc = wr_bprmf_grads(allE, n_users, allE + n_users * (int64_t)D, n_items, D, tu, tp, tn, oc_item, oc_src, B, gOut,
                             gOut + n_users * (int64_t)D, nullptr, nullptr, 0, nullptr, nullptr, step_ws, step_ws_bytes_, stream_)) !=
        WR_OK) return rc;
    // ... back through the propagation (A symmetric: d(mean_l A^l E0) = mean_l A^l gOut) ...
    if ((rc = propagate(gOut, grad)) != WR_OK) return rc;
    // ... plus the EmbLoss gradient on the ego rows
    return wr_embloss_grad(user_tab, item_tab, D, tu, oc_item, oc_src, B, sq3, reg_weight, grad, grad + n_users * (int64_t)D, stream_);
}

}  // extern "C"
