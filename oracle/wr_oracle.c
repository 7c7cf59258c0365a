/*
 * wr_oracle.c — CPU restatement of the WhisprRec embedding-CF training hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under whisprrec_amd/ may import, link or call this file.
 * Allowed callers: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and there only
 * as the checker / the timed CPU baseline — never as the product path.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against vectors
 * produced by running the reference itself in the build container (tests/golden/make_golden.py):
 * g1_bprmf_step, g2_ml100k_curve, g4_lightgcn, g5_sasrec_emb.
 *
 * Each function cites the reference lines (relative to /root/reference) it restates.  The arithmetic
 * is the reference's fp32 arithmetic; reductions (row dots, batch means, gradient sums) are carried in
 * double and rounded once, so that this file sits between any two fp32 summation orders.
 *
 * Build: see oracle/Makefile (gcc -O2 -shared -fPIC, optional -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WR_GAMMA 1e-10f /* src/utils/loss.py:33  BPRLoss(gamma=1e-10) */

/* ------------------------------------------------------------------------------------------------
 * BPRMF.predict forward — src/models/general/BPRMF.py:69-80 and src/utils/loss.py:37-39
 *   user_e = U[u]; pos_e = I[p]; neg_e = I[n]                               (BPRMF.py:74-75)
 *   pos = (user_e*pos_e).sum(1); neg = (user_e*neg_e).sum(1)                (BPRMF.py:76-77)
 *   loss = -log(gamma + sigmoid(pos-neg)).mean()                            (loss.py:38)
 * coef[b] = dL/d(pos_b) = -(1/B) * s(1-s)/(gamma+s): the autograd of loss.py:38 in closed form
 * (what loss.backward() at src/helpers/BaseRunner.py:198 propagates to the three gathers).
 * ------------------------------------------------------------------------------------------------ */
void orc_bpr_fwd(const float *U, const float *I, int64_t D, const int64_t *u, const int64_t *p, const int64_t *n,
                 int64_t B, float *pos_score, float *neg_score, float *coef, float *loss_out) {
    double lsum = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        const float *ue = U + u[b] * D, *pe = I + p[b] * D, *ne = I + n[b] * D;
        double sp = 0.0, sn = 0.0;
        for (int64_t d = 0; d < D; ++d) {
            sp += (double)(ue[d] * pe[d]);
            sn += (double)(ue[d] * ne[d]);
        }
        float fp = (float)sp, fn = (float)sn;
        float x = fp - fn;
        float s = 1.0f / (1.0f + expf(-x));
        lsum += (double)(-logf(WR_GAMMA + s));
        if (pos_score) pos_score[b] = fp;
        if (neg_score) neg_score[b] = fn;
        if (coef) coef[b] = -(s * (1.0f - s) / (WR_GAMMA + s)) / (float)B;
    }
    if (loss_out) *loss_out = (float)(lsum / (double)B);
}

/* Dense gradients of the BPRMF loss w.r.t. both tables — what embedding_dense_backward leaves in
 * user_embeddings.weight.grad / item_embeddings.weight.grad after BaseRunner.py:198.
 *   gU[u_b] += c_b (I[p_b] - I[n_b]);  gI[p_b] += c_b U[u_b];  gI[n_b] -= c_b U[u_b]
 * All gradients are computed from the PRE-step tables (batch-synchronous). */
void orc_bpr_dense_grads(const float *U, int64_t nU, const float *I, int64_t nI, int64_t D, const int64_t *u,
                         const int64_t *p, const int64_t *n, int64_t B, float *gU, float *gI, float *loss_out) {
    double *aU = (double *)calloc((size_t)(nU * D), sizeof(double));
    double *aI = (double *)calloc((size_t)(nI * D), sizeof(double));
    float *coef = (float *)malloc((size_t)B * sizeof(float));
    orc_bpr_fwd(U, I, D, u, p, n, B, NULL, NULL, coef, loss_out);
    for (int64_t b = 0; b < B; ++b) {
        const float *ue = U + u[b] * D, *pe = I + p[b] * D, *ne = I + n[b] * D;
        double *au = aU + u[b] * D, *ap = aI + p[b] * D, *an = aI + n[b] * D;
        float c = coef[b];
        for (int64_t d = 0; d < D; ++d) {
            au[d] += (double)(c * pe[d]) - (double)(c * ne[d]);
            ap[d] += (double)(c * ue[d]);
            an[d] -= (double)(c * ue[d]);
        }
    }
    for (int64_t i = 0; i < nU * D; ++i) gU[i] = (float)aU[i];
    for (int64_t i = 0; i < nI * D; ++i) gI[i] = (float)aI[i];
    free(aU);
    free(aI);
    free(coef);
}

/* torch.optim.SGD.step (momentum=0) as built at src/helpers/BaseRunner.py:120-124:
 *   g <- g + weight_decay * w ; w <- w - lr * g      over EVERY row (dense grads). */
void orc_sgd_dense(float *W, const float *G, int64_t numel, float lr, float l2) {
    for (int64_t i = 0; i < numel; ++i) {
        float g = G[i];
        if (l2 != 0.0f) g = g + l2 * W[i];
        W[i] = W[i] - lr * g;
    }
}

/* torch.optim.Adam.step (amsgrad=False, betas=(0.9,0.999), eps=1e-8; BaseRunner.py:120-124, default
 * optimizer per BaseRunner.py:36).  Restates torch/optim/adam.py::_single_tensor_adam (torch 2.10):
 *   g += wd*w; m.lerp_(g, 1-b1); v = b2*v + (1-b2) g*g
 *   step_size = lr/(1-b1^t); denom = sqrt(v)/sqrt(1-b2^t) + eps; w -= step_size * m/denom        */
void orc_adam_dense(float *W, const float *G, float *M, float *V, int64_t numel, int64_t step, float lr, float l2,
                    float beta1, float beta2, float eps) {
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    float step_size = (float)((double)lr / bc1);
    float bc2_sqrt = (float)sqrt(bc2);
    for (int64_t i = 0; i < numel; ++i) {
        float g = G[i];
        if (l2 != 0.0f) g = g + l2 * W[i];
        float m = M[i] + (1.0f - beta1) * (g - M[i]); /* lerp_ */
        float v = beta2 * V[i] + (1.0f - beta2) * g * g;
        M[i] = m;
        V[i] = v;
        float denom = sqrtf(v) / bc2_sqrt + eps;
        W[i] = W[i] - step_size * (m / denom);
    }
}

/* One BaseRunner.fit iteration for BPRMF with SGD (BaseRunner.py:196-199), dense semantics. */
void orc_bprmf_step_sgd(float *U, int64_t nU, float *I, int64_t nI, int64_t D, const int64_t *u, const int64_t *p,
                        const int64_t *n, int64_t B, float lr, float l2, float *loss_out) {
    float *gU = (float *)malloc((size_t)(nU * D) * sizeof(float));
    float *gI = (float *)malloc((size_t)(nI * D) * sizeof(float));
    orc_bpr_dense_grads(U, nU, I, nI, D, u, p, n, B, gU, gI, loss_out);
    orc_sgd_dense(U, gU, nU * D, lr, l2);
    orc_sgd_dense(I, gI, nI * D, lr, l2);
    free(gU);
    free(gI);
}

/* The same step restated sparsely for l2 == 0 (rows not in the batch have zero gradient and are left
 * bit-identical by SGD): O(B*D) instead of O((nU+nI)*D).  This is the CPU baseline ("port") that
 * bench.py times.  Accumulates per-destination sums in a scratch of touched rows, fp32 like the
 * reference's embedding_dense_backward, then applies w -= lr*g. Scratch: gU/gI are caller-provided
 * ZEROED dense buffers that are returned zeroed. */
void orc_bprmf_step_sgd_sparse(float *U, float *I, int64_t D, const int64_t *u, const int64_t *p, const int64_t *n,
                               int64_t B, float lr, float *gU, float *gI, float *coef, float *loss_out) {
    orc_bpr_fwd(U, I, D, u, p, n, B, NULL, NULL, coef, loss_out);
    for (int64_t b = 0; b < B; ++b) {
        const float *ue = U + u[b] * D, *pe = I + p[b] * D, *ne = I + n[b] * D;
        float *au = gU + u[b] * D, *ap = gI + p[b] * D, *an = gI + n[b] * D;
        float c = coef[b];
        for (int64_t d = 0; d < D; ++d) {
            au[d] += c * (pe[d] - ne[d]);
            float z = c * ue[d];
            ap[d] += z;
            an[d] -= z;
        }
    }
    for (int64_t b = 0; b < B; ++b) { /* apply once per touched row, then clear the scratch row */
        float *w, *g;
        w = U + u[b] * D; g = gU + u[b] * D;
        for (int64_t d = 0; d < D; ++d) { w[d] -= lr * g[d]; g[d] = 0.0f; }
        w = I + p[b] * D; g = gI + p[b] * D;
        for (int64_t d = 0; d < D; ++d) { w[d] -= lr * g[d]; g[d] = 0.0f; }
        w = I + n[b] * D; g = gI + n[b] * D;
        for (int64_t d = 0; d < D; ++d) { w[d] -= lr * g[d]; g[d] = 0.0f; }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Row gather / scatter-add — nn.Embedding forward and embedding_dense_backward with padding_idx
 * (src/models/sequential/SASRec.py:60,84,105-106).  padding_idx < 0 disables masking.
 * ------------------------------------------------------------------------------------------------ */
void orc_gather_rows(const float *W, int64_t D, const int64_t *idx, int64_t n, float *out) {
    for (int64_t k = 0; k < n; ++k) memcpy(out + k * D, W + idx[k] * D, (size_t)D * sizeof(float));
}

void orc_scatter_add_rows(float *G, int64_t nrows, int64_t D, const int64_t *idx, const float *src, int64_t n,
                          int64_t padding_idx) {
    double *acc = (double *)calloc((size_t)(nrows * D), sizeof(double));
    for (int64_t k = 0; k < n; ++k) {
        if (idx[k] == padding_idx) continue;
        for (int64_t d = 0; d < D; ++d) acc[idx[k] * D + d] += (double)src[k * D + d];
    }
    for (int64_t i = 0; i < nrows * D; ++i) G[i] = (float)((double)G[i] + acc[i]);
    free(acc);
}

/* ------------------------------------------------------------------------------------------------
 * LightGCN — src/models/general/LightGCN.py
 * ------------------------------------------------------------------------------------------------ */

/* build_adjmat + csr2tensor (LightGCN.py:54-76, 79-121) in CSR form.
 * Input: user->clicked items CSR (train_clicked_set). Output: symmetric bipartite CSR over
 * N = nU+nI nodes with values d_i^-1/2 d_j^-1/2, d = rowsum + 1e-10 (LightGCN.py:89,92,97),
 * computed in float64 like scipy and rounded to fp32 (LightGCN.py:109).
 * row_ptr has N+1 entries, col/val have 2*nnz entries; columns ascending inside each row. */
void orc_lightgcn_build_adj(int64_t nU, int64_t nI, const int32_t *cptr, const int32_t *cidx, int64_t *row_ptr,
                            int32_t *col, float *val) {
    int64_t N = nU + nI, nnz = cptr[nU];
    int64_t *deg = (int64_t *)calloc((size_t)N, sizeof(int64_t));
    for (int64_t uu = 0; uu < nU; ++uu)
        for (int32_t k = cptr[uu]; k < cptr[uu + 1]; ++k) {
            deg[uu]++;
            deg[nU + cidx[k]]++;
        }
    row_ptr[0] = 0;
    for (int64_t i = 0; i < N; ++i) row_ptr[i + 1] = row_ptr[i] + deg[i];
    double *dis = (double *)malloc((size_t)N * sizeof(double));
    for (int64_t i = 0; i < N; ++i) dis[i] = pow((double)deg[i] + 1e-10, -0.5);
    int64_t *fill = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    memcpy(fill, row_ptr, (size_t)N * sizeof(int64_t));
    /* user rows: item columns ascending because cidx is sorted per user */
    for (int64_t uu = 0; uu < nU; ++uu)
        for (int32_t k = cptr[uu]; k < cptr[uu + 1]; ++k) {
            int64_t j = nU + cidx[k];
            col[fill[uu]] = (int32_t)j;
            val[fill[uu]++] = (float)(dis[uu] * 1.0 * dis[j]);
        }
    /* item rows: users visited in ascending order -> ascending columns */
    for (int64_t uu = 0; uu < nU; ++uu)
        for (int32_t k = cptr[uu]; k < cptr[uu + 1]; ++k) {
            int64_t j = nU + cidx[k];
            col[fill[j]] = (int32_t)uu;
            val[fill[j]++] = (float)(dis[j] * 1.0 * dis[uu]);
        }
    (void)nnz;
    free(deg);
    free(dis);
    free(fill);
}

/* Y = A X for CSR A (the reference multiplies the dense form of the same matrix, LightGCN.py:139) */
void orc_spmm_csr(int64_t N, const int64_t *row_ptr, const int32_t *col, const float *val, const float *X, int64_t D,
                  float *Y) {
    double *acc = (double *)malloc((size_t)D * sizeof(double));
    for (int64_t i = 0; i < N; ++i) {
        for (int64_t d = 0; d < D; ++d) acc[d] = 0.0;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            const float *x = X + (int64_t)col[k] * D;
            float a = val[k];
            for (int64_t d = 0; d < D; ++d) acc[d] += (double)(a * x[d]);
        }
        for (int64_t d = 0; d < D; ++d) Y[i * D + d] = (float)acc[d];
    }
}

/* LightGCN.forward (LightGCN.py:134-148): E_0 = cat(U,I); E_{l+1} = A E_l; out = mean_l E_l.
 * E0 is [N,D] (users then items); out is [N,D]. */
void orc_lightgcn_forward(int64_t N, const int64_t *row_ptr, const int32_t *col, const float *val, const float *E0,
                          int64_t D, int64_t L, float *out) {
    float *cur = (float *)malloc((size_t)(N * D) * sizeof(float));
    float *nxt = (float *)malloc((size_t)(N * D) * sizeof(float));
    double *sum = (double *)malloc((size_t)(N * D) * sizeof(double));
    memcpy(cur, E0, (size_t)(N * D) * sizeof(float));
    for (int64_t i = 0; i < N * D; ++i) sum[i] = (double)E0[i];
    for (int64_t l = 0; l < L; ++l) {
        orc_spmm_csr(N, row_ptr, col, val, cur, D, nxt);
        for (int64_t i = 0; i < N * D; ++i) sum[i] += (double)nxt[i];
        float *t = cur; cur = nxt; nxt = t;
    }
    for (int64_t i = 0; i < N * D; ++i) out[i] = (float)(sum[i] / (double)(L + 1));
    free(cur);
    free(nxt);
    free(sum);
}

/* LightGCN.predict + backward (LightGCN.py:150-175; EmbLoss at src/utils/loss.py:94-98).
 *   loss = BPR(propagated rows) + reg_weight * (||U[u]||_F + ||I[p]||_F + ||I[n]||_F) / B
 * Outputs: loss (the reference returns shape (1,)), dense grad gE0 [N,D] w.r.t. the ego tables.
 * Backward of the propagation uses A^T = A (symmetric): gE0 = (1/(L+1)) sum_l A^l gOut. */
void orc_lightgcn_loss_grads(int64_t nU, int64_t nI, const int64_t *row_ptr, const int32_t *col, const float *val,
                             const float *E0, int64_t D, int64_t L, float reg_weight, const int64_t *u,
                             const int64_t *p, const int64_t *n, int64_t B, float *loss_out, float *gE0) {
    int64_t N = nU + nI;
    float *all = (float *)malloc((size_t)(N * D) * sizeof(float));
    orc_lightgcn_forward(N, row_ptr, col, val, E0, D, L, all);
    const float *Ua = all, *Ia = all + nU * D;
    const float *U0 = E0, *I0 = E0 + nU * D;
    float *coef = (float *)malloc((size_t)B * sizeof(float));
    float mf;
    orc_bpr_fwd(Ua, Ia, D, u, p, n, B, NULL, NULL, coef, &mf);
    /* EmbLoss: un-squared Frobenius norms of the three gathered ego blocks (loss.py:94-97) */
    double nu = 0, np_ = 0, nn = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t d = 0; d < D; ++d) {
            double a = U0[u[b] * D + d], c = I0[p[b] * D + d], e = I0[n[b] * D + d];
            nu += a * a; np_ += c * c; nn += e * e;
        }
    float fu = (float)sqrt(nu), fp = (float)sqrt(np_), fn = (float)sqrt(nn);
    float reg = (fu + fp + fn) / (float)B;
    *loss_out = mf + reg_weight * reg;
    /* gradient w.r.t. propagated tables */
    double *gout = (double *)calloc((size_t)(N * D), sizeof(double));
    for (int64_t b = 0; b < B; ++b) {
        float c = coef[b];
        for (int64_t d = 0; d < D; ++d) {
            float ue = Ua[u[b] * D + d], pe = Ia[p[b] * D + d], ne = Ia[n[b] * D + d];
            gout[u[b] * D + d] += (double)(c * pe) - (double)(c * ne);
            gout[(nU + p[b]) * D + d] += (double)(c * ue);
            gout[(nU + n[b]) * D + d] -= (double)(c * ue);
        }
    }
    float *g = (float *)malloc((size_t)(N * D) * sizeof(float));
    float *t = (float *)malloc((size_t)(N * D) * sizeof(float));
    double *acc = (double *)malloc((size_t)(N * D) * sizeof(double));
    for (int64_t i = 0; i < N * D; ++i) { g[i] = (float)gout[i]; acc[i] = gout[i]; }
    for (int64_t l = 0; l < L; ++l) {
        orc_spmm_csr(N, row_ptr, col, val, g, D, t);
        for (int64_t i = 0; i < N * D; ++i) acc[i] += (double)t[i];
        float *s = g; g = t; t = s;
    }
    for (int64_t i = 0; i < N * D; ++i) acc[i] /= (double)(L + 1);
    /* gradient of the EmbLoss term w.r.t. ego rows: reg_weight/B * x / ||block||_F */
    for (int64_t b = 0; b < B; ++b)
        for (int64_t d = 0; d < D; ++d) {
            if (fu > 0) acc[u[b] * D + d] += (double)reg_weight / (double)B * (double)U0[u[b] * D + d] / (double)fu;
            if (fp > 0) acc[(nU + p[b]) * D + d] += (double)reg_weight / (double)B * (double)I0[p[b] * D + d] / (double)fp;
            if (fn > 0) acc[(nU + n[b]) * D + d] += (double)reg_weight / (double)B * (double)I0[n[b] * D + d] / (double)fn;
        }
    for (int64_t i = 0; i < N * D; ++i) gE0[i] = (float)acc[i];
    free(all); free(coef); free(gout); free(g); free(t); free(acc);
}
