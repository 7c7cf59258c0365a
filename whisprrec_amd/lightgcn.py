"""LightGCN on MI355X behind the reference's model contract (reference src/models/general/LightGCN.py).

Same surface as the reference class: flags ``--embedding_size --gcn_layers --reg_weight`` (LightGCN.py:27-32),
``state_dict`` keys ``user_embedding.weight`` / ``item_embedding.weight``, xavier-uniform init (:52),
``predict(batch) -> loss`` of shape (1,) (:175), ``full_predict`` (:177-187).

What changes underneath:
  * the normalised adjacency D^-1/2 A D^-1/2 (:54-121) is kept in CSR on the device (the reference multiplies its
    DENSE N x N form at run time, :120,139) and built with vectorised NumPy instead of DOK/LIL loops;
  * propagation E_{l+1} = A E_l and the layer mean (:138-143) are ``wr_spmm_csr`` calls with the running sum fused;
  * BPR on the propagated rows uses the same forward / gradient kernels as BPRMF; the backward of the propagation is the
    same SpMM again (A is symmetric);
  * EmbLoss (src/utils/loss.py:94-98, un-squared Frobenius norms) = ``wr_embloss_sumsq`` + row gather / sorted scatter.
Gradients come out dense (like autograd's), so any ``torch.optim`` works; SGD and Adam (the default) run in the HIP
dense optimizers.  No CPU path.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip_ops, host


def build_norm_adj_csr(n_users, n_items, train_clicked_set):
    """CSR of the symmetric normalised bipartite adjacency over N = n_users + n_items nodes.

    Restates build_adjmat + csr2tensor (LightGCN.py:54-76, 79-121): A[u, n_users+i] = A[n_users+i, u] = 1 for every
    train pair, d = rowsum + 1e-10 (:89), value = d_r^-1/2 * 1 * d_c^-1/2 in float64, rounded to fp32 (:109).
    Columns ascend inside each row.  Returns (row_ptr int64 [N+1], col int32 [nnz], val float32 [nnz])."""
    N = n_users + n_items
    us, its = [], []
    for uu, items in train_clicked_set.items():
        if len(items):
            arr = np.fromiter(items, dtype=np.int64, count=len(items))
            us.append(np.full(arr.shape, uu, dtype=np.int64))
            its.append(arr)
    if us:
        uu = np.concatenate(us)
        ii = np.concatenate(its) + n_users
    else:
        uu = ii = np.zeros(0, np.int64)
    rows = np.concatenate([uu, ii])
    cols = np.concatenate([ii, uu])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    deg = np.bincount(rows, minlength=N).astype(np.float64)
    dis = np.power(deg + 1e-10, -0.5)
    val = (dis[rows] * 1.0 * dis[cols]).astype(np.float32)
    row_ptr = np.zeros(N + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=N), out=row_ptr[1:])
    return row_ptr, cols.astype(np.int32), val


SPMM_MFMA_DEFAULT = 0     # decided by measurement: scripts/ab_spmm.py, DESIGN.md section 5


class DenseHipOptimizer:
    """zero_grad/step over dense ``.grad`` tensors with the HIP dense optimizers (torch.optim.SGD / Adam semantics,
    reference BaseRunner.py:120-124).  Adam keeps its step number on the DEVICE (a counter bumped by a one-thread kernel,
    bias-correction constants in a table): no launch argument changes from step to step, so a hipGraph that captured a
    whole training step (HipRunner) replays correctly."""

    MAX_STEPS = 1 << 20

    def __init__(self, params, name, lr, l2):
        if name not in ("SGD", "Adam"):
            raise ValueError(name)
        self.params, self.name, self.lr, self.l2 = list(params), name, float(lr), float(l2)
        self.t = 0
        self.state = {}
        self.step_dev = self.consts = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def prepare(self):
        """allocate everything step() needs (moments, step counter, constants): nothing is allocated inside a graph capture"""
        if self.name == "Adam":
            dev = self.params[0].device
            if self.step_dev is None or self.step_dev.device != dev:
                self.step_dev = torch.full((1,), self.t, dtype=torch.int32, device=dev)
                self.consts = hip_ops.adam_consts(self.MAX_STEPS, self.lr, device=dev)
            for p in self.params:
                if p not in self.state:
                    self.state[p] = (torch.zeros_like(p.data), torch.zeros_like(p.data))

    def sync_step_count(self):
        """after graph replays (which bump the device counter only) bring the host mirror up to date"""
        if self.step_dev is not None:
            self.t = int(self.step_dev.item())

    @torch.no_grad()
    def step(self):
        if self.name == "Adam":
            self.prepare()                     # the device counter starts at the number of steps taken so far
        self.t += 1
        if self.name == "Adam":
            if self.t >= self.MAX_STEPS:
                raise RuntimeError("more than %d Adam steps" % self.MAX_STEPS)
            hip_ops.counter_add(self.step_dev, 1)
        live = [p for p in self.params if p.grad is not None]
        if self.name == "Adam" and len(live) == 2 and live[0].dim() == 2 and live[1].dim() == 2 and \
                live[0].shape[1] == live[1].shape[1]:
            # the two embedding tables of a model: one launch
            a, b = ((p.data, *self.state[p], p.grad.contiguous()) for p in live)
            hip_ops.adam_dense_dev_pair(a, b, self.consts, self.step_dev, self.l2)
            return
        for p in live:
            g = p.grad.contiguous()
            if self.name == "SGD":
                hip_ops.sgd_dense(p.data, g, self.lr, self.l2)
            else:
                m, v = self.state[p]
                hip_ops.adam_dense_dev(p.data, m, v, g, self.consts, self.step_dev, self.l2)


class _LightGcnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user_w, item_w, model, u, p, n):
        ctx.model, ctx.idx = model, (u, p, n)
        ctx.grad = None
        if model._native_step_ok(u):
            # predict and its backward in ONE native call (wr_lightgcn_step): the gradient is ready when backward() asks
            loss, ctx.grad = model._native_step(u, p, n)
            return loss
        loss, ctx.saved = model._forward_loss(u, p, n)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        gE = ctx.grad if ctx.grad is not None else ctx.model._backward(ctx.idx, ctx.saved)
        nU = ctx.model.n_users
        if not getattr(ctx.model, "_unit_root", False):     # HipRunner's captured step calls backward() with a root of ones
            gE.mul_(grad_out.reshape(-1)[0])                 # gE is this call's own buffer: one pass for both tables
        return gE[:nU], gE[nU:], None, None, None, None


def make_lightgcn(general_model_cls):
    class LightGCN(general_model_cls):
        reader = "BaseReader"
        runner = "BaseRunner"
        extra_log_args = ["embedding_size", "gcn_layers", "reg_weight"]
        graph_capturable = True     # with range-checked indices the step never touches the host: HipRunner may capture it

        @staticmethod
        def parse_model_args(parser):
            parser.add_argument("--embedding_size", type=int, default=64, help="Size of embedding vectors.")
            parser.add_argument("--gcn_layers", type=int, default=2, help="Number of LightGCN layers.")
            parser.add_argument("--reg_weight", type=float, default=1e-05, help="The L2 regularization weight.")
            parser.add_argument("--spmm_mfma", type=int, default=SPMM_MFMA_DEFAULT,
                                help="1: the dense head of the adjacency (popular items) runs on the matrix cores "
                                     "(hip_ops.HybridSpmm), the rest on the CSR kernels; 0: CSR kernels only.")
            return general_model_cls.parse_model_args(parser)

        def __init__(self, args, corpus):
            super().__init__(args, corpus)
            self.emb_size = args.embedding_size
            if self.emb_size % 4 != 0:
                raise ValueError("embedding_size must be a multiple of 4 for the HIP kernels (got %d)" % self.emb_size)
            self.gcn_layers = args.gcn_layers
            self.n_users, self.n_items = int(corpus.n_users), int(corpus.n_items)
            self.reg_weight = float(args.reg_weight)
            self.user_embedding = nn.Embedding(self.n_users, self.emb_size)
            self.item_embedding = nn.Embedding(self.n_items, self.emb_size)
            rp, col, val = build_norm_adj_csr(self.n_users, self.n_items, corpus.train_clicked_set)
            # plain attributes like the reference's norm_adj (not buffers: absent from state_dict, LightGCN.py:49-51)
            cptr, crow = hip_ops.spmm_chunks(rp)   # rows cut into chunks of <= 96 non-zeros: hubs do not serialise on one team
            self._csr_host = (cptr, crow, torch.from_numpy(col), torch.from_numpy(val))
            self._csr_full = (rp, col, val)
            self._use_mfma = bool(getattr(args, "spmm_mfma", SPMM_MFMA_DEFAULT)) and self.emb_size in (32, 64, 96, 128)
            self._hybrid = None
            self._csr_dev = None
            self._partials = None
            nn.init.xavier_uniform_(self.user_embedding.weight.data)   # LightGCN.py:52, init.py:32-48
            nn.init.xavier_uniform_(self.item_embedding.weight.data)
            name = getattr(args, "optimizer", None)
            if name in ("SGD", "Adam") and hasattr(args, "lr"):
                self.optimizer = DenseHipOptimizer([self.user_embedding.weight, self.item_embedding.weight], name, args.lr,
                                                   getattr(args, "l2", 0.0))

        # ------------------------------------------------------------------ device state
        def _csr(self):
            dev = self.user_embedding.weight.device
            if self._csr_dev is None or self._csr_dev[0].device != dev:
                self._csr_dev = tuple(t.to(dev) for t in self._csr_host)
                self._csr_dev[1]._wr_levels = self._csr_host[1]._wr_levels      # combine levels of the product (spmm_chunks)
            return self._csr_dev

        def _propagate(self, E0):
            """mean over layers of A^l E0, l = 0..L (LightGCN.py:134-143)."""
            cptr, crow, col, val = self._csr()
            cur = E0
            if self._use_mfma and E0.is_cuda:
                acc = E0.clone()
                if self._hybrid is None or self._hybrid.device != E0.device:
                    self._hybrid = hip_ops.HybridSpmm(*self._csr_full, self.n_users, self.n_items, E0.device)
                if self._hybrid.enabled:
                    for _ in range(self.gcn_layers):
                        cur = self._hybrid.apply(cur, acc=acc)
                    out = torch.empty_like(acc)
                    hip_ops.axpy(out, acc, 1.0 / (self.gcn_layers + 1), overwrite=True)
                    return out
            if self._partials is None or self._partials.device != E0.device:
                self._partials = torch.empty((crow.numel(), E0.shape[1]), dtype=torch.float32, device=E0.device)
            if self.gcn_layers == 0:
                return E0.clone()
            # the layer sum lives in the products themselves: the first one starts it from its input (no copy of E0), the
            # last one scales it to the mean (no scaling pass) — two graph nodes fewer per propagation
            acc = torch.empty_like(E0)
            for l in range(self.gcn_layers):
                cur = hip_ops.spmm_csr_chunked(cptr, crow, col, val, cur, acc=acc, partials=self._partials, acc_from_x=l == 0,
                                               acc_scale=1.0 / (self.gcn_layers + 1) if l == self.gcn_layers - 1 else 1.0)
            return acc

        def forward(self):
            E0 = torch.cat([self.user_embedding.weight.data, self.item_embedding.weight.data], dim=0)
            allE = self._propagate(E0)
            return allE[:self.n_users], allE[self.n_users:]

        # ------------------------------------------------------------------ loss / gradients
        def _forward_loss(self, u, p, n):
            U0, I0 = self.user_embedding.weight.data, self.item_embedding.weight.data
            E0 = torch.cat([U0, I0], dim=0)
            allE = self._propagate(E0)
            Ua, Ia = allE[:self.n_users], allE[self.n_users:]
            # BPR on the propagated rows + EmbLoss on the ego rows, folded to the (1,)-shaped loss of the reference
            # (loss.py:94) on the device: two launches
            loss, sq = hip_ops.lightgcn_loss(Ua, Ia, U0, I0, u, p, n, self.reg_weight)
            return loss, (allE, sq)

        def _backward(self, idx, saved):
            u, p, n = idx
            allE, sq = saved
            nU, D, B = self.n_users, self.emb_size, u.numel()
            Ua, Ia = allE[:nU].contiguous(), allE[nU:].contiguous()
            # gradient w.r.t. the propagated tables: BPRMF gradient kernels on (Ua, Ia).  One small batch: the radix-sort
            # builder and the sequential run path need no host round trip besides the index check.
            tabs = hip_ops.BprmfTables(Ua, Ia)
            # indices already range-checked for the whole epoch (HipRunner): no per-batch read-back, the step never waits on the host
            plan = hip_ops.BatchPlan(u, p, n, B, nU, self.n_items, builder="small" if B <= 4096 else "generic", hot=False,
                                     validate=not getattr(self, "_trusted_indices", False))
            gOut = torch.zeros(nU + self.n_items, D, device=allE.device)
            tabs.grads(plan, 0, gOut[:nU], gOut[nU:], stamps=False)     # gOut is zero-filled and read densely
            # back through the propagation: A is symmetric, d(mean_l A^l E0) = mean_l A^l gOut
            gE = self._propagate(gOut)
            # EmbLoss: d/dx ||X||_F = x/||X||_F per gathered row, duplicates add up (loss.py:94-98); the plan's runs give
            # the multiplicities, the norms stay on the device
            hip_ops.embloss_grad(self.user_embedding.weight.data, self.item_embedding.weight.data, plan, 0, sq,
                                 self.reg_weight, gE[:nU], gE[nU:])
            return gE

        NATIVE_STEP = True      # predict + backward as one native call where it applies (CSR kernels, B <= 4,096)

        def _native_step_ok(self, u):
            return self.NATIVE_STEP and u.is_cuda and not self._use_mfma and self.gcn_layers >= 1 and \
                u.numel() <= int(hip_ops.abi.lib().wr_bprmf_plan_small_max_batch())

        def _native_step(self, u, p, n):
            trusted = bool(getattr(self, "_trusted_indices", False))
            loss, grad, err = hip_ops.lightgcn_step(self.user_embedding.weight.data, self.item_embedding.weight.data, self._csr(),
                                                    self.gcn_layers, u, p, n, self.reg_weight, trusted=trusted)
            if err is not None and int(err.item()) != 0:      # nn.Embedding would have raised (one read-back per untrusted batch)
                raise IndexError("index out of range in batch (user_id >= n_users or item id >= n_items)")
            return loss, grad

        def _batch(self, feed_dict):
            dev = self.user_embedding.weight.device
            return tuple(feed_dict[k].to(dev).reshape(-1) for k in ("user_id", "pos_item", "neg_items"))

        def predict(self, feed_dict):
            u, p, n = self._batch(feed_dict)
            return _LightGcnLoss.apply(self.user_embedding.weight, self.item_embedding.weight, self, u, p, n)

        def eval_factors(self):
            """propagated (user, item) matrices: their inner products are the ranking scores (LightGCN.py:177-187)"""
            Ua, Ia = self.forward()
            return Ua.contiguous(), Ia.contiguous()

        def full_predict(self, feed_dict):
            dev = self.user_embedding.weight.device
            Ua, Ia = self.forward()
            user_e = hip_ops.gather_rows(Ua.contiguous(), feed_dict["user_id"].to(dev))
            return torch.matmul(user_e, Ia.t())

    LightGCN.__qualname__ = "LightGCN"
    return LightGCN


LightGCN = make_lightgcn(host.GeneralModel)


def bind(reference_general_model_cls):
    return make_lightgcn(reference_general_model_cls)
