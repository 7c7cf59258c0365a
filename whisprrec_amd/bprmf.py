"""BPRMF on MI355X behind the reference's model contract (reference src/models/general/BPRMF.py).

Same surface as the reference class — ``reader``/``runner``/``extra_log_args``, ``parse_model_args`` (flag
``--embedding_size``, BPRMF.py:24), ``__init__(args, corpus)``, ``predict(batch) -> loss``, ``full_predict(batch) ->
[B, n_items]``, ``state_dict`` keys ``user_embeddings.weight`` / ``item_embeddings.weight`` — but the arithmetic of
``predict`` + ``loss.backward()`` + ``optimizer.step()`` (reference src/helpers/BaseRunner.py:196-199) runs in the
hand-written HIP kernels of libwhisprrec_hip.so.  There is no CPU path: using the model on a CPU device raises.

Two ways to drive it:
  * unchanged ``BaseRunner.fit``: the model pre-installs ``self.optimizer`` (the runner only builds one while it is
    ``None``, BaseRunner.py:182-183).  ``predict`` records the batch, ``loss.backward()`` is a marker, and
    ``optimizer.step()`` launches the fused step, which also writes the loss value into the tensor ``predict`` returned —
    it is read by the runner only after ``step()`` (``loss.detach().cpu()``, BaseRunner.py:200).  ``--fused 0`` computes
    the loss eagerly in ``predict`` instead (one extra gather pass).
  * ``HipRunner`` (runner.py): plans a whole epoch on the device and runs the steps from native code.
Other ``--optimizer`` names (Adagrad, Adadelta, ...) work through ``torch.optim`` on the dense gradients the backward
kernels emit.
"""
import torch
import torch.nn as nn

from . import hip_ops, host


class FusedOptimizer:
    """Duck-types what BaseRunner.fit needs from ``model.optimizer`` (``zero_grad``/``step``) and applies
    torch.optim.SGD / torch.optim.Adam semantics (reference BaseRunner.py:120-124) inside the HIP step.

    Both optimizers are DENSE in the reference whenever they carry state or weight decay: Adam moves every row at every
    step, SGD with ``--l2`` shrinks every row.  ``lazy`` selects how that is computed: 0 = table passes every step
    (wr_adam_dense / wr_sgd_decay_untouched), 1 = exact lazy rows (hip_ops.LazyOptimizerState: a row is replayed when a
    batch needs it, all rows in ``flush()``; same bits, no table passes), -1 = lazy when the tables hold at least 8 rows
    per batch row and 65536 rows in all."""

    SPARSE_STATEFUL = ("Adagrad", "Adadelta")     # fused when l2 == 0 (hip_ops.StatefulSparseState)

    @staticmethod
    def supports(name, l2):
        return name in ("SGD", "Adam") or (name in FusedOptimizer.SPARSE_STATEFUL and float(l2) == 0.0)

    def __init__(self, model, name, lr, l2, lazy=-1):
        if not self.supports(name, l2):
            raise ValueError("FusedOptimizer supports SGD, Adam and (with l2 = 0) Adagrad / Adadelta, got %r l2=%r" % (name, l2))
        self.model, self.name, self.lr, self.l2 = model, name, float(lr), float(l2)
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.adam_step = 0
        self.state = None
        self.lazy = int(lazy)
        self.lazy_state = None
        self._lazy_decided = None

    def zero_grad(self, set_to_none=True):
        self.model._pending = None

    def _adam_state(self):
        if self.state is None:
            m = self.model
            z = torch.zeros_like
            self.state = {"gU": z(m.user_embeddings.weight), "gI": z(m.item_embeddings.weight),
                          "mU": z(m.user_embeddings.weight), "vU": z(m.user_embeddings.weight),
                          "mI": z(m.item_embeddings.weight), "vI": z(m.item_embeddings.weight)}
        return self.state

    def _sparse_state(self, tabs):
        if self.lazy_state is None:
            self.lazy_state = hip_ops.StatefulSparseState(tabs, self.name, self.lr)
        elif self.lazy_state.tabs.U.data_ptr() != tabs.U.data_ptr() or self.lazy_state.tabs.I.data_ptr() != tabs.I.data_ptr():
            raise RuntimeError("the embedding tables were re-allocated after optimizer steps had been taken")
        return self.lazy_state

    def _use_lazy(self, tabs, batch):
        """decided at the first step and kept: the two representations of the optimizer state are not mixed"""
        if self._lazy_decided is None:
            rows = tabs.U.shape[0] + tabs.I.shape[0]
            stateful = self.name == "Adam" or self.l2 != 0.0
            self._lazy_decided = stateful and (self.lazy == 1 or (self.lazy == -1 and rows >= 65536 and rows >= 8 * batch))
        return self._lazy_decided

    @torch.no_grad()
    def step_batch(self, tabs, plan, k, loss_out=None):
        """zero_grad / predict / backward / optimizer.step of BaseRunner.py:196-199 on batch k of the plan"""
        if self.name in self.SPARSE_STATEFUL:
            self.adam_step += 1
            return self._sparse_state(tabs).step(plan, k, loss_out=loss_out)
        if self._use_lazy(tabs, plan.batch_size):
            if self.lazy_state is None:
                self.lazy_state = hip_ops.LazyOptimizerState(tabs, self.name, self.lr, self.l2, self.betas, self.eps)
            elif self.lazy_state.tabs.U.data_ptr() != tabs.U.data_ptr() or self.lazy_state.tabs.I.data_ptr() != tabs.I.data_ptr():
                raise RuntimeError("the embedding tables were re-allocated after optimizer steps had been taken")
            self.adam_step += 1
            return self.lazy_state.step(plan, k, loss_out=loss_out)
        if self.name == "SGD":
            return tabs.step_sgd(plan, k, self.lr, self.l2, loss_out=loss_out)
        st = self._adam_state()
        self.adam_step += 1
        loss, sid = tabs.grads(plan, k, st["gU"], st["gI"], loss_out=loss_out)
        hip_ops.adam_dense(tabs.U, st["mU"], st["vU"], st["gU"], self.adam_step, self.lr, self.l2, self.betas[0],
                           self.betas[1], self.eps, stamp=tabs.stamp_u, step_id=sid)
        hip_ops.adam_dense(tabs.I, st["mI"], st["vI"], st["gI"], self.adam_step, self.lr, self.l2, self.betas[0],
                           self.betas[1], self.eps, stamp=tabs.stamp_i, step_id=sid)
        return loss

    @torch.no_grad()
    def run_batches(self, tabs, plan, first, count, losses):
        """`count` consecutive step_batch calls; the lazy optimizers issue them from native code"""
        if self.name in self.SPARSE_STATEFUL:
            self.adam_step += count
            return self._sparse_state(tabs).run(plan, first, count, losses)
        if self._use_lazy(tabs, plan.batch_size):
            if self.lazy_state is None:
                self.lazy_state = hip_ops.LazyOptimizerState(tabs, self.name, self.lr, self.l2, self.betas, self.eps)
            elif self.lazy_state.tabs.U.data_ptr() != tabs.U.data_ptr() or self.lazy_state.tabs.I.data_ptr() != tabs.I.data_ptr():
                raise RuntimeError("the embedding tables were re-allocated after optimizer steps had been taken")
            self.adam_step += count
            return self.lazy_state.run(plan, first, count, losses)
        for k in range(count):
            self.step_batch(tabs, plan, first + k, loss_out=losses[k])
        return losses

    @torch.no_grad()
    def step(self):
        m = self.model
        if m._pending is None:
            raise RuntimeError("optimizer.step() without a preceding predict()/backward()")
        u, p, n, loss_buf = m._pending
        m._pending = None
        tabs = m._tables()
        plan = hip_ops.BatchPlan(u, p, n, u.numel(), m.user_num, m.item_num)
        self.step_batch(tabs, plan, 0, loss_out=loss_buf)

    def flush(self):
        """bring every row up to the current step (no-op for the dense representation); called by the model before
        anything but a training step reads the tables"""
        if self.lazy_state is not None:
            self.lazy_state.flush()


class _DeferredBprLoss(torch.autograd.Function):
    """The loss tensor handed back by ``predict`` when the fused optimizer owns the update: backward only records
    that it ran (the gradient never materialises; the step kernels consume the batch directly)."""

    @staticmethod
    def forward(ctx, user_w, item_w, model, loss_buf):
        ctx.model = model
        return loss_buf

    @staticmethod
    def backward(ctx, grad_out):
        ctx.model._backward_seen = True
        return None, None, None, None


class _DenseGradBprLoss(torch.autograd.Function):
    """Loss with real (dense) gradients for arbitrary torch optimizers: forward = wr_bpr_fwd, backward = the two step
    kernels in gradient-emitting mode (embedding_dense_backward of BaseRunner.py:198)."""

    @staticmethod
    def forward(ctx, user_w, item_w, model, u, p, n):
        out = hip_ops.bpr_fwd(user_w.detach(), item_w.detach(), u, p, n, scores=False)
        ctx.model, ctx.idx = model, (u, p, n)
        return out["loss"]

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        u, p, n = ctx.idx
        tabs = m._tables()
        plan = hip_ops.BatchPlan(u, p, n, u.numel(), m.user_num, m.item_num)
        gU = torch.zeros_like(tabs.U)
        gI = torch.zeros_like(tabs.I)
        tabs.grads(plan, 0, gU, gI, stamps=False)
        return gU * grad_out, gI * grad_out, None, None, None, None


def make_bprmf(general_model_cls):
    """Builds the BPRMF class on top of a GeneralModel base: ``host.GeneralModel`` stand-alone, or the reference's own
    ``models.BaseModel.GeneralModel`` when dropped into its tree (so its Dataset / reader plumbing is reused)."""

    class BPRMF(general_model_cls):
        reader = "BaseReader"
        runner = "BaseRunner"
        extra_log_args = ["embedding_size"]

        @staticmethod
        def parse_model_args(parser):
            parser.add_argument("--embedding_size", type=int, default=64, help="Size of embedding vectors.")
            parser.add_argument("--fused", type=int, default=1,
                                help="1: loss value is produced by the fused step at optimizer.step(); 0: eagerly in predict().")
            parser.add_argument("--lazy_optimizer", type=int, default=-1,
                                help="Adam / SGD with --l2: 1 = exact lazy row updates, 0 = dense table passes, -1 = by table size.")
            return general_model_cls.parse_model_args(parser)

        def __init__(self, args, corpus):
            super().__init__(args, corpus)
            self.emb_size = args.embedding_size
            if self.emb_size % 4 != 0:
                raise ValueError("embedding_size must be a multiple of 4 for the HIP kernels (got %d)" % self.emb_size)
            # same construction order and initialiser as the reference (BPRMF.py:35-40, init.py:13-29), so the same
            # torch seed gives bit-identical initial tables
            self.user_embeddings = nn.Embedding(self.user_num, self.emb_size)
            self.item_embeddings = nn.Embedding(self.item_num, self.emb_size)
            nn.init.xavier_normal_(self.user_embeddings.weight.data)
            nn.init.xavier_normal_(self.item_embeddings.weight.data)
            self.fused = int(getattr(args, "fused", 1))
            self._pending = None
            self._backward_seen = False
            self._tabs = None
            name = getattr(args, "optimizer", None)
            if name is not None and hasattr(args, "lr") and FusedOptimizer.supports(name, getattr(args, "l2", 0.0)):
                self.optimizer = FusedOptimizer(self, name, args.lr, getattr(args, "l2", 0.0),
                                                getattr(args, "lazy_optimizer", -1))

        # ------------------------------------------------------------------ plumbing
        def _tables(self):
            U, I = self.user_embeddings.weight.data, self.item_embeddings.weight.data
            if self._tabs is None or self._tabs.U.data_ptr() != U.data_ptr() or self._tabs.I.data_ptr() != I.data_ptr():
                self._tabs = hip_ops.BprmfTables(U, I)  # raises on CPU tensors: there is no CPU path
            return self._tabs

        def _batch(self, feed_dict):
            dev = self.user_embeddings.weight.device
            return tuple(feed_dict[k].to(dev).reshape(-1) for k in ("user_id", "pos_item", "neg_items"))

        def _sync_tables(self):
            """lazy optimizer rows -> current (a no-op otherwise); every reader of the tables outside a step calls it"""
            opt = getattr(self, "optimizer", None)
            if isinstance(opt, FusedOptimizer):
                opt.flush()

        def train(self, mode=True):
            if not mode:
                self._sync_tables()           # model.eval() precedes every evaluation (BaseRunner.py:229)
            return super().train(mode)

        def state_dict(self, *a, **kw):
            self._sync_tables()               # save_model (BaseModel.py:48-53)
            return super().state_dict(*a, **kw)

        def load_state_dict(self, *a, **kw):
            self._sync_tables()               # load_model (BaseModel.py:55-59): pending replays belong to the old weights
            return super().load_state_dict(*a, **kw)

        # ------------------------------------------------------------------ reference surface
        def forward(self, user, item):
            self._sync_tables()
            return (hip_ops.gather_rows(self.user_embeddings.weight.data, user),
                    hip_ops.gather_rows(self.item_embeddings.weight.data, item))

        def predict(self, feed_dict):
            u, p, n = self._batch(feed_dict)
            if self.training and isinstance(self.optimizer, FusedOptimizer):
                loss_buf = torch.zeros((), dtype=torch.float32, device=u.device)
                if not self.fused:
                    loss_buf = hip_ops.bpr_fwd(self.user_embeddings.weight.data, self.item_embeddings.weight.data, u, p, n,
                                               scores=False)["loss"]
                self._pending = (u, p, n, loss_buf if self.fused else None)
                return _DeferredBprLoss.apply(self.user_embeddings.weight, self.item_embeddings.weight, self, loss_buf)
            return _DenseGradBprLoss.apply(self.user_embeddings.weight, self.item_embeddings.weight, self, u, p, n)

        def full_predict(self, feed_dict):
            """scores[B, n_items] = U[user] @ I^T (BPRMF.py:82-91).  Evaluation only: a plain GEMM, left to rocBLAS."""
            self._sync_tables()
            dev = self.user_embeddings.weight.device
            user_e = hip_ops.gather_rows(self.user_embeddings.weight.data, feed_dict["user_id"].to(dev))
            return torch.matmul(user_e, self.item_embeddings.weight.data.t())

        def eval_factors(self):
            """(user matrix, item matrix) whose inner products are the ranking scores (full_predict, BPRMF.py:82-91)."""
            self._sync_tables()
            return self.user_embeddings.weight.data, self.item_embeddings.weight.data

        # ------------------------------------------------------------------ native epoch (used by HipRunner)
        def check_step_stream(self):
            """after an epoch's losses have been read (the device is idle anyway): raises if a bounded wait inside a chained
            step launch expired during the epoch (hip_ops.BprmfTables.check_chain; never observed)"""
            tabs = getattr(self, "_step_stream_tabs", None)
            if tabs is not None:
                tabs.check_chain()

        @torch.no_grad()
        def train_epoch(self, u, p, n, batch_size, lr, l2=0.0, optimizer="SGD", chunk=64, prep=None):
            """Runs one epoch over triplets already in batch order on the device; returns the per-batch losses
            (device tensor), i.e. the list BaseRunner.fit averages (BaseRunner.py:200-201).  prep: a hip_ops.EpochPrep whose
            ``cols`` are u, p, n — the columns are then produced chunk by chunk on the device (shuffle + negative sampling)
            beside the steps instead of before them."""
            tabs = self._tables()
            N = u.numel()
            nb = (N + batch_size - 1) // batch_size
            losses = torch.empty(nb, dtype=torch.float32, device=u.device)
            opt = self.optimizer if isinstance(self.optimizer, FusedOptimizer) else None
            if optimizer == "SGD" and l2 == 0.0:
                # stateless update: plans built on a side stream one chunk ahead of the steps, steps issued natively
                if getattr(self, "_pipe", None) is None:
                    self._pipe = hip_ops.PipelinedSgd(chunk)
                handle = self._pipe.plan(tabs.U, [(tabs.I, u, p, n)], batch_size, lr=lr, prep=prep)
                self._pipe.run(handle, 0, lr, losses)
                self._step_stream_tabs = handle["segs"][0]["tabs"]      # check_step_stream(): the chained launches' flag
                return losses
            if not FusedOptimizer.supports(optimizer, l2):
                raise ValueError("train_epoch supports SGD and Adam; use BaseRunner.fit for %r" % optimizer)
            if opt is None or opt.name != optimizer or opt.lr != float(lr) or opt.l2 != float(l2):
                if opt is not None and opt.adam_step > 0 and opt.name == optimizer:
                    raise ValueError("lr / l2 changed after optimizer steps had been taken")
                opt = self.optimizer = FusedOptimizer(self, optimizer, lr, l2, getattr(opt, "lazy", -1))
            # updates with optimizer state: the same plan pipeline (plans built on a side stream one chunk ahead, skewed ids
            # switched to load-balanced buckets after the first overflow), steps issued by the fused optimizer
            if getattr(self, "_pipe", None) is None:
                self._pipe = hip_ops.PipelinedSgd(chunk)
            def runner(plan, first, count, out):
                return opt.run_batches(tabs, plan, first, count, out)
            runner.wants_chain_marks = optimizer == "Adam"      # folded Adam steps go out as one launch per step
            handle = self._pipe.plan(tabs.U, [(tabs.I, u, p, n)], batch_size, prep=prep, runner=runner)
            self._pipe.run(handle, 0, lr, losses)
            return losses

    BPRMF.__qualname__ = "BPRMF"
    return BPRMF


BPRMF = make_bprmf(host.GeneralModel)


def bind(reference_general_model_cls):
    """``BPRMFHip = bind(GeneralModel)`` inside the reference tree."""
    return make_bprmf(reference_general_model_cls)
