"""SGL (self-supervised graph learning) on MI355X behind the reference's model contract (reference
src/models/general/SGL.py, src/utils/augmentor.py) — SURVEY §8(f)4: graph views built on CSR.

Same surface as the reference class: flags ``--embedding_size --gcn_layers --type --reg_weight --ssl_tau --ssl_weight
--drop_ratio`` (SGL.py:25-41), ``state_dict`` keys ``user_embedding.weight`` / ``item_embedding.weight``, xavier-uniform
init (:65), ``graph_construction()`` called from ``Dataset.actions_before_epoch`` (:67-79, :262), ``predict`` (:232-246),
``full_predict`` (:248-254).

What changes underneath:
  * the adjacency is an edge list in the row-major order ``scipy``'s ``nonzero()`` gives the reference (:81-103), never a
    DOK/LIL matrix and never dense (the reference multiplies ``to_dense()`` N x N views on CPU, :145-146);
  * the views draw from the SAME ``random.sample`` calls in the same order (augmentor.py:52-53,94), so a given ``random``
    seed drops the same edges / nodes; the construction around them is vectorised NumPy (no ``sp.diags`` products);
  * views are NOT symmetric (the reference drops (u,i) and (i,u) independently, and normalises by row sums on both
    sides, SGL.py:117-124) — so each view keeps a CSR of its transpose for the backward propagation;
  * the three propagations (:232-239) and their backward run on the chunked CSR SpMM kernel (``wr_spmm_csr_chunked``), row
    gathers / the deterministic sorted scatter-add / EmbLoss on the HIP row kernels; the InfoNCE term (:196-230) is two
    plain [B, n] GEMMs with exp/sum around them and stays on rocBLAS through ``torch.matmul``.
No CPU path.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops, host
from .lightgcn import DenseHipOptimizer


# ---------------------------------------------------------------------------------------------------- views (host)
def adjacency_edges(n_users, n_items, train_clicked_set):
    """(rows, cols) int64 of the symmetric bipartite adjacency over N = n_users + n_items nodes in row-major order —
    the order ``adj_matrix.nonzero()`` enumerates them in the reference (build_adjmat, SGL.py:81-103; augmentor.py:45,84)."""
    us, its = [], []
    for uu, items in train_clicked_set.items():
        if len(items):
            arr = np.fromiter(items, dtype=np.int64, count=len(items))
            us.append(np.full(arr.shape, uu, dtype=np.int64))
            its.append(arr)
    uu = np.concatenate(us) if us else np.zeros(0, np.int64)
    ii = (np.concatenate(its) + n_users) if its else np.zeros(0, np.int64)
    rows, cols = np.concatenate([uu, ii]), np.concatenate([ii, uu])
    order = np.lexsort((cols, rows))
    return rows[order], cols[order]


def edge_dropout_edges(rows, cols, dropout_rate):
    """augmentor.edge_dropout (augmentor.py:75-111): keep int(E*(1-rate)) edges chosen by random.sample(range(E), .)"""
    n_keep = int(rows.size * (1 - dropout_rate))
    keep = np.asarray(random.sample(range(rows.size), n_keep), dtype=np.int64)
    return rows[keep], cols[keep]


def node_dropout_edges(n_nodes, rows, cols, dropout_rate):
    """augmentor.node_dropout (augmentor.py:33-72): the matrix handed in is the N x N adjacency, so int(N*rate) ROW nodes
    and, independently, int(N*rate) COLUMN nodes are dropped; an edge survives when neither its row nor its column is."""
    n_drop = int(n_nodes * dropout_rate)
    drop_r = random.sample(range(n_nodes), n_drop)
    drop_c = random.sample(range(n_nodes), n_drop)
    keep_r = np.ones(n_nodes, bool); keep_r[drop_r] = False
    keep_c = np.ones(n_nodes, bool); keep_c[drop_c] = False
    m = keep_r[rows] & keep_c[cols]
    return rows[m], cols[m]


def norm_csr(n_nodes, rows, cols):
    """csr2tensor (SGL.py:105-146) on an edge list: value(r,c) = d_r^-1/2 * 1 * d_c^-1/2 with d = ROW sums + 1e-10, all in
    float32 like the reference's float32 matrices.  Returns (row_ptr int64 [N+1], col int32, val float32), columns
    ascending inside a row."""
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    cnt = np.bincount(rows, minlength=n_nodes)
    deg = cnt.astype(np.float32) + np.float32(1e-10)
    dis = np.power(deg, np.float32(-0.5)).astype(np.float32)
    val = (dis[rows] * np.float32(1.0)) * dis[cols]
    row_ptr = np.zeros(n_nodes + 1, np.int64)
    np.cumsum(cnt, out=row_ptr[1:])
    return row_ptr, cols.astype(np.int32), val.astype(np.float32)


def transpose_csr(n_nodes, row_ptr, col, val):
    rows = np.repeat(np.arange(n_nodes, dtype=np.int64), np.diff(row_ptr))
    order = np.lexsort((rows, col))
    cnt = np.bincount(col, minlength=n_nodes)
    tp = np.zeros(n_nodes + 1, np.int64)
    np.cumsum(cnt, out=tp[1:])
    return tp, rows[order].astype(np.int32), val[order]


class CsrGraph:
    """One normalised view on the device: chunked CSR of A (forward) and of A^T (backward; shared when symmetric)."""

    def __init__(self, n_nodes, row_ptr, col, val, device, symmetric=False):
        self.n = n_nodes
        self.fwd = self._upload(row_ptr, col, val, device)
        self.bwd = self.fwd if symmetric else self._upload(*transpose_csr(n_nodes, row_ptr, col, val), device)
        self._partials = None

    @staticmethod
    def _upload(row_ptr, col, val, device):
        cptr, crow = hip_ops.spmm_chunks(row_ptr)
        out = tuple(t.to(device) for t in (cptr, crow, torch.from_numpy(np.ascontiguousarray(col)),
                                           torch.from_numpy(np.ascontiguousarray(val))))
        out[1]._wr_levels = crow._wr_levels                                     # combine levels of the product (spmm_chunks)
        return out

    def propagate(self, E0, layers, transpose=False):
        """mean over l = 0..layers of A^l E0 (SGL.forward, SGL.py:148-164), or of (A^T)^l E0"""
        cptr, crow, col, val = self.bwd if transpose else self.fwd
        if layers == 0:
            return E0.clone()
        cur = E0
        need = (crow.numel(), E0.shape[1])
        if self._partials is None or tuple(self._partials.shape) != need or self._partials.device != E0.device:
            self._partials = torch.empty(need, dtype=torch.float32, device=E0.device)
        acc = torch.empty_like(E0)       # started by the first product from its input, scaled to the mean by the last one
        for l in range(layers):
            cur = hip_ops.spmm_csr_chunked(cptr, crow, col, val, cur, acc=acc, partials=self._partials, acc_from_x=l == 0,
                                           acc_scale=1.0 / (layers + 1) if l == layers - 1 else 1.0)
        return acc


class _IdentityView:
    """type 'RW': graph_construction leaves [] (SGL.py:69,75), forward() over an empty list returns the embeddings"""

    def propagate(self, E0, layers, transpose=False):
        return E0.clone()


# ---------------------------------------------------------------------------------------------------- model
class _SglLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user_w, item_w, model, u, p, n):
        loss, gE0 = model._loss_and_grad(u, p, n)
        ctx.gE0, ctx.nU = gE0, model.n_users
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        s = grad_out.reshape(-1)[0]
        return ctx.gE0[:ctx.nU] * s, ctx.gE0[ctx.nU:] * s, None, None, None, None


def make_sgl(general_model_cls):
    class SGL(general_model_cls):
        reader = "BaseReader"
        runner = "BaseRunner"
        extra_log_args = ["embedding_size", "gcn_layers", "reg_weight", "type", "ssl_tau", "ssl_weight", "drop_ratio"]
        graph_capturable = True     # HipRunner may capture the step in a hipGraph; re-captured when the views change (graph_key)

        @staticmethod
        def parse_model_args(parser):
            parser.add_argument("--embedding_size", type=int, default=64, help="Size of embedding vectors.")
            parser.add_argument("--gcn_layers", type=int, default=2, help="Number of SGL layers.")
            parser.add_argument("--type", type=str, default="ED", help="The type to generate views. Range in ['ED', 'ND', 'RW'].")
            parser.add_argument("--reg_weight", type=float, default=1e-4, help="The L2 regularization weight.")
            parser.add_argument("--ssl_tau", type=float, default=0.1, help="The temperature in softmax.")
            parser.add_argument("--ssl_weight", type=float, default=0.05, help="The hyperparameter to control the strengths of SSL.")
            parser.add_argument("--drop_ratio", type=float, default=0.1, help="The dropout ratio.")
            return general_model_cls.parse_model_args(parser)

        def __init__(self, args, corpus):
            super().__init__(args, corpus)
            self.n_users, self.n_items = int(corpus.n_users), int(corpus.n_items)
            self.emb_size = args.embedding_size
            if self.emb_size % 4 != 0:
                raise ValueError("embedding_size must be a multiple of 4 for the HIP kernels (got %d)" % self.emb_size)
            self.gcn_layers = args.gcn_layers
            self.reg_weight = float(args.reg_weight)
            self.type = str(args.type)
            self.ssl_weight, self.ssl_tau, self.drop_ratio = args.ssl_weight, args.ssl_tau, args.drop_ratio
            self.user_embedding = nn.Embedding(self.n_users, self.emb_size)
            self.item_embedding = nn.Embedding(self.n_items, self.emb_size)
            # plain attributes like the reference's graphs (absent from state_dict)
            self._edges = adjacency_edges(self.n_users, self.n_items, corpus.train_clicked_set)
            self._graphs = {}            # name -> host CSR; uploaded lazily on the embeddings' device
            self._graphs_dev = {}
            self._graphs["train"] = (norm_csr(self.n_users + self.n_items, *self._edges), True)
            nn.init.xavier_uniform_(self.user_embedding.weight.data)    # self.apply(xavier_uniform_initialization), SGL.py:65
            nn.init.xavier_uniform_(self.item_embedding.weight.data)
            name = getattr(args, "optimizer", None)
            if name in ("SGD", "Adam") and hasattr(args, "lr"):
                self.optimizer = DenseHipOptimizer([self.user_embedding.weight, self.item_embedding.weight], name, args.lr,
                                                   getattr(args, "l2", 0.0))

        # ------------------------------------------------------------------ views
        def graph_key(self):
            return getattr(self, "_views_version", 0)

        def graph_construction(self):
            """two views per epoch from Python's ``random`` stream, in the reference's order (SGL.py:67-79)"""
            self._views_version = getattr(self, "_views_version", 0) + 1
            N = self.n_users + self.n_items
            rows, cols = self._edges
            for name in ("sub1", "sub2"):
                self._graphs_dev.pop(name, None)
                if self.type == "ND":
                    r, c = node_dropout_edges(N, rows, cols, self.drop_ratio)
                elif self.type == "ED":
                    r, c = edge_dropout_edges(rows, cols, self.drop_ratio)
                else:
                    self._graphs[name] = None
                    continue
                self._graphs[name] = (norm_csr(N, r, c), False)

        def _graph(self, name):
            dev = self.user_embedding.weight.device
            g = self._graphs_dev.get(name)
            if g is None or getattr(g, "device", dev) != dev:
                spec = self._graphs.get(name)
                if name != "train" and name not in self._graphs:
                    raise AttributeError("graph_construction() has not been called (SGL.Dataset.actions_before_epoch)")
                if spec is None:
                    g = _IdentityView()
                else:
                    (rp, col, val), sym = spec
                    g = CsrGraph(self.n_users + self.n_items, rp, col, val, dev, symmetric=sym)
                g.device = dev
                self._graphs_dev[name] = g
            return g

        def forward(self, graph="train"):
            E0 = torch.cat([self.user_embedding.weight.data, self.item_embedding.weight.data], dim=0)
            allE = self._graph(graph).propagate(E0, self.gcn_layers)
            return allE[:self.n_users], allE[self.n_users:]

        # ------------------------------------------------------------------ loss + gradient
        def _ssl(self, u, p, E1, E2):
            """calc_ssl_loss (SGL.py:196-230): InfoNCE of each batch row against ALL rows of the second view"""
            nU, tau = self.n_users, self.ssl_tau

            def side(idx, A, Bm):
                e1 = F.normalize(A[idx], dim=1)
                e2 = F.normalize(Bm[idx], dim=1)
                all2 = F.normalize(Bm, dim=1)
                v1 = torch.exp(torch.sum(e1 * e2, dim=1) / tau)
                v2 = torch.sum(torch.exp(e1.matmul(all2.T) / tau), dim=1)
                return -torch.sum(torch.log(v1 / v2))

            return (side(p, E1[nU:], E2[nU:]) + side(u, E1[:nU], E2[:nU])) * self.ssl_weight

        @torch.no_grad()
        def _loss_and_grad(self, u, p, n):
            nU, L, B = self.n_users, self.gcn_layers, u.numel()
            U0, I0 = self.user_embedding.weight.data, self.item_embedding.weight.data
            E0 = torch.cat([U0, I0], dim=0)
            gm, g1, g2 = self._graph("train"), self._graph("sub1"), self._graph("sub2")
            Em, E1, E2 = gm.propagate(E0, L), g1.propagate(E0, L), g2.propagate(E0, L)
            # --- BPR on the main view: sum of -logsigmoid(pos - neg) (calc_bpr_loss, SGL.py:166-194)
            idx = torch.cat([u, p + nU, n + nU])
            rows = hip_ops.gather_rows(Em, idx)
            ue, pe, ne = rows[:B], rows[B:2 * B], rows[2 * B:]
            x = (ue * pe).sum(dim=1) - (ue * ne).sum(dim=1)
            l1 = F.softplus(-x).sum()
            coef = -torch.sigmoid(-x).unsqueeze(1)
            gEm = torch.zeros_like(Em)
            hip_ops.scatter_add_rows(gEm, idx, torch.cat([coef * (pe - ne), coef * ue, -coef * ue]))
            # --- EmbLoss on the raw rows (utils/loss.py:94-98)
            sq = hip_ops.embloss_sumsq(U0, I0, u, p, n)
            reg = torch.sqrt(sq).sum() / B
            # --- InfoNCE between the two views: plain GEMMs, autograd over the two propagated tables only
            with torch.enable_grad():
                E1r, E2r = E1.detach().requires_grad_(True), E2.detach().requires_grad_(True)
                ssl = self._ssl(u, p, E1r, E2r)
                gE1, gE2 = torch.autograd.grad(ssl, [E1r, E2r])
            loss = l1 + reg * self.reg_weight + ssl.detach()
            # --- back through the three propagations (views are not symmetric: transposed CSR)
            gE0 = gm.propagate(gEm, L, transpose=True)
            gE0 += g1.propagate(gE1.contiguous(), L, transpose=True)
            gE0 += g2.propagate(gE2.contiguous(), L, transpose=True)
            # indices already range-checked for the whole epoch (HipRunner): no per-batch read-back, the step never waits on the host
            plan = hip_ops.BatchPlan(u, p, n, B, nU, self.n_items, builder="small" if B <= 4096 else "generic", hot=False,
                                     validate=not getattr(self, "_trusted_indices", False))
            hip_ops.embloss_grad(U0, I0, plan, 0, sq, self.reg_weight, gE0[:nU], gE0[nU:])
            return loss, gE0

        def _batch(self, feed_dict):
            dev = self.user_embedding.weight.device
            return tuple(feed_dict[k].to(dev).reshape(-1) for k in ("user_id", "pos_item", "neg_items"))

        def predict(self, feed_dict):
            u, p, n = self._batch(feed_dict)
            return _SglLoss.apply(self.user_embedding.weight, self.item_embedding.weight, self, u, p, n)

        def eval_factors(self):
            Ua, Ia = self.forward("train")
            return Ua.contiguous(), Ia.contiguous()

        def full_predict(self, feed_dict):
            dev = self.user_embedding.weight.device
            Ua, Ia = self.forward("train")
            user_e = hip_ops.gather_rows(Ua.contiguous(), feed_dict["user_id"].to(dev))
            return torch.matmul(user_e, Ia.t())

        class Dataset(general_model_cls.Dataset):
            def actions_before_epoch(self):
                super().actions_before_epoch()
                self.model.graph_construction()          # SGL.py:258-262

    SGL.__qualname__ = "SGL"
    return SGL


SGL = make_sgl(host.GeneralModel)


def bind(reference_general_model_cls):
    return make_sgl(reference_general_model_cls)
