"""Row-sharded BPRMF step across the GPUs of one node: one process per GPU, RCCL all-to-all over xGMI.

The reference is single-device; this module is new design, not a translation.  Semantics are those of ONE
BaseRunner.fit iteration (reference src/helpers/BaseRunner.py:196-199) over the union of all ranks' batches: synchronous SGD,
every gradient taken from the pre-step tables, loss = mean over the global batch, negatives drawn from ALL items
(src/models/BaseModel.py:168,174).

Layout
  * both tables are row-sharded cyclically: row r lives on rank r % G at local index r // G (cyclic keeps popular — low-id —
    items spread over all shards; a checkpoint is re-assembled by interleaving, ``gather_full``).
  * interactions are partitioned by user owner, so the user row of every triplet is local and only ITEM rows cross links.
    Each rank draws its own batches (size B) from its partition; the global batch is their union.
  * a rank's item buffer holds its own rows followed by 2 B "slot" rows: the rows a step receives from their owners land
    right behind the shard, so the step kernels address local and received rows alike ("virtual" item ids).
One rank (world = 1): every item row is local — the step stream is the single-GPU one (hip_ops.PipelinedSgd: group plans, one
launch per step), nothing is exchanged.
Per chunk of steps (index work only — it never depends on table values; native kernels, no host-side sort)
  * wr_shard_route: local items keep their local row; a step's distinct REMOTE items get slots in ascending (owner, row) order
    (LDS bitmap of the routed keys + prefix popcounts); per owner the requested rows go into a padded send buffer;
  * TWO fixed-size all-to-alls carry the request lists and their lengths of all steps of the chunk; wr_shard_pack turns what
    arrived into per-step lists of rows to serve; ONE read-back brings the lengths to the host (the split sizes of the
    steps' row exchanges);
  * the group plan of the chunk (wr_group_plan_build) on the virtual ids: which rows recur inside a batch.
Per step (hot path)
  1. owner: wr_gather_rows of the rows requested from it            -> all-to-all (rows to the requesters' slot rows)
  2. requester: wr_bprmf_shard_step_group — user rows and LOCAL item rows updated in place, one summed gradient row per slot
                                                                    -> all-to-all (gradient rows back to the owners)
  3. owner: wr_scatter_add_rows(alpha = -lr): rows requested by several ranks are summed in a fixed order
  No collective carries the loss: per-step partials are all-reduced once per chunk.
A row that its owner also trains on is updated twice in a step (the owner's own contribution in place, the others' in 3.):
the same sum, rounded once more — within the 1e-5 of the north star, and the same bits from run to run.
Link budget per rank and step: 2 * D*4 * (unique remote items) bytes each way — see DESIGN.md §6.
"""
import math
import time

import numpy as np
import torch
import torch.distributed as dist


class HipBackend:
    """Local compute through the C-ABI.  (Tests substitute an oracle-backed object with the same methods to exercise the
    exchange logic on CPU/gloo; the product never does.)"""

    def __init__(self):
        from . import abi, hip_ops
        self.abi, self.ops = abi, hip_ops

    def route(self, u, p, n, batch, world, rank, n_users, n_items, M, nL, C):
        dev, N = u.device, u.numel()
        nb = (N + batch - 1) // batch
        i32 = dict(dtype=torch.int32, device=dev)
        vu, vp, vn = torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N, **i32)
        send_rows, send_cnt = torch.empty(world * nb * C, **i32), torch.empty(world * nb, **i32)
        err = torch.zeros(2, **i32)
        u, p, n = (t.to(torch.int32).contiguous() for t in (u, p, n))
        self.abi.check(self.abi.lib().wr_shard_route(u.data_ptr(), p.data_ptr(), n.data_ptr(), N, batch, world, rank, n_users,
                                                     n_items, M, nL, C, vu.data_ptr(), vp.data_ptr(), vn.data_ptr(),
                                                     send_rows.data_ptr(), send_cnt.data_ptr(), err.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "wr_shard_route")
        return vu, vp, vn, send_rows, send_cnt, err

    def pack(self, recv_rows, recv_cnt, nb, world, C, nL):
        dev = recv_rows.device
        serve_rows = torch.empty(nb * world * C, dtype=torch.int64, device=dev)
        serve_off = torch.empty(nb * (world + 1), dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.abi.check(self.abi.lib().wr_shard_pack(recv_rows.data_ptr(), recv_cnt.data_ptr(), nb, world, C, nL,
                                                    serve_rows.data_ptr(), world * C, serve_off.data_ptr(), err.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "wr_shard_pack")
        return serve_rows.view(nb, world * C), serve_off.view(nb, world + 1), err

    GROUP_MIN_ROWS_PER_TRIPLET = 12      # as hip_ops.PipelinedSgd: below this the lists of shared rows outgrow their capacity

    def plan_local_begin(self, vu, vp, vn, batch, n_user_rows, n_ext_rows, n_local_items, D=64):
        """which rows recur inside a batch: the group plan (no sort) where the shard is large against the batch and rows are
        whole 128-byte lines; otherwise — and for any chunk whose lists overflow (popularity-skewed ids, decided in
        plan_local_end when the plan's words have been read) — the sorted batch plan (wr_plan*.hip) and its two-kernel step
        with the hot-row path.  Nothing is read back here."""
        ops = self.ops
        tabs_ok = (int(D) * 4) % 128 == 0          # rows are whole 128-byte lines (torch allocations are 256-byte aligned)
        if tabs_ok and batch <= 131072 and min(n_user_rows, n_local_items) >= self.GROUP_MIN_ROWS_PER_TRIPLET * batch:
            return ops.GroupPlan(vu, vp, vn, batch, n_user_rows, n_ext_rows, defer=True)
        return None

    def plan_local_meta(self, plan):
        """the three words plan_local_end wants to see (zeros when no group plan was started)"""
        if plan is None:
            return torch.zeros(3, dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        return plan.buf[:3]

    def plan_local_end(self, plan, meta, vu, vp, vn, batch, n_user_rows, n_ext_rows):
        if plan is not None:
            plan.finish_from(meta)
            plan.validate()
            if not plan.overflow and not plan.long_run:
                return plan
        return self.ops.BatchPlan(vu, vp, vn, batch, n_user_rows, n_ext_rows, validate=True)

    def plan_local(self, vu, vp, vn, batch, n_user_rows, n_ext_rows, n_local_items, D=64):
        plan = self.plan_local_begin(vu, vp, vn, batch, n_user_rows, n_ext_rows, n_local_items, D)
        meta = plan.buf[:3].cpu().numpy() if plan is not None else None
        return self.plan_local_end(plan, meta, vu, vp, vn, batch, n_user_rows, n_ext_rows)

    def gather_rows(self, tab, idx):
        return self.ops.gather_rows(tab, idx)

    def local_step(self, U, I_ext, nL, plan, k, global_batch, lr, grad_slots, loss_out):
        L, ops = self.abi.lib(), self.ops
        if isinstance(plan, ops.BatchPlan):
            import ctypes
            off, B = k * plan.batch_size, plan.batch_len(k)
            nbytes = self.abi.check_size(L.wr_bprmf_step_workspace_bytes(plan.batch_size, U.shape[1]), "workspace")
            ws = ops.workspace(U.device, "step").get(nbytes)
            hot = plan.hot_struct(k)
            self.abi.check(L.wr_bprmf_shard_step(U.data_ptr(), U.shape[0], I_ext.data_ptr(), I_ext.shape[0], nL, U.shape[1],
                                                 plan.tu.data_ptr() + 4 * off, plan.tp.data_ptr() + 4 * off,
                                                 plan.tn.data_ptr() + 4 * off, plan.oc_item.data_ptr() + 8 * off,
                                                 plan.oc_src.data_ptr() + 8 * off, B, global_batch, lr, grad_slots.data_ptr(),
                                                 loss_out.data_ptr(), ctypes.addressof(hot) if hot is not None else None,
                                                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                           "wr_bprmf_shard_step")
            return
        nbytes = self.abi.check_size(L.wr_bprmf_group_workspace_bytes(plan.batch_size, U.shape[1]), "workspace")
        ws = ops.workspace(U.device, "group_step").get(nbytes)
        sync = ops.BprmfTables(U, I_ext)._group_sync(1)
        self.abi.check(L.wr_bprmf_shard_step_group(U.data_ptr(), U.shape[0], I_ext.data_ptr(), I_ext.shape[0], nL, U.shape[1],
                                                   plan.u.data_ptr(), plan.p.data_ptr(), plan.n.data_ptr(), plan.n_triplets,
                                                   plan.batch_size, plan.buf.data_ptr(), plan.buf.numel(), k, global_batch, lr,
                                                   grad_slots.data_ptr(), loss_out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   sync.data_ptr(), sync.numel(), torch.cuda.current_stream().cuda_stream),
                       "wr_bprmf_shard_step_group")

    def scatter_add(self, tab, idx, src, alpha):
        self.ops.scatter_add_rows(tab, idx, src, alpha=alpha)

    def apply_plan(self, serve_rows, serve_len, n_rows):
        """row plan of a chunk's gradient-row applications (wr_scatter.hip): which of the rows a step serves were requested
        by several ranks, and in which order their gradient rows are summed — index work, once per chunk; None where it
        does not apply (more than 2^18 rows served per step)"""
        try:
            return self.ops.ScatterPlan(serve_rows, n_rows, seg_len=serve_len)
        except ValueError:
            return None

    def check(self, U, I_ext):
        self.ops.BprmfTables(U, I_ext).check_chain()


def n_local_rows(n_rows, rank, world):
    return (n_rows - rank + world - 1) // world


class ChunkPlan:
    pass


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ShardedBprmf:
    def __init__(self, n_users, n_items, emb_size, device, backend=None, group=None, loopback=0):
        """loopback = G > 1 (measurement only, with ONE real rank): this rank does rank 0's work of a G-rank job — routing,
        request lists, gathers, the step on received rows, gradient rows back, scatter-add — with every exchange replaced
        by a device copy from itself (the other ranks are taken to be clones: what it requests from owner o is what
        requester o is taken to request from it).  Prices the per-rank work beside the links; no result to compare."""
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.loopback = int(loopback) if int(loopback) > 1 else 0
        if self.loopback:
            assert self.world == 1, "loopback runs with one real rank"
            self.rank, self.world = 0, self.loopback
        self.n_users, self.n_items, self.D = int(n_users), int(n_items), int(emb_size)
        self.device = device
        self.backend = backend if backend is not None else HipBackend()
        # rows per owner in the routed key space (owner * M + local row): a multiple of 32, so that a word of the routing
        # bitmap belongs to one owner
        self.M = ((self.n_items + self.world - 1) // self.world + 31) // 32 * 32
        self.nL = n_local_rows(self.n_items, self.rank, self.world)
        self.U = torch.zeros(n_local_rows(self.n_users, self.rank, self.world), self.D, device=device)
        self.I_ext = torch.zeros(max(self.nL, 1), self.D, device=device)       # grows by 2 B slot rows once the batch size is known
        self.I = self.I_ext[:self.nL]
        self._local = None               # world = 1: the single-GPU step stream
        self._side = None                # index work of the next chunk runs on this stream beside the steps

    def _ensure_slots(self, batch):
        """the item buffer = the shard's rows + 2 * batch slot rows (a step receives at most that many distinct rows)"""
        need = self.nL + 2 * int(batch)
        if self.I_ext.shape[0] < need:
            ext = torch.zeros(need, self.D, device=self.device)
            ext[:self.nL].copy_(self.I)
            self.I_ext, self.I = ext, ext[:self.nL]

    # ------------------------------------------------------------------ tables
    def load_full(self, U_full, I_full):
        """Take this rank's rows out of full tables (the reference's single-tensor layout)."""
        self.U.copy_(U_full[self.rank::self.world].to(self.device))
        self.I.copy_(I_full[self.rank::self.world].to(self.device))

    def init_xavier(self, seed):
        """N(0, 2/(rows+D)) like xavier_normal_ on the full tables (reference src/models/init.py:25); each shard draws
        its own rows from a per-rank stream."""
        g = torch.Generator(device=self.device)
        g.manual_seed(seed * 1000 + self.rank)
        self.U.normal_(0.0, math.sqrt(2.0 / (self.n_users + self.D)), generator=g)
        self.I.normal_(0.0, math.sqrt(2.0 / (self.n_items + self.D)), generator=g)

    def gather_full(self):
        """All-gather the shards into the reference's checkpoint layout (user_embeddings.weight, item_embeddings.weight)."""
        if self.device.type == "cuda" and hasattr(self.backend, "check"):
            self.backend.check(self.U, self.I_ext)      # never a checkpoint of tables a step launch has declared invalid
        assert not self.loopback, "loopback objects hold one rank's share only"
        out = []
        for tab, n in ((self.U, self.n_users), (self.I, self.n_items)):
            cap = (n + self.world - 1) // self.world
            padded = torch.zeros(cap, self.D, device=self.device)
            padded[:tab.shape[0]] = tab
            parts = [torch.empty_like(padded) for _ in range(self.world)]
            dist.all_gather(parts, padded, group=self.group)
            full = torch.stack(parts, dim=1).reshape(cap * self.world, self.D)[:n]
            out.append(full)
        return out

    # ------------------------------------------------------------------ plan (index work, once per chunk)
    def list_cap(self, batch):
        """entries of one (owner, step) request list: a step's 2 B item occurrences spread over the G owners, with room"""
        G = self.world
        return 2 * batch if G <= 2 else min(2 * batch, (2 * batch // G) * 3 // 2 + 1024)

    def plan_chunk(self, u, p, n, batch):
        """u, p, n: this rank's triplets of the chunk in batch order (global ids, every u % world == rank).
        All ranks must call this with the same number of steps."""
        return self.plan_chunk_end(self.plan_chunk_begin(u, p, n, batch))

    def plan_chunk_begin(self, u, p, n, batch):
        """queues the chunk's index work — on a side stream when the tables live on a GPU, so that it runs beside the steps of
        the chunk before (call it BEFORE run_chunk of that chunk, plan_chunk_end after); nothing is read back here"""
        G, dev = self.world, u.device
        N = u.numel()
        nb = (N + batch - 1) // batch
        cp = ChunkPlan()
        cp.nb, cp.batch, cp.N = nb, batch, N
        cp.ready = None
        if G == 1:
            # every item row is local: the single-GPU step stream on this rank's tables
            cp.u, cp.p, cp.n = (t.to(torch.int32).contiguous() for t in (u, p, n))
            return cp
        self._ensure_slots(batch)
        C = cp.C = self.list_cap(batch)
        side = None
        if dev.type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(dev))      # the chunk before last is done with: its arrays may be reused
        with torch.cuda.stream(side) if side is not None else _null():
            vu, vp, vn, send_rows, send_cnt, err = self.backend.route(u, p, n, batch, G, self.rank, self.n_users, self.n_items,
                                                                      self.M, self.nL, C)
            # the request lists of the whole chunk: two fixed-size all-to-alls (per peer: nb lists of C rows, nb lengths)
            recv_rows, recv_cnt = torch.empty_like(send_rows), torch.empty_like(send_cnt)
            self._a2a(recv_cnt, send_cnt)
            self._a2a(recv_rows, send_rows)
            if self.loopback:
                recv_rows.clamp_(max=max(self.nL - 1, 0))      # the clones' shards may be one row shorter than rank 0's
            serve_rows, serve_off, err2 = self.backend.pack(recv_rows, recv_cnt, nb, G, C, self.nL)
            cp.ids = (vu, vp, vn)
            words = [send_cnt.view(G, nb).t().reshape(-1), serve_off.reshape(-1), err, err2]
            cp.local = cp.apply = None
            if hasattr(self.backend, "plan_local_begin"):
                cp.local = self.backend.plan_local_begin(vu, vp, vn, batch, self.U.shape[0], self.I_ext.shape[0], self.nL, self.D)
                words.append(self.backend.plan_local_meta(cp.local))
                cp.apply = self.backend.apply_plan(serve_rows, serve_off[:, G].contiguous(), self.nL)
                words.append(cp.apply.meta[1:2] if cp.apply is not None else err2[:0])
            # ONE read-back per chunk: the lengths (split sizes of the steps' row exchanges) and the error words
            cat = torch.cat(words)
            if side is not None:
                cp.host = torch.empty(cat.numel(), dtype=cat.dtype, pin_memory=True)
                cp.host.copy_(cat, non_blocking=True)
                cp.ready = torch.cuda.Event()
                cp.ready.record(side)
            else:
                cp.host = cat
            cp.serve_rows = serve_rows
            cp.keep = (send_rows, send_cnt, recv_rows, recv_cnt, serve_off, cat)      # alive until the chunk is dropped
        return cp

    def plan_chunk_end(self, cp):
        """waits for the chunk's index work, reads its lengths and error words; the calling stream is made to wait for it"""
        if self.world == 1:
            return cp
        G, nb, C = self.world, cp.nb, cp.C
        if cp.ready is not None:
            cp.ready.synchronize()
            torch.cuda.current_stream(self.device).wait_event(cp.ready)
        host = cp.host.numpy()
        req = host[:nb * G].reshape(nb, G)                                        # rows I request from owner o at step k
        off = host[nb * G:nb * G + nb * (G + 1)].reshape(nb, G + 1)               # where requester s's rows start in my serve list
        e = host[nb * G + nb * (G + 1):]
        if e[0] != 0:
            raise IndexError("row-sharded step: id out of range, or a triplet whose user this rank does not own")
        if e[1] != 0:
            raise RuntimeError("row-sharded step: a request list exceeds its capacity (%d rows per owner and step)" % C)
        if e[2] != 0:
            raise RuntimeError("row-sharded step: a peer requested a row outside this shard")
        vu, vp, vn = cp.ids
        if cp.local is not None or hasattr(self.backend, "plan_local_begin"):
            cp.local = self.backend.plan_local_end(cp.local, e[3:6], vu, vp, vn, cp.batch, self.U.shape[0], self.I_ext.shape[0])
            if cp.apply is not None and e[6] != 0:
                cp.apply = None    # a few rows draw most of the requests: the per-step sorted scatter-add is made for that
        else:
            cp.local = self.backend.plan_local(vu, vp, vn, cp.batch, self.U.shape[0], self.I_ext.shape[0], self.nL, self.D)
        cp.req_splits = req.tolist()
        cp.serve_splits = np.diff(off, axis=1).tolist()
        cp.nq = req.sum(1).tolist()
        cp.ns = off[:, G].tolist()
        cp.max_nq, cp.max_ns = max(cp.nq + [1]), max(cp.ns + [1])
        return cp

    # ------------------------------------------------------------------ hot path
    def run_chunk(self, cp, lr, global_batch=None):
        """Runs the cp.nb steps; returns this rank's per-step loss shares (sum over ranks = global mean loss)."""
        D, dev = self.D, self.device
        if self.world == 1:
            return self._run_local(cp, lr, global_batch)
        losses = torch.zeros(cp.nb, dtype=torch.float32, device=dev)
        grad_slots = torch.empty(cp.max_nq, D, device=dev)
        grad_recv = torch.empty(cp.max_ns, D, device=dev)
        nL = self.nL
        for k in range(cp.nb):
            Bk = min(cp.batch, cp.N - k * cp.batch)
            gb = global_batch if global_batch is not None else Bk * self.world
            nq, ns = cp.nq[k], cp.ns[k]
            send = self.backend.gather_rows(self.I, cp.serve_rows[k, :ns]) if ns > 0 else grad_recv[:0]
            self._a2a(self.I_ext[nL:nL + nq], send, cp.req_splits[k], cp.serve_splits[k])
            gs = grad_slots[:nq]
            self.backend.local_step(self.U, self.I_ext, nL, cp.local, k, gb, lr, grad_slots, losses[k:k + 1])
            gr = grad_recv[:ns]
            self._a2a(gr, gs, cp.serve_splits[k], cp.req_splits[k])
            if ns > 0 and cp.apply is not None:
                cp.apply.apply(self.I, k, ns, gr, alpha=-lr)
            elif ns > 0:
                self.backend.scatter_add(self.I, cp.serve_rows[k, :ns], gr, -lr)
        return losses

    def _a2a(self, out, inp, out_splits=None, in_splits=None):
        if self.loopback:
            out.copy_(inp)          # the clone's answer has the same shape: a device copy stands in for the links
            return
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=self.group)

    def _run_local(self, cp, lr, global_batch):
        """one rank: the single-GPU step stream (plans one chunk ahead on a side stream, one launch per step)"""
        if hasattr(self.backend, "local_stream"):
            return self.backend.local_stream(self, cp, lr, global_batch)
        if global_batch is not None and global_batch != cp.batch:
            # a caller that prices this rank's batch as a share of a larger global batch: gradients and loss scale with
            # batch / global_batch — the same step at a learning rate scaled alike, its loss share scaled afterwards
            scale = cp.batch / float(global_batch)
        else:
            scale = 1.0
        if self._local is None:
            from .hip_ops import PipelinedSgd
            self._local = PipelinedSgd(chunk=max(cp.nb, 1), min_triplets=1)
        losses = torch.empty(cp.nb, dtype=torch.float32, device=self.device)
        handle = self._local.plan(self.U, [(self.I, cp.u, cp.p, cp.n)], cp.batch)
        self._local.run(handle, 0, lr * scale, losses)
        return losses * scale if scale != 1.0 else losses

    def global_losses(self, local_losses):
        out = local_losses.clone()
        if self.loopback:
            return out
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out


# ---------------------------------------------------------------------------------------------------- bench (N > 1)
def bench_run(args, rank, world, dev):
    """bench.py --gpus N (N > 1), mode 'alltoall': weak scaling, every rank runs batches of args.batch triplets whose users
    it owns; tables are row-sharded over the N GPUs; negatives uniform over ALL items as in the reference
    (src/models/BaseModel.py:168).  The process group exists already; returns the result dict on rank 0."""
    from .rotating import dedup_step_bytes
    B, D, K, W = args.batch, args.emb, args.steps, args.warmup
    chunk = args.chunk if args.chunk > 0 else max(1, min(64, K))
    model = ShardedBprmf(args.users, args.items, D, dev)
    model.init_xavier(3407)
    g = torch.Generator(device=dev)
    g.manual_seed(3407 * 7919 + rank)
    spare = chunk         # one chunk behind the timed steps that is planned inside the region and never trained (see below)
    n_trip = (K + W + spare) * B
    u = (torch.randint(0, model.U.shape[0], (n_trip,), generator=g, device=dev) * world + rank).to(torch.int32)   # users this rank owns
    p = torch.randint(0, args.items, (n_trip,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, args.items, (n_trip,), generator=g, device=dev, dtype=torch.int32)

    import gc
    if world == 1:
        # one rank: every row is local and the step stream is the single-GPU one, consumed exactly as bench.py's N = 1 path
        # consumes it — ONE stream of warm-up | timed steps | one spare chunk nobody trains on, so that the plan of the first
        # timed chunk is built beside the warm-up and K batches' worth of plan builds run inside the timed region
        from .hip_ops import PipelinedSgd
        pipe = model._local = PipelinedSgd(chunk=chunk, min_triplets=1)
        lw = torch.empty(max(W, 1), dtype=torch.float32, device=dev)
        lt = torch.empty(K, dtype=torch.float32, device=dev)
        gc.disable()
        try:
            handle = pipe.plan(model.U, [(model.I, u, p, n)], B, first_chunk=[W - W // 2, W // 2] if W > 0 else None)
            if W > 0:
                pipe.run_steps(handle, W, args.lr, lw)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run_steps(handle, K, args.lr, lt)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        finally:
            gc.enable()
        res = [lt]
    else:
        # ONE sequence of chunks — warm-up | timed steps | one spare chunk nobody trains on — through the chunk pipeline: the
        # index work of chunk c + 1 is queued (side stream) before the steps of chunk c and read back after them.  The plan of
        # the first timed chunk is therefore built beside the warm-up, and the timed region carries the index work of the
        # chunks behind its own (the last of them the spare one): K steps' worth of planning for K trained steps, the N = 1
        # arrangement.
        def cut(first, count):
            spans, done = [], 0
            while done < count:
                c = min(chunk, count - done)
                spans.append(((first + done) * B, c))
                done += c
            return spans
        warm, timed, tail = cut(0, W), cut(W, K), cut(W + K, spare)
        spans = warm + timed + tail
        begin = lambda sp: model.plan_chunk_begin(u[sp[0]:sp[0] + sp[1] * B], p[sp[0]:sp[0] + sp[1] * B],
                                                  n[sp[0]:sp[0] + sp[1] * B], B)
        res, nxt = [], begin(spans[0])
        t0 = t1 = None
        gc.disable()            # no cyclic-GC pass of the interpreter inside a sub-millisecond timed region (see bench.py)
        try:
            for i in range(len(warm) + len(timed)):
                if i == len(warm):
                    torch.cuda.synchronize()
                    dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                cp = model.plan_chunk_end(nxt)
                nxt = begin(spans[i + 1])
                out = model.run_chunk(cp, args.lr, global_batch=B * world)
                if i >= len(warm):
                    res.append(out)
            torch.cuda.synchronize()      # the steps and the spare chunk's index work
            t1 = time.perf_counter()
        finally:
            gc.enable()
        model.plan_chunk_end(nxt)         # all ranks consume the spare chunk's read-back alike; it is not trained
    loop = None
    if world == 1 and getattr(args, "loopback_world", 0) > 1:
        loop = loopback_run(args, dev, chunk)
    # every rank started behind the same barrier; the job's time is the MAX over ranks of (own completion - start), which is
    # what a closing barrier would measure without that barrier's own launch + rendezvous latency (~0.1 ms of a 0.7 ms region)
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    model.backend.check(model.U, model.I_ext)
    losses = model.global_losses(torch.cat(res))
    dt = float(dt.item())
    if rank != 0:
        return None
    lv = losses.cpu().numpy()
    assert np.all(np.isfinite(lv)), "non-finite loss"
    value = world * K * B / dt
    lo = W * B
    step_bytes, uu, ui = dedup_step_bytes(u[lo:lo + K * B], p[lo:lo + K * B], n[lo:lo + K * B], B, D)   # per GPU, duplicates counted once
    link_mb = 2 * ui * (world - 1) / world * D * 4 / 1e6
    return {"value": value, "ms_per_step": dt / K * 1e3, "loss_first": float(lv[0]), "loss_last": float(lv[-1]), "loopback": loop,
            "parallelism": "row-sharded tables x%d, RCCL all-to-all of item rows + gradient rows" % world
                           if world > 1 else "row-sharded tables x1: every row local, the single-GPU step stream, no exchange",
            "sampling": "reference rule: negatives uniform over all items (src/models/BaseModel.py:168)",
            "plan_chunk_batches": chunk,
            "exchange": "per step and rank: ~%.1f MB of item rows in and as many gradient-row bytes out over xGMI "
                        "(2 x all_to_all_single), index exchange once per chunk (2 fixed-size all-to-alls)" % link_mb,
            "roofline": {"bound": "hbm", "achieved": step_bytes * K / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": step_bytes * K / dt / 1e9 / 8000.0, "traffic": None,
                         "kernel": "whole sharded step per GPU (exchange-bound for N > 1; see DESIGN.md 6)",
                         "algorithmic_bytes_per_step_per_gpu": step_bytes, "uniq_users_per_step": uu,
                         "uniq_items_per_step": ui,
                         "definition": "2*D*4*(unique users + unique items of the local batch) + 12*B, as at N=1"}}


def loopback_run(args, dev, chunk):
    """one GPU doing rank 0's work of a G-rank job with every exchange replaced by a device copy (ShardedBprmf(loopback=G)):
    what the row-sharded step costs a rank BESIDE the links — routing + request lists + group plan per chunk, and per step
    gather, step on local + received rows, scatter-add of the returned gradient rows"""
    G, B, D, K, W = int(args.loopback_world), args.batch, args.emb, args.steps, args.warmup
    model = ShardedBprmf(args.users, args.items, D, dev, loopback=G)
    model.init_xavier(3407)
    g = torch.Generator(device=dev)
    g.manual_seed(3407 * 31 + G)
    n_trip = (K + W) * B
    u = (torch.randint(0, model.U.shape[0], (n_trip,), generator=g, device=dev) * G).to(torch.int32)
    p = torch.randint(0, args.items, (n_trip,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, args.items, (n_trip,), generator=g, device=dev, dtype=torch.int32)

    def run_range(first, count):
        # as bench_run: the index work of chunk c + 1 is queued before the steps of chunk c and read back after them;
        # t_wait = host time spent waiting for index work (what the side stream did not hide)
        spans, done, t_wait = [], 0, 0.0
        while done < count:
            c = min(chunk, count - done)
            spans.append(((first + done) * B, c))
            done += c
        begin = lambda sp: model.plan_chunk_begin(u[sp[0]:sp[0] + sp[1] * B], p[sp[0]:sp[0] + sp[1] * B],
                                                  n[sp[0]:sp[0] + sp[1] * B], B)
        nxt = begin(spans[0])
        for i in range(len(spans)):
            t0 = time.perf_counter()
            cp = model.plan_chunk_end(nxt)
            t_wait += time.perf_counter() - t0
            nxt = begin(spans[i + 1]) if i + 1 < len(spans) else None
            model.run_chunk(cp, args.lr, global_batch=B * G)
        return t_wait

    run_range(0, max(W, 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_plan = run_range(W, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.backend.check(model.U, model.I_ext)
    return {"virtual_world": G, "us_per_step": dt / K * 1e6, "plan_wait_us_per_step": t_plan / K * 1e6,
            "what": "rank 0's work of a %d-rank job on one GPU, exchanges replaced by device copies (no link traffic)" % G}
