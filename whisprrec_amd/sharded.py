"""Row-sharded BPRMF step across the GPUs of one node: one process per GPU, RCCL all-to-all over xGMI.

The reference is single-device; this module is new design, not a translation.  Semantics are those
of ONE BaseRunner.fit iteration (reference src/helpers/BaseRunner.py:196-199) over the union of all ranks' batches:
synchronous SGD, every gradient taken from the pre-step tables, loss = mean over the global batch.

Layout
  * both tables are row-sharded cyclically: row r lives on rank r % G at local index r // G (cyclic keeps popular —
    low-id — items spread over all shards; a checkpoint is re-assembled by interleaving, ``gather_full``).
  * interactions are partitioned by user owner, so the user row of every triplet is local and only ITEM rows cross
    links.  Each rank draws its own batches (size B) from its partition; the global batch is their union.
Per chunk of steps (plan time, amortised; index work only — it never depends on table values)
  * per step: unique (owner, local row) item keys of the local batch -> slot ids; the triplets are rewritten to
    (local user row, pos slot, neg slot) and planned with the ordinary BatchPlan;
  * ONE all-to-all of the request lists of all steps of the chunk tells every owner which rows to serve at each
    step; the owner pre-sorts, per step, the rows it will receive gradients for (fixed summation order).
Per step (hot path)
  1. owner: wr_gather_rows of the requested rows            -> all-to-all (rows to requesters)
  2. requester: wr_bprmf_shard_step on (local user shard, received rows): user rows updated in place, one reduced
     gradient row per slot                                   -> all-to-all (gradient rows back to owners)
  3. owner: wr_apply_rows_sorted(alpha = -lr)               (segmented, duplicate rows from several ranks summed)
  No collective carries the loss: per-step partials are all-reduced once per chunk.
Link budget per rank and step: 2 * D*4 * (unique remote items) bytes each way — see DESIGN.md §6.
"""
import math
import time

import numpy as np
import torch
import torch.distributed as dist


def slotify_generic(u_loc, rk_p, rk_n, batch, GM):
    """Per step: unique routed item keys (ascending) and the slot (rank of its key inside the step) of every occurrence.
    Device-agnostic restatement with torch.unique; HipBackend.slotify gets the same result from the batch plan's sort.
    Returns (slot_p, slot_n, uniq_key, uniq_step, nq)."""
    dev = u_loc.device
    N = u_loc.numel()
    nb = (N + batch - 1) // batch
    step_of = torch.arange(N, device=dev) // batch
    ck = torch.cat([step_of * GM + rk_p, step_of * GM + rk_n])
    uniq, inv = torch.unique(ck, return_inverse=True)                      # sorted by (step, routed key)
    step_starts = torch.searchsorted(uniq, torch.arange(nb + 1, device=dev) * GM)
    slots = inv - step_starts[torch.cat([step_of, step_of])]
    return slots[:N], slots[N:], uniq % GM, uniq // GM, step_starts[1:] - step_starts[:-1]


class HipBackend:
    """Local compute through the C-ABI.  (Tests substitute an oracle-backed object with the same methods to exercise
    the exchange logic on CPU/gloo; the product never does.)"""

    def __init__(self):
        from . import abi, hip_ops
        self.abi, self.ops = abi, hip_ops

    def plan_slots(self, u_loc, rk_p, rk_n, batch, n_user_rows, GM):
        """Batch plan + slot ids in one go: the plan is built on the ROUTED item keys (a bijection of item ids, so runs and
        flags are the same), the heads of the sorted runs are the step's unique keys, a prefix sum over the head flags
        gives every occurrence its slot, and tp/tn/oc_item are rewritten from keys to slots.  The sort is the plan
        builder's (hand-written kernels); what is left here is flag/prefix/scatter glue on int32 vectors."""
        dev = u_loc.device
        N, B = u_loc.numel(), batch
        plan = self.ops.BatchPlan(u_loc.to(torch.int32), rk_p.to(torch.int32), rk_n.to(torch.int32), B, n_user_rows, GM,
                                  validate=True)
        nb = plan.n_batches
        oi = plan.oc_item
        pos = torch.arange(2 * N, device=dev)
        bidx = pos // (2 * B)
        head = torch.ones(2 * N, dtype=torch.bool, device=dev)
        head[1:] = oi[1:] != oi[:-1]
        head[bidx * (2 * B) == pos] = True
        csum = torch.cumsum(head.to(torch.int32), 0)
        slot = (csum - csum[bidx * (2 * B)]).to(torch.int32)               # 0-based rank of the run inside its batch
        uniq_key = oi[head].to(torch.int64)
        uniq_step = bidx[head]
        nq = torch.bincount(uniq_step, minlength=nb)
        t_idx = bidx * B + (plan.oc_src >> 1).to(torch.int64)
        neg_side = (plan.oc_src & 1).bool()
        slot_p = torch.empty(N, dtype=torch.int32, device=dev)
        slot_n = torch.empty(N, dtype=torch.int32, device=dev)
        slot_p[t_idx[~neg_side]] = slot[~neg_side]
        slot_n[t_idx[neg_side]] = slot[neg_side]
        flag = torch.tensor(-2 ** 31, dtype=torch.int32, device=dev)
        plan.tp = (plan.tp & flag) | slot_p                                    # keep bit 31 (several occurrences), swap key for slot
        plan.tn = (plan.tn & flag) | slot_n
        plan.oc_item = slot.contiguous()
        return plan, uniq_key, uniq_step, nq

    def gather_rows(self, tab, idx):
        return self.ops.gather_rows(tab, idx)

    def local_step(self, U, item_rows, plan, k, global_batch, lr, grad_slots, loss_out):
        L, ops = self.abi.lib(), self.ops
        off = k * plan.batch_size
        B = plan.batch_len(k)
        nbytes = self.abi.check_size(L.wr_bprmf_step_workspace_bytes(plan.batch_size, U.shape[1]), "workspace")
        ws = ops.workspace(U.device, "step").get(nbytes)
        import ctypes
        hot = plan.hot_struct(k)
        self.abi.check(L.wr_bprmf_shard_step(U.data_ptr(), U.shape[0], item_rows.data_ptr(), item_rows.shape[0], U.shape[1],
                                             plan.tu.data_ptr() + 4 * off, plan.tp.data_ptr() + 4 * off,
                                             plan.tn.data_ptr() + 4 * off, plan.oc_item.data_ptr() + 8 * off,
                                             plan.oc_src.data_ptr() + 8 * off, B, global_batch, lr, grad_slots.data_ptr(),
                                             loss_out.data_ptr(), ctypes.addressof(hot) if hot is not None else None,
                                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                       "wr_bprmf_shard_step")

    def apply_sorted(self, tab, sorted_rows, perm, src, alpha):
        self.abi.check(self.abi.lib().wr_apply_rows_sorted(tab.data_ptr(), tab.shape[0], tab.shape[1], sorted_rows.data_ptr(),
                                                           perm.data_ptr(), src.data_ptr(), sorted_rows.numel(), alpha,
                                                           torch.cuda.current_stream().cuda_stream), "wr_apply_rows_sorted")


def n_local_rows(n_rows, rank, world):
    return (n_rows - rank + world - 1) // world


class ChunkPlan:
    pass


class ShardedBprmf:
    def __init__(self, n_users, n_items, emb_size, device, backend=None, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n_users, self.n_items, self.D = int(n_users), int(n_items), int(emb_size)
        self.device = device
        self.backend = backend if backend is not None else HipBackend()
        self.M = (self.n_items + self.world - 1) // self.world          # slots of the routed key per owner
        self.U = torch.zeros(n_local_rows(self.n_users, self.rank, self.world), self.D, device=device)
        self.I = torch.zeros(n_local_rows(self.n_items, self.rank, self.world), self.D, device=device)

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
    def plan_chunk(self, u, p, n, batch):
        """u, p, n: this rank's triplets of the chunk in batch order (global ids, every u % world == rank).
        All ranks must call this with the same number of steps."""
        G, M, dev = self.world, self.M, u.device
        N = u.numel()
        nb = (N + batch - 1) // batch
        u, p, n = u.to(torch.int64), p.to(torch.int64), n.to(torch.int64)
        GM = G * M
        # routed key of an item: (owner, local row) — a bijection of the item id that makes each owner's rows contiguous
        rk_p, rk_n = (p % G) * M + p // G, (n % G) * M + n // G
        local_plan, u_key, u_step, nq = self.backend.plan_slots(u // G, rk_p, rk_n, batch, self.U.shape[0], GM)
        u_owner, u_row = u_key // M, u_key % M
        req_counts = torch.bincount(u_step * G + u_owner, minlength=nb * G).reshape(nb, G)   # rows requested per owner
        # ---- exchange the request lists of the whole chunk: destination-major, then step
        order = torch.sort(u_owner * nb + u_step, stable=True)[1]
        send_rows = u_row[order]
        send_counts = req_counts.t().contiguous()                              # [G, nb]
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts.view(-1), send_counts.view(-1), group=self.group)   # nb counts per peer
        send_tot = send_counts.sum(1).tolist()
        recv_tot = recv_counts.sum(1).tolist()
        recv_rows = torch.empty(int(sum(recv_tot)), dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv_rows, send_rows, output_split_sizes=recv_tot, input_split_sizes=send_tot, group=self.group)
        # recv_rows is (source, step)-major; per step we serve the concatenation over sources: one stable sort by step
        # regroups the whole chunk, a second one (by (step, row)) fixes every step's summation order
        rc = recv_counts.cpu()                                                 # [G, nb]
        seg_step = torch.arange(nb, device=dev).repeat(G)                      # step of each (source, step) segment
        recv_step = torch.repeat_interleave(seg_step, recv_counts.reshape(-1))
        order1 = torch.sort(recv_step, stable=True)[1]
        serve_all = recv_rows[order1]                                          # step-major, sources in rank order inside a step
        step_sorted = recv_step[order1]
        serve_cnt = rc.sum(0)                                                  # rows served per step
        serve_off = torch.zeros(nb + 1, dtype=torch.int64)
        serve_off[1:] = torch.cumsum(serve_cnt, 0)
        L = max(int(self.I.shape[0]), 1)
        srt_key, order2 = torch.sort(step_sorted * L + serve_all, stable=True)
        apply_rows_all = (srt_key % L).to(torch.int32)
        apply_perm_all = (order2 - serve_off.to(dev)[step_sorted]).to(torch.int32)   # sorting keeps every element in its step
        cp = ChunkPlan()
        cp.nb, cp.batch, cp.N = nb, batch, N
        cp.local = local_plan
        cp.nq = nq.tolist()
        cp.req_splits = req_counts.tolist()                                    # [nb][G] what I receive from owner d
        cp.serve_splits = rc.t().contiguous().tolist()                         # [nb][G] what I send to requester s
        so = serve_off.tolist()
        cp.serve_rows = [serve_all[so[k]:so[k + 1]] for k in range(nb)]
        cp.apply_rows = [apply_rows_all[so[k]:so[k + 1]] for k in range(nb)]
        cp.apply_perm = [apply_perm_all[so[k]:so[k + 1]] for k in range(nb)]
        cp.max_nq = max(cp.nq) if cp.nq else 0
        cp.max_serve = max((r.numel() for r in cp.serve_rows), default=0)
        return cp

    # ------------------------------------------------------------------ hot path
    def run_chunk(self, cp, lr, global_batch=None):
        """Runs the cp.nb steps; returns this rank's per-step loss shares (sum over ranks = global mean loss)."""
        D, dev = self.D, self.device
        losses = torch.zeros(cp.nb, dtype=torch.float32, device=dev)
        recv_rows = torch.empty(max(cp.max_nq, 1), D, device=dev)
        grad_slots = torch.empty(max(cp.max_nq, 1), D, device=dev)
        grad_recv = torch.empty(max(cp.max_serve, 1), D, device=dev)
        for k in range(cp.nb):
            Bk = min(cp.batch, cp.N - k * cp.batch)
            gb = global_batch if global_batch is not None else Bk * self.world
            nq, ns = cp.nq[k], cp.serve_rows[k].numel()
            send = self.backend.gather_rows(self.I, cp.serve_rows[k]) if ns > 0 else grad_recv[:0]
            rr = recv_rows[:nq]
            dist.all_to_all_single(rr, send, output_split_sizes=cp.req_splits[k], input_split_sizes=cp.serve_splits[k],
                                   group=self.group)
            gs = grad_slots[:nq]
            self.backend.local_step(self.U, rr, cp.local, k, gb, lr, gs, losses[k:k + 1])
            gr = grad_recv[:ns]
            dist.all_to_all_single(gr, gs, output_split_sizes=cp.serve_splits[k], input_split_sizes=cp.req_splits[k],
                                   group=self.group)
            if ns > 0:
                self.backend.apply_sorted(self.I, cp.apply_rows[k], cp.apply_perm[k], gr, -lr)
        return losses

    def global_losses(self, local_losses):
        out = local_losses.clone()
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
    n_trip = (K + W) * B
    u = torch.randint(0, model.U.shape[0], (n_trip,), generator=g, device=dev) * world + rank   # users this rank owns
    p = torch.randint(0, args.items, (n_trip,), generator=g, device=dev)
    n = torch.randint(1, args.items, (n_trip,), generator=g, device=dev)

    def run_range(first, count):
        out, done = [], 0
        while done < count:
            c = min(chunk, count - done)
            lo = (first + done) * B
            cp = model.plan_chunk(u[lo:lo + c * B], p[lo:lo + c * B], n[lo:lo + c * B], B)
            out.append(model.run_chunk(cp, args.lr, global_batch=B * world))
            done += c
        return out

    if W > 0:
        run_range(0, W)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc.disable()            # no cyclic-GC pass of the interpreter inside a sub-millisecond timed region (see bench.py)
    t0 = time.perf_counter()
    res = run_range(W, K)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gc.enable()
    # every rank started behind the same barrier; the job's time is the MAX over ranks of (own completion - start), which is
    # what a closing barrier would measure without that barrier's own launch + rendezvous latency (~0.1 ms of a 0.7 ms region)
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    losses = model.global_losses(torch.cat(res))
    dt = float(dt.item())
    if rank != 0:
        return None
    lv = losses.cpu().numpy()
    assert np.all(np.isfinite(lv)), "non-finite loss"
    value = world * K * B / dt
    lo = W * B
    step_bytes, uu, ui = dedup_step_bytes(u[lo:], p[lo:], n[lo:], B, D)      # per GPU, duplicates counted once
    link_mb = 2 * ui * (world - 1) / world * D * 4 / 1e6
    return {"value": value, "ms_per_step": dt / K * 1e3, "loss_first": float(lv[0]), "loss_last": float(lv[-1]),
            "parallelism": "row-sharded tables x%d, RCCL all-to-all of item rows + gradient rows" % world,
            "sampling": "reference rule: negatives uniform over all items (src/models/BaseModel.py:168)",
            "plan_chunk_batches": chunk,
            "exchange": "per step and rank: ~%.1f MB of item rows in and as many gradient-row bytes out over xGMI "
                        "(2 x all_to_all_single), index exchange once per chunk" % link_mb,
            "roofline": {"bound": "hbm", "achieved": step_bytes * K / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": step_bytes * K / dt / 1e9 / 8000.0, "traffic": None,
                         "kernel": "whole sharded step per GPU (exchange-bound for N > 1; see DESIGN.md 6)",
                         "algorithmic_bytes_per_step_per_gpu": step_bytes, "uniq_users_per_step": uu,
                         "uniq_items_per_step": ui,
                         "definition": "2*D*4*(unique users + unique items of the local batch) + 12*B, as at N=1"}}
