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
import os
import time

import numpy as np
import torch
import torch.distributed as dist


class HipBackend:
    """Local compute through the C-ABI.  (Tests substitute an oracle-backed object with the same methods to exercise
    the exchange logic on CPU/gloo; the product never does.)"""

    def __init__(self):
        from . import abi, hip_ops
        self.abi, self.ops = abi, hip_ops

    def prepare_chunk(self, u_loc, slot_p, slot_n, batch, n_user_rows, max_slots):
        return self.ops.BatchPlan(u_loc.to(torch.int32), slot_p.to(torch.int32), slot_n.to(torch.int32), batch, n_user_rows,
                                  max_slots, validate=True)

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
        step_of = torch.arange(N, device=dev) // batch
        # routed key of an item: (owner, local row); composite with the step so one unique() serves the whole chunk
        def routed(x):
            return (x % G) * M + x // G
        GM = G * M
        ck = torch.cat([step_of * GM + routed(p), step_of * GM + routed(n)])
        uniq, inv = torch.unique(ck, return_inverse=True)                      # sorted by (step, owner, local row)
        step_starts = torch.searchsorted(uniq, torch.arange(nb + 1, device=dev) * GM)   # first uniq index of each step
        slots = inv - step_starts[torch.cat([step_of, step_of])]
        slot_p, slot_n = slots[:N], slots[N:]
        nq = (step_starts[1:] - step_starts[:-1])                              # unique items per step
        bounds = (torch.arange(nb, device=dev)[:, None] * GM + torch.arange(G + 1, device=dev)[None, :] * M).reshape(-1)
        cuts = torch.searchsorted(uniq, bounds).reshape(nb, G + 1)
        req_counts = (cuts[:, 1:] - cuts[:, :-1])                              # [nb, G] rows requested from each owner
        # ---- exchange the request lists of the whole chunk: destination-major, then step
        u_owner = (uniq % GM) // M
        u_step = uniq // GM
        u_row = (uniq % GM) % M
        order = torch.sort(u_owner * nb + u_step, stable=True)[1]
        send_rows = u_row[order]
        send_counts = req_counts.t().contiguous()                              # [G, nb]
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts.view(-1), send_counts.view(-1), group=self.group)   # nb counts per peer
        send_tot = send_counts.sum(1).tolist()
        recv_tot = recv_counts.sum(1).tolist()
        recv_rows = torch.empty(int(sum(recv_tot)), dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv_rows, send_rows, output_split_sizes=recv_tot, input_split_sizes=send_tot, group=self.group)
        # recv_rows is (source, step)-major; per step we serve the concatenation over sources
        rc = recv_counts.cpu()                                                 # [G, nb]
        src_off = torch.zeros(G, nb + 1, dtype=torch.int64)
        src_off[:, 1:] = torch.cumsum(rc, 1)
        base = torch.zeros(G, dtype=torch.int64)
        base[1:] = torch.cumsum(rc.sum(1), 0)[:-1]
        cp = ChunkPlan()
        cp.nb, cp.batch, cp.N = nb, batch, N
        cp.local = self.backend.prepare_chunk(u // G, slot_p, slot_n, batch, self.U.shape[0], int(nq.max().item()))
        cp.nq = nq.tolist()
        cp.req_splits = req_counts.tolist()                                    # [nb][G] what I receive from owner d
        cp.serve_splits = rc.t().contiguous().tolist()                         # [nb][G] what I send to requester s
        cp.serve_rows, cp.apply_rows, cp.apply_perm = [], [], []
        for k in range(nb):
            segs = [recv_rows[int(base[s] + src_off[s, k]): int(base[s] + src_off[s, k + 1])] for s in range(G)]
            rows_k = torch.cat(segs) if segs else recv_rows[:0]
            cp.serve_rows.append(rows_k)
            srt, perm = torch.sort(rows_k, stable=True)
            cp.apply_rows.append(srt.to(torch.int32))
            cp.apply_perm.append(perm.to(torch.int32))
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
def bench_main(args, rank, world, local_rank):
    """bench.py --gpus N (N > 1), launched by torch.distributed.run with one rank per GPU.  Weak scaling: every rank runs
    batches of args.batch triplets whose users it owns; tables are row-sharded over the N GPUs."""
    import json
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    B, D, K, W = args.batch, args.emb, args.steps, args.warmup
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
            c = min(args.chunk, count - done)
            lo = (first + done) * B
            cp = model.plan_chunk(u[lo:lo + c * B], p[lo:lo + c * B], n[lo:lo + c * B], B)
            out.append(model.run_chunk(cp, args.lr, global_batch=B * world))
            done += c
        return out

    run_range(0, W)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_range(W, K)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    losses = model.global_losses(torch.cat(res))
    dt = float(dt.item())
    if rank == 0:
        lv = losses.cpu().numpy()
        assert np.all(np.isfinite(lv)), "non-finite loss"
        value = world * K * B / dt
        step_bytes = (6 * D * 4 + 12) * B * world          # upper bound (no in-batch duplicates), all GPUs
        out = {"metric": "BPR training triplets/sec", "value": value, "unit": "triplets/s", "n_gpus": world, "steps": K,
               "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BPRMF emb_size=%d, synthetic %d users x %d items (uniform ids), batch %d per GPU "
                                      "(global %d), SGD l2=0, tables row-sharded over %d GPUs, RCCL all-to-all of item "
                                      "rows and gradient rows, plan build in timed region" %
                                      (D, args.users, args.items, B, B * world, world),
                          "batch_per_gpu": B, "global_batch": B * world, "emb_size": D, "optimizer": "SGD", "l2": 0.0,
                          "lr": args.lr, "plan_chunk_batches": args.chunk, "tables": "row-sharded, cyclic"},
               "loss_first": float(lv[0]), "loss_last": float(lv[-1]),
               "roofline": {"bound": "hbm", "achieved": step_bytes * K / dt / 1e9 / world, "peak": 8000.0, "unit": "GB/s",
                            "frac": step_bytes * K / dt / 1e9 / world / 8000.0, "traffic": None,
                            "kernel": "whole sharded step per GPU (exchange-bound; see DESIGN.md §6)"}}
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()
