"""Stratified multi-GPU BPRMF: user rows fixed per rank, item-row blocks ROTATING around the ring of GPUs.

Why a second multi-GPU mode.  The all-to-all mode of ``sharded.py`` moves two item rows and two gradient rows per triplet
over xGMI (≈450 B each way per triplet at D=64): with ≈50–60 GB/s per point-to-point link that is several times the
HBM time of the step itself.  xGMI is point-to-point, so the MI355X-first layout moves SHARDS, not rows:

  * users are partitioned over the G ranks (owner = u % G) and never move;
  * items are partitioned into G blocks (block = i % G, local index i // G); at sub-epoch r rank g HOLDS block (g + r) % G
    and trains only on the stratum (users of g) x (items of the held block) — positives AND negatives from that block;
  * the G local batches of one step touch disjoint user rows and disjoint item rows, so their union is one exact
    batch-synchronous SGD step over the global batch (same semantics as BaseRunner.fit, reference
    src/helpers/BaseRunner.py:196-199) with NO per-step communication — each rank simply runs the single-GPU fused step;
  * after the stratum the block moves to rank g-1 (ring send/recv over one xGMI link): per epoch every block visits every
    rank once, (G-1)/G of the item table crosses each link once per epoch instead of ≈1 KB per triplet.
This is the stratified SGD schedule of DSGD (Gemulla et al., KDD 2011) applied to BPR triplets.  What changes with
respect to the reference's sampling: the negative of a triplet is drawn from the positive's block (a fixed pseudo-random
1/G subset of the items, i % G) instead of from all items, and batches are drawn per stratum.

Overlap.  Each block is split into ``parts`` contiguous row ranges; a stratum is trained part by part and a part is sent on
a side stream as soon as its last step is enqueued, while the next part trains; an incoming part is waited for right before
its first step of the next stratum.  With 2 parts every transfer (one half of a block) runs beside the compute of a half.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from .sharded import n_local_rows


class DistTransport:
    """The collective side of the schedule over torch.distributed (backend nccl = RCCL on the GPUs, gloo in the CPU tests).
    Kept behind this small interface so that tests can run several virtual ranks on ONE GPU (threads + in-process copies,
    tests/test_hip_rotating_loopback.py) and exercise the stream / event ordering of the rotation for real."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def exchange(self, send_buf, dst, recv_buf, src):
        """send_buf -> dst and recv_buf <- src (either may be None) as one grouped P2P batch; returns when the CURRENT
        stream has the transfers ordered before whatever is enqueued next (no host wait on the GPU)"""
        ops = []
        if send_buf is not None:
            ops.append(dist.P2POp(dist.isend, send_buf, dst, group=self.group))
        if recv_buf is not None:
            ops.append(dist.P2POp(dist.irecv, recv_buf, src, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def all_gather(self, t):
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t.contiguous(), group=self.group)
        return out

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def HipLocal(chunk=64):
    """local training of one stratum = the single-GPU pipelined run (hip_ops.PipelinedSgd)"""
    from . import hip_ops
    return hip_ops.PipelinedSgd(chunk)


class RotatingBprmf:
    def __init__(self, n_users, n_items, emb_size, device, parts=2, local=None, group=None, transport=None):
        self.tx = transport if transport is not None else DistTransport(group)
        self.rank, self.world = self.tx.rank, self.tx.world
        self.n_users, self.n_items, self.D = int(n_users), int(n_items), int(emb_size)
        self.device, self.parts = device, max(1, int(parts))
        self.local = local if local is not None else HipLocal()
        self.U = torch.zeros(n_local_rows(self.n_users, self.rank, self.world), self.D, device=device)
        self.cap = (self.n_items + self.world - 1) // self.world          # rows of the largest item block
        self.held = self.rank                                               # block currently held
        self.I = torch.zeros(self.cap, self.D, device=device)              # held block (rows beyond its size unused)
        self.I_in = torch.zeros(self.cap, self.D, device=device)           # landing buffer for the next block
        from .hip_ops import side_stream
        self.comm_stream = side_stream(device) if device.type == "cuda" else None
        self._incoming = {}      # part -> event: that part of the NEXT block has landed in I_in
        self._ready = {}         # part -> event still to be waited for before the held block's part is touched
        self._deferred = None    # part of the held block whose hand-over was left to the next call (run_strata defer_last)

    # ------------------------------------------------------------------ layout helpers
    def block_rows(self, block):
        return n_local_rows(self.n_items, block, self.world)

    def part_range(self, block, part):
        """contiguous local-row range [lo, hi) of part `part` of `block`"""
        n = self.block_rows(block)
        per = (n + self.parts - 1) // self.parts
        return min(n, part * per), min(n, (part + 1) * per)

    def load_full(self, U_full, I_full):
        self.U.copy_(U_full[self.rank::self.world].to(self.device))
        blk = I_full[self.held::self.world].to(self.device)
        self.I[:blk.shape[0]].copy_(blk)

    def init_xavier(self, seed):
        g = torch.Generator(device=self.device)
        g.manual_seed(seed * 1000 + self.rank)
        self.U.normal_(0.0, math.sqrt(2.0 / (self.n_users + self.D)), generator=g)
        self.I.normal_(0.0, math.sqrt(2.0 / (self.n_items + self.D)), generator=g)

    def gather_full(self):
        """Re-assemble the reference's checkpoint layout (user table, item table) on every rank."""
        self.complete_rotation()
        self._drain()
        self.check_steps()       # a checkpoint never holds tables a step launch has declared invalid
        G, D = self.world, self.D
        capu = (self.n_users + G - 1) // G
        pu = torch.zeros(capu, D, device=self.device)
        pu[:self.U.shape[0]] = self.U
        us = self.tx.all_gather(pu)
        U_full = torch.stack(us, dim=1).reshape(capu * G, D)[:self.n_users]
        hs = self.tx.all_gather(torch.tensor([self.held], device=self.device))
        bs = self.tx.all_gather(self.I)
        by_block = [None] * G
        for r in range(G):
            by_block[int(hs[r].item())] = bs[r]
        I_full = torch.stack(by_block, dim=1).reshape(self.cap * G, D)[:self.n_items]
        return U_full, I_full

    # ------------------------------------------------------------------ ring rotation
    def _send_part(self, part):
        """enqueue: part `part` of the held block -> rank-1, same part of the next block <- rank+1 (side stream)"""
        G = self.world
        if G == 1:
            return
        lo, hi = self.part_range(self.held, part)
        nxt = (self.held + 1) % G
        lo2, hi2 = self.part_range(nxt, part)
        dst, src = (self.rank - 1) % G, (self.rank + 1) % G
        send_buf = self.I[lo:hi] if hi > lo else None
        recv_buf = self.I_in[lo2:hi2] if hi2 > lo2 else None
        if send_buf is None and recv_buf is None:
            return
        if self.comm_stream is not None:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))          # the part's last step is enqueued
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(done)
                self.tx.exchange(send_buf, dst, recv_buf, src)            # comm stream ordered behind the transfer
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
            self._incoming[part] = ev
        else:
            self.tx.exchange(send_buf, dst, recv_buf, src)

    def _finish_rotation(self):
        """every part of the held block has been handed on and every part of the next one is on its way: the landing buffer
        becomes the held block.  Nothing is waited for here — a part is waited for right before it is first touched
        (`_await_part`), so the transfer of a stratum's LAST part hides behind the next stratum's first part."""
        if self.world == 1:
            return
        self._drain()                                   # parts of the outgoing block nobody waited for (no steps on them)
        self.I, self.I_in = self.I_in, self.I
        self.held = (self.held + 1) % self.world
        self._ready, self._incoming = self._incoming, {}

    def _await_part(self, part):
        ev = self._ready.pop(part, None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def _drain(self):
        for part in list(self._ready):
            self._await_part(part)

    def check_steps(self):
        """raises if a bounded wait inside a step launch ever expired on this device (hip_ops.BprmfTables.check_chain;
        synchronises) — called before the tables leave this object (gather_full) and at the end of a benchmark"""
        if self.U.device.type == "cuda":
            from . import hip_ops
            hip_ops.BprmfTables(self.U, self.I[:1] if self.I.shape[0] else self.I).check_chain()

    # ------------------------------------------------------------------ training
    def run_subepoch(self, u, p, n, steps_per_part, batch, lr):
        """One stratum: u, p, n are LOCAL indices in batch order (u rows of the user shard, p and n rows of the held block);
        the first steps_per_part[0] batches use items of part 0 only, the next steps_per_part[1] batches items of part 1, ...
        As soon as the steps of a part are enqueued the part is handed to the ring.  Every rank must pass the same
        steps_per_part.  Returns the per-step local losses (mean over ranks = loss of the global batch)."""
        return self.run_strata([(u, p, n, steps_per_part)], batch, lr)

    def complete_rotation(self):
        """hand over the part a previous run_strata(defer_last=True) kept back — and finish the rotation if it was the
        stratum's last part"""
        if self._deferred is not None:
            (k, stratum_done), self._deferred = self._deferred, None
            self._send_part(k)
            if stratum_done:
                self._finish_rotation()

    def _rotation_pending(self):
        return self._deferred is not None and self._deferred[1] and self.world > 1

    def run_strata(self, strata, batch, lr, part_relative=False, defer_last=False):
        """Consecutive strata [(u, p, n, steps_per_part), ...] (see run_subepoch), one block rotation after each.  All
        their batches go to the local runner as ONE list of segments, so its plan pipeline runs across the rotations: the
        first plan of stratum r+1 is built while stratum r trains (a plan needs the indices only, not the block that is
        still on its way).  With 8 GPUs a stratum is a few dozen steps: an exposed plan build per stratum would cost a
        third of it.  part_relative=True: p and n already count rows from the start of their part (no offset pass).
        defer_last=True: the LAST part of the last stratum is not handed to the ring here but at the start of the next call
        (or by complete_rotation / gather_full) — in a long run the hand-over of a stratum's last part hides behind the next
        stratum's first part; a caller that cuts the run into pieces (bench.py: warm-up, timed steps) keeps that overlap
        across its pieces this way instead of ending each piece on an exposed transfer."""
        return self.run_prepared(self.prepare(strata, batch, part_relative), lr, defer_last=defer_last)

    def prepare(self, strata, batch, part_relative=False):
        """Index-side preparation of run_strata — segments, and the local runner's plan handle with the plan of the first
        chunk already queued (a plan depends on the indices only).  Does not touch the ring: a hand-over left pending by
        the previous call (defer_last) is accounted for (the strata start on the block that hand-over brings in) and is
        carried out by run_prepared.  Lets a caller build the first plan of a piece of work outside that piece's timed
        region, the way every later plan is built beside steps."""
        B = int(batch)
        pending = self._rotation_pending()
        # stratum r trains on bufs[r % 2] (every rotation swaps I and I_in; a pending one swaps them before stratum 0)
        bufs = (self.I_in, self.I) if pending else (self.I, self.I_in)
        held = (self.held + 1) % self.world if pending else self.held
        segments, counts, ends = [], [], []
        r = 0                                            # strata completed so far in this call (buffer parity)
        for stratum in strata:
            u, p, n, steps_per_part = stratum[:4]
            # optional 5th element: per part, whether its steps END in this entry (default: every part does).  A part that
            # goes on is not handed over yet; an entry that leaves its stratum open (last part not ended) is followed, if at
            # all, by the entry that continues the SAME stratum (a run cut into pieces: bench.py's timed piece + spare piece)
            part_ends = [bool(e) for e in stratum[4]] if len(stratum) > 4 else [True] * self.parts
            if len(part_ends) != self.parts:
                raise ValueError("one end flag per part")
            per_part = [int(steps_per_part[k]) if k < len(steps_per_part) else 0 for k in range(self.parts)]
            table = bufs[r % 2] if self.world > 1 else self.I
            first = 0
            for k, st in enumerate(per_part):
                lo, hi = self.part_range(held, k)
                sl = slice(first * B, min(u.numel(), (first + st) * B))
                segments.append((table[lo:hi], u[sl], p[sl] if part_relative else p[sl] - lo,
                                 n[sl] if part_relative else n[sl] - lo))
                first += st
            counts.append(per_part)
            ends.append(part_ends)
            if part_ends[-1]:                            # the stratum is complete: the next entry trains on the next block
                held = (held + 1) % self.world
                r += 1
        handle = self.local.plan(self.U, segments, B)
        return {"handle": handle, "counts": counts, "ends": ends, "B": B, "pending": pending}

    def run_prepared(self, prep, lr, n_strata=None, defer_last=False):
        """the steps of the first n_strata strata (default: all) of a prepare()d schedule; see run_strata"""
        if prep["pending"] != self._rotation_pending():
            raise RuntimeError("the ring's state changed between prepare() and run_prepared()")
        self.complete_rotation()
        # the global batch is the union of the G local batches: its mean loss has 1/(G*B) coefficients.  The local kernels
        # use 1/B, and the local batches touch disjoint rows, so dividing the learning rate by G gives the same update.
        lr = lr / self.world
        counts = prep["counts"] if n_strata is None else prep["counts"][:n_strata]
        ends = prep["ends"][:len(counts)]
        handle = prep["handle"]
        losses = torch.zeros(sum(sum(c) for c in counts), dtype=torch.float32, device=self.device)
        first, seg = 0, 0
        open_stratum = False
        # defer_last: the LAST hand-over of the call — a part whose steps end the call — is left to the next call
        tail = None
        if defer_last and self.world > 1 and counts:
            done = [k for k in range(self.parts) if ends[-1][k]]
            if done and all(counts[-1][k2] == 0 for k2 in range(done[-1] + 1, self.parts)):
                tail = done[-1]
        for si, (per_part, part_ends) in enumerate(zip(counts, ends)):
            last_stratum = si == len(counts) - 1
            for k, st in enumerate(per_part):
                if st > 0:
                    self._await_part(k)                 # the held block's part k has landed (it came in during the last stratum)
                    self.local.run(handle, seg, lr, losses[first:first + st])
                    first += st
                seg += 1
                if not part_ends[k]:
                    continue                            # the part goes on in the next call: not handed over yet
                if last_stratum and k == tail:
                    self._deferred = (k, bool(part_ends[-1]) and k == self.parts - 1)   # handed over at the start of the next call
                else:
                    self._send_part(k)
            if not part_ends[-1]:
                open_stratum = True                     # the stratum goes on in the next call: no rotation yet
            elif self._deferred is None:
                self._finish_rotation()
        if self._deferred is None and not open_stratum:
            self._drain()                               # leave with the held block complete
        return losses

    def global_losses(self, local_losses):
        return self.tx.all_reduce_sum(local_losses.clone()) / self.world


def stratum_steps(n_interactions, world, batch):
    """steps per (rank, sub-epoch) stratum when an epoch of n_interactions is spread over world^2 strata"""
    return max(1, int(round(n_interactions / float(world * world * batch))))


# ---------------------------------------------------------------------------------------------------- bench (N > 1)
SAMPLING_NOTE = ("stratified: a triplet's negative is drawn uniformly from the PART of the item block its positive lies in (a "
                 "block = a fixed pseudo-random 1/G of the items, i % G; a part = one of `parts` contiguous row ranges of it) "
                 "and batches are drawn per (user shard, item part) stratum; the "
                 "reference draws negatives from ALL items (src/models/BaseModel.py:168) — waived in this mode, kept in "
                 "mode 'alltoall'; the arithmetic of a step (batch-synchronous SGD on the global batch) is unchanged")


def dedup_step_bytes(u, p, n, B, D, steps=8):
    """algorithmic bytes of one local step, in-batch duplicates counted once (the N=1 definition):
    2*D*4*(unique users + unique items) + 12*B, averaged over the first `steps` whole batches of (u, p, n)"""
    nb = min(int(steps), u.numel() // B)
    if nb <= 0:
        return float((6 * D * 4 + 12) * B), float(B), float(2 * B)
    uu = ui = 0
    for k in range(nb):
        sl = slice(k * B, (k + 1) * B)
        uu += torch.unique(u[sl]).numel()
        ui += torch.unique(torch.cat([p[sl], n[sl]])).numel()
    uu, ui = uu / nb, ui / nb
    return 2.0 * D * 4 * (uu + ui) + 12.0 * B, uu, ui


def bench_run(args, rank, world, dev):
    """bench.py --gpus N (N > 1), mode 'rotate': weak scaling, batch args.batch per GPU, stratified schedule.  The block
    rotation happens as often as a full epoch of args.interactions interactions would require (every stratum_steps steps),
    inside the timed region.  The process group exists already; returns the result dict on rank 0."""
    import time
    B, D, K, W = args.batch, args.emb, args.steps, args.warmup
    S = stratum_steps(args.interactions, world, B)
    from .hip_ops import PipelinedSgd
    chunk = args.chunk if args.chunk > 0 else max(1, min(64, K))
    model = RotatingBprmf(args.users, args.items, D, dev, parts=args.parts, local=PipelinedSgd(chunk, min_triplets=1))
    model.init_xavier(3407)
    g = torch.Generator(device=dev)
    g.manual_seed(3407 * 7919 + rank)

    part_len = [S // model.parts + (1 if k < S % model.parts else 0) for k in range(model.parts)]   # steps of a part
    part_end = np.cumsum(part_len)                                                               # offsets inside a stratum

    def make_pieces(g0, count):
        """The strata pieces of global steps [g0, g0 + count) of ONE continuous run: stratum s = steps [s S, (s+1) S) trains
        on block (rank + s) % G, part k of it on steps [part_end[k] - part_len[k], part_end[k]) of the stratum.  A part is
        handed to the ring in the piece that contains its last step, the block rotates in the piece that contains the
        stratum's last step — so a window shorter than a part (the driver's 20 steps at G = 2: a part lasts 190) carries no
        hand-over, as in the long run it is cut from, instead of a whole block per window.  Indices are generated BEFORE
        the timed region like the N=1 bench: uniform users of this rank, positives and negatives uniform inside the part
        (item 0, which the reference never draws as a negative, is local row 0 of block 0); they count rows from the
        start of their part."""
        pieces, gs, g1 = [], g0, g0 + count
        while gs < g1:
            s_idx, o0 = gs // S, gs % S
            o1 = min(S, o0 + (g1 - gs))
            held = (rank + s_idx) % world
            per_part, ends, cols = [], [], []
            for k in range(model.parts):
                a, b = int(part_end[k] - part_len[k]), int(part_end[k])
                st = max(0, min(o1, b) - max(o0, a))
                per_part.append(st)
                ends.append(bool(o0 < b <= o1) if part_len[k] > 0 else bool(o1 == S))
                if st:
                    lo, hi = model.part_range(held, k)
                    width = max(hi - lo, 1)
                    cnt = st * B
                    uu = torch.randint(0, model.U.shape[0], (cnt,), generator=g_idx, device=dev, dtype=torch.int32)
                    pp = torch.randint(0, width, (cnt,), generator=g_idx, device=dev, dtype=torch.int32)
                    n0 = 1 if (held == 0 and lo == 0 and width > 1) else 0
                    nn_ = torch.randint(n0, width, (cnt,), generator=g_idx, device=dev, dtype=torch.int32)
                    cols.append((uu, pp, nn_))
            if cols:
                u_c, p_c, n_c = (torch.cat([c[j] for c in cols]) for j in range(3))
                pieces.append((u_c, p_c, n_c, per_part, ends))
            gs += o1 - o0
        return pieces

    g_idx = g
    w1 = W - W // 2
    warm = make_pieces(0, w1) if W > 0 else []
    warm2 = make_pieces(w1, W - w1) if W >= 2 else None
    timed = make_pieces(W, K)
    spare = make_pieces(W + K, min(chunk, S))
    # Same shape as the N=1 bench.  The hand-over of a piece's last part (when the piece ends on a part's last step) is left
    # to the next piece (defer_last): it then runs beside that piece's first part, as it would beside the next part in one
    # long run.  The timed piece's index work is prepare()d before the clock starts (its first plan is built outside the
    # region, like every plan but a run's first is built beside steps), with a spare piece of `chunk` steps behind it that is
    # planned — inside the region, beside the last timed steps — but never trained: K steps' worth of plan builds between the
    # two timestamps, K steps trained.
    torch.cuda.synchronize()
    if W > 0:
        model.run_strata(warm, B, args.lr, part_relative=True, defer_last=True)
        if warm2:       # in two pieces when it can be: every host path of a piece has then run twice before the clock starts
            model.run_strata(warm2, B, args.lr, part_relative=True, defer_last=True)
    prepared = model.prepare(timed + spare, B, part_relative=True)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    import gc
    gc.disable()            # no cyclic-GC pass of the interpreter inside a sub-millisecond timed region (see bench.py)
    try:
        t0 = time.perf_counter()
        local_losses = model.run_prepared(prepared, args.lr, n_strata=len(timed), defer_last=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    finally:
        gc.enable()
    # every rank started behind the same barrier; the job's time is the MAX over ranks of (own completion - start), which is
    # what a closing barrier would measure without that barrier's own launch + rendezvous latency (~0.1 ms of a 0.7 ms region)
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    model.complete_rotation()
    model._drain()
    model.check_steps()
    losses = model.global_losses(local_losses)
    dt = float(dt.item())
    if rank != 0:
        return None
    lv = losses.cpu().numpy()
    assert np.all(np.isfinite(lv)), "non-finite loss"
    value = world * K * B / dt
    step_bytes, uu, ui = dedup_step_bytes(timed[0][0], timed[0][1], timed[0][2], B, D)   # per GPU, duplicates counted once
    block_mb = model.cap * D * 4 / 1e6
    work_mb = (model.U.shape[0] * D * 4 + model.cap * D * 4) / 1e6
    return {"value": value, "ms_per_step": dt / K * 1e3, "loss_first": float(lv[0]), "loss_last": float(lv[-1]),
            "parallelism": "stratified-rotation x%d" % world, "sampling": SAMPLING_NOTE,
            "steps_per_stratum": S, "block_MB": block_mb, "parts": model.parts, "plan_chunk_batches": chunk,
            "exchange": "one item block (%.0f MB) per rank moves to its ring neighbour every %d steps "
                        "(batch_isend_irecv, 1 xGMI link per rank), no per-step collective" % (block_mb, S),
            "roofline": {"bound": "hbm", "achieved": step_bytes * K / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                         "frac": step_bytes * K / dt / 1e9 / 8000.0, "traffic": None,
                         "kernel": "whole step per GPU incl. plan build and block rotation (per-kernel numbers: N=1 run)",
                         "algorithmic_bytes_per_step_per_gpu": step_bytes, "uniq_users_per_step": uu,
                         "uniq_items_per_step": ui,
                         "definition": "2*D*4*(unique users + unique items of the local batch) + 12*B, as at N=1",
                         "working_set_MB_per_gpu": work_mb,
                         "note": "a rank trains on its user shard and ONE item block: %.0f MB here — %s the 256 MB Infinity "
                                 "Cache, so this fraction is %s comparable with the HBM-bound N=1 figure"
                                 % (work_mb, "inside" if work_mb < 256 else "beyond", "NOT" if work_mb < 256 else "")}}
