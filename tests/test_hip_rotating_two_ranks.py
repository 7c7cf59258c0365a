"""The stratified (rotating item blocks) schedule with TWO and THREE real ranks on ONE GPU: every rank is a process running the
product's step stream (hip_ops.PipelinedSgd, real kernels on cuda:0), the ring transfers go through torch.distributed over
gloo, staged through the host (RCCL refuses several ranks on one device; on a node every rank has its own GPU and the backend
is RCCL, whose point-to-point calls are ordered with the streams).  A whole
epoch — every item block visits every rank — must equal the single-process oracle on the global batches.  (Threads as ranks
with an in-process transport: tests/test_hip_rotating_loopback.py; the schedule alone on CPU: tests/test_rotating_gloo.py.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from conftest import rel_err
from test_sharded_gloo import _free_port

pytestmark = pytest.mark.gpu


class StagedGlooTransport:
    """rotating.DistTransport's interface over gloo with HOST staging: gloo's point-to-point calls take the device pointer as
    it is (the host reads the card's memory through the PCIe aperture, unordered with the GPU's streams), so every buffer is
    brought to the host behind a stream synchronisation, exchanged as a CPU tensor, and copied back on the current stream —
    the ordering RCCL's stream-ordered P2P gives on a node."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def exchange(self, send_buf, dst, recv_buf, src):
        dist = self.dist
        cur = torch.cuda.current_stream()
        ops, host_recv = [], None
        if send_buf is not None:
            cur.synchronize()
            ops.append(dist.P2POp(dist.isend, send_buf.cpu(), dst))
        if recv_buf is not None:
            host_recv = torch.empty(recv_buf.shape, dtype=recv_buf.dtype)
            ops.append(dist.P2POp(dist.irecv, host_recv, src))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        if host_recv is not None:
            recv_buf.copy_(host_recv)

    def all_gather(self, t):
        torch.cuda.current_stream().synchronize()
        out = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
        self.dist.all_gather(out, t.cpu().contiguous())
        return [o.to(t.device) for o in out]

    def all_reduce_sum(self, t):
        torch.cuda.current_stream().synchronize()
        h = t.cpu()
        self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
        t.copy_(h)
        return t


def _worker(rank, world, port, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from whisprrec_amd import hip_ops
        from whisprrec_amd.rotating import RotatingBprmf
        nU, nI, D, B, lr, parts, chunk = payload["shape"]
        local = hip_ops.PipelinedSgd(chunk)
        local.PLAN_TRIPLETS = 1                       # plans of exactly `chunk` batches: chunk ends inside and across the strata
        m = RotatingBprmf(nU, nI, D, dev, parts=parts, local=local, transport=StagedGlooTransport())
        m.load_full(torch.from_numpy(payload["U"]), torch.from_numpy(payload["I"]))
        t = lambda a: torch.from_numpy(a.astype(np.int32)).to(dev)
        sched = [(t(u), t(p), t(n), pp) for (u, p, n, pp) in payload["strata"][rank]]
        losses = m.run_strata(sched, B, lr)
        assert m.held == rank                         # a full epoch brings every block home
        gl = m.global_losses(losses)
        Uf, If = m.gather_full()
        torch.cuda.synchronize()
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), U=Uf.cpu().numpy(), I=If.cpu().numpy(), loss=gl.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,parts,chunk", [(2, 2, 2), (3, 2, 64)])
def test_rotating_epoch_real_ranks_one_gpu_equals_single_process(tmp_path, world, parts, chunk):
    from whisprrec_amd.sharded import n_local_rows
    rng = np.random.RandomState(world * 10 + parts)
    nU, nI, D, B, lr = 6000, 4000 + world, 64, 2048, 0.3
    steps_per_part = [3] * parts
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)

    def part_range(block, k):
        n = n_local_rows(nI, block, world); per = (n + parts - 1) // parts
        return min(n, k * per), min(n, (k + 1) * per)

    strata = [[None] * world for _ in range(world)]
    glob = [[] for _ in range(world)]
    for rank in range(world):
        n_loc_u = n_local_rows(nU, rank, world)
        for r in range(world):
            held = (rank + r) % world
            us, ps, ns = [], [], []
            for k in range(parts):
                lo, hi = part_range(held, k)
                cnt = steps_per_part[k] * B
                us.append(rng.randint(0, n_loc_u, cnt)); ps.append(rng.randint(lo, hi, cnt)); ns.append(rng.randint(lo, hi, cnt))
            u, p, n = np.concatenate(us), np.concatenate(ps), np.concatenate(ns)
            strata[rank][r] = (u, p, n, steps_per_part)
            glob[r].append((u * world + rank, p * world + held, n * world + held))      # back to global ids
    payload = dict(shape=(nU, nI, D, B, lr, parts, chunk), U=U, I=I, strata=strata)
    mp.spawn(_worker, args=(world, _free_port(), payload, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    Uo, Io = U.copy(), I.copy()
    ref = []
    for r in range(world):
        for k in range(sum(steps_per_part)):
            sl = slice(k * B, (k + 1) * B)
            gu = np.concatenate([glob[r][rank][0][sl] for rank in range(world)])
            gp = np.concatenate([glob[r][rank][1][sl] for rank in range(world)])
            gn = np.concatenate([glob[r][rank][2][sl] for rank in range(world)])
            ref.append(oracle.bprmf_step_sgd(Uo, Io, gu, gp, gn, lr, 0.0))              # ONE step over the global batch
    assert rel_err(got["loss"], np.asarray(ref)) < 1e-5
    assert rel_err(got["U"], Uo) < 1e-5
    assert rel_err(got["I"], Io) < 1e-5
