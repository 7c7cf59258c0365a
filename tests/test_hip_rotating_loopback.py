"""The stratified multi-GPU schedule with several VIRTUAL ranks on one GPU: every rank is a thread with its own streams,
the ring transfers are in-process device copies ordered by events exactly like the RCCL P2P calls (the receiver's stream
waits for the sender's data, the sender's stream waits for the receiver's copy).  Unlike the CPU/gloo test this runs the real
kernels asynchronously, so a missing wait in the rotation (a part trained before it landed, a part overwritten before it was
sent) shows up as a mismatch with the single-process oracle on the GLOBAL batches."""
import queue
import threading

import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


class _Shared:
    def __init__(self, world):
        self.world = world
        self.q = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.ack = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
        self.barrier = threading.Barrier(world, timeout=120)
        self.slots = [None] * world


class LoopbackTransport:
    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def exchange(self, send_buf, dst, recv_buf, src):
        cur = torch.cuda.current_stream()
        if send_buf is not None:
            ev = torch.cuda.Event(); ev.record(cur)
            self.sh.q[(self.rank, dst)].put((send_buf, ev))
        if recv_buf is not None:
            buf, ev = self.sh.q[(src, self.rank)].get(timeout=120)
            cur.wait_event(ev)
            torch.cuda._sleep(40_000_000)     # a slow link (~20 ms): whoever touches the part without waiting reads stale rows
            recv_buf.copy_(buf)
            ack = torch.cuda.Event(); ack.record(cur)
            self.sh.ack[(src, self.rank)].put(ack)
        if send_buf is not None:
            cur.wait_event(self.sh.ack[(self.rank, dst)].get(timeout=120))   # send buffer free from here on (stream order)

    def all_gather(self, t):
        torch.cuda.current_stream().synchronize()
        self.sh.slots[self.rank] = t.clone()
        self.sh.barrier.wait()
        out = [self.sh.slots[r].clone() for r in range(self.world)]
        self.sh.barrier.wait()
        return out

    def all_reduce_sum(self, t):
        return torch.stack(self.all_gather(t)).sum(0)


@pytest.mark.parametrize("world,parts,chunk", [(2, 2, 2), (3, 2, 64), (4, 3, 1)])
def test_virtual_ranks_epoch_equals_single_process(world, parts, chunk):
    from whisprrec_amd import hip_ops
    from whisprrec_amd.rotating import RotatingBprmf
    from whisprrec_amd.sharded import n_local_rows
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(world * 7 + parts)
    nU, nI, D, B, lr, epochs = 6000, 4000 + world, 64, 2048, 0.3, 2
    steps_per_part = [3] * parts
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)

    def part_range(block, k):
        n = n_local_rows(nI, block, world); per = (n + parts - 1) // parts
        return min(n, k * per), min(n, (k + 1) * per)

    n_strata = world * epochs
    strata = [[None] * n_strata for _ in range(world)]
    glob = [[] for _ in range(n_strata)]
    for rank in range(world):
        n_loc_u = n_local_rows(nU, rank, world)
        for r in range(n_strata):
            held = (rank + r) % world
            us, ps, ns = [], [], []
            for k in range(parts):
                lo, hi = part_range(held, k)
                cnt = steps_per_part[k] * B
                us.append(rng.randint(0, n_loc_u, cnt)); ps.append(rng.randint(lo, hi, cnt)); ns.append(rng.randint(lo, hi, cnt))
            u, p, n = np.concatenate(us), np.concatenate(ps), np.concatenate(ns)
            strata[rank][r] = (u, p, n)
            glob[r].append((u * world + rank, p * world + held, n * world + held))
    shared = _Shared(world)
    results, errors = [None] * world, []

    def worker(rank):
        try:
            main = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(main):
                local = hip_ops.PipelinedSgd(chunk)
                local.PLAN_TRIPLETS = 1      # plans of exactly `chunk` batches: chunk ends fall inside and across the strata
                m = RotatingBprmf(nU, nI, D, dev, parts=parts, local=local, transport=LoopbackTransport(shared, rank))
                m.load_full(torch.from_numpy(U), torch.from_numpy(I))
                t = lambda a: torch.from_numpy(a.astype(np.int32)).to(dev)
                sched = [(t(u), t(p), t(n), steps_per_part) for (u, p, n) in strata[rank]]
                losses = m.run_strata(sched, B, lr)
                gl = m.global_losses(losses)
                Uf, If = m.gather_full()
                main.synchronize()
                results[rank] = (gl.cpu().numpy(), Uf.cpu().numpy(), If.cpu().numpy(), m.held)
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))
            try:
                shared.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    assert all(not th.is_alive() for th in threads), "a virtual rank hangs"
    Uo, Io = U.copy(), I.copy()
    ref = []
    nsteps = sum(steps_per_part)
    for r in range(n_strata):
        for k in range(nsteps):
            sl = slice(k * B, (k + 1) * B)
            gu = np.concatenate([glob[r][rank][0][sl] for rank in range(world)])
            gp = np.concatenate([glob[r][rank][1][sl] for rank in range(world)])
            gn = np.concatenate([glob[r][rank][2][sl] for rank in range(world)])
            ref.append(oracle.bprmf_step_sgd(Uo, Io, gu, gp, gn, lr, 0.0))
    for rank in range(world):
        gl, Uf, If, held = results[rank]
        assert held == rank
        assert rel_err(gl, np.asarray(ref)) < 1e-5
        assert rel_err(Uf, Uo) < 1e-5 and rel_err(If, Io) < 1e-5
