"""world_size-2 (and 3) CPU/gloo test of the row-sharded step's EXCHANGE LOGIC (whisprrec_amd/sharded.py): routing of item
rows to owners, all-to-all of rows and gradient rows, owner-side segmented application, loss shares.  The local
arithmetic is injected from the CPU oracle (tests may do that; the product's backend is the HIP library).  The result
must equal the single-process oracle run on the union of the ranks' batches."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import rel_err


class OracleBackend:
    class _Plan:
        pass

    def plan_slots(self, u_loc, rk_p, rk_n, batch, n_user_rows, GM):
        from whisprrec_amd.sharded import slotify_generic
        slot_p, slot_n, u_key, u_step, nq = slotify_generic(u_loc, rk_p, rk_n, batch, GM)
        pl = self._Plan()
        pl.u, pl.p, pl.n, pl.batch_size = u_loc.numpy(), slot_p.numpy(), slot_n.numpy(), batch
        assert pl.u.max() < n_user_rows
        return pl, u_key, u_step, nq

    def gather_rows(self, tab, idx):
        return torch.from_numpy(oracle.gather_rows(tab.numpy(), idx.numpy()))

    def local_step(self, U, item_rows, plan, k, global_batch, lr, grad_slots, loss_out):
        lo = k * plan.batch_size
        u, p, n = plan.u[lo:lo + plan.batch_size], plan.p[lo:lo + plan.batch_size], plan.n[lo:lo + plan.batch_size]
        B = len(u)
        Un, R = U.numpy(), item_rows.numpy()
        _, _, coef, loss = oracle.bpr_fwd(Un, R, u, p, n)
        coef = coef.astype(np.float64) * B / global_batch            # mean over the GLOBAL batch
        loss_out[0] = loss * B / global_batch
        gU = np.zeros(Un.shape, np.float64)
        gS = np.zeros(R.shape, np.float64)
        np.add.at(gU, u, coef[:, None] * (R[p].astype(np.float64) - R[n]))
        np.add.at(gS, p, coef[:, None] * Un[u])
        np.add.at(gS, n, -coef[:, None] * Un[u])
        Un -= (lr * gU).astype(np.float32)
        grad_slots.copy_(torch.from_numpy(gS.astype(np.float32)))

    def apply_sorted(self, tab, sorted_rows, perm, src, alpha):
        acc = np.zeros(tab.shape, np.float64)
        np.add.at(acc, sorted_rows.numpy().astype(np.int64), src.numpy()[perm.numpy().astype(np.int64)].astype(np.float64))
        tab.numpy()[...] += (alpha * acc).astype(np.float32)


def _worker(rank, world, port, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from whisprrec_amd.sharded import ShardedBprmf
        nU, nI, D, B, steps, lr = payload["shape"]
        m = ShardedBprmf(nU, nI, D, torch.device("cpu"), backend=OracleBackend())
        m.load_full(torch.from_numpy(payload["U"]), torch.from_numpy(payload["I"]))
        u, p, n = (torch.from_numpy(payload[k][rank]) for k in ("u", "p", "n"))
        assert bool((u % world == rank).all())
        cp = m.plan_chunk(u, p, n, B)
        losses = m.global_losses(m.run_chunk(cp, lr, global_batch=B * world))
        Uf, If = m.gather_full()
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), U=Uf.numpy(), I=If.numpy(), loss=losses.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,nI", [(2, 61), (3, 40)])
def test_sharded_step_equals_single_process(tmp_path, world, nI):
    rng = np.random.RandomState(world)
    nU, D, B, steps, lr = 47, 16, 64, 3, 0.2
    U = (rng.standard_normal((nU, D)) * 0.5).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.5).astype(np.float32)
    per_rank = {"u": [], "p": [], "n": []}
    for r in range(world):
        owned = np.arange(r, nU, world)
        per_rank["u"].append(rng.choice(owned, size=steps * B).astype(np.int64))
        per_rank["p"].append(rng.randint(0, nI, steps * B).astype(np.int64))   # few items: heavy cross-rank duplicates
        per_rank["n"].append(rng.randint(1, nI, steps * B).astype(np.int64))
    payload = dict(shape=(nU, nI, D, B, steps, lr), U=U, I=I, **per_rank)
    mp.spawn(_worker, args=(world, _free_port(), payload, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    # single-process reference: the global batch of step k is the union of the ranks' k-th batches
    Uo, Io = U.copy(), I.copy()
    ref_loss = []
    for k in range(steps):
        sl = slice(k * B, (k + 1) * B)
        gu = np.concatenate([per_rank["u"][r][sl] for r in range(world)])
        gp = np.concatenate([per_rank["p"][r][sl] for r in range(world)])
        gn = np.concatenate([per_rank["n"][r][sl] for r in range(world)])
        ref_loss.append(oracle.bprmf_step_sgd(Uo, Io, gu, gp, gn, lr, 0.0))
    assert rel_err(got["loss"], np.asarray(ref_loss)) < 1e-5
    assert rel_err(got["U"], Uo) < 1e-5
    assert rel_err(got["I"], Io) < 1e-5
