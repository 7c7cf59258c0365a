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


def route_numpy(u, p, n, batch, world, rank, n_users, n_items, M, nL, C):
    """NumPy restatement of wr_shard_route (whisprrec_amd/csrc/wr_shard.hip): local items keep their local row, a step's
    distinct remote items get slots in ascending (owner, local row) order; per owner the requested rows, padded to C"""
    u, p, n = (np.asarray(a, dtype=np.int64) for a in (u, p, n))
    N = u.size
    nb = (N + batch - 1) // batch
    assert (u % world == rank).all() and u.max() < n_users and max(p.max(), n.max()) < n_items
    vu, vp, vn = u // world, np.zeros(N, np.int64), np.zeros(N, np.int64)
    send_rows = np.zeros((world, nb, C), np.int32)
    send_cnt = np.zeros((world, nb), np.int32)
    for k in range(nb):
        sl = slice(k * batch, min(N, (k + 1) * batch))
        ids = np.concatenate([p[sl], n[sl]])
        owner, row = ids % world, ids // world
        remote = owner != rank
        keys = np.unique(owner[remote] * M + row[remote])
        v = np.where(remote, nL + np.searchsorted(keys, owner * M + row), row)
        vp[sl], vn[sl] = v[:sl.stop - sl.start], v[sl.stop - sl.start:]
        for o in range(world):
            rows_o = keys[keys // M == o] - o * M
            assert rows_o.size <= C
            send_rows[o, k, :rows_o.size] = rows_o
            send_cnt[o, k] = rows_o.size
    return vu, vp, vn, send_rows, send_cnt


class OracleBackend:
    """the backend's methods (whisprrec_amd.sharded.HipBackend) on CPU tensors, arithmetic from the CPU oracle"""

    class _Plan:
        pass

    def route(self, u, p, n, batch, world, rank, n_users, n_items, M, nL, C):
        vu, vp, vn, sr, sc = route_numpy(u.numpy(), p.numpy(), n.numpy(), batch, world, rank, n_users, n_items, M, nL, C)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        return (t(vu, torch.int32), t(vp, torch.int32), t(vn, torch.int32), t(sr.reshape(-1), torch.int32),
                t(sc.reshape(-1), torch.int32), torch.zeros(2, dtype=torch.int32))

    def pack(self, recv_rows, recv_cnt, nb, world, C, nL):
        rr, rc = recv_rows.numpy().reshape(world, nb, C), recv_cnt.numpy().reshape(world, nb)
        serve = np.zeros((nb, world * C), np.int64)
        off = np.zeros((nb, world + 1), np.int32)
        for k in range(nb):
            for s in range(world):
                c = int(rc[s, k])
                serve[k, off[k, s]:off[k, s] + c] = rr[s, k, :c]
                off[k, s + 1] = off[k, s] + c
        assert serve.max() < max(nL, 1)
        return torch.from_numpy(serve), torch.from_numpy(off), torch.zeros(1, dtype=torch.int32)

    def plan_local(self, vu, vp, vn, batch, n_user_rows, n_ext_rows, n_local_items, D=64):
        pl = self._Plan()
        pl.u, pl.p, pl.n, pl.batch_size = vu.numpy().astype(np.int64), vp.numpy().astype(np.int64), vn.numpy().astype(np.int64), batch
        assert pl.u.max() < n_user_rows and max(pl.p.max(), pl.n.max()) < n_ext_rows
        return pl

    def gather_rows(self, tab, idx):
        return torch.from_numpy(oracle.gather_rows(tab.numpy(), idx.numpy()))

    def local_step(self, U, I_ext, nL, plan, k, global_batch, lr, grad_slots, loss_out):
        lo = k * plan.batch_size
        u, p, n = plan.u[lo:lo + plan.batch_size], plan.p[lo:lo + plan.batch_size], plan.n[lo:lo + plan.batch_size]
        B = len(u)
        Un, R = U.numpy(), I_ext.numpy()
        _, _, coef, loss = oracle.bpr_fwd(Un, R, u, p, n)
        coef = coef.astype(np.float64) * B / global_batch            # mean over the GLOBAL batch
        loss_out[0] = loss * B / global_batch
        gU = np.zeros(Un.shape, np.float64)
        gI = np.zeros(R.shape, np.float64)
        np.add.at(gU, u, coef[:, None] * (R[p].astype(np.float64) - R[n]))
        np.add.at(gI, p, coef[:, None] * Un[u])
        np.add.at(gI, n, -coef[:, None] * Un[u])
        Un -= (lr * gU).astype(np.float32)
        R[:nL] -= (lr * gI[:nL]).astype(np.float32)                   # the shard's own rows: in place
        ns = min(grad_slots.shape[0], R.shape[0] - nL)
        grad_slots[:ns].copy_(torch.from_numpy(gI[nL:nL + ns].astype(np.float32)))     # received rows: gradients go back

    def scatter_add(self, tab, idx, src, alpha):
        acc = np.zeros(tab.shape, np.float64)
        np.add.at(acc, idx.numpy(), src.numpy().astype(np.float64))
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
        if payload.get("pipelined"):
            # bench_run's order: the index work of chunk c + 1 (its two index all-to-alls included) is queued BEFORE the steps
            # of chunk c and read back after them — the same sequence of collectives on every rank
            cut = (steps // 2) * B
            spans = [(0, cut), (cut, steps * B)]
            nxt, parts = m.plan_chunk_begin(u[:cut], p[:cut], n[:cut], B), []
            for i, (lo, hi) in enumerate(spans):
                cp = m.plan_chunk_end(nxt)
                nxt = m.plan_chunk_begin(u[spans[i + 1][0]:spans[i + 1][1]], p[spans[i + 1][0]:spans[i + 1][1]],
                                         n[spans[i + 1][0]:spans[i + 1][1]], B) if i + 1 < len(spans) else None
                parts.append(m.run_chunk(cp, lr, global_batch=B * world))
            losses = m.global_losses(torch.cat(parts))
        else:
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


@pytest.mark.parametrize("world,nI,pipelined", [(2, 61, False), (3, 40, False), (2, 53, True)])
def test_sharded_step_equals_single_process(tmp_path, world, nI, pipelined):
    rng = np.random.RandomState(world)
    nU, D, B, steps, lr = 47, 16, 64, 4 if pipelined else 3, 0.2
    U = (rng.standard_normal((nU, D)) * 0.5).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.5).astype(np.float32)
    per_rank = {"u": [], "p": [], "n": []}
    for r in range(world):
        owned = np.arange(r, nU, world)
        per_rank["u"].append(rng.choice(owned, size=steps * B).astype(np.int64))
        per_rank["p"].append(rng.randint(0, nI, steps * B).astype(np.int64))   # few items: heavy cross-rank duplicates
        per_rank["n"].append(rng.randint(1, nI, steps * B).astype(np.int64))
    payload = dict(shape=(nU, nI, D, B, steps, lr), U=U, I=I, pipelined=pipelined, **per_rank)
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
