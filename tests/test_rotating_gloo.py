"""world_size 2 and 3 CPU/gloo test of the stratified (rotating item blocks) multi-GPU schedule, whisprrec_amd/rotating.py:
one full epoch — every item block visits every rank — must equal the single-process oracle applied to the global batches
(union of the ranks' k-th batches).  Local arithmetic injected from the oracle; ring send/recv through gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from conftest import rel_err


class OracleLocal:
    def plan(self, U, segments, batch):
        return U.numpy(), [(rows.numpy(), u.numpy().astype(np.int64), p.numpy().astype(np.int64), n.numpy().astype(np.int64))
                           for rows, u, p, n in segments], batch

    def run(self, handle, seg, lr, losses):
        U, segs, B = handle
        I, u, p, n = segs[seg]
        assert U.flags.c_contiguous and I.flags.c_contiguous
        for k in range((len(u) + B - 1) // B):
            sl = slice(k * B, (k + 1) * B)
            losses[k] = oracle.bprmf_step_sgd(U, I, u[sl], p[sl], n[sl], lr, 0.0)


def _worker(rank, world, port, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from whisprrec_amd.rotating import RotatingBprmf
        nU, nI, D, B, lr, parts = payload["shape"]
        m = RotatingBprmf(nU, nI, D, torch.device("cpu"), parts=parts, local=OracleLocal())
        m.load_full(torch.from_numpy(payload["U"]), torch.from_numpy(payload["I"]))
        T = torch.from_numpy
        if payload["whole_epoch"] == "deferred":                    # one call per stratum, last hand-over left to the next call
            losses = []
            for r in range(world):
                u, p, n, per_part = payload["strata"][rank][r]
                losses.append(m.run_strata([(T(u), T(p), T(n), per_part)], B, lr, defer_last=True))
                assert m._deferred == (parts - 1, True) and m.held == (rank + r) % world     # the rotation is still open
            m.complete_rotation()
        elif payload["whole_epoch"] == "pieces":                    # ONE continuous run cut at arbitrary steps (bench.py N > 1)
            S = sum(payload["strata"][rank][0][3])
            spp = payload["strata"][rank][0][3]
            part_end = np.cumsum(spp)
            losses, g0 = [], 0
            for length in payload["cuts"]:
                pieces, gs, g1 = [], g0, g0 + length
                while gs < g1:
                    r, o0 = gs // S, gs % S
                    o1 = min(S, o0 + (g1 - gs))
                    u, p, n, _ = payload["strata"][rank][r]
                    per_part = [max(0, min(o1, int(part_end[k])) - max(o0, int(part_end[k] - spp[k]))) for k in range(parts)]
                    ends = [bool(o0 < part_end[k] <= o1) for k in range(parts)]
                    sl = slice(o0 * B, o1 * B)
                    pieces.append((T(u[sl]), T(p[sl]), T(n[sl]), per_part, ends))
                    gs += o1 - o0
                losses.append(m.run_strata(pieces, B, lr, defer_last=True))
                g0 += length
            assert g0 == world * S
            m.complete_rotation()
        elif payload["whole_epoch"]:                                # all strata in one call: plans pipelined across rotations
            losses = [m.run_strata([(T(u), T(p), T(n), pp) for (u, p, n, pp) in payload["strata"][rank]], B, lr)]
        else:
            losses = []
            for r in range(world):
                u, p, n, per_part = payload["strata"][rank][r]
                assert m.held == (rank + r) % world
                losses.append(m.run_subepoch(T(u), T(p), T(n), per_part, B, lr))
        assert m.held == rank                                       # a full epoch brings every block home
        gl = m.global_losses(torch.cat(losses))
        Uf, If = m.gather_full()
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), U=Uf.numpy(), I=If.numpy(), loss=gl.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize("world,nI,parts,whole", [(2, 61, 2, False), (3, 100, 2, True), (2, 40, 1, True), (3, 37, 3, False),
                                                  (2, 61, 2, "deferred"), (3, 50, 1, "deferred"), (2, 61, 2, "pieces"),
                                                  (3, 70, 2, "pieces")])
def test_rotating_epoch_equals_single_process(tmp_path, world, nI, parts, whole):
    from whisprrec_amd.sharded import n_local_rows
    rng = np.random.RandomState(world * 10 + parts)
    nU, D, B, lr = 53, 16, 32, 0.3
    steps_per_part = [2] * parts
    U = (rng.standard_normal((nU, D)) * 0.5).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.5).astype(np.float32)

    def part_range(block, k):
        n = n_local_rows(nI, block, world); per = (n + parts - 1) // parts
        return min(n, k * per), min(n, (k + 1) * per)

    strata = [[None] * world for _ in range(world)]
    glob = [[] for _ in range(world)]                               # per sub-epoch: list over local steps of per-rank batches
    for rank in range(world):
        n_loc_u = n_local_rows(nU, rank, world)
        for r in range(world):
            held = (rank + r) % world
            us, ps, ns = [], [], []
            for k in range(parts):
                lo, hi = part_range(held, k)
                cnt = steps_per_part[k] * B
                us.append(rng.randint(0, n_loc_u, cnt)); ps.append(rng.randint(lo, max(hi, lo + 1), cnt)); ns.append(rng.randint(lo, max(hi, lo + 1), cnt))
            u, p, n = np.concatenate(us), np.concatenate(ps), np.concatenate(ns)
            strata[rank][r] = (u.astype(np.int64), p.astype(np.int64), n.astype(np.int64), steps_per_part)
            glob[r].append((u * world + rank, p * world + held, n * world + held))   # back to global ids
    total = world * sum(steps_per_part)
    cuts = [3, 1, 2] + [total - 6] if total > 6 else [total]       # pieces that end inside parts, on part ends, on stratum ends
    payload = dict(shape=(nU, nI, D, B, lr, parts), U=U, I=I, strata=strata, whole_epoch=whole, cuts=cuts)
    mp.spawn(_worker, args=(world, _free_port(), payload, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    Uo, Io = U.copy(), I.copy()
    ref = []
    nsteps = sum(steps_per_part)
    for r in range(world):
        for k in range(nsteps):
            sl = slice(k * B, (k + 1) * B)
            gu = np.concatenate([glob[r][rank][0][sl] for rank in range(world)])
            gp = np.concatenate([glob[r][rank][1][sl] for rank in range(world)])
            gn = np.concatenate([glob[r][rank][2][sl] for rank in range(world)])
            ref.append(oracle.bprmf_step_sgd(Uo, Io, gu, gp, gn, lr, 0.0))           # ONE step over the global batch
    assert rel_err(got["loss"], np.asarray(ref)) < 1e-5
    assert rel_err(got["U"], Uo) < 1e-5
    assert rel_err(got["I"], Io) < 1e-5
