"""The row-sharded step (whisprrec_amd/sharded.py) with TWO (and three) ranks on ONE GPU: the product's HIP backend — real
kernels on cuda:0 in every rank — and torch.distributed over gloo, staged through the host (RCCL refuses several ranks on one
device; on a node each rank has its own GPU and the backend is RCCL).  Exchange logic, kernels and owner-side application together must equal the
single-process oracle on the union of the ranks' batches.  (The exchange logic alone, with the oracle as local arithmetic, is
tests/test_sharded_gloo.py on the CPU; one rank over RCCL is tests/test_hip_sharded.py.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from conftest import rel_err
from test_sharded_gloo import _free_port

pytestmark = pytest.mark.gpu


def _stage_collectives_through_host(dist):
    """gloo takes device pointers as they are, unordered with the GPU's streams (the host reads the card's memory through the
    PCIe aperture): the collectives sharded.py calls are wrapped so that device tensors are exchanged as CPU tensors behind
    a stream synchronisation — the ordering RCCL's stream-ordered collectives give on a node.  Test plumbing only."""
    a2a, ar, ag = dist.all_to_all_single, dist.all_reduce, dist.all_gather

    def all_to_all_single(out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        if not inp.is_cuda:
            return a2a(out, inp, output_split_sizes, input_split_sizes, group=group)
        torch.cuda.current_stream().synchronize()
        ho = torch.empty(out.shape, dtype=out.dtype)
        a2a(ho, inp.cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
        out.copy_(ho)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None):
        if not t.is_cuda:
            return ar(t, op=op, group=group)
        torch.cuda.current_stream().synchronize()
        h = t.cpu()
        ar(h, op=op, group=group)
        t.copy_(h)

    def all_gather(outs, t, group=None):
        if not t.is_cuda:
            return ag(outs, t, group=group)
        torch.cuda.current_stream().synchronize()
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        ag(hs, t.cpu().contiguous(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)

    dist.all_to_all_single, dist.all_reduce, dist.all_gather = all_to_all_single, all_reduce, all_gather


def _worker(rank, world, port, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _stage_collectives_through_host(dist)
    try:
        from whisprrec_amd.sharded import ShardedBprmf
        nU, nI, D, B, steps, lr = payload["shape"]
        m = ShardedBprmf(nU, nI, D, dev)                       # HipBackend: the C-ABI library
        m.load_full(torch.from_numpy(payload["U"]), torch.from_numpy(payload["I"]))
        u, p, n = (torch.from_numpy(payload[k][rank]).to(dev) for k in ("u", "p", "n"))
        if payload.get("pipelined"):
            # bench_run's order: chunk c + 1's index work goes to the side stream BEFORE chunk c's steps are queued
            cuts = [0, (steps // 3) * B, (2 * steps // 3) * B, steps * B]
            spans = [(cuts[i], cuts[i + 1]) for i in range(3) if cuts[i + 1] > cuts[i]]
            begin = lambda sp: m.plan_chunk_begin(u[sp[0]:sp[1]], p[sp[0]:sp[1]], n[sp[0]:sp[1]], B)
            nxt, parts = begin(spans[0]), []
            for i in range(len(spans)):
                cp = m.plan_chunk_end(nxt)
                nxt = begin(spans[i + 1]) if i + 1 < len(spans) else None
                parts.append(m.run_chunk(cp, lr, global_batch=B * world))
            losses = m.global_losses(torch.cat(parts))
        else:
            cp = m.plan_chunk(u, p, n, B)
            losses = m.global_losses(m.run_chunk(cp, lr, global_batch=B * world))
        Uf, If = m.gather_full()
        torch.cuda.synchronize()
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), U=Uf.cpu().numpy(), I=If.cpu().numpy(), loss=losses.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nU,nI,D,B,pipelined", [(2, 5001, 3001, 64, 2048, False), (3, 997, 401, 32, 512, False),
                                                       (2, 300_001, 260_001, 64, 4096, True)])
def test_sharded_step_two_ranks_one_gpu_equals_single_process(tmp_path, world, nU, nI, D, B, pipelined):
    """pipelined: three chunks, every chunk's index work queued on the side stream before the steps of the chunk before
    (bench_run's order); shards large enough for group plans"""
    rng = np.random.RandomState(world)
    steps, lr = (6 if pipelined else 3), 0.2
    U = (rng.standard_normal((nU, D)) * 0.5).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.5).astype(np.float32)
    per_rank = {"u": [], "p": [], "n": []}
    for r in range(world):
        owned = np.arange(r, nU, world)
        per_rank["u"].append(rng.choice(owned, size=steps * B).astype(np.int64))
        per_rank["p"].append(rng.randint(0, nI, steps * B).astype(np.int64))      # cross-rank duplicates of item rows
        per_rank["n"].append(rng.randint(1, nI, steps * B).astype(np.int64))
    payload = dict(shape=(nU, nI, D, B, steps, lr), U=U, I=I, pipelined=pipelined, **per_rank)
    mp.spawn(_worker, args=(world, _free_port(), payload, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
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


# ------------------------------------------------------------------------------------------------ configs[3] shard shape
def _row_values(rows, D, salt):
    """deterministic table rows from their global ids (both the ranks and the reference build them): [len(rows), D] float32"""
    r = np.asarray(rows, dtype=np.float64)[:, None]
    d = np.arange(D, dtype=np.float64)[None, :]
    return (0.05 * np.sin(0.37 * r + 1.3 * d + salt) * np.cos(0.011 * r * (d + 1.0))).astype(np.float32)


def _c4_worker(rank, world, port, payload, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _stage_collectives_through_host(dist)
    try:
        from whisprrec_amd.sharded import ShardedBprmf
        nU, nI, D, B, steps, lr = payload["shape"]
        m = ShardedBprmf(nU, nI, D, dev)
        # the shard's rows from their global ids, chunk by chunk (no full table anywhere)
        for tab, n_rows, salt in ((m.U, nU, 0.0), (m.I, nI, 1.0)):
            ids = np.arange(rank, n_rows, world)
            for lo in range(0, ids.size, 1 << 18):
                tab[lo:lo + (1 << 18)].copy_(torch.from_numpy(_row_values(ids[lo:lo + (1 << 18)], D, salt)))
        u, p, n = (torch.from_numpy(payload[k][rank]).to(dev) for k in ("u", "p", "n"))
        cp = m.plan_chunk(u, p, n, B)
        losses = m.global_losses(m.run_chunk(cp, lr, global_batch=B * world))
        torch.cuda.synchronize()
        tu, ti = payload["touched_u"], payload["touched_i"]
        mine_u, mine_i = tu[tu % world == rank], ti[ti % world == rank]
        np.savez(os.path.join(out_dir, "out%d.npz" % rank), u_ids=mine_u, i_ids=mine_i,
                 U=m.U[torch.from_numpy(mine_u // world).to(dev)].cpu().numpy(),
                 I=m.I[torch.from_numpy(mine_i // world).to(dev)].cpu().numpy(), loss=losses.cpu().numpy(),
                 checksum=np.asarray([float(m.U.double().sum().item()), float(m.I.double().sum().item())]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_step_at_the_configs3_shard_shape(tmp_path, world):
    """BASELINE.json configs[3] per-rank shape: D = 128, 1.25 M-row shards of both tables, B = 65,536 per rank — with one rank
    and with two ranks on the one GPU (real kernels, exchange staged through the host), against the single-process oracle
    on the union of the ranks' batches, restricted to the touched rows (compact tables, same arithmetic per row)."""
    rng = np.random.RandomState(40 + world)
    D, B, steps, lr = 128, 65536, 2, 0.1
    nU = nI = 1_250_000 * world
    per_rank = {"u": [], "p": [], "n": []}
    for r in range(world):
        per_rank["u"].append((rng.randint(0, nU // world, steps * B) * world + r).astype(np.int64))
        per_rank["p"].append(rng.randint(0, nI, steps * B).astype(np.int64))
        per_rank["n"].append(rng.randint(1, nI, steps * B).astype(np.int64))
    tu = np.unique(np.concatenate(per_rank["u"]))
    ti = np.unique(np.concatenate(per_rank["p"] + per_rank["n"]))
    payload = dict(shape=(nU, nI, D, B, steps, lr), touched_u=tu, touched_i=ti, **per_rank)
    mp.spawn(_c4_worker, args=(world, _free_port(), payload, str(tmp_path)), nprocs=world, join=True)
    # reference on compact tables of the touched rows
    Uc, Ic = _row_values(tu, D, 0.0), _row_values(ti, D, 1.0)
    ref_loss = []
    for k in range(steps):
        sl = slice(k * B, (k + 1) * B)
        gu = np.searchsorted(tu, np.concatenate([per_rank["u"][r][sl] for r in range(world)]))
        gp = np.searchsorted(ti, np.concatenate([per_rank["p"][r][sl] for r in range(world)]))
        gn = np.searchsorted(ti, np.concatenate([per_rank["n"][r][sl] for r in range(world)]))
        ref_loss.append(oracle.bprmf_step_sgd(Uc, Ic, gu, gp, gn, lr, 0.0))
    for r in range(world):
        got = np.load(tmp_path / ("out%d.npz" % r))
        assert rel_err(got["loss"], np.asarray(ref_loss)) < 1e-5
        assert rel_err(got["U"], Uc[np.searchsorted(tu, got["u_ids"])]) < 1e-5
        assert rel_err(got["I"], Ic[np.searchsorted(ti, got["i_ids"])]) < 1e-5
