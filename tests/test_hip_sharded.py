"""GPU test of the row-sharded step through RCCL with a single rank (the only multi-process-free configuration a 1-GPU
box allows): exercises HipBackend — wr_gather_rows, wr_bprmf_shard_step (MODE 2), wr_apply_rows_sorted — and
torch.distributed's nccl(=RCCL) all_to_all_single, against the oracle.  The multi-rank exchange logic itself is covered on
CPU/gloo by tests/test_sharded_gloo.py."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_sharded_world1_matches_oracle(pg):
    from whisprrec_amd.sharded import ShardedBprmf
    dev = pg
    rng = np.random.RandomState(11)
    nU, nI, D, B, steps, lr = 3000, 2000, 64, 4096, 4, 0.3
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    u = rng.randint(0, nU, steps * B); p = rng.randint(0, nI, steps * B); n = rng.randint(1, nI, steps * B)
    m = ShardedBprmf(nU, nI, D, dev)
    m.load_full(torch.from_numpy(U), torch.from_numpy(I))
    cp = m.plan_chunk(torch.from_numpy(u).to(dev), torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev), B)
    losses = m.global_losses(m.run_chunk(cp, lr)).cpu().numpy()
    Uf, If = m.gather_full()
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
           for k in range(steps)]
    assert rel_err(losses, np.asarray(ref)) < 1e-5
    assert rel_err(Uf.cpu().numpy(), Uo) < 1e-5
    assert rel_err(If.cpu().numpy(), Io) < 1e-5


def test_shard_step_global_batch_scaling(pg):
    """coefficients and loss share are scaled by 1/global_batch, not 1/local batch"""
    from whisprrec_amd.sharded import ShardedBprmf
    dev = pg
    rng = np.random.RandomState(12)
    nU, nI, D, B = 500, 400, 32, 1024
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    u = rng.randint(0, nU, B); p = rng.randint(0, nI, B); n = rng.randint(1, nI, B)
    m = ShardedBprmf(nU, nI, D, dev)
    m.load_full(torch.from_numpy(U), torch.from_numpy(I))
    cp = m.plan_chunk(torch.from_numpy(u).to(dev), torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev), B)
    losses = m.run_chunk(cp, 0.4, global_batch=4 * B).cpu().numpy()
    Uo, Io = U.copy(), I.copy()
    lo = oracle.bprmf_step_sgd(Uo, Io, u, p, n, 0.4 / 4, 0.0)   # grads scale by 1/4 <=> lr/4
    assert abs(losses[0] - lo / 4) / (lo / 4) < 1e-5
    Uf, If = m.gather_full()
    assert rel_err(Uf.cpu().numpy(), Uo) < 1e-5 and rel_err(If.cpu().numpy(), Io) < 1e-5


def test_plan_slots_equals_generic_slotify(pg):
    """HipBackend.plan_slots (slots from the batch plan's sort) must give the unique-key lists and slot ids of the
    torch.unique restatement that the CPU/gloo tests exercise"""
    from whisprrec_amd.sharded import HipBackend, slotify_generic
    dev = pg
    rng = np.random.RandomState(5)
    G, nI, nUloc, B, N = 4, 5003, 700, 1024, 3 * 1024 + 300
    M = (nI + G - 1) // G
    u_loc = torch.from_numpy(rng.randint(0, nUloc, N)).to(dev)
    p = torch.from_numpy(np.minimum((rng.pareto(1.0, N) * 40).astype(np.int64), nI - 1)).to(dev)
    n = torch.from_numpy(rng.randint(1, nI, N)).to(dev)
    rk_p, rk_n = (p % G) * M + p // G, (n % G) * M + n // G
    sp, sn, key, step, nq = slotify_generic(u_loc, rk_p, rk_n, B, G * M)
    plan, key2, step2, nq2 = HipBackend().plan_slots(u_loc, rk_p, rk_n, B, nUloc, G * M)
    assert torch.equal(key, key2) and torch.equal(step, step2) and torch.equal(nq, nq2)
    # the plan is sorted by user: undo through the slots it carries
    tp = (plan.tp & 0x7FFFFFFF).cpu().numpy(); tn = (plan.tn & 0x7FFFFFFF).cpu().numpy(); tu = plan.tu.cpu().numpy()
    for k in range(plan.n_batches):
        lo, hi = k * B, min(N, (k + 1) * B)
        order = lo + np.argsort(u_loc.cpu().numpy()[lo:hi], kind="stable")
        assert np.array_equal(tu[lo:hi], u_loc.cpu().numpy()[order])
        assert np.array_equal(tp[lo:hi], sp.cpu().numpy()[order]) and np.array_equal(tn[lo:hi], sn.cpu().numpy()[order])


def test_rotating_world1_matches_oracle(pg):
    """stratified schedule with one rank: the ring send is a self-copy through RCCL P2P, the local arithmetic is the
    single-GPU run path on the held block (whisprrec_amd/rotating.py)"""
    from whisprrec_amd.rotating import RotatingBprmf
    dev = pg
    rng = np.random.RandomState(21)
    nU, nI, D, B, lr, parts = 3000, 2001, 64, 4096, 0.3, 2
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    m = RotatingBprmf(nU, nI, D, dev, parts=parts)
    m.load_full(torch.from_numpy(U), torch.from_numpy(I))
    Uo, Io = U.copy(), I.copy()
    ref, got = [], []
    for sub in range(3):
        per_part = [2, 3]
        us, ps, ns = [], [], []
        for k in range(parts):
            lo, hi = m.part_range(m.held, k)
            cnt = per_part[k] * B
            us.append(rng.randint(0, nU, cnt)); ps.append(rng.randint(lo, hi, cnt)); ns.append(rng.randint(max(lo, 1), hi, cnt))
        u, p, n = np.concatenate(us), np.concatenate(ps), np.concatenate(ns)
        t = lambda a: torch.from_numpy(a).to(dev)
        got.append(m.global_losses(m.run_subepoch(t(u), t(p), t(n), per_part, B, lr)).cpu().numpy())
        for k in range(sum(per_part)):
            sl = slice(k * B, (k + 1) * B)
            ref.append(oracle.bprmf_step_sgd(Uo, Io, u[sl], p[sl], n[sl], lr, 0.0))
    Uf, If = m.gather_full()
    assert rel_err(np.concatenate(got), np.asarray(ref)) < 1e-5
    assert rel_err(Uf.cpu().numpy(), Uo) < 1e-5
    assert rel_err(If.cpu().numpy(), Io) < 1e-5
