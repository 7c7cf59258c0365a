"""GPU tests of the row-sharded step with a single rank over RCCL (the only multi-process-free configuration a 1-GPU box
allows): with one rank every row is local and the step stream is the single-GPU one; the index kernels of the multi-rank
path (wr_shard_route, wr_shard_pack) are checked here against their NumPy restatement.  The multi-rank exchange logic is
covered on CPU/gloo by tests/test_sharded_gloo.py and with real kernels by tests/test_hip_sharded_two_ranks.py."""
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


def test_route_and_pack_match_their_numpy_restatement(pg):
    """wr_shard_route / wr_shard_pack (index work of the row-sharded step) against the NumPy restatement the CPU/gloo tests
    exercise: virtual ids, request lists per owner, per-step serve lists — bit-exact, incl. a key space of several passes"""
    from whisprrec_amd.sharded import HipBackend
    from test_sharded_gloo import route_numpy
    dev = pg
    rng = np.random.RandomState(5)
    hb = HipBackend()
    for G, rank, nU, nI, B, N in ((4, 1, 2800, 5003, 1024, 3 * 1024 + 300), (3, 2, 999, 1_300_000, 4096, 2 * 4096)):
        M = ((nI + G - 1) // G + 31) // 32 * 32
        nL = (nI - rank + G - 1) // G
        C = min(2 * B, (2 * B // G) * 3 // 2 + 1024)
        u = rng.choice(np.arange(rank, nU, G), N).astype(np.int32)
        p = np.minimum((rng.pareto(1.0, N) * 40).astype(np.int64), nI - 1).astype(np.int32) if nI < 10_000 else \
            rng.randint(0, nI, N).astype(np.int32)
        n = rng.randint(1, nI, N).astype(np.int32)
        t = lambda a: torch.from_numpy(a).to(dev)
        vu, vp, vn, sr, sc, err = hb.route(t(u), t(p), t(n), B, G, rank, nU, nI, M, nL, C)
        rvu, rvp, rvn, rsr, rsc = route_numpy(u, p, n, B, G, rank, nU, nI, M, nL, C)
        assert not err.cpu().numpy().any()
        nb = (N + B - 1) // B
        assert np.array_equal(vu.cpu().numpy(), rvu) and np.array_equal(vp.cpu().numpy(), rvp) and np.array_equal(vn.cpu().numpy(), rvn)
        got_cnt = sc.cpu().numpy().reshape(G, nb)
        assert np.array_equal(got_cnt, rsc)
        got_rows = sr.cpu().numpy().reshape(G, nb, C)
        for o in range(G):
            for k in range(nb):
                assert np.array_equal(got_rows[o, k, :rsc[o, k]], rsr[o, k, :rsc[o, k]])
        # pack: what arrived (here: the lists this rank sent, read as if they had been requested FROM it)
        serve, off, err2 = hb.pack(sr, sc, nb, G, C, max(nL, M))
        off = off.cpu().numpy()
        assert not err2.cpu().numpy().any() and np.array_equal(np.diff(off, axis=1), rsc.T)
        for k in range(nb):
            want = np.concatenate([rsr[s, k, :rsc[s, k]] for s in range(G)])
            assert np.array_equal(serve[k, :off[k, G]].cpu().numpy(), want)
    # a user of another rank and an id out of range are reported
    bad_u = u.copy()
    bad_u[3] += 1
    assert hb.route(t(bad_u), t(p), t(n), B, G, rank, nU, nI, M, nL, C)[5].cpu().numpy()[0] != 0


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
