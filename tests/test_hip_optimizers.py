"""torch.optim.Adagrad / Adadelta (weight_decay = 0) fused into the step kernels (wr_bprmf_run_stateful) — the rest of the
reference's --optimizer flag surface (src/helpers/BaseRunner.py:34-37,120-124).  Against the trajectories the reference
produced (tests/golden/g10_optimizers.npz), and against the oracle's dense restatements on bigger tables where rows miss
many steps between two updates (Adadelta's state decays at every step; the kernel replays the missed decays)."""
import argparse

import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name", ["Adagrad", "Adadelta"])
def test_trajectories_match_reference_golden(g1, g10, name):
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    tag = name.lower()
    lr = float(g10[tag + "_hp"][0])
    U, I = T(g1["U0"], dev), T(g1["I0"], dev)
    st = hip_ops.StatefulSparseState(hip_ops.BprmfTables(U, I), name, lr)
    for k in range(5):
        plan = hip_ops.BatchPlan(T(g1[f"u{k}"], dev), T(g1[f"p{k}"], dev), T(g1[f"n{k}"], dev), 512, 97, 131)
        loss = st.step(plan, 0)
        assert abs(float(loss) - g10[tag + "_loss"][k]) / g10[tag + "_loss"][k] < TOL
        assert rel_err(U.cpu().numpy(), g10[f"{tag}_U{k + 1}"]) < TOL
        assert rel_err(I.cpu().numpy(), g10[f"{tag}_I{k + 1}"]) < TOL


@pytest.mark.parametrize("name,D,zipf", [("Adagrad", 64, False), ("Adadelta", 64, False), ("Adadelta", 32, True), ("Adagrad", 128, True),
                                         ("Adadelta", 20, False)])
def test_sparse_fused_equals_dense_restatement(name, D, zipf):
    """tables much bigger than the batches: most rows miss many steps between two updates.  Dense restatement = the oracle's
    whole-table step on the oracle's dense gradients (every row, every step), as torch.optim does it."""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5 + D)
    nU, nI, B, steps = 1500, 1200, 128, 14
    lr = 0.05 if name == "Adagrad" else 1.5
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    N = steps * B - 40                                   # short last batch
    u = rng.randint(0, nU, N)
    p = np.minimum((rng.pareto(1.0, N) * 3).astype(np.int64), nI - 1) if zipf else rng.randint(0, nI, N)
    n = rng.randint(1, nI, N)
    if zipf:
        u[::3] = 7                                       # a hot user and hot items: pieces + combine kernels
    plan = hip_ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
    if zipf:
        assert plan.hot is not None
    outs = []
    for native in (True, False, "lag3", "lag1"):              # lagN: a rotating window keeps every row within N steps (Adadelta)
        Ud, Id = T(U, dev), T(I, dev)
        st = hip_ops.StatefulSparseState(hip_ops.BprmfTables(Ud, Id), name, lr,
                                         max_lag=int(native[3:]) if isinstance(native, str) else 0)
        if native:
            losses = torch.cat([st.run(plan, 0, 5), st.run(plan, 5, steps - 5)])
        else:
            losses = torch.stack([st.step(plan, k).clone() for k in range(steps)])
        torch.cuda.synchronize()
        outs.append((Ud, Id, losses, st))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])
    if name == "Adadelta":
        assert int(outs[2][3].last_u.min()) >= steps - 1 - 3 and int(outs[0][3].last_u.min()) == 0
    Uo, Io = U.copy(), I.copy()
    z = np.zeros_like
    su, si, au, ai = z(Uo), z(Io), z(Uo), z(Io)
    ref_loss = []
    for k in range(steps):
        sl = slice(k * B, min(N, (k + 1) * B))
        gU, gI, loss = oracle.bpr_dense_grads(Uo, Io, u[sl], p[sl], n[sl])
        ref_loss.append(loss)
        if name == "Adagrad":
            oracle.adagrad_dense(Uo, gU, su, lr); oracle.adagrad_dense(Io, gI, si, lr)
        else:
            oracle.adadelta_dense(Uo, gU, su, au, lr); oracle.adadelta_dense(Io, gI, si, ai, lr)
    Ud, Id, losses, st = outs[0]
    assert rel_err(losses.cpu().numpy(), np.asarray(ref_loss)) < TOL
    assert rel_err(Ud.cpu().numpy(), Uo) < 1e-4 and rel_err(Id.cpu().numpy(), Io) < 1e-4
    # rows no batch contains: weights bit-identical (a zero gradient moves nothing)
    mu = np.ones(nU, bool); mu[u] = False
    mi = np.ones(nI, bool); mi[p] = False; mi[n] = False
    assert np.array_equal(Ud.cpu().numpy()[mu], U[mu]) and np.array_equal(Id.cpu().numpy()[mi], I[mi])
    if name == "Adagrad":
        assert rel_err(st.s1_u.cpu().numpy(), su) < 1e-4 and rel_err(st.s1_i.cpu().numpy(), si) < 1e-4


@pytest.mark.parametrize("name,l2,fused", [("Adagrad", 0.0, True), ("Adadelta", 0.0, True), ("Adagrad", 1e-4, False)])
def test_model_takes_the_fused_optimizer_only_without_weight_decay(name, l2, fused):
    """with --l2 the two optimizers move every row (weight decay): that stays on torch.optim over the dense gradients"""
    from whisprrec_amd import host
    from whisprrec_amd.bprmf import BPRMF, FusedOptimizer
    dev = torch.device("cuda:0")
    args = argparse.Namespace(device=dev, model_path="/tmp/wr_opt.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64,
                              optimizer=name, lr=0.05, l2=l2)
    m = BPRMF(args, host.Corpus(50, 40, {})).to(dev)
    assert isinstance(m.optimizer, FusedOptimizer) == fused
    rng = np.random.RandomState(1)
    batch = {"user_id": T(rng.randint(0, 50, 64), dev), "pos_item": T(rng.randint(0, 40, 64), dev),
             "neg_items": T(rng.randint(1, 40, 64), dev).unsqueeze(1), "batch_size": 64, "phase": "train"}
    opt = m.optimizer if fused else getattr(torch.optim, name)(m.parameters(), lr=0.05, weight_decay=l2)
    m.train()
    w0 = m.user_embeddings.weight.detach().clone()
    for _ in range(2):
        opt.zero_grad()
        loss = m.predict(batch)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss.detach())) and not torch.equal(m.user_embeddings.weight.detach(), w0)
