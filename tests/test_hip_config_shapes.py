"""GPU parity at the shapes of BASELINE.json configs[2], [3], [4] (C3, C4, C5 of SURVEY.md 8a) — the sizes the other GPU
tests stay below:
  C3  LightGCN gcn_layers=2, emb_size=64 on an ml-1m-shaped graph (6,040 users x 3,706 items, ~0.7 M train pairs with a
      power-law item popularity: the head items are rated by a large share of the users), B = 2,048: predict + backward
      against the oracle's restatement (reference src/models/general/LightGCN.py:134-175);
  C4  BPRMF emb_size=128 on 10M x 10M tables, B = 65,536 (one rank's batch of the 8-GPU configuration, here on one GPU):
      one fused step against the oracle on the touched rows, every other row bit-identical, loss = the forward-only kernel's
      (reference src/models/general/BPRMF.py:69-80, src/helpers/BaseRunner.py:196-199);
  C5  SASRec's item-embedding slice at B = 2,048, history_max = 20, 3,706 items: gather bit-exact, scatter-add against the
      oracle, padding row 0 without gradient (reference src/models/sequential/SASRec.py:84,105-106).
The real ml-1m .inter file is not in the reference tree, hence the synthetic graph of its shape."""
import argparse

import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err
from whisprrec_amd import host

pytestmark = pytest.mark.gpu
TOL = 1e-5


def ml1m_shaped_pairs(seed=3407, n_users=6040, n_items=3706, n_pairs=700_000):
    """distinct (user, item) train pairs with ml-1m's shape: ~0.7 M of them, every user has >= 20, item popularity is a
    power law whose head reaches ~40 % of the users (ml-1m's most rated films are rated by > 40 % of the users)"""
    rng = np.random.RandomState(seed)
    w = 1.0 / (np.arange(n_items) + 25) ** 0.8
    w = rng.permutation(w / w.sum())                      # popular items spread over the id range
    per_user = np.maximum(20, rng.lognormal(np.log(85.0), 0.9, n_users)).astype(np.int64)
    per_user = np.minimum((per_user * (1.1 * n_pairs / per_user.sum())).astype(np.int64) + 1, n_items // 2)
    users = np.repeat(np.arange(n_users), np.maximum(per_user, 24))
    items = rng.choice(n_items, size=users.size, p=w)
    key = np.unique(users.astype(np.int64) * n_items + items)          # distinct pairs, user-major
    return key // n_items, key % n_items


def test_c3_lightgcn_ml1m_shape_loss_and_grads_match_oracle():
    from whisprrec_amd.lightgcn import LightGCN
    dev = torch.device("cuda:0")
    nU, nI, D, B, L = 6040, 3706, 64, 2048, 2
    uu, ii = ml1m_shaped_pairs()
    assert uu.size > 500_000
    pop = np.bincount(ii, minlength=nI)
    assert pop.max() > 0.25 * nU                                        # the head rows are dense, as in ml-1m
    ptr = np.zeros(nU + 1, np.int64)
    np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
    tcs = {u: set(ii[ptr[u]:ptr[u + 1]].tolist()) for u in range(nU)}
    corpus = host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []},
                                  "test": {"user_id": [], "item_id": []}}, tcs, {})
    args = argparse.Namespace(device=dev, model_path="/tmp/wr_lgcn_c3.pt", buffer=1, num_neg=1, test_all=1, embedding_size=D,
                              gcn_layers=L, reg_weight=1e-5)
    m = LightGCN(args, corpus).to(dev)
    rng = np.random.RandomState(7)
    E0 = (rng.standard_normal((nU + nI, D)) * 0.1).astype(np.float32)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(E0[:nU]))
        m.item_embedding.weight.copy_(torch.from_numpy(E0[nU:]))
    rows = rng.randint(0, uu.size, B)                                   # a training batch: observed pairs + negatives
    u, p, n = uu[rows], ii[rows], rng.randint(1, nI, B)
    rp, col, val = oracle.lightgcn_build_adj(nU, nI, ptr, ii.astype(np.int32))
    assert rp[-1] == 2 * uu.size
    loss_ref, g_ref = oracle.lightgcn_loss_grads(nU, nI, rp, col, val, E0, L, 1e-5, u, p, n)
    fwd_ref = oracle.lightgcn_forward(rp, col, val, E0, L)
    ua, ia = m.forward()
    assert rel_err(ua.detach().cpu().numpy(), fwd_ref[:nU]) < TOL and rel_err(ia.detach().cpu().numpy(), fwd_ref[nU:]) < TOL
    m.train()
    batch = {"user_id": torch.from_numpy(u).to(dev), "pos_item": torch.from_numpy(p).to(dev),
             "neg_items": torch.from_numpy(n).to(dev)}
    loss = m.predict(batch)
    assert loss.shape == (1,) and abs(float(loss.detach()) - loss_ref) / abs(loss_ref) < TOL
    loss.backward()
    assert rel_err(m.user_embedding.weight.grad.cpu().numpy(), g_ref[:nU]) < TOL
    assert rel_err(m.item_embedding.weight.grad.cpu().numpy(), g_ref[nU:]) < TOL


def test_c4_bprmf_d128_10m_tables_step_matches_oracle_on_touched_rows():
    from whisprrec_amd import hip_ops as ops
    dev = torch.device("cuda:0")
    nU = nI = 10_000_000
    D, B, lr = 128, 65536, 0.05
    g = torch.Generator(device=dev).manual_seed(3407)
    U = torch.randn(nU, D, generator=g, device=dev) * 0.05            # 5.12 GB each, generated on the device
    I = torch.randn(nI, D, generator=g, device=dev) * 0.05
    rng = np.random.RandomState(3407)
    u, p, n = rng.randint(0, nU, B), rng.randint(0, nI, B), rng.randint(1, nI, B)
    p[:64] = p[64:128]                                                  # some shared item rows and users even at 10M rows
    n[200:232] = p[300:332]
    u[:16] = u[16:32]
    T = lambda a: torch.from_numpy(a.astype(np.int32)).to(dev)
    fwd = ops.bpr_fwd(U, I, torch.from_numpy(u).to(dev), torch.from_numpy(p).to(dev), torch.from_numpy(n).to(dev), scores=False)
    U0, I0 = U.clone(), I.clone()
    tabs = ops.BprmfTables(U, I)
    plan = ops.BatchPlan(T(u), T(p), T(n), B, nU, nI)
    loss = tabs.step_sgd(plan, 0, lr)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(fwd["loss"])) / float(fwd["loss"]) < 1e-6
    # the oracle on the touched rows: compact tables (rows renumbered in id order), same arithmetic per row
    ku, inv_u = np.unique(u, return_inverse=True)
    ki, inv_i = np.unique(np.concatenate([p, n]), return_inverse=True)
    Uc = U0[torch.from_numpy(ku).to(dev)].cpu().numpy()
    Ic = I0[torch.from_numpy(ki).to(dev)].cpu().numpy()
    loss_ref = oracle.bprmf_step_sgd(Uc, Ic, inv_u, inv_i[:B], inv_i[B:], lr, 0.0)
    assert abs(float(loss) - loss_ref) / loss_ref < TOL
    assert rel_err(tabs.U[torch.from_numpy(ku).to(dev)].cpu().numpy(), Uc) < TOL
    assert rel_err(tabs.I[torch.from_numpy(ki).to(dev)].cpu().numpy(), Ic) < TOL
    # rows outside the batch: bit-identical (exactly sparse update, SGD l2 = 0)
    for tab, tab0, keys in ((tabs.U, U0, ku), (tabs.I, I0, ki)):
        changed = (tab != tab0).any(dim=1)
        touched = torch.zeros(tab.shape[0], dtype=torch.bool, device=dev)
        touched[torch.from_numpy(keys).to(dev)] = True
        assert not bool((changed & ~touched).any())
        assert int(changed.sum()) > 0.99 * keys.size
    # a second run from the same tables: bitwise equal
    tabs2 = ops.BprmfTables(U0, I0)
    loss2 = tabs2.step_sgd(plan, 0, lr)
    assert torch.equal(tabs2.U, tabs.U) and torch.equal(tabs2.I, tabs.I) and float(loss2) == float(loss)


def test_c5_sasrec_item_embedding_slice_ml1m_shape():
    from whisprrec_amd.sasrec import HipEmbedding
    dev = torch.device("cuda:0")
    nI, D, B, T = 3706, 64, 2048, 20
    rng = np.random.RandomState(5)
    W = (rng.standard_normal((nI, D)) * 0.1).astype(np.float32)
    lengths = rng.randint(1, T + 1, B)
    hist = rng.randint(1, nI, (B, T)).astype(np.int64)
    hist[np.arange(T)[None, :] >= lengths[:, None]] = 0                 # right-padded with 0 (reference BaseModel.py:119)
    pos, neg = rng.randint(1, nI, B).astype(np.int64), rng.randint(1, nI, B).astype(np.int64)
    emb = HipEmbedding(nI, D, padding_idx=0).to(dev)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(W))
    th, tp, tn = (torch.from_numpy(a).to(dev) for a in (hist, pos, neg))
    gh, gp, gn = emb(th), emb(tp), emb(tn)
    assert np.array_equal(gh.detach().cpu().numpy(), oracle.gather_rows(W, hist))      # a gather is a copy: bit-exact
    assert np.array_equal(gp.detach().cpu().numpy(), oracle.gather_rows(W, pos))
    assert np.array_equal(gn.detach().cpu().numpy(), oracle.gather_rows(W, neg))
    ch, cp, cn = (rng.standard_normal(s).astype(np.float32) for s in ((B, T, D), (B, D), (B, D)))
    ((gh * torch.from_numpy(ch).to(dev)).sum() + (gp * torch.from_numpy(cp).to(dev)).sum()
     + (gn * torch.from_numpy(cn).to(dev)).sum()).backward()
    got = emb.weight.grad.cpu().numpy()
    idx = np.concatenate([hist.reshape(-1), pos, neg])
    src = np.concatenate([ch.reshape(-1, D), cp, cn])
    ref = oracle.scatter_add_rows(nI, idx, src, padding_idx=0)
    assert rel_err(got, ref) < TOL
    assert not got[0].any()                                             # row 0: all three uses drop its gradient
    assert idx.size == B * (T + 2) and (idx == 0).sum() > 1000          # ~45 K rows, thousands of them padding


@pytest.mark.parametrize("D,min_density", [(64, 0.12), (64, 0.05), (32, 0.2), (128, 0.12)])
def test_c3_hybrid_mfma_spmm_matches_oracle_csr_product(D, min_density):
    """the dense head of the adjacency on the matrix cores + the rest on the CSR kernels = the oracle's CSR product
    (reference LightGCN.py:139), layer sum included; bitwise reproducible"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    nU, nI = 6040, 3706
    uu, ii = ml1m_shaped_pairs()
    ptr = np.zeros(nU + 1, np.int64)
    np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
    rp, col, val = oracle.lightgcn_build_adj(nU, nI, ptr, ii.astype(np.int32))
    rng = np.random.RandomState(D)
    X = (rng.standard_normal((nU + nI, D)) * 0.1).astype(np.float32)
    A0 = (rng.standard_normal((nU + nI, D)) * 0.1).astype(np.float32)
    ref = oracle.spmm_csr(rp, col, val, X)
    hy = hip_ops.HybridSpmm(rp, col, val, nU, nI, dev, min_density=min_density)
    assert hy.enabled and hy.H >= 32 and hy.density > 0.5 * min_density
    Xd = torch.from_numpy(X).to(dev)
    acc = torch.from_numpy(A0).to(dev)
    Y = hy.apply(Xd, acc=acc)
    assert rel_err(Y.cpu().numpy(), ref) < TOL
    assert rel_err(acc.cpu().numpy(), A0 + ref) < TOL
    acc2 = torch.from_numpy(A0).to(dev)
    Y2 = hy.apply(Xd, acc=acc2)
    assert torch.equal(Y, Y2) and torch.equal(acc, acc2)
    # a graph without a dense head: the hybrid declines
    flat = hip_ops.HybridSpmm(rp, col, val, nU, nI, dev, min_density=0.9)
    assert not flat.enabled


def test_c3_lightgcn_with_the_mfma_product_matches_oracle():
    """LightGCN.predict + backward with --spmm_mfma 1 on the ml-1m-shaped graph against the oracle (as the CSR-only test)"""
    from whisprrec_amd.lightgcn import LightGCN
    dev = torch.device("cuda:0")
    nU, nI, D, B, L = 6040, 3706, 64, 2048, 2
    uu, ii = ml1m_shaped_pairs()
    ptr = np.zeros(nU + 1, np.int64)
    np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
    tcs = {u: set(ii[ptr[u]:ptr[u + 1]].tolist()) for u in range(nU)}
    corpus = host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []},
                                  "test": {"user_id": [], "item_id": []}}, tcs, {})
    args = argparse.Namespace(device=dev, model_path="/tmp/wr_lgcn_c3m.pt", buffer=1, num_neg=1, test_all=1, embedding_size=D,
                              gcn_layers=L, reg_weight=1e-5, spmm_mfma=1)
    m = LightGCN(args, corpus).to(dev)
    rng = np.random.RandomState(8)
    E0 = (rng.standard_normal((nU + nI, D)) * 0.1).astype(np.float32)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(E0[:nU]))
        m.item_embedding.weight.copy_(torch.from_numpy(E0[nU:]))
    rows = rng.randint(0, uu.size, B)
    u, p, n = uu[rows], ii[rows], rng.randint(1, nI, B)
    rp, col, val = oracle.lightgcn_build_adj(nU, nI, ptr, ii.astype(np.int32))
    loss_ref, g_ref = oracle.lightgcn_loss_grads(nU, nI, rp, col, val, E0, L, 1e-5, u, p, n)
    m.train()
    loss = m.predict({"user_id": torch.from_numpy(u).to(dev), "pos_item": torch.from_numpy(p).to(dev),
                      "neg_items": torch.from_numpy(n).to(dev)})
    assert m._hybrid is not None and m._hybrid.enabled
    assert abs(float(loss.detach()) - loss_ref) / abs(loss_ref) < TOL
    loss.backward()
    assert rel_err(m.user_embedding.weight.grad.cpu().numpy(), g_ref[:nU]) < TOL
    assert rel_err(m.item_embedding.weight.grad.cpu().numpy(), g_ref[nU:]) < TOL
