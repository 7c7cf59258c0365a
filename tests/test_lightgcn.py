"""LightGCN drop-in (whisprrec_amd/lightgcn.py) against what the reference produced (tests/golden/g4_lightgcn.npz):
adjacency (CPU), then on the GPU propagation, loss, dense gradients and 3 optimizer steps."""
import argparse

import numpy as np
import pytest
import torch

from conftest import rel_err
from whisprrec_amd import host

TOL = 1e-5


def _corpus(g4):
    ptr, idx = g4["clicked_ptr"], g4["clicked_idx"]
    nU, nI = g4["U0"].shape[0], g4["I0"].shape[0]
    tcs = {u: set(idx[ptr[u]:ptr[u + 1]].tolist()) for u in range(nU) if ptr[u + 1] > ptr[u]}
    return host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []},
                                "test": {"user_id": [], "item_id": []}}, tcs, {})


def _args(dev, **kw):
    base = dict(device=dev, model_path="/tmp/wr_lgcn.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, gcn_layers=2,
                reg_weight=1e-5)
    base.update(kw)
    return argparse.Namespace(**base)


def test_adjacency_matches_reference(g4):
    from whisprrec_amd.lightgcn import build_norm_adj_csr
    c = _corpus(g4)
    rp, col, val = build_norm_adj_csr(c.n_users, c.n_items, c.train_clicked_set)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    got = {(int(r), int(cc)): v for r, cc, v in zip(rows, col, val)}
    ref = {(int(r), int(cc)): v for r, cc, v in zip(g4["adj_row"], g4["adj_col"], g4["adj_val"])}
    assert set(got) == set(ref)
    assert max(abs(got[k] - ref[k]) / abs(ref[k]) for k in ref) < 1e-6
    assert all(np.all(np.diff(col[rp[i]:rp[i + 1]]) > 0) for i in range(len(rp) - 1))   # columns ascend
    assert rp[4 + 1] == rp[4]                                                            # isolated user: empty row


def test_flags_and_state_dict(g4):
    from whisprrec_amd.lightgcn import LightGCN
    p = argparse.ArgumentParser()
    LightGCN.parse_model_args(p)
    a = p.parse_args([])
    assert (a.embedding_size, a.gcn_layers, a.reg_weight) == (64, 2, 1e-5)
    m = LightGCN(_args(torch.device("cpu")), _corpus(g4))
    assert list(m.state_dict().keys()) == ["user_embedding.weight", "item_embedding.weight"]
    assert LightGCN.extra_log_args == ["embedding_size", "gcn_layers", "reg_weight"]


def _model(g4, dev, **kw):
    from whisprrec_amd.lightgcn import LightGCN
    m = LightGCN(_args(dev, **kw), _corpus(g4)).to(dev)
    with torch.no_grad():
        m.user_embedding.weight.copy_(torch.from_numpy(g4["U0"]))
        m.item_embedding.weight.copy_(torch.from_numpy(g4["I0"]))
    return m


def _batch(g4, dev):
    return {k: torch.from_numpy(g4[s]).to(dev) for k, s in (("user_id", "u"), ("pos_item", "p"), ("neg_items", "n"))}


@pytest.mark.gpu
def test_forward_loss_grads_match_reference(g4):
    dev = torch.device("cuda:0")
    m = _model(g4, dev)
    ua, ia = m.forward()
    assert rel_err(ua.cpu().numpy(), g4["user_all"]) < TOL and rel_err(ia.cpu().numpy(), g4["item_all"]) < TOL
    m.train()
    loss = m.predict(_batch(g4, dev))
    assert loss.shape == (1,)
    assert abs(float(loss.detach()) - float(g4["loss"][0])) / float(g4["loss"][0]) < TOL
    loss.backward()
    assert rel_err(m.user_embedding.weight.grad.cpu().numpy(), g4["gU"]) < TOL
    assert rel_err(m.item_embedding.weight.grad.cpu().numpy(), g4["gI"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("opt,tag", [("SGD", "sgd"), ("Adam", "adam")])
def test_three_steps_match_reference(g4, opt, tag):
    """zero_grad / predict / backward / step exactly as BaseRunner.fit issues them (BaseRunner.py:196-199)"""
    dev = torch.device("cuda:0")
    m = _model(g4, dev, optimizer=opt, lr=float(g4[tag + "_lr"][0]), l2=0.0)
    m.train()
    batch = _batch(g4, dev)
    for k in range(3):
        m.optimizer.zero_grad()
        loss = m.predict(batch)
        loss.backward()
        m.optimizer.step()
        assert abs(float(loss.detach()) - g4[tag + "_loss"][k]) / g4[tag + "_loss"][k] < TOL
    tol = TOL if opt == "SGD" else 1e-4
    assert rel_err(m.user_embedding.weight.detach().cpu().numpy(), g4[tag + "_U3"]) < tol
    assert rel_err(m.item_embedding.weight.detach().cpu().numpy(), g4[tag + "_I3"]) < tol


@pytest.mark.gpu
def test_torch_optimizer_path_and_full_predict(g4):
    """model.optimizer stays None without lr/optimizer in args -> a runner builds torch.optim on our dense grads"""
    dev = torch.device("cuda:0")
    m = _model(g4, dev)
    assert m.optimizer is None
    opt = torch.optim.SGD(m.parameters(), lr=float(g4["sgd_lr"][0]))
    m.train()
    for k in range(3):
        opt.zero_grad()
        loss = m.predict(_batch(g4, dev))
        loss.backward()
        opt.step()
    assert rel_err(m.user_embedding.weight.detach().cpu().numpy(), g4["sgd_U3"]) < TOL
    m.eval()
    s = m.full_predict({"user_id": torch.tensor([0, 3], device=dev)})
    ua, ia = m.forward()
    assert s.shape == (2, g4["I0"].shape[0]) and torch.allclose(s, ua[[0, 3]] @ ia.t(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("D,L", [(64, 32), (32, 8), (128, 64)])
def test_chunked_spmm_matches_oracle_on_power_law_graph(D, L):
    """hub rows (thousands of neighbours), empty rows and rows of exactly L non-zeros; the chunked kernel pair must equal
    the oracle's CSR product, be reproducible, and agree with the plain row-per-team kernel"""
    import oracle
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(D)
    N = 3000
    deg = np.minimum((rng.pareto(0.9, N) * 3).astype(np.int64), N - 1)
    deg[5] = 0; deg[6] = L; deg[7] = L + 1; deg[8] = 2500
    rp = np.zeros(N + 1, np.int64); np.cumsum(deg, out=rp[1:])
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg]).astype(np.int32)
    val = rng.standard_normal(len(col)).astype(np.float32)
    X = rng.standard_normal((N, D)).astype(np.float32)
    ref = oracle.spmm_csr(rp, col, val, X)
    cptr, crow = hip_ops.spmm_chunks(rp, L)
    assert int(crow.max()) == N - 1 and len(np.unique(crow.numpy())) == N and int(np.diff(cptr.numpy()).max()) <= L
    t = lambda a: torch.from_numpy(a).to(dev)
    acc = torch.ones(N, D, device=dev)
    Y = hip_ops.spmm_csr_chunked(cptr.to(dev), crow.to(dev), t(col), t(val), t(X), acc=acc)
    Y2 = hip_ops.spmm_csr_chunked(cptr.to(dev), crow.to(dev), t(col), t(val), t(X))
    assert torch.equal(Y, Y2)
    assert rel_err(Y.cpu().numpy(), ref) < TOL
    assert rel_err(acc.cpu().numpy(), ref + 1.0) < TOL
    assert not Y[5].any()
    Y3 = hip_ops.spmm_csr(t(rp), t(col), t(val), t(X))
    assert rel_err(Y3.cpu().numpy(), ref) < TOL
    # both combine forms (one level: the row's first chunk adds all the others; two: groups of 16 chunk slots first), whatever
    # spmm_chunks picked for this graph; each reproducible
    assert crow._wr_levels == (1 if -(-2500 // L) <= hip_ops.SPMM_ONE_LEVEL_MAX_CHUNKS else 2)
    for lv in (1, 2):
        Ya = hip_ops.spmm_csr_chunked(cptr.to(dev), crow.to(dev), t(col), t(val), t(X), levels=lv)
        Yb = hip_ops.spmm_csr_chunked(cptr.to(dev), crow.to(dev), t(col), t(val), t(X), levels=lv)
        assert torch.equal(Ya, Yb) and rel_err(Ya.cpu().numpy(), ref) < TOL


@pytest.mark.gpu
def test_hip_runner_batches_equal_reference_loop(g4):
    """HipRunner.fit drives LightGCN with batches sliced from the epoch's device columns; with the reference streams
    (--device_epoch_prep 0) they are the batches BaseRunner.fit's DataLoader collates sample by sample: same losses"""
    import random
    from whisprrec_amd import runner
    from whisprrec_amd.lightgcn import LightGCN
    dev = torch.device("cuda:0")
    corpus = _corpus(g4)
    rng = np.random.RandomState(0)
    tu, ti = [], []
    for uu, items in corpus.train_clicked_set.items():
        if len(items) > 40:
            continue                      # user 0 clicked every item: the reference's rejection sampler would never return
        for it in items:
            tu.append(uu); ti.append(it)
    corpus.data_df["train"] = {"user_id": np.asarray(tu), "item_id": np.asarray(ti)}
    out = []
    for cls in (runner.BaseRunner, runner.HipRunner):
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        args = _args(dev, optimizer="Adam", lr=2e-3, l2=0.0, epoch=1, check_epoch=1, test_epoch=-1, early_stop=10, batch_size=64,
                     eval_batch_size=256, num_workers=0, pin_memory=0, topk="10", metric="NDCG", device_epoch_prep=0)
        m = LightGCN(args, corpus).to(dev)
        ds = LightGCN.Dataset(m, corpus, "train")
        r = cls(args)
        out.append((r.fit(ds, epoch=1), r.fit(ds, epoch=2), m.user_embedding.weight.detach().clone()))
    assert abs(out[0][0] - out[1][0]) < 1e-6 and abs(out[0][1] - out[1][1]) < 1e-6
    assert torch.allclose(out[0][2], out[1][2], rtol=0, atol=1e-7)


@pytest.mark.gpu
def test_captured_step_graph_equals_eager_loop():
    """HipRunner replays a hipGraph of the whole LightGCN training step; losses and tables must equal the eager loop's bits"""
    import random
    from whisprrec_amd import runner
    from whisprrec_amd.lightgcn import LightGCN
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)
    nU, nI, B = 500, 300, 256
    sets, tu, ti = {}, [], []
    for uu in range(nU):
        items = np.unique(rng.randint(0, nI - 40, rng.randint(4, 30)))
        sets[uu] = set(items.tolist()); tu += [uu] * len(items); ti += items.tolist()
    frames = {"train": {"user_id": np.asarray(tu), "item_id": np.asarray(ti)},
              "dev": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)},
              "test": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)}}
    corpus = host.Corpus(nU, nI, frames, sets, {u: set() for u in sets})
    assert len(tu) >= 10 * B
    out = []
    for graphs in (0, 1):
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        args = _args(dev, optimizer="Adam", lr=2e-3, l2=0.0, epoch=1, check_epoch=1, test_epoch=-1, early_stop=10, batch_size=B,
                     eval_batch_size=256, num_workers=0, pin_memory=0, topk="10", metric="NDCG", device_epoch_prep=0, hip_graphs=graphs)
        m = LightGCN(args, corpus).to(dev)
        ds = LightGCN.Dataset(m, corpus, "train")
        r = runner.HipRunner(args)
        l1, l2 = r.fit(ds, epoch=1), r.fit(ds, epoch=2)
        assert (getattr(r, "_graph_cache", None) is not None) == bool(graphs)
        out.append((l1, l2, m.user_embedding.weight.detach().clone(), m.item_embedding.weight.detach().clone(), m.optimizer.t))
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][3], out[1][3])
    assert out[0][4] == out[1][4]


@pytest.mark.gpu
def test_pair_adam_launch_equals_two_single_launches():
    """wr_adam_dense_dev_pair (both embedding tables in one launch) leaves the bits of two wr_adam_dense_dev launches"""
    from whisprrec_amd import hip_ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    consts = ops.adam_consts(64, 1e-3, device=dev)
    step = torch.full((1,), 3, dtype=torch.int32, device=dev)
    for rows_a, rows_b, D in ((6040, 3706, 64), (7, 300001, 32), (1, 1, 4)):
        mk = lambda r: [torch.randn(r, D, device=dev, generator=g) for _ in range(2)] + \
                       [torch.rand(r, D, device=dev, generator=g), torch.randn(r, D, device=dev, generator=g)]
        a, b = mk(rows_a), mk(rows_b)
        a1, b1 = [t.clone() for t in a], [t.clone() for t in b]
        ops.adam_dense_dev_pair(a, b, consts, step, l2=1e-4)
        for t in (a1, b1):
            ops.adam_dense_dev(t[0], t[1], t[2], t[3], consts, step, l2=1e-4)
        for x, y in zip(a + b, a1 + b1):
            assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("D,L,N", [(64, 96, 9746), (32, 8, 3000), (128, 64, 3000), (96, 32, 2000)])
def test_fused_combine_equals_two_launch_product(D, L, N, monkeypatch):
    """wr_spmm_csr_chunked_fused (the last chunk of a cut row to finish adds the row's partials, one launch) leaves the bits
    of the chunk + combine launches: products repeated on the same counters, with the layer sum started from the input
    and scaled, hub rows of hundreds of chunks"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(D + L)
    deg = np.minimum((rng.pareto(0.8, N) * 20).astype(np.int64), N - 1)
    deg[3] = 0; deg[4] = L; deg[9] = L + 1; deg[11] = N - 1
    rp = np.zeros(N + 1, np.int64); np.cumsum(deg, out=rp[1:])
    col = np.concatenate([np.sort(rng.choice(N, d, replace=False)) for d in deg]).astype(np.int32)
    val = rng.standard_normal(len(col)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    cptr, crow = hip_ops.spmm_chunks(rp, L)
    cptr, crow, col, val = cptr.to(dev), crow.to(dev), t(col), t(val)
    X = t(rng.standard_normal((N, D)).astype(np.float32))
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(hip_ops, "SPMM_FUSED_COMBINE", fused)
        acc = torch.empty(N, D, device=dev)
        cur, ys = X, []
        for layer in range(3):                         # a propagation: the layer sum starts from the input, ends scaled
            cur = hip_ops.spmm_csr_chunked(cptr, crow, col, val, cur, acc=acc, levels=1, acc_from_x=layer == 0,
                                           acc_scale=0.25 if layer == 2 else 1.0)
            ys.append(cur)
        outs[fused] = ys + [acc, hip_ops.spmm_csr_chunked(cptr, crow, col, val, X, levels=1)]
    assert hasattr(crow, "_wr_fuse") and not crow._wr_fuse[1].any()      # the counters are back at zero
    for a, b in zip(outs[True], outs[False]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_native_step_equals_the_call_by_call_step(g4):
    """wr_lightgcn_step (predict + backward as ONE native call) runs the kernels of the call-by-call form in the same order on
    the same kind of buffers: loss and both gradient tables bit for bit — on the golden graph and on an ml-1m-shaped one"""
    from whisprrec_amd.lightgcn import LightGCN
    dev = torch.device("cuda:0")
    outs = []
    for native in (True, False):
        m = _model(g4, dev)
        m.NATIVE_STEP = native
        m.train()
        loss = m.predict(_batch(g4, dev))
        loss.backward()
        outs.append((loss.detach().clone(), m.user_embedding.weight.grad.clone(), m.item_embedding.weight.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert abs(float(outs[0][0]) - float(g4["loss"][0])) / float(g4["loss"][0]) < TOL
    assert rel_err(outs[0][1].cpu().numpy(), g4["gU"]) < TOL and rel_err(outs[0][2].cpu().numpy(), g4["gI"]) < TOL
    # an id out of range raises like nn.Embedding
    bad = _batch(g4, dev)
    bad["pos_item"] = bad["pos_item"].clone()
    bad["pos_item"][0] = 10 ** 6
    m = _model(g4, dev)
    with pytest.raises(IndexError):
        m.predict(bad)


@pytest.mark.gpu
def test_hybrid_product_small_graphs_and_padding_columns():
    """HybridSpmm (--spmm_mfma 1): a graph with no more users than one K split holds stays on the CSR kernels; a user
    count that is no multiple of the K split pads the item-side tiles with masked columns (exact zeros whatever the
    table holds).  Zero ENTRIES of the dense tiles are still multiplied: a non-finite value in a user row reaches every head
    item (0 x inf), which the CSR kernels would not do — one more reason the hybrid is off by default."""
    from whisprrec_amd import hip_ops
    from whisprrec_amd.lightgcn import build_norm_adj_csr
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)

    def graph(nU, nI, dens):
        item_dens = np.where(np.arange(nI) < nI // 2, dens, 0.02)        # head items, then a sparse tail for the CSR part
        clicked = {uu: set(np.nonzero(rng.rand(nI) < item_dens)[0].tolist()) for uu in range(nU)}
        clicked[nU - 1] = {nI - 1}                               # the last user rated one (tail) item only
        return build_norm_adj_csr(nU, nI, clicked)

    rp, col, val = graph(100, 256, 0.4)
    assert not hip_ops.HybridSpmm(rp, col, val, 100, 256, dev).enabled
    nU, nI, D = 300, 256, 64
    rp, col, val = graph(nU, nI, 0.4)
    hy = hip_ops.HybridSpmm(rp, col, val, nU, nI, dev)
    assert hy.enabled
    X = rng.standard_normal((nU + nI, D)).astype(np.float32)
    ref = oracle_spmm(rp, col, val, X)
    Y = hy.apply(torch.from_numpy(X).to(dev))
    assert rel_err(Y.cpu().numpy(), ref) < TOL


def oracle_spmm(rp, col, val, X):
    import oracle
    return oracle.spmm_csr(rp, col, val, X)
