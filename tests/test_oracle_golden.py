"""Pins the CPU oracle (oracle/) against vectors produced by the reference itself (tests/golden/*.npz).

Tolerance: 1e-5 relative (BASELINE.json north_star) — in practice the oracle sits within ~1e-6.
Index outputs (sampler) are compared bit-exactly.
"""
import numpy as np
import pytest

import oracle
from conftest import rel_err

TOL = 1e-5


def test_g1_forward(g1):
    pos, neg, coef, loss = oracle.bpr_fwd(g1["U0"], g1["I0"], g1["u0"], g1["p0"], g1["n0"])
    assert rel_err(pos, g1["pos_score"]) < TOL
    assert rel_err(neg, g1["neg_score"]) < TOL
    assert abs(loss - float(g1["loss0"][0])) / abs(float(g1["loss0"][0])) < TOL


def test_g1_dense_grads(g1):
    gU, gI, _ = oracle.bpr_dense_grads(g1["U0"], g1["I0"], g1["u0"], g1["p0"], g1["n0"])
    assert rel_err(gU, g1["gU"]) < TOL
    assert rel_err(gI, g1["gI"]) < TOL
    # rows not in the batch have exactly zero gradient
    untouched = np.setdiff1d(np.arange(g1["U0"].shape[0]), g1["u0"])
    assert np.all(gU[untouched] == 0)


@pytest.mark.parametrize("tag,nsteps", [("sgd", 5), ("sgdl2", 3)])
def test_g1_sgd_trajectory(g1, tag, nsteps):
    lr, l2 = g1[tag + "_hp"]
    U, I = g1["U0"].copy(), g1["I0"].copy()
    for k in range(nsteps):
        loss = oracle.bprmf_step_sgd(U, I, g1[f"u{k}"], g1[f"p{k}"], g1[f"n{k}"], lr, l2)
        assert abs(loss - g1[tag + "_loss"][k]) / abs(g1[tag + "_loss"][k]) < TOL
        assert rel_err(U, g1[f"{tag}_U{k + 1}"]) < TOL
        assert rel_err(I, g1[f"{tag}_I{k + 1}"]) < TOL


def test_g1_sparse_baseline_matches_dense(g1):
    lr, _ = g1["sgd_hp"]
    U, I = g1["U0"].copy(), g1["I0"].copy()
    bl = oracle.SparseSgdBaseline(U, I, 512)
    for k in range(5):
        bl.step(g1[f"u{k}"], g1[f"p{k}"], g1[f"n{k}"], lr)
    assert rel_err(U, g1["sgd_U5"]) < TOL
    assert rel_err(I, g1["sgd_I5"]) < TOL
    assert not bl.gU.any() and not bl.gI.any()


@pytest.mark.parametrize("tag", ["adam", "adaml2"])
def test_g1_adam_trajectory(g1, tag):
    lr, l2 = g1[tag + "_hp"]
    U, I = g1["U0"].copy(), g1["I0"].copy()
    mU, vU, mI, vI = (np.zeros_like(U), np.zeros_like(U), np.zeros_like(I), np.zeros_like(I))
    for k in range(3):
        gU, gI, loss = oracle.bpr_dense_grads(U, I, g1[f"u{k}"], g1[f"p{k}"], g1[f"n{k}"])
        oracle.adam_dense(U, gU, mU, vU, k + 1, lr, l2)
        oracle.adam_dense(I, gI, mI, vI, k + 1, lr, l2)
        assert abs(loss - g1[tag + "_loss"][k]) / abs(g1[tag + "_loss"][k]) < TOL
        assert rel_err(U, g1[f"{tag}_U{k + 1}"]) < TOL
        assert rel_err(I, g1[f"{tag}_I{k + 1}"]) < TOL


def _batches(g, tag):
    off = 0
    for bsz in g[tag + "_bsz"]:
        yield (g[tag + "_bu"][off:off + bsz], g[tag + "_bp"][off:off + bsz], g[tag + "_bn"][off:off + bsz])
        off += bsz


def test_g2_ml100k_sgd_curve(g2):
    """BaseRunner.fit loss curve on ml-100k, 2 epochs, incl. the short last batch (480 rows)."""
    lr, l2 = g2["sgd_hp"]
    U, I = g2["sgd_U0"].copy(), g2["sgd_I0"].copy()
    losses = []
    for (u, p, n) in _batches(g2, "sgd"):
        losses.append(oracle.bprmf_step_sgd(U, I, u, p, n, lr, l2))
    assert rel_err(np.asarray(losses), g2["sgd_loss"]) < TOL
    assert rel_err(U, g2["sgd_Uend"]) < TOL
    assert rel_err(I, g2["sgd_Iend"]) < TOL
    # fit() returns the unweighted mean of per-batch means (BaseRunner.py:201)
    assert abs(np.mean(losses[:33]) - g2["sgd_epoch_mean"][0]) < 1e-6


def test_g2_ml100k_adam_curve(g2):
    lr, l2 = g2["adam_hp"]
    U, I = g2["adam_U0"].copy(), g2["adam_I0"].copy()
    mU, vU, mI, vI = (np.zeros_like(U), np.zeros_like(U), np.zeros_like(I), np.zeros_like(I))
    losses = []
    for k, (u, p, n) in enumerate(_batches(g2, "adam")):
        gU, gI, loss = oracle.bpr_dense_grads(U, I, u, p, n)
        oracle.adam_dense(U, gU, mU, vU, k + 1, lr, l2)
        oracle.adam_dense(I, gI, mI, vI, k + 1, lr, l2)
        losses.append(loss)
    assert rel_err(np.asarray(losses), g2["adam_loss"]) < TOL
    # Adam divides by sqrt(v)+eps with v ~ 1e-12 on this data: compare on the table scale
    assert rel_err(U, g2["adam_Uend"]) < 1e-4
    assert rel_err(I, g2["adam_Iend"]) < 1e-4


def test_g2_batches_cover_train_set(g2):
    """Every epoch is a permutation of the train pairs (DataLoader(shuffle=True), BaseRunner.py:188-193)."""
    n = len(g2["train_user"])
    key = g2["train_user"].astype(np.int64) * 100000 + g2["train_item"]
    ep = g2["sgd_bu"][:n].astype(np.int64) * 100000 + g2["sgd_bp"][:n]
    assert np.array_equal(np.sort(key), np.sort(ep))


def test_g3_sampler_bit_exact(g3):
    np.random.seed(int(g3["seed"][0]))
    n1 = oracle.sample_negatives(g3["users"], int(g3["n_items"][0]), g3["clicked_ptr"], g3["clicked_idx"])
    n2 = oracle.sample_negatives(g3["users"], int(g3["n_items"][0]), g3["clicked_ptr"], g3["clicked_idx"])
    assert np.array_equal(n1, g3["neg_epoch1"])
    assert np.array_equal(n2, g3["neg_epoch2"])
    assert n1.min() >= 1  # item 0 is never a negative (BaseModel.py:168)


def test_g2_sampler_respects_train_set(g2):
    ptr, idx = g2["clicked_ptr"], g2["clicked_idx"]
    n = len(g2["train_user"])
    for tag in ("sgd",):
        u, neg = g2[tag + "_bu"][:n], g2[tag + "_bn"][:n]
        for uu, nn in zip(u[:5000], neg[:5000]):
            assert nn not in idx[ptr[uu]:ptr[uu + 1]]


def _g4_graph(g4):
    nU, nI = g4["U0"].shape[0], g4["I0"].shape[0]
    return nU, nI, oracle.lightgcn_build_adj(nU, nI, g4["clicked_ptr"], g4["clicked_idx"])


def test_g4_lightgcn_adjacency(g4):
    nU, nI, (rp, col, val) = _g4_graph(g4)
    N = nU + nI
    rows = np.repeat(np.arange(N), np.diff(rp))
    got = {(int(r), int(c)): v for r, c, v in zip(rows, col, val)}
    ref = {(int(r), int(c)): v for r, c, v in zip(g4["adj_row"], g4["adj_col"], g4["adj_val"])}
    assert set(got) == set(ref)
    assert max(abs(got[k] - ref[k]) / abs(ref[k]) for k in ref) < 1e-6


def test_g4_lightgcn_forward_and_grads(g4):
    nU, nI, (rp, col, val) = _g4_graph(g4)
    L, reg = int(g4["hp"][0]), float(g4["hp"][1])
    E0 = np.concatenate([g4["U0"], g4["I0"]])
    out = oracle.lightgcn_forward(rp, col, val, E0, L)
    assert rel_err(out[:nU], g4["user_all"]) < TOL
    assert rel_err(out[nU:], g4["item_all"]) < TOL
    loss, g = oracle.lightgcn_loss_grads(nU, nI, rp, col, val, E0, L, reg, g4["u"], g4["p"], g4["n"])
    assert abs(loss - float(g4["loss"][0])) / abs(float(g4["loss"][0])) < TOL
    assert rel_err(g[:nU], g4["gU"]) < TOL
    assert rel_err(g[nU:], g4["gI"]) < TOL


def test_g4_lightgcn_sgd_3steps(g4):
    nU, nI, (rp, col, val) = _g4_graph(g4)
    L, reg = int(g4["hp"][0]), float(g4["hp"][1])
    E = np.concatenate([g4["U0"], g4["I0"]]).copy()
    lr = float(g4["sgd_lr"][0])
    for k in range(3):
        loss, g = oracle.lightgcn_loss_grads(nU, nI, rp, col, val, E, L, reg, g4["u"], g4["p"], g4["n"])
        assert abs(loss - g4["sgd_loss"][k]) / abs(g4["sgd_loss"][k]) < TOL
        oracle.sgd_dense(E, g, lr, 0.0)
    assert rel_err(E[:nU], g4["sgd_U3"]) < TOL
    assert rel_err(E[nU:], g4["sgd_I3"]) < TOL


def test_g5_embedding_gather_scatter(g5):
    W = g5["W0"]
    assert np.array_equal(oracle.gather_rows(W, g5["hist"]), g5["gather_hist"])
    assert np.array_equal(oracle.gather_rows(W, g5["pos"]), g5["gather_pos"])
    idx = np.concatenate([g5["hist"].reshape(-1), g5["pos"]])
    src = np.concatenate([g5["g_his"].reshape(-1, W.shape[1]), g5["g_pos"]])
    G = oracle.scatter_add_rows(W.shape[0], idx, src, padding_idx=0)
    assert rel_err(G, g5["gW_slice"]) < TOL
    assert not G[0].any() and not g5["gW_full"][0].any()  # padding row gets no gradient (SASRec.py:60)


def test_g6_evaluate_method(g6):
    res = oracle.evaluate_method(g6["predictions"], [5, 10, 20], ["NDCG", "HR", "RECALL", "PRECISION"])
    for k, v in zip(g6["keys"], g6["values"]):
        assert abs(res[str(k)] - v) < 1e-12


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64, 1000, 4097, 100003])
def test_epoch_permutation_is_a_keyed_bijection(n):
    """oracle.epoch_permutation — the restatement of the device's epoch order (wr_epoch_shuffle; reference counterpart:
    DataLoader(shuffle=True), BaseRunner.py:188-193): every row exactly once, another key another order, subsets of
    positions evaluate to the same values, and no visible structure (what a shuffle is for)."""
    p = oracle.epoch_permutation(n, 3407, 1)
    assert p.dtype == np.int64 and np.array_equal(np.sort(p), np.arange(n))
    assert np.array_equal(p, oracle.epoch_permutation(n, 3407, 1))
    if n >= 64:
        probe = np.array([0, 1, n // 2, n - 1])
        assert np.array_equal(oracle.epoch_permutation_at(probe, n, 3407, 1), p[probe])
    if n >= 1000:
        q = oracle.epoch_permutation(n, 3407, 2)
        assert (p == q).mean() < 0.01 and (p == np.arange(n)).mean() < 0.01
        assert abs(np.corrcoef(np.arange(n), p)[0, 1]) < 0.1
        assert abs(np.abs(np.diff(p)).mean() / n - 1 / 3) < 0.03      # consecutive rows land far apart


@pytest.mark.parametrize("name", ["adagrad", "adadelta"])
def test_adagrad_adadelta_restatements_match_reference_trajectories(g1, g10, name):
    """oracle.adagrad_dense / adadelta_dense on the oracle's dense gradients vs torch.optim.Adagrad / Adadelta driven by the
    reference's BPRMF.predict + backward (tests/golden/make_golden.py g1 -> g10_optimizers.npz)"""
    lr = float(g10[name + "_hp"][0])
    U, I = g1["U0"].copy(), g1["I0"].copy()
    z = np.zeros_like
    su, si, au, ai = z(U), z(I), z(U), z(I)
    for k in range(5):
        gU, gI, loss = oracle.bpr_dense_grads(U, I, g1[f"u{k}"], g1[f"p{k}"], g1[f"n{k}"])
        assert abs(loss - g10[name + "_loss"][k]) / g10[name + "_loss"][k] < 1e-5
        if name == "adagrad":
            oracle.adagrad_dense(U, gU, su, lr); oracle.adagrad_dense(I, gI, si, lr)
        else:
            oracle.adadelta_dense(U, gU, su, au, lr); oracle.adadelta_dense(I, gI, si, ai, lr)
        assert rel_err(U, g10[f"{name}_U{k + 1}"]) < 1e-5 and rel_err(I, g10[f"{name}_I{k + 1}"]) < 1e-5
