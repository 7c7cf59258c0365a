"""GPU parity tests of the BPRMF hot path: HIP kernels (through the C-ABI) vs the golden vectors produced by the
reference and vs the CPU oracle on seeded inputs.

Bar (BASELINE.json north_star): indices bit-exact; fp32 embeddings / loss within 1e-5 relative.
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from whisprrec_amd import hip_ops
    info = hip_ops.device_info()
    assert info["wave_size"] == 64
    return hip_ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------------------------------------ forward
def test_fwd_matches_reference_golden(ops, dev, g1):
    out = ops.bpr_fwd(T(g1["U0"], dev), T(g1["I0"], dev), T(g1["u0"], dev), T(g1["p0"], dev), T(g1["n0"], dev), coef=True)
    assert rel_err(out["pos_score"].cpu().numpy(), g1["pos_score"]) < TOL
    assert rel_err(out["neg_score"].cpu().numpy(), g1["neg_score"]) < TOL
    assert abs(float(out["loss"]) - float(g1["loss0"][0])) / float(g1["loss0"][0]) < TOL
    _, _, coef, _ = oracle.bpr_fwd(g1["U0"], g1["I0"], g1["u0"], g1["p0"], g1["n0"])
    assert rel_err(out["coef"].cpu().numpy(), coef) < TOL


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64, 100, 128, 200, 256, 512])
def test_fwd_embedding_sizes(ops, dev, D):
    rng = np.random.RandomState(D)
    nU, nI, B = 301, 411, 1000
    U = (rng.standard_normal((nU, D)) / np.sqrt(D) * 2).astype(np.float32)
    I = (rng.standard_normal((nI, D)) / np.sqrt(D) * 2).astype(np.float32)
    u, p, n = rng.randint(0, nU, B), rng.randint(0, nI, B), rng.randint(0, nI, B)
    out = ops.bpr_fwd(T(U, dev), T(I, dev), T(u, dev), T(p, dev), T(n, dev))
    pos, neg, _, loss = oracle.bpr_fwd(U, I, u, p, n)
    assert rel_err(out["pos_score"].cpu().numpy(), pos) < TOL
    assert rel_err(out["neg_score"].cpu().numpy(), neg) < TOL
    assert abs(float(out["loss"]) - loss) / loss < TOL


def test_fwd_saturated_scores(ops, dev):
    """sigmoid saturation on both sides: loss -> -log(1e-10) for x << 0 (loss.py:38), coef -> 0 for |x| large."""
    D = 64
    U = np.zeros((2, D), np.float32); I = np.zeros((3, D), np.float32)
    U[0, 0] = 10.0; I[1, 0] = 10.0; I[2, 0] = -10.0
    u = np.array([0, 0, 1]); p = np.array([1, 2, 1]); n = np.array([2, 1, 2])
    out = ops.bpr_fwd(T(U, dev), T(I, dev), T(u, dev), T(p, dev), T(n, dev), coef=True)
    _, _, coef, loss = oracle.bpr_fwd(U, I, u, p, n)
    assert abs(float(out["loss"]) - loss) / loss < TOL
    assert np.allclose(out["coef"].cpu().numpy(), coef, rtol=1e-5, atol=1e-12)
    assert np.isfinite(float(out["loss"]))


# ------------------------------------------------------------------------------------------------ plan
def _check_plan(plan, u, p, n, B):
    N = len(u)
    tu, tp_raw, tn_raw = plan.tu.cpu().numpy(), plan.tp.cpu().numpy(), plan.tn.cpu().numpy()
    tp, tn = tp_raw & 0x7FFFFFFF, tn_raw & 0x7FFFFFFF
    torig = plan.torig.cpu().numpy()
    oi, osrc = plan.oc_item.cpu().numpy(), plan.oc_src.cpu().numpy()
    for k in range(plan.n_batches):
        lo, hi = k * B, min(N, (k + 1) * B)
        order = lo + np.argsort(u[lo:hi], kind="stable")         # stable sort by user inside the batch
        assert np.array_equal(torig[lo:hi], order)                # bit-exact permutation
        assert np.array_equal(tu[lo:hi], u[order]) and np.array_equal(tp[lo:hi], p[order]) and np.array_equal(tn[lo:hi], n[order])
        Bk = hi - lo
        items = np.concatenate([tp[lo:hi], tn[lo:hi]])
        src = np.concatenate([np.arange(Bk) << 1, (np.arange(Bk) << 1) | 1])
        o2 = np.argsort(items, kind="stable")
        assert np.array_equal(oi[2 * lo:2 * lo + 2 * Bk], items[o2])
        assert np.array_equal(osrc[2 * lo:2 * lo + 2 * Bk], src[o2])
        # bit 31 of tp/tn: the item row has more than one occurrence in this batch
        uniq, inv, cnt = np.unique(items, return_inverse=True, return_counts=True)
        assert np.array_equal(tp_raw[lo:hi] < 0, cnt[inv[:Bk]] > 1)
        assert np.array_equal(tn_raw[lo:hi] < 0, cnt[inv[Bk:]] > 1)


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
def test_plan_is_a_stable_sort(ops, dev, dtype):
    rng = np.random.RandomState(7)
    nU, nI, N, B = 50, 70, 1000, 128  # 8 batches, last one short (104)
    u = rng.randint(0, nU, N).astype(dtype); p = rng.randint(0, nI, N).astype(dtype); n = rng.randint(0, nI, N).astype(dtype)
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True)
    assert plan.n_batches == 8 and plan.batch_len(7) == 104
    _check_plan(plan, u, p, n, B)


@pytest.mark.parametrize("shape", [(50, 70, 1000, 128), (1000, 3000, 20000, 4096), (1 << 20, 1 << 20, 3 * 65536 + 777, 65536),
                                   (943, 1574, 66016, 2048), (5, 7, 300, 64), (10_000_000, 12_345_678, 2 * 65536 + 5, 65536),
                                   (3_000_000, 1_000_000, 2 * 262144 + 1000, 262144),
                                   (1_000_000, 1_000_000, 2 * 1048576 + 7, 1048576)])
def test_fast_and_generic_builders_agree_bitwise(ops, dev, shape):
    """the hand-written bucket/LDS-sort builder must emit exactly the generic radix-sort builder's arrays"""
    nU, nI, N, B = shape
    rng = np.random.RandomState(N)
    u = rng.randint(0, nU, N).astype(np.int32); p = rng.randint(0, nI, N).astype(np.int32); n = rng.randint(0, nI, N).astype(np.int32)
    a = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True, builder="generic")
    try:
        b = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True, builder="fast")
    except Exception as e:  # overflow is legal for tiny id ranges: then "auto" must fall back
        assert "overflow" in str(e)
        b = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True, builder="auto")
        assert b.builder == "generic"
        return
    assert a.builder == "generic" and b.builder == "fast"
    for name in ("tu", "tp", "tn", "torig", "oc_item", "oc_src"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name


def test_fast_builder_overflow_falls_back(ops, dev):
    """all triplets on one user / one item: every pair lands in one bucket -> overflow -> generic builder"""
    N, B = 8192, 4096
    u = np.zeros(N, np.int32); p = np.full(N, 3, np.int32); n = np.full(N, 5, np.int32)
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, 100000, 100000, keep_orig=True, builder="auto")
    assert plan.builder == "generic"
    _check_plan(plan, u, p, n, B)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_auto_builder_on_skewed_ids(ops, dev, seed):
    """power-law users and items, a short last batch, hot rows: whichever builder "auto" ends up with, the plan must be
    the stable sort (memory safety of the fast builder after a bucket overflow is part of what this exercises)"""
    rng = np.random.RandomState(seed)
    nU, nI, B = [(50000, 80000, 4096), (1 << 20, 1 << 20, 16384), (3000, 100, 2048), (200000, 200000, 8192)][seed]
    N = 5 * B + 123
    u = np.minimum((rng.pareto(1.1, N) * 20).astype(np.int64), nU - 1).astype(np.int32)
    p = np.minimum((rng.pareto(0.9, N) * 5).astype(np.int64), nI - 1).astype(np.int32)
    n = rng.randint(1, nI, N).astype(np.int32)
    n[::7] = 1
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True, builder="auto")
    assert plan.builder in ("fast", "generic")
    _check_plan(plan, u, p, n, B)


def test_fast_builder_rejects_out_of_range(ops, dev):
    u = np.array([0, 1, 5] * 100, np.int32); p = np.array([0, 1, 2] * 100, np.int32); n = np.array([1, 1, 1] * 100, np.int32)
    with pytest.raises(IndexError):
        ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), 64, 5, 3, builder="auto")


def test_plan_wide_keys(ops, dev):
    """composite (batch,row) keys beyond 32 bits take the 64-bit path"""
    rng = np.random.RandomState(8)
    nU, nI, N, B = (1 << 30) + 5, (1 << 29) + 3, 4096, 4   # 1024 batches x 31 bits
    u = rng.randint(0, nU, N); p = rng.randint(0, nI, N); n = rng.randint(0, nI, N)
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, keep_orig=True)
    _check_plan(plan, u, p, n, B)


def test_plan_rejects_out_of_range(ops, dev):
    u = np.array([0, 1, 5]); p = np.array([0, 1, 2]); n = np.array([1, 1, 1])
    with pytest.raises(IndexError):
        ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), 4, 5, 3)
    with pytest.raises(IndexError):
        ops.BatchPlan(T(p, dev), T(u, dev), T(n, dev), 4, 5, 3)


# ------------------------------------------------------------------------------------------------ step
def test_sgd_trajectory_matches_reference_golden(ops, dev, g1):
    lr, l2 = g1["sgd_hp"]
    tabs = ops.BprmfTables(T(g1["U0"], dev), T(g1["I0"], dev))
    for k in range(5):
        plan = ops.BatchPlan(T(g1[f"u{k}"], dev), T(g1[f"p{k}"], dev), T(g1[f"n{k}"], dev), 512, 97, 131)
        loss = tabs.step_sgd(plan, 0, float(lr), float(l2))
        assert abs(float(loss) - g1["sgd_loss"][k]) / g1["sgd_loss"][k] < TOL
        assert rel_err(tabs.U.cpu().numpy(), g1[f"sgd_U{k + 1}"]) < TOL
        assert rel_err(tabs.I.cpu().numpy(), g1[f"sgd_I{k + 1}"]) < TOL


def test_sgd_weight_decay_matches_reference_golden(ops, dev, g1):
    lr, l2 = g1["sgdl2_hp"]
    tabs = ops.BprmfTables(T(g1["U0"], dev), T(g1["I0"], dev))
    for k in range(3):
        plan = ops.BatchPlan(T(g1[f"u{k}"], dev), T(g1[f"p{k}"], dev), T(g1[f"n{k}"], dev), 512, 97, 131)
        loss = tabs.step_sgd(plan, 0, float(lr), float(l2))
        assert abs(float(loss) - g1["sgdl2_loss"][k]) / g1["sgdl2_loss"][k] < TOL
        assert rel_err(tabs.U.cpu().numpy(), g1[f"sgdl2_U{k + 1}"]) < TOL
        assert rel_err(tabs.I.cpu().numpy(), g1[f"sgdl2_I{k + 1}"]) < TOL


def test_dense_grads_match_reference_golden(ops, dev, g1):
    tabs = ops.BprmfTables(T(g1["U0"], dev), T(g1["I0"], dev))
    plan = ops.BatchPlan(T(g1["u0"], dev), T(g1["p0"], dev), T(g1["n0"], dev), 512, 97, 131)
    gU = torch.zeros_like(tabs.U); gI = torch.zeros_like(tabs.I)
    loss, sid = tabs.grads(plan, 0, gU, gI)
    assert abs(float(loss) - float(g1["loss0"][0])) / float(g1["loss0"][0]) < TOL
    assert rel_err(gU.cpu().numpy(), g1["gU"]) < TOL
    assert rel_err(gI.cpu().numpy(), g1["gI"]) < TOL
    # stamps mark exactly the rows of the batch
    su = tabs.stamp_u.cpu().numpy() == sid
    assert np.array_equal(np.nonzero(su)[0], np.unique(g1["u0"]))
    si = tabs.stamp_i.cpu().numpy() == sid
    assert np.array_equal(np.nonzero(si)[0], np.unique(np.concatenate([g1["p0"], g1["n0"]])))
    # tables untouched in gradient mode
    assert np.array_equal(tabs.U.cpu().numpy(), g1["U0"]) and np.array_equal(tabs.I.cpu().numpy(), g1["I0"])


@pytest.mark.parametrize("tag", ["adam", "adaml2"])
def test_adam_trajectory_matches_reference_golden(ops, dev, g1, tag):
    lr, l2 = g1[tag + "_hp"]
    tabs = ops.BprmfTables(T(g1["U0"], dev), T(g1["I0"], dev))
    gU = torch.zeros_like(tabs.U); gI = torch.zeros_like(tabs.I)
    mU, vU, mI, vI = (torch.zeros_like(tabs.U), torch.zeros_like(tabs.U), torch.zeros_like(tabs.I), torch.zeros_like(tabs.I))
    for k in range(3):
        plan = ops.BatchPlan(T(g1[f"u{k}"], dev), T(g1[f"p{k}"], dev), T(g1[f"n{k}"], dev), 512, 97, 131)
        loss, sid = tabs.grads(plan, 0, gU, gI)
        ops.adam_dense(tabs.U, mU, vU, gU, k + 1, float(lr), float(l2), stamp=tabs.stamp_u, step_id=sid)
        ops.adam_dense(tabs.I, mI, vI, gI, k + 1, float(lr), float(l2), stamp=tabs.stamp_i, step_id=sid)
        assert abs(float(loss) - g1[tag + "_loss"][k]) / g1[tag + "_loss"][k] < TOL
        assert rel_err(tabs.U.cpu().numpy(), g1[f"{tag}_U{k + 1}"]) < TOL
        assert rel_err(tabs.I.cpu().numpy(), g1[f"{tag}_I{k + 1}"]) < TOL


def test_ml100k_sgd_curve_matches_reference_golden(ops, dev, g2):
    """BaseRunner.fit over ml-100k, 2 epochs x 33 batches (last batch 480 rows), via the native multi-step loop."""
    lr, _ = g2["sgd_hp"]
    nU, nI = int(g2["n_users"][0]), int(g2["n_items"][0])
    tabs = ops.BprmfTables(T(g2["sgd_U0"], dev), T(g2["sgd_I0"], dev))
    n_ep = len(g2["train_user"])
    losses = []
    for ep in range(2):
        sl = slice(ep * n_ep, (ep + 1) * n_ep)
        plan = ops.BatchPlan(T(g2["sgd_bu"][sl], dev), T(g2["sgd_bp"][sl], dev), T(g2["sgd_bn"][sl], dev), 2048, nU, nI)
        assert plan.n_batches == 33 and plan.batch_len(32) == 480
        losses.append(tabs.run_sgd(plan, 0, plan.n_batches, float(lr)).cpu().numpy())
    losses = np.concatenate(losses)
    assert rel_err(losses, g2["sgd_loss"]) < TOL
    assert rel_err(tabs.U.cpu().numpy(), g2["sgd_Uend"]) < TOL
    assert rel_err(tabs.I.cpu().numpy(), g2["sgd_Iend"]) < TOL
    assert abs(float(np.mean(losses[:33])) - g2["sgd_epoch_mean"][0]) < 1e-6  # fit() return value (BaseRunner.py:201)


def test_ml100k_adam_curve_matches_reference_golden(ops, dev, g2):
    lr, l2 = g2["adam_hp"]
    nU, nI = int(g2["n_users"][0]), int(g2["n_items"][0])
    tabs = ops.BprmfTables(T(g2["adam_U0"], dev), T(g2["adam_I0"], dev))
    plan = ops.BatchPlan(T(g2["adam_bu"], dev), T(g2["adam_bp"], dev), T(g2["adam_bn"], dev), 2048, nU, nI)
    gU = torch.zeros_like(tabs.U); gI = torch.zeros_like(tabs.I)
    mU, vU, mI, vI = (torch.zeros_like(tabs.U), torch.zeros_like(tabs.U), torch.zeros_like(tabs.I), torch.zeros_like(tabs.I))
    losses = []
    for k in range(plan.n_batches):
        loss, sid = tabs.grads(plan, k, gU, gI)
        ops.adam_dense(tabs.U, mU, vU, gU, k + 1, float(lr), float(l2), stamp=tabs.stamp_u, step_id=sid)
        ops.adam_dense(tabs.I, mI, vI, gI, k + 1, float(lr), float(l2), stamp=tabs.stamp_i, step_id=sid)
        losses.append(loss)
    losses = torch.stack(losses).cpu().numpy()
    assert rel_err(losses, g2["adam_loss"]) < TOL
    assert rel_err(tabs.U.cpu().numpy(), g2["adam_Uend"]) < 1e-4  # same bar as the oracle test (v ~ 1e-12 under the sqrt)
    assert rel_err(tabs.I.cpu().numpy(), g2["adam_Iend"]) < 1e-4


@pytest.mark.parametrize("D", [4, 8, 32, 64, 96, 128, 256, 500, 1024])   # 4 and 1024: the smallest and largest supported rows
def test_step_vs_oracle_embedding_sizes(ops, dev, D):
    rng = np.random.RandomState(100 + D)
    nU, nI, B = 203, 157, 2048   # heavy duplication: every row appears ~10x
    U = (rng.standard_normal((nU, D)) / np.sqrt(D) * 3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) / np.sqrt(D) * 3).astype(np.float32)
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    Uo, Io = U.copy(), I.copy()
    for k in range(3):
        u, p, n = rng.randint(0, nU, B), rng.randint(0, nI, B), rng.randint(0, nI, B)
        plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
        loss = tabs.step_sgd(plan, 0, 0.2)
        lo = oracle.bprmf_step_sgd(Uo, Io, u, p, n, 0.2, 0.0)
        assert abs(float(loss) - lo) / lo < TOL
    assert rel_err(tabs.U.cpu().numpy(), Uo) < TOL
    assert rel_err(tabs.I.cpu().numpy(), Io) < TOL


def test_single_hot_row_and_untouched_rows(ops, dev):
    """One user and one item take every occurrence (longest possible runs); every other row must stay bit-identical
    under SGD with l2 = 0 (the only exactly-sparse case)."""
    rng = np.random.RandomState(5)
    nU, nI, D, B = 64, 64, 64, 4096
    U = rng.standard_normal((nU, D)).astype(np.float32) * 0.2
    I = rng.standard_normal((nI, D)).astype(np.float32) * 0.2
    u = np.full(B, 3); p = np.full(B, 9); n = rng.randint(10, 20, B)
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
    loss = tabs.step_sgd(plan, 0, 0.5)
    Uo, Io = U.copy(), I.copy()
    lo = oracle.bprmf_step_sgd(Uo, Io, u, p, n, 0.5, 0.0)
    assert abs(float(loss) - lo) / lo < TOL
    Ug, Ig = tabs.U.cpu().numpy(), tabs.I.cpu().numpy()
    assert rel_err(Ug, Uo) < TOL and rel_err(Ig, Io) < TOL
    keep_u = np.setdiff1d(np.arange(nU), [3]); keep_i = np.setdiff1d(np.arange(nI), np.concatenate([[9], np.arange(10, 20)]))
    assert np.array_equal(Ug[keep_u], U[keep_u]) and np.array_equal(Ig[keep_i], I[keep_i])


@pytest.mark.parametrize("D", [64, 32, 128])
def test_hot_rows_power_law_items(ops, dev, D):
    """Power-law positives: a handful of item rows take hundreds to thousands of occurrences per batch.  Those runs are cut
    into pieces by the plan and summed by many workgroups (bprmf_item_hot_*); result must match the oracle, be bitwise
    reproducible, and the gradient-emitting mode must agree too."""
    rng = np.random.RandomState(D)
    nU, nI, B = 20000, 30000, 16384
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    N = 2 * B + 1000                                            # third batch is short
    u = rng.randint(0, nU, N)
    p = np.minimum((rng.pareto(0.8, N) * 2).astype(np.int64), nI - 1)
    n = rng.randint(1, nI, N)
    n[::5] = 7                                                  # a hot NEGATIVE row as well (sign handling)
    runs = []
    for rep in range(2):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
        assert plan.hot is not None and int(plan.hot["counts_host"].view(-1, 4)[:, 1].min()) >= 2   # every batch has hot runs
        losses = tabs.run_sgd(plan, 0, plan.n_batches, 0.2).cpu().numpy()
        runs.append((tabs.U.cpu().numpy().copy(), tabs.I.cpu().numpy().copy(), losses))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], 0.2, 0.0)
           for k in range(3)]
    assert rel_err(runs[0][2], np.asarray(ref)) < TOL
    assert rel_err(runs[0][0], Uo) < TOL and rel_err(runs[0][1], Io) < TOL
    # single-step entry point and gradient mode on the first batch
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    plan = ops.BatchPlan(T(u[:B], dev), T(p[:B], dev), T(n[:B], dev), B, nU, nI)
    gU = torch.zeros_like(tabs.U); gI = torch.zeros_like(tabs.I)
    tabs.grads(plan, 0, gU, gI)
    rU, rI, _ = oracle.bpr_dense_grads(U, I, u[:B], p[:B], n[:B])
    assert rel_err(gU.cpu().numpy(), rU) < TOL and rel_err(gI.cpu().numpy(), rI) < TOL
    tabs.step_sgd(plan, 0, 0.2)
    U1, I1 = U.copy(), I.copy()
    oracle.bprmf_step_sgd(U1, I1, u[:B], p[:B], n[:B], 0.2, 0.0)
    assert rel_err(tabs.I.cpu().numpy(), I1) < TOL


@pytest.mark.parametrize("case", range(12))
def test_random_shapes_vs_oracle(ops, dev, case):
    """seeded random table sizes, embedding sizes, batch sizes and id distributions (uniform, clustered, power-law, with a
    short last batch) through plan + multi-step driver, against the oracle; both plan builders must give the same bits"""
    rng = np.random.RandomState(1000 + case)
    D = int(rng.choice([4, 8, 12, 16, 32, 48, 64, 64, 96, 128, 256]))
    nU, nI = int(rng.randint(3, 5000)), int(rng.randint(3, 5000))
    B = int(rng.choice([7, 64, 100, 1000, 2048, 5000]))
    steps = int(rng.randint(1, 4))
    N = B * steps - int(rng.randint(0, B // 2 + 1))
    kind = case % 3
    if kind == 0:
        u, p, n = rng.randint(0, nU, N), rng.randint(0, nI, N), rng.randint(0, nI, N)
    elif kind == 1:
        u = rng.randint(0, max(1, nU // 50), N); p = rng.randint(0, max(1, nI // 50), N); n = rng.randint(0, nI, N)
    else:
        u = np.minimum((rng.pareto(1.0, N) * 3).astype(np.int64), nU - 1)
        p = np.minimum((rng.pareto(0.7, N) * 2).astype(np.int64), nI - 1)
        n = rng.randint(0, nI, N)
    U = (rng.standard_normal((nU, D)) / np.sqrt(D) * 2).astype(np.float32)
    I = (rng.standard_normal((nI, D)) / np.sqrt(D) * 2).astype(np.float32)
    lr = float(rng.choice([0.01, 0.3]))
    out = []
    for builder in ("generic", "auto"):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, builder=builder)
        losses = tabs.run_sgd(plan, 0, plan.n_batches, lr).cpu().numpy()
        out.append((tabs.U.cpu().numpy(), tabs.I.cpu().numpy(), losses))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
           for k in range((N + B - 1) // B)]
    assert rel_err(out[0][2], np.asarray(ref)) < TOL
    assert rel_err(out[0][0], Uo) < TOL and rel_err(out[0][1], Io) < TOL


@pytest.mark.parametrize("D", [64, 16])
def test_hot_rows_power_law_users(ops, dev, D):
    """Power-law USERS: a few users own hundreds to thousands of the batch's triplets (their runs are cut into pieces and
    worked on by many workgroups, bprmf_user_hot_*), combined with hot items.  Oracle parity, bitwise reproducibility,
    gradient mode."""
    rng = np.random.RandomState(40 + D)
    nU, nI, B = 30000, 20000, 16384
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    N = 2 * B + 700
    u = np.minimum((rng.pareto(0.8, N) * 2).astype(np.int64), nU - 1)
    p = np.minimum((rng.pareto(1.0, N) * 3).astype(np.int64), nI - 1)
    n = rng.randint(1, nI, N)
    runs = []
    for rep in range(2):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
        c = plan.hot["counts_host"].view(-1, 4)
        assert int(c[:, 3].min()) >= 2 and int(c[:, 1].min()) >= 1          # hot users and hot items in every batch
        losses = tabs.run_sgd(plan, 0, plan.n_batches, 0.2).cpu().numpy()
        runs.append((tabs.U.cpu().numpy().copy(), tabs.I.cpu().numpy().copy(), losses))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], 0.2, 0.0)
           for k in range(3)]
    assert rel_err(runs[0][2], np.asarray(ref)) < TOL
    assert rel_err(runs[0][0], Uo) < TOL and rel_err(runs[0][1], Io) < TOL
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    plan = ops.BatchPlan(T(u[:B], dev), T(p[:B], dev), T(n[:B], dev), B, nU, nI)
    gU = torch.zeros_like(tabs.U); gI = torch.zeros_like(tabs.I)
    loss, _ = tabs.grads(plan, 0, gU, gI)
    rU, rI, rl = oracle.bpr_dense_grads(U, I, u[:B], p[:B], n[:B])
    assert rel_err(gU.cpu().numpy(), rU) < TOL and rel_err(gI.cpu().numpy(), rI) < TOL and abs(float(loss) - rl) / rl < TOL


def test_step_is_bitwise_reproducible(ops, dev):
    rng = np.random.RandomState(77)
    nU, nI, D, B = 5000, 3000, 64, 16384
    U = rng.standard_normal((nU, D)).astype(np.float32) * 0.1
    I = rng.standard_normal((nI, D)).astype(np.float32) * 0.1
    u, p, n = rng.randint(0, nU, B), rng.randint(0, nI, B), rng.randint(0, nI, B)
    res = []
    for _ in range(2):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
        loss = tabs.step_sgd(plan, 0, 0.3)
        res.append((tabs.U.cpu().numpy().copy(), tabs.I.cpu().numpy().copy(), float(loss)))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]


def test_full_size_step_c2(ops, dev):
    """BASELINE.json configs[1] shapes: 1M x 1M tables, D=64, B=65,536 uniform ids.  One step against the oracle's
    sparse restatement, plus size-independent properties: untouched rows bit-identical, loss equals the forward-only
    kernel's loss, second run bitwise identical."""
    nU = nI = 1_000_000
    D, B = 64, 65536
    g = torch.Generator(device="cpu").manual_seed(3407)
    U = torch.randn(nU, D, generator=g) * 0.05
    I = torch.randn(nI, D, generator=g) * 0.05
    rng = np.random.RandomState(3407)
    u, p, n = rng.randint(0, nU, B), rng.randint(0, nI, B), rng.randint(1, nI, B)
    Ud, Id = U.to(dev), I.to(dev)
    fwd = ops.bpr_fwd(Ud, Id, T(u, dev), T(p, dev), T(n, dev), scores=False)
    tabs = ops.BprmfTables(Ud.clone(), Id.clone())
    plan = ops.BatchPlan(T(u.astype(np.int32), dev), T(p.astype(np.int32), dev), T(n.astype(np.int32), dev), B, nU, nI)
    loss = tabs.step_sgd(plan, 0, 0.05)
    assert abs(float(loss) - float(fwd["loss"])) / float(fwd["loss"]) < 1e-6
    Uo, Io = U.numpy().copy(), I.numpy().copy()
    bl = oracle.SparseSgdBaseline(Uo, Io, B)
    lo = bl.step(u, p, n, 0.05)
    assert abs(float(loss) - lo) / lo < TOL
    Ug, Ig = tabs.U.cpu().numpy(), tabs.I.cpu().numpy()
    assert rel_err(Ug, Uo) < TOL and rel_err(Ig, Io) < TOL
    mask_u = np.ones(nU, bool); mask_u[u] = False
    mask_i = np.ones(nI, bool); mask_i[p] = False; mask_i[n] = False
    assert np.array_equal(Ug[mask_u], U.numpy()[mask_u]) and np.array_equal(Ig[mask_i], I.numpy()[mask_i])
    tabs2 = ops.BprmfTables(Ud.clone(), Id.clone())
    tabs2.step_sgd(plan, 0, 0.05)
    assert torch.equal(tabs2.U, tabs.U) and torch.equal(tabs2.I, tabs.I)


@pytest.mark.parametrize("hot", [False, True])
def test_long_runs_with_and_without_hot_list(ops, dev, hot):
    """rows with 33..2000 occurrences in one batch: with the plan's hot-run list (pieces + combine) and without it (a plan
    built with hot=False, as LightGCN's backward does: the item phase must then walk the whole run itself)"""
    rng = np.random.RandomState(77)
    nU, nI, D, B = 500, 400, 64, 4096
    U = (rng.standard_normal((nU, D)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.2).astype(np.float32)
    u = rng.randint(0, nU, B); p = rng.randint(0, nI, B); n = rng.randint(0, nI, B)
    p[:2000] = 7; n[2000:2040] = 9; p[2100:2133] = 11; u[:600] = 3          # runs of 2000+, 40+, 33+ and a hot user
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, hot=hot)
    assert (plan.hot is not None) == hot
    loss = tabs.step_sgd(plan, 0, 0.2)
    Uo, Io = U.copy(), I.copy()
    lo = oracle.bprmf_step_sgd(Uo, Io, u, p, n, 0.2, 0.0)
    assert abs(float(loss) - lo) / lo < TOL
    assert rel_err(tabs.U.cpu().numpy(), Uo) < TOL and rel_err(tabs.I.cpu().numpy(), Io) < TOL
    gU = torch.zeros(nU, D, device=dev); gI = torch.zeros(nI, D, device=dev)
    tabs2 = ops.BprmfTables(T(U, dev), T(I, dev))
    tabs2.grads(plan, 0, gU, gI)
    rU, rI, _ = oracle.bpr_dense_grads(U, I, u, p, n)
    assert rel_err(gU.cpu().numpy(), rU) < TOL and rel_err(gI.cpu().numpy(), rI) < TOL


@pytest.mark.parametrize("zipf", [False, True])
def test_phase_events_attached_to_dispatches(ops, dev, zipf):
    """wr_bprmf_run_sgd's timing hooks (include/whisprrec_hip.h): four events per step attached to the kernels as start / stop
    events, any of them None; they must not change the result, and the intervals must be ordered and positive — also when the
    phases have several kernels (hot rows: pieces + combine)."""
    rng = np.random.RandomState(5)
    nU, nI, D, B, NB = 3000, 2000, 64, 4096, 3
    U = (rng.standard_normal((nU, D)) * 0.1).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.1).astype(np.float32)
    u, p, n = rng.randint(0, nU, NB * B), rng.randint(0, nI, NB * B), rng.randint(1, nI, NB * B)
    if zipf:
        p[: NB * B // 2] = 7          # a hot item row in every batch
        u[::3] = 11                   # and a hot user row
    plan = ops.BatchPlan(T(u.astype(np.int32), dev), T(p.astype(np.int32), dev), T(n.astype(np.int32), dev), B, nU, nI)
    assert (plan.hot is not None) == zipf
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    l_ref = ref.run_sgd(plan, 0, NB, 0.1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * NB)]
    ev[4] = None                      # holes are allowed
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    l_ev = tabs.run_sgd(plan, 0, NB, 0.1, phase_events=ev)
    torch.cuda.synchronize()
    assert torch.equal(tabs.U, ref.U) and torch.equal(tabs.I, ref.I) and torch.equal(l_ev, l_ref)
    for k in range(NB):
        if ev[4 * k] is not None:
            assert ev[4 * k].elapsed_time(ev[4 * k + 1]) > 0             # user phase
        assert ev[4 * k + 2].elapsed_time(ev[4 * k + 3]) > 0             # item phase
        assert ev[4 * k + 1].elapsed_time(ev[4 * k + 3]) > 0             # user phase ends before the item phase does
    with pytest.raises(ValueError):
        tabs.run_sgd(plan, 0, NB, 0.1, phase_events=ev[:3 * NB])


def _skewed_epoch(rng, nU, nI, N, user_a=1.1, item_a=0.9):
    """power-law users and items whose popular ids sit at the low end of the id space (worst case for row-range buckets),
    uniform negatives as the reference draws them (BaseModel.py:168)"""
    u = np.minimum((rng.pareto(user_a, N) * 20).astype(np.int64), nU - 1)
    p = np.minimum((rng.pareto(item_a, N) * 5).astype(np.int64), nI - 1)
    n = rng.randint(1, nI, N)
    perm = rng.permutation(N)            # batches are random samples of the epoch, as after DataLoader(shuffle=True)
    return u[perm], p[perm], n[perm]


@pytest.mark.parametrize("dtype", [np.int32, np.int64])
@pytest.mark.parametrize("shape", [(50000, 80000, 4096, 20), (1 << 20, 1 << 20, 65536, 6), (3000, 1574, 2048, 33),
                                   (200000, 200000, 16384, 9), (1000, 1000, 65536, 3)])
def test_mapped_builder_agrees_with_generic_bitwise_on_skewed_ids(ops, dev, shape, dtype):
    """hand-written builder with a bucket map (load-balanced row ranges, heavy rows split by position) against the radix-sort
    builder on power-law ids, short last batch included: the same arrays bit for bit, without falling back"""
    nU, nI, B, nb = shape
    N = nb * B - B // 3
    rng = np.random.RandomState(B + nb)
    u, p, n = (x.astype(dtype) for x in _skewed_epoch(rng, nU, nI, N))
    ud, pd, nd = T(u, dev), T(p, dev), T(n, dev)
    bmap = ops.BucketMap(ud, pd, nU, nI, B)
    a = ops.BatchPlan(ud, pd, nd, B, nU, nI, keep_orig=True, builder="generic")
    b = ops.BatchPlan(ud, pd, nd, B, nU, nI, keep_orig=True, builder="auto", bucket_map=bmap)
    if nU > 2000:      # tables smaller than the batch need more than 1024 buckets (most rows are heavy): no map, generic builder
        assert bmap.users is not None and bmap.items is not None
        assert b.builder == "fast+map", "the mapped builder overflowed"
    for name in ("tu", "tp", "tn", "torig", "oc_item", "oc_src"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    _check_plan(b, u, p, n, B)
    # the plain bucket builder gives up on these ids (that is what the map is for) ...
    c = ops.BatchPlan(ud, pd, nd, B, nU, nI, builder="auto")
    if B * 8 < min(nU, nI):
        assert c.builder == "generic" and c.fast_overflowed
    ops._FAST_BACKOFF.clear()
    # ... and a map on one side only leaves the other side's equal-width buckets in place
    uu = rng.randint(0, nU, N).astype(dtype)
    half = ops.BucketMap(T(uu, dev), pd, nU, nI, B)
    half.users = None
    d = ops.BatchPlan(T(uu, dev), pd, nd, B, nU, nI, builder="auto", bucket_map=half)
    e = ops.BatchPlan(T(uu, dev), pd, nd, B, nU, nI, builder="generic")
    if d.builder == "fast+map":
        for name in ("tu", "tp", "tn", "oc_item", "oc_src"):
            assert torch.equal(getattr(d, name), getattr(e, name)), name
    ops._FAST_BACKOFF.clear()


def test_mapped_builder_overflow_falls_back(ops, dev):
    """a map built from one epoch, batches that are NOT samples of it (every triplet on one cold row): overflow -> generic"""
    nU = nI = 100000; B = 4096; N = 3 * B
    rng = np.random.RandomState(3)
    u, p, n = (x.astype(np.int32) for x in _skewed_epoch(rng, nU, nI, N))
    bmap = ops.BucketMap(T(u, dev), T(p, dev), nU, nI, B)
    u2 = np.full(N, 77777, np.int32); p2 = np.full(N, 55555, np.int32)
    plan = ops.BatchPlan(T(u2, dev), T(p2, dev), T(n, dev), B, nU, nI, keep_orig=True, builder="auto", bucket_map=bmap)
    assert plan.builder == "generic" and plan.fast_overflowed
    _check_plan(plan, u2, p2, n, B)
    ops._FAST_BACKOFF.clear()


def test_pipelined_sgd_switches_to_the_bucket_map_on_skewed_ids(ops, dev):
    """PipelinedSgd: the first chunk overflows the equal-width buckets, the following chunks run on the hand-written builder
    with a map of the epoch; the trained tables equal those of per-batch generic plans bit for bit"""
    nU, nI, D, B, nbat = 60000, 40000, 64, 4096, 10
    rng = np.random.RandomState(9)
    u, p, n = (x.astype(np.int32) for x in _skewed_epoch(rng, nU, nI, nbat * B))
    U = (rng.standard_normal((nU, D)) * 0.1).astype(np.float32); I = (rng.standard_normal((nI, D)) * 0.1).astype(np.float32)
    ops._FAST_BACKOFF.clear()
    Ud, Id = T(U, dev), T(I, dev)
    pipe = ops.PipelinedSgd(chunk=3)
    pipe.PLAN_TRIPLETS = 1               # plans of exactly 3 batches (the default would put this small epoch into one plan)
    h = pipe.plan(Ud, [(Id, T(u, dev), T(p, dev), T(n, dev))], B)
    losses = torch.empty(nbat, dtype=torch.float32, device=dev)
    seen = []
    orig = ops.BatchPlan.__init__
    def spy(self, *a, **k):
        orig(self, *a, **k); seen.append(self)
    ops.BatchPlan.__init__ = spy
    try:
        pipe.run(h, 0, 0.05, losses)
    finally:
        ops.BatchPlan.__init__ = orig
    torch.cuda.synchronize()
    seen = [pl.builder for pl in seen]      # plans are built deferred: the builder is known once they were used
    assert h["map"] and "fast+map" in seen, seen
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    plan = ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, builder="generic")
    l_ref = ref.run_sgd(plan, 0, nbat, 0.05)
    assert torch.equal(ref.U, Ud) and torch.equal(ref.I, Id) and torch.equal(l_ref, losses)
    ops._FAST_BACKOFF.clear()


@pytest.mark.parametrize("form", ["count", "bitonic"])
@pytest.mark.parametrize("B,nbat,nU,nI,short", [(2048, 3, 900, 1500, 100), (4096, 2, 100000, 50, 0), (512, 5, 40, 30, 511), (1, 3, 5, 5, 0),
                                                (3000, 2, 5000, 7000, 1), (2048, 1, 6040, 3706, 0), (4096, 2, 1, 1, 3),
                                                (4096, 1, 8000, 8000, 0), (4096, 2, 3, 12000, 0)])
def test_small_single_launch_builder_agrees_bitwise(ops, dev, monkeypatch, form, B, nbat, nU, nI, short):
    """one workgroup per batch — LDS counting sort where a counter per row fits LDS (form "count"; tables too large for it
    take the bitonic sorts either way), LDS bitonic sorts otherwise: the arrays of the radix-sort builder, bit for bit
    (duplicates up to a batch of one single row, short last batch, non-power-of-two batches, int64 and int32 inputs)"""
    if form == "bitonic":
        monkeypatch.setenv("WR_PLAN_SMALL", "bitonic")
    rng = np.random.RandomState(B + nbat)
    N = nbat * B - short
    u, p, n = rng.randint(0, nU, N), rng.randint(0, nI, N), rng.randint(0, nI, N)
    for cast in (np.int64, np.int32):
        a = ops.BatchPlan(T(u.astype(cast), dev), T(p.astype(cast), dev), T(n.astype(cast), dev), B, nU, nI, keep_orig=True,
                          builder="generic", hot=False)
        b = ops.BatchPlan(T(u.astype(cast), dev), T(p.astype(cast), dev), T(n.astype(cast), dev), B, nU, nI, keep_orig=True,
                          builder="small", hot=False)
        assert b.builder == "small"
        for x, y in ((a.tu, b.tu), (a.tp, b.tp), (a.tn, b.tn), (a.torig, b.torig), (a.oc_item, b.oc_item), (a.oc_src, b.oc_src)):
            assert torch.equal(x, y)
    _check_plan(b, u, p, n, B)
    bad = u.copy(); bad[N // 2] = nU
    with pytest.raises(IndexError):
        ops.BatchPlan(T(bad, dev), T(p, dev), T(n, dev), B, nU, nI, builder="small", hot=False)
    with pytest.raises(Exception):
        ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), 5000, nU, nI, builder="small", hot=False)
