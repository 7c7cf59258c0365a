"""Exact lazy optimizers (wr_adam_rows_lazy / wr_sgd_rows_lazy / wr_*_catchup_all, whisprrec_amd/csrc/wr_lazy.hip): after
flush() the tables must hold the SAME BITS as the dense torch.optim semantics kernels (wr_adam_dense, fused SGD step +
wr_sgd_decay_untouched), which tests/test_hip_bprmf.py pins to the reference's goldens (BaseRunner.py:120-124,199)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(nU, nI, D, B, steps, seed, zipf=False):
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(seed)
    U = torch.randn(nU, D, generator=g, device=dev) * 0.1
    I = torch.randn(nI, D, generator=g, device=dev) * 0.1
    rng = np.random.RandomState(seed)
    u = rng.randint(0, nU, steps * B)
    p = np.minimum((rng.pareto(1.0, steps * B) * 3).astype(np.int64), nI - 1) if zipf else rng.randint(0, nI, steps * B)
    n = rng.randint(1, nI, steps * B)
    if zipf:
        u[::5] = 7                        # a hot USER too (every batch: > 32 triplets of user 7): pieces + combine on that side
    t = lambda a: torch.from_numpy(a).to(dev)
    plan = hip_ops.BatchPlan(t(u), t(p), t(n), B, nU, nI)
    if zipf:
        assert plan.hot is not None and int(plan.hot["counts_host"].view(-1, 4)[:, 3].min()) >= 1
    return hip_ops, U, I, plan


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("l2", [0.0, 1e-3])
@pytest.mark.parametrize("D,zipf", [(64, False), (32, True), (128, False), (20, False), (32, False)])
def test_lazy_adam_bit_identical_to_dense(l2, D, zipf, fold):
    """fold=True: the catch-up happens inside the step kernels' row loads (wr_bprmf_step_adam_folded); fold=False: in a pass
    of its own (wr_adam_rows_lazy) before wr_bprmf_step_adam.  Both must leave the dense optimizer's bits."""
    nU, nI, B, steps, lr = 3000, 2500, 256, 24, 1e-2
    hip_ops, U, I, plan = _setup(nU, nI, D, B, steps, 7 + D, zipf)
    # dense (what FusedOptimizer did before)
    Ud, Id = U.clone(), I.clone()
    td = hip_ops.BprmfTables(Ud, Id)
    z = torch.zeros_like
    gU, gI, mU, vU, mI, vI = z(Ud), z(Id), z(Ud), z(Ud), z(Id), z(Id)
    dense_loss = []
    for k in range(steps):
        loss, sid = td.grads(plan, k, gU, gI)
        dense_loss.append(loss.clone())
        hip_ops.adam_dense(Ud, mU, vU, gU, k + 1, lr, l2, stamp=td.stamp_u, step_id=sid)
        hip_ops.adam_dense(Id, mI, vI, gI, k + 1, lr, l2, stamp=td.stamp_i, step_id=sid)
    # lazy
    Ul, Il = U.clone(), I.clone()
    tl = hip_ops.BprmfTables(Ul, Il)
    st = hip_ops.LazyOptimizerState(tl, "Adam", lr, l2, fold=fold)
    lazy_loss = []
    for k in range(steps):
        lazy_loss.append(st.step(plan, k).clone())
        if k == steps // 2:
            st.flush()                                    # a flush in the middle (evaluation) must change nothing
    assert not torch.equal(Ul, Ud)                        # untouched rows are behind until the flush
    st.flush()
    assert torch.equal(torch.stack(lazy_loss), torch.stack(dense_loss))
    for a, b in ((Ul, Ud), (Il, Id), (st.m_u, mU), (st.v_u, vU), (st.m_i, mI), (st.v_i, vI)):
        assert torch.equal(a, b)
    assert int(st.last_u.min()) == steps and int(st.last_i.min()) == steps


@pytest.mark.parametrize("max_lag", [1, 5, 64])
@pytest.mark.parametrize("l2,D,zipf", [(0.0, 64, False), (1e-3, 32, True), (0.0, 20, False)])
def test_bounded_lag_bit_identical_to_dense(l2, D, zipf, max_lag):
    """wr_bprmf_run_adam_lazy_bounded: a rotating window of rows / max_lag rows per table is replayed before every step, so no
    row misses more than max_lag steps — and nothing else changes: the dense optimizer's bits, native loop in two calls."""
    nU, nI, B, steps, lr = 3000, 2500, 256, 24, 1e-2
    hip_ops, U, I, plan = _setup(nU, nI, D, B, steps, 11 + D, zipf)
    Ud, Id = U.clone(), I.clone()
    td = hip_ops.BprmfTables(Ud, Id)
    z = torch.zeros_like
    gU, gI, mU, vU, mI, vI = z(Ud), z(Id), z(Ud), z(Ud), z(Id), z(Id)
    dense_loss = []
    for k in range(steps):
        loss, sid = td.grads(plan, k, gU, gI)
        dense_loss.append(loss.clone())
        hip_ops.adam_dense(Ud, mU, vU, gU, k + 1, lr, l2, stamp=td.stamp_u, step_id=sid)
        hip_ops.adam_dense(Id, mI, vI, gI, k + 1, lr, l2, stamp=td.stamp_i, step_id=sid)
    Ul, Il = U.clone(), I.clone()
    st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(Ul, Il), "Adam", lr, l2, fold=False, max_lag=max_lag)
    l1 = st.run(plan, 0, 10)
    l2_ = st.run(plan, 10, steps - 10)
    # the invariant the window maintains: after step t every row stands at step >= t - 1 - max_lag (rows of the last
    # window were brought to t - 1 before the step, the window comes round every max_lag steps)
    assert int(st.last_u.min()) >= steps - 1 - max_lag and int(st.last_i.min()) >= steps - 1 - max_lag
    if max_lag >= 5:
        assert int(st.last_u.min()) < steps                   # still lazy: not every row is current
    st.flush()
    assert torch.equal(torch.cat([l1, l2_]), torch.stack(dense_loss))
    for a, b in ((Ul, Ud), (Il, Id), (st.m_u, mU), (st.v_u, vU), (st.m_i, mI), (st.v_i, vI)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("max_lag", [0, 3])
@pytest.mark.parametrize("D", [64, 24])
def test_lazy_sgd_weight_decay_bit_identical_to_dense(D, max_lag):
    """max_lag = 3: with the rotating window (wr_bprmf_run_sgd_lazy_bounded) no row misses more than 3 decay steps"""
    nU, nI, B, steps, lr, l2 = 4000, 1500, 512, 20, 0.1, 1e-2
    hip_ops, U, I, plan = _setup(nU, nI, D, B, steps, 3 + D, True)
    Ud, Id = U.clone(), I.clone()
    td = hip_ops.BprmfTables(Ud, Id)
    dense_loss = [td.step_sgd(plan, k, lr, l2).clone() for k in range(steps)]
    Ul, Il = U.clone(), I.clone()
    st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(Ul, Il), "SGD", lr, l2, max_lag=max_lag)
    lazy_loss = [st.step(plan, k).clone() for k in range(steps)]
    if max_lag:
        assert int(st.last_u.min()) >= steps - 1 - max_lag and int(st.last_i.min()) >= steps - 1 - max_lag
    st.flush()
    assert torch.equal(torch.stack(lazy_loss), torch.stack(dense_loss))
    assert torch.equal(Ul, Ud) and torch.equal(Il, Id)


def test_lazy_adam_consts_table_grows():
    nU, nI, D, B, steps = 300, 200, 16, 32, 3
    hip_ops, U, I, plan = _setup(nU, nI, D, B, steps, 1)
    st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, 0.0)
    st.t = st.n_consts - 2                                # pretend many steps were taken (rows replay them: all-zero moments)
    st.last_u.fill_(st.t); st.last_i.fill_(st.t); st.flushed_at = st.t
    n0 = st.n_consts
    for k in range(steps):
        st.step(plan, k)
    st.flush()
    assert st.n_consts == 2 * n0 and torch.isfinite(U).all()


@pytest.mark.parametrize("name,l2,zipf", [("Adam", 0.0, False), ("Adam", 1e-3, True), ("Adam", 1e-3, False), ("SGD", 1e-2, True)])
def test_native_multi_batch_loop_equals_per_batch_calls(name, l2, zipf):
    """wr_bprmf_run_adam_lazy / wr_bprmf_run_sgd_lazy: same bits as calling step() batch by batch (short last batch, hot rows)"""
    nU, nI, D, B, steps, lr = 3000, 2500, 64, 256, 9, 1e-2
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(5)
    U = torch.randn(nU, D, generator=g, device=dev) * 0.1; I = torch.randn(nI, D, generator=g, device=dev) * 0.1
    rng = np.random.RandomState(5)
    N = steps * B - 100                                  # the last batch is short
    u = rng.randint(0, nU, N)
    p = np.minimum((rng.pareto(1.0, N) * 3).astype(np.int64), nI - 1) if zipf else rng.randint(0, nI, N)
    n = rng.randint(1, nI, N)
    t = lambda a: torch.from_numpy(a).to(dev)
    plan = hip_ops.BatchPlan(t(u), t(p), t(n), B, nU, nI)
    outs = []
    for native in (False, True, "unfolded"):
        Ua, Ia = U.clone(), I.clone()
        st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(Ua, Ia), name, lr, l2, fold=native != "unfolded")
        if native:
            l1 = st.run(plan, 0, 4); l2_ = st.run(plan, 4, steps - 4)
            losses = torch.cat([l1, l2_])
        else:
            losses = torch.stack([st.step(plan, k).clone() for k in range(steps)])
        st.flush()
        outs.append((losses, Ua, Ia))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])


@pytest.mark.parametrize("l2,D,nI", [(0.0, 64, 200_000), (1e-3, 64, 60_000), (0.0, 128, 120_000), (0.0, 32, 25_000)])
def test_chained_folded_adam_equals_two_launch_folded_adam_bitwise(l2, D, nI):
    """round 3: the folded Adam step as ONE launch per step (wr_bprmf_run_adam_folded_chain: the item phase of step k-1 —
    weights, both moments and the step stamp of the rows that recur, stored write-through — inside the launch of step k's
    user phase) leaves the bits of the two-launch folded step in tables, moments and stamps (losses to 1e-6: a few more
    partials); cut into calls of several lengths; nI = 25,000: most steps defer more runs than the list holds and go out as two launches in between"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    nU, B, nb, lr = 70_000, 8192, 9, 1e-2
    g = torch.Generator(device=dev); g.manual_seed(11 + D)
    U0 = torch.randn(nU, D, generator=g, device=dev) * 0.1
    I0 = torch.randn(nI, D, generator=g, device=dev) * 0.1
    N = nb * B - 1000                                                 # short last batch
    u = torch.randint(0, nU, (N,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, nI, (N,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, nI, (N,), generator=g, device=dev, dtype=torch.int32)
    arena = hip_ops.PlanArena(dev, N, B, overlap_items=nI)
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
    assert plan.overlap is not None and plan.hot is None
    out = []
    for chain in (False, True):
        st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U0.clone(), I0.clone()), "Adam", lr, l2, fold=True)
        st.chain = chain
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        for first, count in ((0, 4), (4, 1), (5, 4)):                 # a call of one step is never chained
            st.run(plan, first, count, losses[first:first + count])
        torch.cuda.synchronize()
        st.tabs.check_chain()
        assert (st.chain_calls > 0) == chain
        out.append((st.tabs.U, st.tabs.I, st.m_u, st.v_u, st.m_i, st.v_i, st.last_u, st.last_i, losses))
    for a, b in zip(out[0][:-1], out[1][:-1]):
        assert torch.equal(a, b)
    # a chained step's loss is summed over a few more partials (one per chunk of deferred runs): same terms, another grouping
    assert torch.allclose(out[0][-1], out[1][-1], rtol=1e-6, atol=0)
