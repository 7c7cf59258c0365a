"""The step stream without a per-batch sort (wr_group.hip: wr_group_plan_build + wr_bprmf_run_sgd_group).  Semantics are the
reference loop's (src/helpers/BaseRunner.py:194-200: strictly sequential, batch-synchronous SGD steps, l2 = 0), so the
checks are: the plan's index arrays bit-exact against their NumPy restatement (oracle.group_plan); tables and losses after
N steps equal to the oracle's (1e-5, north_star) and to the sorted-plan stream's (rounding); tables bitwise reproducible and
independent of how the steps are cut into calls (a stale read of a row handed over inside a launch would break exactly
that); unusable plans say so (list overflow, id out of range)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from whisprrec_amd import hip_ops
    return hip_ops


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _epoch(seed, nU, nI, n):
    rng = np.random.RandomState(seed)
    return (rng.randint(0, nU, n).astype(np.int32), rng.randint(0, nI, n).astype(np.int32),
            rng.randint(1, nI, n).astype(np.int32))


def _tables(seed, nU, nI, D):
    rng = np.random.RandomState(seed)
    return ((rng.standard_normal((nU, D)) * 0.2).astype(np.float32), (rng.standard_normal((nI, D)) * 0.2).astype(np.float32))


def _same_plan(got, ref):
    assert got["R_u"] == ref["R_u"] and got["R_i"] == ref["R_i"]
    assert np.array_equal(got["flags"], ref["flags"])
    for side in ("users", "items"):
        assert set(got[side]) == set(ref[side])
        for key, (rows, src) in ref[side].items():
            assert np.array_equal(got[side][key][0], rows) and np.array_equal(got[side][key][1], src), (side, key)


@pytest.mark.parametrize("nU,nI,B,n", [(5_000, 4_000, 1024, 3 * 1024 + 100),          # one range, dense sharing
                                       (300_000, 1_000_000, 8192, 4 * 8192 - 77),       # 2 and 4 ranges, short last batch
                                       (70_000, 200_000, 1001, 5 * 1001),               # batch starts not 16-byte aligned
                                       (3_000_000, 5_000_000, 16384, 2 * 16384 + 5)])   # hashed rows (tables beyond 2^21 rows)
def test_plan_arrays_match_the_numpy_restatement(ops, nU, nI, B, n):
    dev = torch.device("cuda:0")
    u, p, neg = _epoch(11, nU, nI, n)
    plan = ops.GroupPlan(T(u, dev), T(p, dev), T(neg, dev), B, nU, nI)
    assert not plan.bad_index and not plan.overflow
    _same_plan(plan.decode(), oracle.group_plan(u, p, neg, B, nU, nI))


def test_plan_reports_overflow_long_runs_and_bad_ids(ops):
    dev = torch.device("cuda:0")
    u, p, n = _epoch(5, 50_000, 3_000, 2 * 32768)                 # 65,536 occurrences on 3,000 item rows: every row shared
    plan = ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), 32768, 50_000, 3_000)
    assert plan.overflow
    u, p, n = _epoch(6, 50_000, 40_000, 8192)
    p[:100] = 7                                                    # one row with 100 occurrences
    plan = ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), 4096, 50_000, 40_000)
    assert plan.long_run and not plan.overflow
    p[5] = 40_000
    with pytest.raises(IndexError):
        ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), 4096, 50_000, 40_000).validate()


@pytest.mark.parametrize("nU,nI,D,B", [(70_000, 200_000, 64, 8192), (70_000, 60_000, 64, 4096), (50_000, 60_000, 128, 8192),
                                       (400_000, 400_000, 64, 32768), (3_000, 2_500, 64, 256), (2_200_000, 2_500_000, 32, 8192)])
def test_steps_match_oracle_and_are_reproducible(ops, nU, nI, D, B):
    dev = torch.device("cuda:0")
    nb, lr = 7, 0.1
    u, p, n = _epoch(2 + D, nU, nI, nb * B - B // 3)                  # short last batch
    U, I = _tables(3, nU, nI, D)
    du, dp, dn = T(u, dev), T(p, dev), T(n, dev)
    plan = ops.GroupPlan(du, dp, dn, B, nU, nI)
    assert not plan.overflow and not plan.bad_index
    outs = []
    for cuts in ([nb], [3, nb - 3], [1, 1, nb - 2]):                  # one call; two calls; steps that start unchained
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        assert tabs.group_supported()
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        at = 0
        for c in cuts:
            tabs.run_sgd_group(plan, at, c, lr, losses[at:at + c])
            at += c
        torch.cuda.synchronize()
        tabs.check_chain()
        outs.append((tabs.U.clone(), tabs.I.clone(), losses.clone()))
    for o in outs[1:]:                                                # same tables bit for bit however the steps are cut
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
        assert rel_err(o[2].cpu().numpy(), outs[0][2].cpu().numpy()) < 1e-6
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))                      # and run to run, losses included
    l2 = tabs.run_sgd_group(plan, 0, nb, lr)
    torch.cuda.synchronize()
    assert torch.equal(tabs.U, outs[0][0]) and torch.equal(tabs.I, outs[0][1]) and torch.equal(l2, outs[0][2])
    Uo, Io = U.copy(), I.copy()
    lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(outs[0][2].cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(outs[0][0].cpu().numpy(), Uo) < TOL and rel_err(outs[0][1].cpu().numpy(), Io) < TOL
    # the sorted-plan stream on the same batches: other summation order, same step
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    ref.run_sgd(ops.BatchPlan(du, dp, dn, B, nU, nI), 0, nb, lr)
    torch.cuda.synchronize()
    assert rel_err(outs[0][0].cpu().numpy(), ref.U.cpu().numpy()) < 1e-6
    assert rel_err(outs[0][1].cpu().numpy(), ref.I.cpu().numpy()) < 1e-6


def test_hot_rows_take_the_slow_paths(ops):
    """degenerate batches: a user with 150 triplets in one batch and an item row with 200 occurrences — runs far beyond the
    tiles' staged window (slow paths, same results; the plan says `long_run`).  More than 256 occurrences of one row are
    beyond what the plan orders: `overflow`, and the caller takes the sorted plan."""
    dev = torch.device("cuda:0")
    nU, nI, D, B, nb, lr = 40_000, 50_000, 64, 2048, 3, 0.05
    u, p, n = _epoch(9, nU, nI, nb * B)
    u[B:B + 150] = 123
    p[:200] = 77
    n[2 * B:2 * B + 40] = 77
    U, I = _tables(4, nU, nI, D)
    plan = ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
    assert plan.long_run and not plan.overflow
    _same_plan(plan.decode(), oracle.group_plan(u, p, n, B, nU, nI))
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    losses = tabs.run_sgd_group(plan, 0, nb, lr)
    torch.cuda.synchronize()
    tabs.check_chain()
    Uo, Io = U.copy(), I.copy()
    lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(losses.cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(tabs.U.cpu().numpy(), Uo) < TOL and rel_err(tabs.I.cpu().numpy(), Io) < TOL
    u[B:2 * B] = 123                                                # one user for a whole batch
    assert ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI).overflow


def test_headline_shape_rows_outside_the_batch_untouched(ops):
    """configs[1] shapes (1M x 1M, D = 64, B = 65,536): touched rows against the oracle's sparse restatement, every other
    row bit-identical, second run bitwise equal"""
    dev = torch.device("cuda:0")
    nU = nI = 1_000_000
    D, B, nb, lr = 64, 65536, 6, 0.05
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    U0 = torch.randn(nU, D, generator=g, device=dev) * 0.05
    I0 = torch.randn(nI, D, generator=g, device=dev) * 0.05
    u, p, n = _epoch(21, nU, nI, nb * B)
    plan = ops.GroupPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI)
    assert not plan.overflow and not plan.long_run
    tabs = ops.BprmfTables(U0.clone(), I0.clone())
    losses = tabs.run_sgd_group(plan, 0, nb, lr)
    torch.cuda.synchronize()
    tabs.check_chain()
    tu, ti = np.unique(u), np.unique(np.concatenate([p, n]))
    Uc, Ic = U0[T(tu.astype(np.int64), dev)].cpu().numpy(), I0[T(ti.astype(np.int64), dev)].cpu().numpy()
    cu, cp, cn = np.searchsorted(tu, u), np.searchsorted(ti, p), np.searchsorted(ti, n)
    lo_ref = [oracle.bprmf_step_sgd(Uc, Ic, cu[k * B:(k + 1) * B], cp[k * B:(k + 1) * B], cn[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(losses.cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(tabs.U[T(tu.astype(np.int64), dev)].cpu().numpy(), Uc) < TOL
    assert rel_err(tabs.I[T(ti.astype(np.int64), dev)].cpu().numpy(), Ic) < TOL
    mu = torch.ones(nU, dtype=torch.bool, device=dev)
    mu[T(tu.astype(np.int64), dev)] = False
    mi = torch.ones(nI, dtype=torch.bool, device=dev)
    mi[T(ti.astype(np.int64), dev)] = False
    assert torch.equal(tabs.U[mu], U0[mu]) and torch.equal(tabs.I[mi], I0[mi])
    again = ops.BprmfTables(U0.clone(), I0.clone())
    l2 = again.run_sgd_group(plan, 0, nb, lr)
    torch.cuda.synchronize()
    assert torch.equal(again.U, tabs.U) and torch.equal(again.I, tabs.I) and torch.equal(l2, losses)


def test_expired_wait_is_raised_within_one_chunk(ops, g2):
    """The only failure of the product whose detection could lag its damage: a bounded wait inside a step launch that
    expires sets a sticky word.  The step stream copies that word to pinned memory behind every chunk and reads it when the
    next chunk starts (and when a run ends): with the word set by hand, the stream raises one chunk later — not at the end of
    the epoch — and HipRunner.fit raises before anything could save or evaluate the tables."""
    from whisprrec_amd import abi
    dev = torch.device("cuda:0")
    nU, nI, D, B, nb, lr = 120_000, 150_000, 64, 8192, 16, 0.05
    u, p, n = _epoch(31, nU, nI, nb * B)
    U, I = _tables(5, nU, nI, D)
    pipe = ops.PipelinedSgd(chunk=4, min_triplets=1)
    Ud, Id = T(U, dev), T(I, dev)
    handle = pipe.plan(Ud, [(Id, T(u, dev), T(p, dev), T(n, dev))], B)
    assert handle["group"]
    losses = torch.empty(nb, dtype=torch.float32, device=dev)
    pipe.run_steps(handle, 4, lr, losses[:4])                       # one chunk: nothing wrong
    tabs = handle["segs"][0]["tabs"]
    words = tabs.sticky_words()
    assert words
    try:
        for w in words:
            w.fill_(1)
        with pytest.raises(abi.WhisprRecHipError, match="expired"):
            pipe.run_steps(handle, 12, lr, losses[4:])              # raised when the chunk after the next one starts
        assert handle["pos"] <= 12                                   # ... not after all 16 steps
        # the model path HipRunner.fit takes (BPRMF.train_epoch -> PipelinedSgd.run): raises before fit returns
        from test_hip_integration import _setup
        from whisprrec_amd import runner
        args, corpus, model, ds = _setup(g2)
        with pytest.raises(abi.WhisprRecHipError):
            runner.HipRunner(args).fit(ds, epoch=1)
    finally:
        for w in words:
            w.zero_()
        torch.cuda.synchronize()


def test_pipeline_falls_back_to_sorted_plans_on_skewed_ids(ops):
    """PipelinedSgd starts with group plans where the SHAPE qualifies; ids it is not made for (a hot item row: the lists
    overflow; a row with dozens of occurrences: `long_run`) send the stream to sorted plans — the overflowing chunk is
    re-planned, later chunks are planned sorted from the start — and the result is the oracle's either way."""
    dev = torch.device("cuda:0")
    nU, nI, D, B, nb, lr = 120_000, 150_000, 64, 8192, 9, 0.05
    for hot in (600, 100):                                            # 600 occurrences per batch: overflow; 100: long run
        u, p, n = _epoch(50 + hot, nU, nI, nb * B)
        for k in range(nb):
            p[k * B:k * B + hot] = 4242
        U, I = _tables(6, nU, nI, D)
        pipe = ops.PipelinedSgd(chunk=3, min_triplets=1)
        Ud, Id = T(U, dev), T(I, dev)
        handle = pipe.plan(Ud, [(Id, T(u, dev), T(p, dev), T(n, dev))], B)
        assert handle["group"]
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(handle, 0, lr, losses)
        torch.cuda.synchronize()
        handle["segs"][0]["tabs"].check_chain()
        assert not handle["group"] and pipe.stats["group_fallbacks"] >= 1
        Uo, Io = U.copy(), I.copy()
        lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
                  for k in range(nb)]
        assert rel_err(losses.cpu().numpy(), np.asarray(lo_ref)) < TOL
        assert rel_err(Ud.cpu().numpy(), Uo) < TOL and rel_err(Id.cpu().numpy(), Io) < TOL


def test_int64_epoch_columns_take_the_group_path(ops):
    """the reference hands int64 index columns (src/models/BaseModel.py:96-127): the step stream narrows them chunk by chunk
    on the plan stream and runs on group plans all the same — same tables as with int32 columns, bit for bit; an id beyond
    2^31 is an IndexError (as nn.Embedding's), not an alias of a valid row"""
    dev = torch.device("cuda:0")
    nU, nI, D, B, nb, lr = 130_000, 160_000, 64, 8192, 7, 0.05
    u, p, n = _epoch(77, nU, nI, nb * B - 1000)
    U, I = _tables(8, nU, nI, D)
    out = {}
    for dt in (torch.int32, torch.int64):
        pipe = ops.PipelinedSgd(chunk=3, min_triplets=1)
        Ud, Id = T(U, dev), T(I, dev)
        handle = pipe.plan(Ud, [(Id, T(u, dev).to(dt), T(p, dev).to(dt), T(n, dev).to(dt))], B)
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(handle, 0, lr, losses)
        torch.cuda.synchronize()
        assert pipe.stats["group_calls"] > 0 and pipe.stats["group_fallbacks"] == 0
        out[dt] = (Ud, Id, losses)
    for a, b in zip(out[torch.int32], out[torch.int64]):
        assert torch.equal(a, b)
    Uo, Io = U.copy(), I.copy()
    lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(out[torch.int64][2].cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(out[torch.int64][0].cpu().numpy(), Uo) < TOL and rel_err(out[torch.int64][1].cpu().numpy(), Io) < TOL
    bad = T(p, dev).to(torch.int64)
    bad[5] = (1 << 32) + 5
    pipe = ops.PipelinedSgd(chunk=3, min_triplets=1)
    with pytest.raises(IndexError):
        handle = pipe.plan(T(U, dev), [(T(I, dev), T(u, dev).to(torch.int64), bad, T(n, dev).to(torch.int64))], B)
        pipe.run(handle, 0, lr, torch.empty(nb, dtype=torch.float32, device=dev))
        torch.cuda.synchronize()
