"""The chained step launch (wr_bprmf_run_sgd_chain): ONE launch per step — the item phase of step k-1 rides in the launch of
step k's user phase; the runs of batch k that read a row that item phase rewrites wait, inside the launch, for a counter
the item workgroups add to after their write-through row stores.  Semantics are the reference loop's
(src/helpers/BaseRunner.py:194-200: strictly sequential, batch-synchronous steps), so the checks are: tables after N chained
steps BIT-IDENTICAL to the two-launch stream's (a stale read of a handed-over row would break exactly that) and equal to the
oracle's; losses equal to rounding and bitwise reproducible; steps with too many deferred runs, the first step of a call and
plans that do not qualify take the two-launch form; no wait ever expires."""
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


def _arena_plan(ops, dev, u, p, n, B, nU, nI):
    arena = ops.PlanArena(dev, u.size, B, overlap_items=nI)
    return ops.BatchPlan(T(u, dev), T(p, dev), T(n, dev), B, nU, nI, arena=arena, overlap=True), arena


def _tables(seed, nU, nI, D):
    rng = np.random.RandomState(seed)
    return ((rng.standard_normal((nU, D)) * 0.2).astype(np.float32), (rng.standard_normal((nI, D)) * 0.2).astype(np.float32))


@pytest.mark.parametrize("nI,D,B", [(200_000, 64, 8192), (60_000, 64, 8192), (60_000, 128, 8192), (25_000, 64, 8192),
                                    (400_000, 64, 32768)])
def test_chained_steps_equal_two_launch_steps_bitwise(ops, nI, D, B):
    dev = torch.device("cuda:0")
    nU, nb, lr = 70_000, 9, 0.1
    u, p, n = _epoch(2 + D, nU, nI, nb * B - 1000)                     # short last batch
    U, I = _tables(3, nU, nI, D)
    plan, _ = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    assert plan.overlap is not None
    dc, cap = plan.overlap["def_count_np"], plan.overlap["cap"]
    if nI == 25_000:
        assert dc[1:].min() > cap        # ~28 % of the runs read a row shared in the batch before: every step as two launches
    else:
        assert 0 < dc[1:].min() and dc.max() <= cap
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    l_ref = ref.run_sgd(plan, 0, nb, lr)
    outs = []
    for rep in range(2):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        assert tabs.chain_supported()
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        tabs.run_sgd_chain(plan, 0, 4, lr, losses[:4])                  # two calls: a two-launch step in the middle of the plan
        tabs.run_sgd_chain(plan, 4, nb - 4, lr, losses[4:])
        torch.cuda.synchronize()
        tabs.check_chain()
        outs.append((tabs.U.clone(), tabs.I.clone(), losses.clone()))
    assert torch.equal(outs[0][0], ref.U) and torch.equal(outs[0][1], ref.I)           # same tables, bit for bit
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))                     # and reproducible, losses too
    assert rel_err(outs[0][2].cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
    Uo, Io = U.copy(), I.copy()
    lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(outs[0][2].cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(outs[0][0].cpu().numpy(), Uo) < TOL and rel_err(outs[0][1].cpu().numpy(), Io) < TOL


def test_mixed_chained_and_two_launch_steps(ops):
    """def_limit between the batches' counts: some steps chained, some not, in one call — same tables"""
    dev = torch.device("cuda:0")
    nU, nI, D, B, nb, lr = 50_000, 120_000, 64, 8192, 12, 0.1
    u, p, n = _epoch(21, nU, nI, nb * B)
    U, I = _tables(5, nU, nI, D)
    plan, _ = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    dc = plan.overlap["def_count_np"][1:]
    limit = int(np.median(dc))
    assert (dc <= limit).any() and (dc > limit).any()
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    l_ref = ref.run_sgd(plan, 0, nb, lr)
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * nb)]
    l_mix = tabs.run_sgd_chain(plan, 0, nb, lr, def_limit=limit, phase_events=ev)
    torch.cuda.synchronize()
    tabs.check_chain()
    assert torch.equal(tabs.U, ref.U) and torch.equal(tabs.I, ref.I)
    assert rel_err(l_mix.cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
    assert all(ev[4 * k].elapsed_time(ev[4 * k + 1]) > 0 for k in range(nb))          # every step's launch carried its events


def test_headline_shape_many_steps(ops):
    """1M x 1M x 64, B = 65,536 (BASELINE.json configs[1]): 24 chained steps against 24 two-launch steps, every bit; the
    loss of each step equals the forward-only kernel's on the tables before the step"""
    dev = torch.device("cuda:0")
    nU = nI = 1_000_000
    D, B, nb, lr = 64, 65536, 24, 0.05
    g = torch.Generator(device=dev).manual_seed(7)
    u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    U0 = torch.randn(nU, D, device=dev, generator=g) * 0.1
    I0 = torch.randn(nI, D, device=dev, generator=g) * 0.1
    arena = ops.PlanArena(dev, nb * B, B, overlap_items=nI)
    plan = ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
    assert plan.overlap is not None and plan.overlap["fits"]
    ref = ops.BprmfTables(U0.clone(), I0.clone())
    l_ref = ref.run_sgd(plan, 0, nb, lr)
    for rep in range(3):                                             # the hand-off under the load of the whole chip, three times
        tabs = ops.BprmfTables(U0.clone(), I0.clone())
        l_chain = tabs.run_sgd_chain(plan, 0, nb, lr)
        torch.cuda.synchronize()
        tabs.check_chain()
        assert torch.equal(tabs.U, ref.U) and torch.equal(tabs.I, ref.I)
        assert rel_err(l_chain.cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
    fwd = ops.bpr_fwd(U0, I0, u[:B].long(), p[:B].long(), n[:B].long(), scores=False)["loss"]
    assert abs(float(fwd) - float(l_ref[0])) < 1e-6 * abs(float(fwd))


def test_chain_entry_refuses_rows_that_are_not_whole_lines(ops):
    from whisprrec_amd import abi
    dev = torch.device("cuda:0")
    nU, nI, B, nb = 20_000, 80_000, 8192, 3
    u, p, n = _epoch(4, nU, nI, nb * B)
    plan, _ = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    tabs = ops.BprmfTables(torch.zeros(nU, 16, device=dev), torch.zeros(nI, 16, device=dev))    # 64-B rows
    assert not tabs.chain_supported()
    with pytest.raises(abi.WhisprRecHipError, match="whole 128-B lines"):
        tabs.run_sgd_chain(plan, 0, nb, 0.1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", ["qualifies", "small_item_table", "hot_rows", "narrow_rows"])
def test_pipeline_picks_the_form_per_plan_and_results_do_not_depend_on_it(ops, case):
    dev = torch.device("cuda:0")
    nU, B, nb, lr = 60_000, 8192, 11, 0.05
    D = 16 if case == "narrow_rows" else 64
    nI = 9000 if case == "small_item_table" else 150_000
    u, p, n = _epoch(11, nU, nI, nb * B)
    if case == "hot_rows":
        p[::50] = 7                                         # one item row with ~160 occurrences per batch
    U, I = _tables(5, nU, nI, D)
    res, used = [], []
    for chain in (True, False):
        pipe = ops.PipelinedSgd(chunk=4, min_triplets=1, chain=chain)
        Ud, Id = T(U, dev), T(I, dev)
        h = pipe.plan(Ud, [(Id, T(u, dev), T(p, dev), T(n, dev))], B)
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(h, 0, lr, losses)
        torch.cuda.synchronize()
        h["segs"][0]["tabs"].check_chain()
        if chain and case == "hot_rows":
            assert h["chain"] is False          # after the first plan with hot rows the following plans skip the marks
        res.append((Ud, Id, losses))
        used.append(dict(pipe.stats))
    assert used[1]["chain_calls"] == 0 and used[1]["plain_calls"] == 3
    assert used[0]["chain_calls"] == (3 if case == "qualifies" else 0), (case, used)
    assert used[0]["chain_calls"] + used[0]["plain_calls"] == 3
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) < 1e-6
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
           for k in range(nb)]
    assert rel_err(res[0][2].cpu().numpy(), np.asarray(ref)) < TOL
    assert rel_err(res[0][0].cpu().numpy(), Uo) < TOL and rel_err(res[0][1].cpu().numpy(), Io) < TOL


def test_multi_segment_handle_chains_inside_segments_only(ops):
    """the stratified schedule's shape: several segments, each with its own view of the item rows, one plan sequence across
    them — chained launches inside a segment, a two-launch step at every segment start; same tables as without chaining"""
    dev = torch.device("cuda:0")
    nU, rows, D, B, lr = 40_000, 60_000, 64, 8192, 0.05
    U, I = _tables(9, nU, 3 * rows, D)
    segs_np = []
    for k, nb in enumerate((5, 3, 4)):
        u, p, n = _epoch(30 + k, nU, rows, nb * B)
        segs_np.append((k, u, p, n))
    res = []
    for chain in (True, False):
        pipe = ops.PipelinedSgd(chunk=4, min_triplets=1, chain=chain)
        Ud, Id = T(U, dev), T(I, dev)
        # consecutive slices of one array per column, as the rotating schedule passes them
        ua, pa, na = (T(np.concatenate([s[j] for s in segs_np]), dev) for j in (1, 2, 3))
        segments, off = [], 0
        for k, u, p, n in segs_np:
            segments.append((Id[k * rows:(k + 1) * rows], ua[off:off + u.size], pa[off:off + u.size], na[off:off + u.size]))
            off += u.size
        h = pipe.plan(Ud, segments, B)
        assert h["chain"] == chain
        losses = []
        for k in range(3):
            l = torch.empty(h["segs"][k]["nb"], dtype=torch.float32, device=dev)
            pipe.run(h, k, lr, l)
            losses.append(l)
        torch.cuda.synchronize()
        for sg in h["segs"]:
            sg["tabs"].check_chain()
        res.append((Ud, Id, torch.cat(losses), dict(pipe.stats)))
    assert res[0][3]["chain_calls"] >= 3 and res[1][3]["chain_calls"] == 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) < 1e-6
    Uo, Io = U.copy(), I.copy()
    ref = []
    for k, u, p, n in segs_np:
        view = Io[k * rows:(k + 1) * rows]
        for j in range(u.size // B):
            ref.append(oracle.bprmf_step_sgd(Uo, view, u[j * B:(j + 1) * B], p[j * B:(j + 1) * B], n[j * B:(j + 1) * B], lr, 0.0))
    assert rel_err(res[0][2].cpu().numpy(), np.asarray(ref)) < TOL
    assert rel_err(res[0][0].cpu().numpy(), Uo) < TOL and rel_err(res[0][1].cpu().numpy(), Io) < TOL


def test_c4_shape_chained(ops):
    """BASELINE.json configs[3] tables (10M x 10M, D = 128, B = 65,536; 5.1 GB per table, rows beyond 4 GB): four chained
    steps against four two-launch steps, every bit of both tables"""
    dev = torch.device("cuda:0")
    nU = nI = 10_000_000
    D, B, nb, lr = 128, 65536, 4, 0.05
    g = torch.Generator(device=dev).manual_seed(11)
    u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    p[:64] = nI - 1 - torch.arange(64, device=dev, dtype=torch.int32)          # rows behind the 4 GB mark, for sure
    U0 = torch.randn(nU, D, device=dev, generator=g) * 0.1
    I0 = torch.randn(nI, D, device=dev, generator=g) * 0.1
    arena = ops.PlanArena(dev, nb * B, B, overlap_items=nI)
    plan = ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
    assert plan.overlap is not None and plan._bitmap_ready                    # the builder wrote the bitmap itself (shift 16)
    ref = ops.BprmfTables(U0.clone(), I0.clone())
    l_ref = ref.run_sgd(plan, 0, nb, lr)
    tabs = ops.BprmfTables(U0, I0)
    l_chain = tabs.run_sgd_chain(plan, 0, nb, lr)
    torch.cuda.synchronize()
    tabs.check_chain()
    assert torch.equal(tabs.U, ref.U) and torch.equal(tabs.I, ref.I)
    assert rel_err(l_chain.cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
