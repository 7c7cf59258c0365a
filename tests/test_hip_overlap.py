"""The overlapped step stream (wr_bprmf_plan_overlap_marks + wr_bprmf_run_sgd_overlap): item phase of step k beside the
user phase of step k+1.  Semantics are the reference loop's (src/helpers/BaseRunner.py:194-200: strictly sequential,
batch-synchronous steps), so the checks are: the plan's marks equal a NumPy restatement of their definition; the tables
after N overlapped steps are BIT-IDENTICAL to the ordinary step stream's and match the oracle; losses agree to rounding and
are bitwise reproducible; plans that do not qualify (hot rows, lists beyond capacity) fall back to the ordinary stream."""
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


@pytest.mark.parametrize("nI,B", [(200_000, 8192), (60_000, 8192), (3000, 1024)])
def test_marks_match_their_definition(ops, nI, B):
    dev = torch.device("cuda:0")
    nU, nb = 50_000, 5
    u, p, n = _epoch(1, nU, nI, nb * B - 37)                 # short last batch
    plan, arena = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    torch.cuda.synchronize()
    tu, tp, tn = (x.cpu().numpy() for x in (plan.tu, plan.tp, plan.tn))
    o = arena.overlap
    words, dwords, cap = o["words"], o["dwords"], o["cap"]
    bitmap = o["bitmap"].cpu().numpy().view(np.uint32)[:nb * words].reshape(nb, words)
    tdef = o["tdef"].cpu().numpy().view(np.uint32)[:nb * dwords].reshape(nb, dwords)
    def_q = o["def_q"].cpu().numpy()[:nb * cap].reshape(nb, cap)
    counts = plan.meta_host[2 + 4 * nb:2 + 5 * nb].numpy()
    prev_multi = None
    for b in range(nb):
        lo, hi = b * B, min((b + 1) * B, u.size)
        rows = np.concatenate([tp[lo:hi] & 0x7fffffff, tn[lo:hi] & 0x7fffffff])
        flagged = np.concatenate([tp[lo:hi] < 0, tn[lo:hi] < 0])
        cnt = np.bincount(rows, minlength=nI)
        assert np.array_equal(flagged, cnt[rows] > 1)                                  # bit 31 = several occurrences
        multi = np.zeros(words * 32, bool); multi[:nI] = cnt > 1
        got = np.unpackbits(bitmap[b].view(np.uint8), bitorder="little").astype(bool)
        assert np.array_equal(got, multi)
        heads = np.zeros(B, bool)
        if prev_multi is not None:
            hit = prev_multi[tp[lo:hi] & 0x7fffffff] | prev_multi[tn[lo:hi] & 0x7fffffff]
            uu = tu[lo:hi]
            first_of_run = np.r_[True, uu[1:] != uu[:-1]]
            head_pos = np.maximum.accumulate(np.where(first_of_run, np.arange(hi - lo), 0))
            heads[np.unique(head_pos[hit])] = True
        got_heads = np.unpackbits(tdef[b].view(np.uint8), bitorder="little").astype(bool)[:B]
        assert np.array_equal(got_heads, heads)
        want = np.nonzero(heads)[0]
        assert counts[b] == want.size
        assert np.array_equal(def_q[b, :min(want.size, cap)], want[:cap])
        prev_multi = multi
    if nI == 3000:
        # lists beyond capacity: not for the two-stream form (the chained launch decides per step)
        assert plan.overlap is not None and not plan.overlap["fits"] and counts.max() > cap
    else:
        assert plan.overlap is not None and plan.overlap["fits"] and counts[0] == 0 and counts[1:].min() > 0


@pytest.mark.parametrize("nI,D", [(200_000, 64), (60_000, 64), (60_000, 128), (90_000, 32)])
def test_overlapped_steps_equal_ordinary_steps_bitwise(ops, nI, D):
    dev = torch.device("cuda:0")
    nU, B, nb, lr = 70_000, 8192, 9, 0.1
    u, p, n = _epoch(2 + D, nU, nI, nb * B - 1000)
    rng = np.random.RandomState(3)
    U = (rng.standard_normal((nU, D)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.2).astype(np.float32)
    plan, _ = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    assert plan.overlap is not None
    ref = ops.BprmfTables(T(U, dev), T(I, dev))
    l_ref = ref.run_sgd(plan, 0, nb, lr)
    side = ops.side_stream(dev)
    ev = ops.OverlapEvents(dev)
    outs = []
    for _ in range(2):
        tabs = ops.BprmfTables(T(U, dev), T(I, dev))
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        tabs.run_sgd_overlap(plan, 0, 4, lr, losses[:4], side, ev)           # two calls: a join in the middle of the plan
        tabs.run_sgd_overlap(plan, 4, nb - 4, lr, losses[4:], side, ev)
        torch.cuda.synchronize()
        outs.append((tabs.U.clone(), tabs.I.clone(), losses.clone()))
    assert torch.equal(outs[0][0], ref.U) and torch.equal(outs[0][1], ref.I)           # same tables, bit for bit
    # static form (launches sized for the lists' capacity, counts read on the device: what the hipGraph replays)
    tabs = ops.BprmfTables(T(U, dev), T(I, dev))
    l_static = tabs.run_sgd_overlap(plan, 0, nb, lr, None, side, ev, static=True)
    torch.cuda.synchronize()
    assert torch.equal(tabs.U, ref.U) and torch.equal(tabs.I, ref.I)
    assert rel_err(l_static.cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert rel_err(outs[0][2].cpu().numpy(), l_ref.cpu().numpy()) < 1e-6
    Uo, Io = U.copy(), I.copy()
    lo_ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
              for k in range(nb)]
    assert rel_err(outs[0][2].cpu().numpy(), np.asarray(lo_ref)) < TOL
    assert rel_err(outs[0][0].cpu().numpy(), Uo) < TOL and rel_err(outs[0][1].cpu().numpy(), Io) < TOL


@pytest.mark.parametrize("case", ["qualifies", "lists_overflow", "hot_rows"])
def test_pipeline_picks_the_stream_per_plan_and_results_do_not_depend_on_it(ops, case):
    dev = torch.device("cuda:0")
    nU, B, nb, D, lr = 60_000, 8192, 11, 64, 0.05
    nI = {"qualifies": 150_000, "lists_overflow": 9000, "hot_rows": 150_000}[case]
    u, p, n = _epoch(11, nU, nI, nb * B)
    if case == "hot_rows":
        p[::50] = 7                                         # one item row with ~160 occurrences per batch
    rng = np.random.RandomState(5)
    U = (rng.standard_normal((nU, D)) * 0.2).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.2).astype(np.float32)
    res, used = [], []
    for overlap in (True, False):
        pipe = ops.PipelinedSgd(chunk=4, min_triplets=1, overlap=overlap, chain=False)
        Ud, Id = T(U, dev), T(I, dev)
        h = pipe.plan(Ud, [(Id, T(u, dev), T(p, dev), T(n, dev))], B, lr=lr if case != "hot_rows" else None)
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(h, 0, lr, losses)
        torch.cuda.synchronize()
        res.append((Ud, Id, losses))
        used.append(dict(pipe.stats))
    assert used[1]["graph_replays"] == 0 and used[1]["plain_calls"] == 3
    # plans of 4, 4 and 3 batches: the two full ones replay the captured overlapped stream when they qualify
    assert used[0]["graph_replays"] == (2 if case == "qualifies" else 0), (case, used)
    assert used[0]["graph_replays"] + used[0]["plain_calls"] == 3
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_err(res[0][2].cpu().numpy(), res[1][2].cpu().numpy()) < 1e-6
    Uo, Io = U.copy(), I.copy()
    ref = [oracle.bprmf_step_sgd(Uo, Io, u[k * B:(k + 1) * B], p[k * B:(k + 1) * B], n[k * B:(k + 1) * B], lr, 0.0)
           for k in range(nb)]
    assert rel_err(res[0][2].cpu().numpy(), np.asarray(ref)) < TOL
    assert rel_err(res[0][0].cpu().numpy(), Uo) < TOL and rel_err(res[0][1].cpu().numpy(), Io) < TOL


def test_overlap_entry_refuses_lists_beyond_capacity_and_missing_events(ops):
    from whisprrec_amd import abi
    dev = torch.device("cuda:0")
    nU, nI, B, nb = 20_000, 50_000, 8192, 3
    u, p, n = _epoch(4, nU, nI, nb * B)
    plan, _ = _arena_plan(ops, dev, u, p, n, B, nU, nI)
    assert plan.overlap is not None
    tabs = ops.BprmfTables(torch.zeros(nU, 64, device=dev), torch.zeros(nI, 64, device=dev))
    side, ev = ops.side_stream(dev), ops.OverlapEvents(dev)
    bad = dict(plan.overlap)
    bad["def_count_host"] = torch.full((nb,), plan.overlap["cap"] + 1, dtype=torch.int32)
    plan.overlap = bad
    with pytest.raises(abi.WhisprRecHipError, match="exceed the list capacity"):
        tabs.run_sgd_overlap(plan, 0, nb, 0.1, None, side, ev)
    ev.n = 3
    with pytest.raises(abi.WhisprRecHipError, match="at least 5 events"):
        tabs.run_sgd_overlap(plan, 0, nb, 0.1, None, side, ev)
    torch.cuda.synchronize()
