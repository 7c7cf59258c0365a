"""The plan-time marks of the chained step launch (wr_bprmf_plan_overlap_marks / wr_bprmf_plan_overlap_deferred): which user
runs of batch k+1 read an item row that the item phase of batch k rewrites (src/helpers/BaseRunner.py:194-200: strictly
sequential, batch-synchronous steps — such a run must wait for that item phase).  The marks equal a NumPy restatement of
their definition; the steps that use them are checked in test_hip_chain.py."""
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
