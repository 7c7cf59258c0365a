#!/usr/bin/env python3
"""Bounded randomized sweep of the round-3 paths on the GPU (not part of the test suite):
  * step stream on GROUP plans (no per-batch sort, one launch per step; falls back to sorted plans on skewed ids) against the
    oracle, run twice (tables and losses bitwise equal), random table / batch / chunk sizes, D, id distributions
    (uniform, mildly skewed items, a hot row, hashed table sizes beyond 2^21 rows);
  * the group plan's arrays against the NumPy restatement (oracle.group_plan);
  * scatter-add through the row plan against the sorted path (bitwise), random sizes, padding, clumped and hot rows,
    one call and several segments planned ahead;
  * the folded Adam step as one launch per step against the two-launch form (tables, moments, stamps bitwise), random
    shapes, weight decay, cuts into calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from whisprrec_amd import abi, hip_ops

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, t_note = time.time() + budget, time.time()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n_cases = n_group = n_fallback = n_scatter = n_slow = n_adam_chain = 0


def sorted_reference(table, idx, src, alpha, pad):
    keys = torch.where((idx == pad) | (idx < 0) | (idx >= table.shape[0]), torch.full_like(idx, table.shape[0]), idx)
    srt, perm = torch.sort(keys, stable=True)
    srt, perm = srt.to(torch.int32), perm.to(torch.int32)
    abi.check(abi.lib().wr_apply_rows_sorted(table.data_ptr(), table.shape[0], table.shape[1], srt.data_ptr(), perm.data_ptr(),
                                             src.data_ptr(), idx.numel(), alpha, torch.cuda.current_stream().cuda_stream), "apply")
    return table


while time.time() < t_end:
    # ---- step stream on group plans
    D = int(rng.choice([32, 64, 64, 128]))
    B = int(rng.choice([4096, 8192, 8192, 16384, 12000]))
    big = rng.randint(0, 6) == 0
    nU = int(rng.randint(12 * B, 60 * B)) if not big else int(rng.randint(2_200_000, 3_000_000))
    nI = int(rng.randint(12 * B, 60 * B)) if not big else int(rng.randint(2_200_000, 5_000_000))
    nb = int(rng.randint(2, 8))
    N = nb * B - int(rng.randint(0, B // 2))
    kind = int(rng.randint(0, 4))
    u, p, n = rng.randint(0, nU, N), rng.randint(0, nI, N), rng.randint(1, nI, N)
    if kind == 1:
        p = np.minimum((rng.pareto(1.5, N) * nI / 20).astype(np.int64), nI - 1)          # mild skew: long lists, some long runs
    elif kind == 2:
        p[:: int(rng.randint(40, 400))] = int(rng.randint(0, nI))                         # one hot row
    elif kind == 3:
        u[:: int(rng.randint(3, 9))] = rng.randint(0, nU, 1)[0] // 7 * 7                   # a hot user
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32)
    I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    chunk = int(rng.randint(1, 5))
    res = []
    for rep in range(2):
        pipe = hip_ops.PipelinedSgd(chunk=chunk, min_triplets=1)
        Ud, Id = T(U), T(I)
        h = pipe.plan(Ud, [(Id, T(u).to(torch.int32), T(p).to(torch.int32), T(n).to(torch.int32))], B)
        losses = torch.empty(h["segs"][0]["nb"], dtype=torch.float32, device=dev)
        pipe.run(h, 0, 0.1, losses)
        torch.cuda.synchronize()
        h["segs"][0]["tabs"].check_chain()
        res.append((Ud, Id, losses, dict(pipe.stats)))
    st = res[0][3]
    n_group += int(st["group_calls"] > 0)
    n_fallback += int(st["group_fallbacks"] > 0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]), \
        ("run-to-run", D, B, nU, nI, nb, kind, chunk)
    Uo, Io = U.copy(), I.copy()
    for k in range(res[0][2].numel()):
        sl = slice(k * B, min(N, (k + 1) * B))
        lo = oracle.bprmf_step_sgd(Uo, Io, u[sl], p[sl], n[sl], 0.1, 0.0)
        assert abs(float(res[0][2][k]) - lo) <= 2e-5 * max(abs(lo), 1e-3), ("loss", D, B, nU, nI, kind, k, float(res[0][2][k]), lo)
    e = max(np.abs(res[0][0].cpu().numpy() - Uo).max() / np.abs(Uo).max(), np.abs(res[0][1].cpu().numpy() - Io).max() / np.abs(Io).max())
    assert e < (2e-4 if kind else 2e-5), ("tables vs oracle", D, B, nU, nI, kind, e, st)
    # ---- plan arrays against the restatement (one chunk, uniform ids so that nothing overflows)
    Bp = int(rng.choice([1024, 4096, 8192]))
    nUp, nIp = int(rng.randint(6 * Bp, 40 * Bp)), int(rng.randint(6 * Bp, 40 * Bp))
    if rng.randint(0, 4) == 0:
        nIp = int(rng.randint(2_100_000, 4_500_000))
    Np = int(rng.randint(1, 4)) * Bp - int(rng.randint(0, Bp // 2))
    up, pp, npn = rng.randint(0, nUp, Np).astype(np.int32), rng.randint(0, nIp, Np).astype(np.int32), rng.randint(0, nIp, Np).astype(np.int32)
    plan = hip_ops.GroupPlan(T(up), T(pp), T(npn), Bp, nUp, nIp)
    if not plan.overflow:
        got, ref = plan.decode(), oracle.group_plan(up, pp, npn, Bp, nUp, nIp)
        assert np.array_equal(got["flags"], ref["flags"]), ("plan flags", Bp, nUp, nIp, Np)
        for side in ("users", "items"):
            for key, (rows, srcs) in ref[side].items():
                g = got[side][key]
                assert np.array_equal(g[0], rows) and np.array_equal(g[1], srcs), ("plan list", side, key, Bp, nUp, nIp, Np)
    # ---- scatter-add through the row plan
    n_rows = int(rng.choice([16_384, 20_000, 100_000, 1_000_000, 3_000_000]))
    n = int(rng.choice([17, 1000, 45_056, 114_688, 200_000, 262_144]))
    Ds = int(rng.choice([32, 64, 128]))
    idx = rng.randint(0, n_rows, n).astype(np.int64)
    mode = int(rng.randint(0, 4))
    if mode == 1:
        idx[rng.permutation(n)[: n // 3]] = int(rng.randint(0, n_rows))                  # a hot row
    elif mode == 2:
        k = n // 4
        idx[rng.permutation(n)[:k]] = rng.randint(0, max(1, n_rows // 25), k)             # a clump of rows
    idx[::13] = 0
    if n > 5:
        idx[5] = -1
    src = torch.randn(n, Ds, device=dev)
    idx_d = T(idx)
    base = torch.randn(n_rows, Ds, device=dev) if n_rows <= 100_000 else torch.zeros(n_rows, Ds, device=dev)
    got = hip_ops.scatter_add_rows(base.clone(), idx_d, src, padding_idx=0, alpha=0.5)
    ref = sorted_reference(base.clone(), idx_d, src, 0.5, 0)
    assert torch.equal(got, ref), ("scatter", n_rows, n, Ds, mode)
    S = int(rng.randint(1, 4))
    stride = max(n // S, 1)
    sp = hip_ops.ScatterPlan(idx_d[:S * stride].view(S, stride), n_rows, padding_idx=0)
    n_slow += int(sp.slow)
    tab2, ref2 = base.clone(), base.clone()
    for sgm in range(S):
        sl = slice(sgm * stride, (sgm + 1) * stride)
        sp.apply(tab2, sgm, stride, src[sl].contiguous(), alpha=-1.0)
        sorted_reference(ref2, idx_d[sl].contiguous(), src[sl].contiguous(), -1.0, 0)
    assert torch.equal(tab2, ref2), ("scatter segments", n_rows, n, Ds, mode, S)
    n_scatter += 1
    # ---- folded Adam, one launch per step against two (tables, moments, stamps bitwise)
    Da = int(rng.choice([32, 64, 128]))
    Ba = int(rng.choice([8192, 8192, 16384]))
    nUa, nIa = int(rng.randint(2 * Ba, 30 * Ba)), int(rng.randint(2 * Ba, 30 * Ba))
    nba = int(rng.randint(3, 9))
    Na = nba * Ba - int(rng.randint(0, Ba // 2))
    l2a = float(rng.choice([0.0, 0.0, 1e-3]))
    ua = T(rng.randint(0, nUa, Na).astype(np.int32)); pa = T(rng.randint(0, nIa, Na).astype(np.int32))
    na = T(rng.randint(1, nIa, Na).astype(np.int32))
    arena = hip_ops.PlanArena(dev, Na, Ba, overlap_items=nIa)
    plan_a = hip_ops.BatchPlan(ua, pa, na, Ba, nUa, nIa, arena=arena, overlap=True)
    if plan_a.overlap is not None and plan_a.hot is None:
        U0 = torch.randn(nUa, Da, device=dev) * 0.1
        I0 = torch.randn(nIa, Da, device=dev) * 0.1
        cuts = sorted(set([0, nba] + [int(x) for x in rng.randint(1, nba, int(rng.randint(0, 3)))]))
        outs = []
        for chain in (False, True):
            st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U0.clone(), I0.clone()), "Adam", 1e-2, l2a, fold=True)
            st.chain = chain
            for a, b in zip(cuts[:-1], cuts[1:]):
                st.run(plan_a, a, b - a)
            torch.cuda.synchronize()
            st.tabs.check_chain()
            outs.append((st.tabs.U, st.tabs.I, st.m_u, st.v_u, st.m_i, st.v_i, st.last_u, st.last_i, st.chain_calls))
        for x, y in zip(outs[0][:-1], outs[1][:-1]):
            assert torch.equal(x, y), ("adam chain", Da, Ba, nUa, nIa, nba, l2a, cuts)
        n_adam_chain += int(outs[1][-1] > 0)
    n_cases += 1
    if time.time() - t_note > 30:
        print("%d cases (%d on group plans, %d fell back), %d scatter cases (%d with a brute-force range)" %
              (n_cases, n_group, n_fallback, n_scatter, n_slow), flush=True)
        t_note = time.time()
print("done: %d cases (%d on group plans, %d fell back to sorted plans), %d scatter cases (%d with a brute-force range), "
      "%d Adam cases with chained launches" % (n_cases, n_group, n_fallback, n_scatter, n_slow, n_adam_chain))
