#!/usr/bin/env python3
"""Bounded randomized sweep of the round-2 paths on the GPU (not part of the test suite):
  * step stream with chained launches against the two-launch stream (tables bitwise) and against the oracle, random table
    sizes / batch sizes / chunk lengths / id distributions (uniform, power-law items, a hot row);
  * single-workgroup plan builder, counting and bitonic form, against the radix-sort builder (arrays bitwise);
  * bounded-lag lazy Adam against the unbounded one (tables and moments bitwise after flush)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from whisprrec_amd import hip_ops

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, t_note = time.time() + budget, time.time()
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n_cases = n_chained = 0
while time.time() < t_end:
    # ---- chained step stream
    D = int(rng.choice([32, 64, 64, 96, 128]))
    B = int(rng.choice([8192, 8192, 16384, 12000]))
    nU, nI = int(rng.randint(B // 2, 40 * B)), int(rng.randint(B, 40 * B))
    nb = int(rng.randint(2, 9))
    N = nb * B - int(rng.randint(0, B // 2))
    kind = int(rng.randint(0, 3))
    u, p, n = rng.randint(0, nU, N), rng.randint(0, nI, N), rng.randint(0, nI, N)
    if kind == 1:
        p = np.minimum((rng.pareto(1.2, N) * nI / 50).astype(np.int64), nI - 1)
    elif kind == 2:
        p[:: int(rng.randint(20, 200))] = int(rng.randint(0, nI))
    U = (rng.standard_normal((nU, D)) * 0.3).astype(np.float32); I = (rng.standard_normal((nI, D)) * 0.3).astype(np.float32)
    chunk = int(rng.randint(1, 6))
    res = []
    for chain in (True, False):
        pipe = hip_ops.PipelinedSgd(chunk=chunk, min_triplets=1, chain=chain)
        Ud, Id = T(U), T(I)
        h = pipe.plan(Ud, [(Id, T(u), T(p), T(n))], B)
        losses = torch.empty(h["segs"][0]["nb"], dtype=torch.float32, device=dev)
        pipe.run(h, 0, 0.1, losses)
        torch.cuda.synchronize()
        h["segs"][0]["tabs"].check_chain()
        res.append((Ud, Id, losses, pipe.stats["chain_calls"]))
    n_chained += int(res[0][3] > 0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), ("chain tables", D, B, nU, nI, nb, kind, chunk)
    Uo, Io = U.copy(), I.copy()
    for k in range(res[0][2].numel()):
        sl = slice(k * B, min(N, (k + 1) * B))
        lo = oracle.bprmf_step_sgd(Uo, Io, u[sl], p[sl], n[sl], 0.1, 0.0)
        assert abs(float(res[0][2][k]) - lo) <= 2e-5 * max(abs(lo), 1e-3), ("chain loss", D, B, nU, nI, kind, k)
    e = max(np.abs(res[0][0].cpu().numpy() - Uo).max() / np.abs(Uo).max(), np.abs(res[0][1].cpu().numpy() - Io).max() / np.abs(Io).max())
    assert e < (2e-4 if kind else 2e-5), ("chain tables vs oracle", D, B, nU, nI, kind, e)
    # ---- single-workgroup plan builder, both forms
    Bs = int(rng.choice([1, 33, 512, 2048, 3000, 4096]))
    nUs, nIs = int(rng.choice([1, 7, 900, 6040, 15000, 200000])), int(rng.choice([1, 5, 700, 3706, 15000, 200000]))
    nbs = int(rng.randint(1, 4))
    Ns = max(1, nbs * Bs - int(rng.randint(0, Bs)))
    us, ps, ns = rng.randint(0, nUs, Ns), rng.randint(0, nIs, Ns), rng.randint(0, nIs, Ns)
    if rng.randint(0, 2):
        ps = np.minimum((rng.pareto(0.9, Ns) * 3).astype(np.int64), nIs - 1)
    ref = hip_ops.BatchPlan(T(us), T(ps), T(ns), Bs, nUs, nIs, builder="generic", hot=False, keep_orig=True)
    for form in ("count", "bitonic"):
        if form == "bitonic":
            os.environ["WR_PLAN_SMALL"] = "bitonic"
        else:
            os.environ.pop("WR_PLAN_SMALL", None)
        sm = hip_ops.BatchPlan(T(us), T(ps), T(ns), Bs, nUs, nIs, builder="small", hot=False, keep_orig=True)
        for nm in ("tu", "tp", "tn", "torig", "oc_item", "oc_src"):
            assert torch.equal(getattr(sm, nm), getattr(ref, nm)), ("small builder", form, nm, Bs, nUs, nIs, Ns)
    os.environ.pop("WR_PLAN_SMALL", None)
    # ---- bounded-lag lazy Adam against the unbounded one
    Bl, nbl = int(rng.choice([64, 257, 1024])), int(rng.randint(4, 40))
    nUl, nIl, Dl = int(rng.randint(50, 20000)), int(rng.randint(50, 20000)), int(rng.choice([16, 32, 64]))
    ul, pl, nl = rng.randint(0, nUl, Bl * nbl), rng.randint(0, nIl, Bl * nbl), rng.randint(0, nIl, Bl * nbl)
    Ul0 = (rng.standard_normal((nUl, Dl)) * 0.3).astype(np.float32); Il0 = (rng.standard_normal((nIl, Dl)) * 0.3).astype(np.float32)
    plan = hip_ops.BatchPlan(T(ul), T(pl), T(nl), Bl, nUl, nIl)
    l2 = float(rng.choice([0.0, 1e-3]))
    outs = []
    for lag in (0, int(rng.choice([2, 5, 16, 64]))):
        Ux, Ix = T(Ul0), T(Il0)
        st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(Ux, Ix), "Adam", 1e-2, l2, fold=False, max_lag=lag)
        half = nbl // 2
        st.run(plan, 0, half); st.run(plan, half, nbl - half); st.flush()
        outs.append((Ux, Ix, st.m_u, st.v_u, st.m_i, st.v_i))
    for a, b in zip(*outs):
        assert torch.equal(a, b), ("bounded lag", Bl, nbl, nUl, nIl, Dl, l2)
    n_cases += 1
    if time.time() - t_note > 30:
        t_note = time.time()
        print("stress_round2: %d cases so far (%d with chained launches)" % (n_cases, n_chained), flush=True)
print("stress_round2: %d random cases ok (%d with chained launches)" % (n_cases, n_chained))
