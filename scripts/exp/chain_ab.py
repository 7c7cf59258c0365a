"""A/B of the chained step launch (wr_bprmf_run_sgd_chain) against the two-launch step (wr_bprmf_run_sgd): tables must be
bit-identical; steps-only time per step, plans prebuilt.  Usage: python scripts/exp/chain_ab.py [users items D B nb]"""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, ".")
from whisprrec_amd import abi  # noqa: E402
abi.LIB_PATH = os.path.abspath(os.environ.get("WR_LIB", abi.LIB_PATH))
from whisprrec_amd import hip_ops as ops  # noqa: E402

nU, nI, D, B, nb = (int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else (1_000_000, 1_000_000, 64, 65536, 32)))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
U0 = torch.randn(nU, D, device=dev, generator=g) * 0.1
I0 = torch.randn(nI, D, device=dev, generator=g) * 0.1
arena = ops.PlanArena(dev, nb * B, B, overlap_items=nI)
plan = ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
torch.cuda.synchronize()
assert plan.overlap is not None, "plan does not qualify"
dc = plan.overlap["def_count_host"].numpy()
print("deferred runs per batch: min %d max %d (cap %d)" % (dc[1:].min(), dc.max(), plan.overlap["cap"]))
lr = 0.05


def run(kind, reps=1):
    tabs = ops.BprmfTables(U0.clone(), I0.clone())
    losses = torch.empty(nb, dtype=torch.float32, device=dev)
    fn = tabs.run_sgd if kind == "plain" else tabs.run_sgd_chain
    fn(plan, 0, nb, lr, losses)
    torch.cuda.synchronize()
    out = (tabs.U.clone(), tabs.I.clone(), losses.clone())
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(plan, 0, nb, lr, losses)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / nb * 1e6)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * nb)]
    fn(plan, 0, nb, lr, losses, phase_events=ev)
    torch.cuda.synchronize()
    ks = [ev[4 * k].elapsed_time(ev[4 * k + 1]) * 1e3 for k in range(1, nb)]
    print("  %s: kernel carrying the user phase, us: mean %.2f min %.2f max %.2f" % (kind, sum(ks) / len(ks), min(ks), max(ks)))
    if kind != "plain":
        tabs.check_chain()
    return out, ts


a, ta = run("plain", 5)
b, tb = run("chain", 5)
print("tables equal:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]),
      "loss max rel diff: %.2e" % float(((a[2] - b[2]).abs() / a[2].abs()).max()))
print("plain us/step:", " ".join("%.2f" % t for t in ta))
print("chain us/step:", " ".join("%.2f" % t for t in tb))
