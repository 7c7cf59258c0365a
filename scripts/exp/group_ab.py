"""A/B on the GPU box: the sorted-plan step stream (chained launches) against the group-plan stream (wr_group.hip), steps only
and plan builds alone, at the headline shape.  python scripts/exp/group_ab.py [--users N --items N --batch B --emb D --steps K]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from whisprrec_amd import abi  # noqa: E402
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
from whisprrec_amd import hip_ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=1_000_000)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--emb", type=int, default=64)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--plan-only", action="store_true")
ap.add_argument("--sort-users", action="store_true", help="each batch's triplets pre-sorted by user (access-order experiment)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, D, K = a.batch, a.emb, a.steps
g = torch.Generator(device=dev)
g.manual_seed(3407)
U = torch.randn(a.users, D, generator=g, device=dev) * float(np.sqrt(2.0 / (a.users + D)))
I = torch.randn(a.items, D, generator=g, device=dev) * float(np.sqrt(2.0 / (a.items + D)))
n = K * B
u = torch.randint(0, a.users, (n,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, a.items, (n,), generator=g, device=dev, dtype=torch.int32)
ng = torch.randint(1, a.items, (n,), generator=g, device=dev, dtype=torch.int32)
if a.sort_users:
    o = torch.argsort(u.view(K, B).long(), dim=1) + (torch.arange(K, device=dev) * B)[:, None]
    o = o.reshape(-1)
    u, p, ng = u[o].contiguous(), p[o].contiguous(), ng[o].contiguous()
tabs = hip_ops.BprmfTables(U, I)
losses = torch.empty(K, dtype=torch.float32, device=dev)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


arena = hip_ops.PlanArena(dev, n, B, overlap_items=a.items)
t_plan1 = timed(lambda: hip_ops.BatchPlan(u, p, ng, B, a.users, a.items, arena=arena, overlap=True, validate=False), a.reps)
plan1 = hip_ops.BatchPlan(u, p, ng, B, a.users, a.items, arena=arena, overlap=True)
garena = hip_ops.GroupArena(dev, n, B, a.users, a.items)
t_plan2 = timed(lambda: hip_ops.GroupPlan(u, p, ng, B, a.users, a.items, arena=garena).finish(), a.reps)
plan2 = hip_ops.GroupPlan(u, p, ng, B, a.users, a.items, arena=garena)
print("plan build per batch: sorted %.2f us (median %.2f), group %.2f us (median %.2f); overflow=%s long_run=%s" %
      (t_plan1[0] / K * 1e6, t_plan1[1] / K * 1e6, t_plan2[0] / K * 1e6, t_plan2[1] / K * 1e6, plan2.overflow, plan2.long_run))
if a.plan_only:
    sys.exit(0)
if plan1.overlap is not None:
    t1 = timed(lambda: tabs.run_sgd_chain(plan1, 0, K, 0.05, losses), a.reps)
    print("sorted plan, chained launches: %.2f us/step (median %.2f)" % (t1[0] / K * 1e6, t1[1] / K * 1e6))
t0 = timed(lambda: tabs.run_sgd(plan1, 0, K, 0.05, losses), a.reps)
print("sorted plan, two launches:     %.2f us/step (median %.2f)" % (t0[0] / K * 1e6, t0[1] / K * 1e6))
t2 = timed(lambda: tabs.run_sgd_group(plan2, 0, K, 0.05, losses), a.reps)
print("group plan:                    %.2f us/step (median %.2f)" % (t2[0] / K * 1e6, t2[1] / K * 1e6))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K)]
for e in evs:
    e.record()
torch.cuda.synchronize()
tabs.run_sgd_group(plan2, 0, K, 0.05, losses, events=evs)
torch.cuda.synchronize()
kt = [evs[2 * k].elapsed_time(evs[2 * k + 1]) * 1e3 for k in range(K)]
print("group step launches (attached events): first %.2f us, others mean %.2f min %.2f max %.2f" %
      (kt[0], float(np.mean(kt[1:])), min(kt[1:]), max(kt[1:])))
tabs.check_chain()
d = plan2.decode()
fl = d["flags"]
pc = lambda x: int(np.unpackbits(np.ascontiguousarray(x).view(np.uint8)).sum())
nb = fl.shape[0]
print("per batch: user-shared %.0f, p-shared %.0f, n-shared %.0f, deferred %.0f; user list %.0f, item list %.0f" %
      (pc(fl[:, :, 0]) / nb, pc(fl[:, :, 1]) / nb, pc(fl[:, :, 2]) / nb, pc(fl[:, :, 3]) / max(nb - 1, 1),
       sum(len(v[0]) for v in d["users"].values()) / nb, sum(len(v[0]) for v in d["items"].values()) / nb))
