#!/usr/bin/env python3
"""Micro-experiments on the GPU box: time individual entry points on the configs[1] shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops, abi

dev = torch.device("cuda:0")
nU = nI = 1_000_000; D = 64; B = 65536; NB = 40
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nU, D, generator=g, device=dev) * 0.01
I = torch.randn(nI, D, generator=g, device=dev) * 0.01
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev)

def timeit(fn, iters):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

# E1: forward only (reads 3 rows per triplet)
def fwd(k):
    sl = slice(k * B, (k + 1) * B)
    hip_ops.bpr_fwd(U, I, u[sl], p[sl], n[sl], scores=False)
fwd(0)
t = timeit(fwd, NB)
print("fwd-only (2 launches): %.2f us/step  -> %.2f TB/s of row reads" % (t, 768 * B / t / 1e6))

tabs = hip_ops.BprmfTables(U, I)
plan = hip_ops.BatchPlan(u.to(torch.int32), p.to(torch.int32), n.to(torch.int32), B, nU, nI)
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
L = abi.lib()
for rep in range(2):
    for v in variants:
        if hasattr(L, "wr_internal_set_user_slots"):
            L.wr_internal_set_user_slots(v)
        tabs.run_sgd(plan, 0, 4, 0.05)
        t = timeit(lambda k: tabs.run_sgd(plan, 0, NB, 0.05), 1) / NB
        print("variant %d: step (user+item phase): %.2f us" % (v, t))

# the chained step launch (one launch per step: user phase of step k + item phase of step k-1)
arena = hip_ops.PlanArena(dev, NB * B, B, overlap_items=nI)
plan_c = hip_ops.BatchPlan(u.to(torch.int32), p.to(torch.int32), n.to(torch.int32), B, nU, nI, arena=arena, overlap=True)
tabs.run_sgd_chain(plan_c, 0, 4, 0.05)
for rep in range(2):
    t = timeit(lambda k: tabs.run_sgd_chain(plan_c, 0, NB, 0.05), 1) / NB
    print("chained step launch: %.2f us/step" % t)
tabs.check_chain()

# the group-plan step stream (round 3: no per-batch sort; one launch per step = triplets of batch k + tiles of batch k-1)
u32, p32, n32 = u.to(torch.int32), p.to(torch.int32), n.to(torch.int32)
garena = hip_ops.GroupArena(dev, NB * B, B, nU, nI)
for rep in range(2):
    t = timeit(lambda k: hip_ops.GroupPlan(u32, p32, n32, B, nU, nI, arena=garena).finish(), 1) / NB
    print("group plan build: %.2f us/batch" % t)
plan_g = hip_ops.GroupPlan(u32, p32, n32, B, nU, nI, arena=garena)
tabs.run_sgd_group(plan_g, 0, 4, 0.05)
for rep in range(2):
    t = timeit(lambda k: tabs.run_sgd_group(plan_g, 0, NB, 0.05), 1) / NB
    print("group step launch: %.2f us/step" % t)
tabs.check_chain()
