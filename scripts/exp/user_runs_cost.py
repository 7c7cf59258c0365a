"""How much of the user phase is the serial walk over users with several triplets in a batch?  Same tables and batch size,
fresh rows every step; users drawn with replacement (headline data: ~3 % of the triplets sit in multi-triplet runs) against
users drawn without replacement inside each batch (no runs at all), items random in both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
nU = nI = 1_000_000; D = 64; B = 65536; NB = 64
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
u_rep = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
u_dis = torch.cat([torch.randperm(nU, generator=g, device=dev)[:B].to(torch.int32) for _ in range(NB)])
for name, u in (("users with replacement", u_rep), ("users distinct inside a batch", u_dis)):
    tabs = hip_ops.BprmfTables(U, I)
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
    tabs.run_sgd(plan, 0, NB, 0.05); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * NB)]
    tabs.run_sgd(plan, 0, NB, 0.05, phase_events=ev); torch.cuda.synchronize()
    ua = sum(ev[4 * k].elapsed_time(ev[4 * k + 1]) for k in range(NB)) / NB * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tabs.run_sgd(plan, 0, NB, 0.05); e1.record(); torch.cuda.synchronize()
    print("%-32s step %.2f us, user phase (events) %.2f us" % (name, e0.elapsed_time(e1) / NB * 1e3, ua))
