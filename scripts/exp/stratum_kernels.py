"""per-kernel times of the two-launch step at the 8-GPU schedule's per-rank shape (user shard 125 K rows, item part 62.5 K
rows, B = 65,536) and neighbours; plans prebuilt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops as ops
dev = torch.device("cuda:0")
D, B, nb = 64, 65536, 32
for nU, nI in ((125_000, 62_500), (125_000, 125_000), (250_000, 125_000), (500_000, 250_000), (1_000_000, 1_000_000)):
    g = torch.Generator(device=dev).manual_seed(1)
    u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
    U = torch.randn(nU, D, device=dev, generator=g) * 0.1
    I = torch.randn(nI, D, device=dev, generator=g) * 0.1
    plan = ops.BatchPlan(u, p, n, B, nU, nI)
    tabs = ops.BprmfTables(U, I)
    tabs.run_sgd(plan, 0, nb, 0.05)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tabs.run_sgd(plan, 0, nb, 0.05); e1.record(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * nb)]
    tabs.run_sgd(plan, 0, nb, 0.05, phase_events=ev); torch.cuda.synchronize()
    ua = sum(ev[4 * k].elapsed_time(ev[4 * k + 1]) for k in range(nb)) / nb * 1e3
    ia = sum(ev[4 * k + 2].elapsed_time(ev[4 * k + 3]) for k in range(nb)) / nb * 1e3
    multi = float(((plan.tp < 0).sum() + (plan.tn < 0).sum()).item()) / (2 * nb * B)
    print("%8d x %8d: step %.2f us (user %.2f, item %.2f); occurrences on shared item rows %.0f %%, hot: %s" %
          (nU, nI, e0.elapsed_time(e1) / nb * 1e3, ua, ia, 100 * multi, plan.hot is not None))
