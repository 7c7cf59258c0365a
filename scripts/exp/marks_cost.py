"""plan build time with and without the chained launch's marks (64 batches of 65,536; 1M x 1M ids), in isolation"""
import sys
import time
import torch
sys.path.insert(0, ".")
from whisprrec_amd import hip_ops as ops  # noqa: E402
dev = torch.device("cuda:0")
nU = nI = 1_000_000
B, nb = 65536, 64
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
for marks in (False, True, False, True):
    arena = ops.PlanArena(dev, nb * B, B, overlap_items=nI if marks else 0)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=marks, defer=True, validate=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6 / nb)
        plan.finish()
    print("marks=%s: us per batch %s" % (marks, " ".join("%.2f" % t for t in ts)))
