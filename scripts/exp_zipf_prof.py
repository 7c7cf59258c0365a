import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
nU = nI = 1_000_000; D = 64; B = 65536; NB = 32
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
r = torch.rand(NB * B, generator=g, device=dev, dtype=torch.float64)
p = (torch.exp(r * np.log(nI)).to(torch.int64) - 1).clamp_(0, nI - 1).to(torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
tabs = hip_ops.BprmfTables(U, I)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
print("builder", plan.builder, "hot counts batch0", plan.hot["counts_host"][:4].tolist() if plan.hot else None)
for rep in range(3):
    tabs.run_sgd(plan, 0, NB, 0.05)
torch.cuda.synchronize()
