import sys, os
sys.path.insert(0, "/root/repo")
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
nU = nI = 1_000_000; D = 64; B = 65536; NB = 48
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
tabs = hip_ops.BprmfTables(U, I)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
st = hip_ops.LazyOptimizerState(tabs, "Adam", 1e-3, 0.0)
for rep in range(2):
    for k in range(NB):
        st.step(plan, k)
st.flush()
torch.cuda.synchronize()
