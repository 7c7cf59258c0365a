"""folded lazy Adam at the headline shape: run under rocprofv3 --kernel-trace for per-kernel durations"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
nU = nI = 1_000_000; D = 64; B = int(os.environ.get("WR_B", "65536")); NB = int(os.environ.get("WR_NB", "48"))
g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
U = torch.randn(nU, D, generator=g, device=dev) * 0.01
I = torch.randn(nI, D, generator=g, device=dev) * 0.01
st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, 0.0, fold=(os.environ.get("FOLD", "1") == "1"))
st.run(plan, 0, NB)
torch.cuda.synchronize()
