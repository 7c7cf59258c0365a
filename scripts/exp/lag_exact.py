"""bounded lag at scale: tables after N lazy Adam steps must be bit-identical whatever the window (0 = unbounded)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
nU, nI, D, B, NB = 300_000, 200_000, 64, 2048, 700
g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
U0 = torch.randn(nU, D, generator=g, device=dev) * 0.01
I0 = torch.randn(nI, D, generator=g, device=dev) * 0.01
outs = {}
for lag in (0, 0, 64, 64, 7, 300):
    U, I = U0.clone(), I0.clone()
    st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, 0.0, fold=False, max_lag=lag)
    l1 = st.run(plan, 0, 300); l2 = st.run(plan, 300, NB - 300)
    st.flush()
    torch.cuda.synchronize()
    key = (lag, len([k for k in outs if k[0] == lag]))
    outs[key] = (U, I, torch.cat([l1, l2]))
ref = outs[(0, 0)]
for k, v in outs.items():
    print(k, "U equal:", torch.equal(v[0], ref[0]), "I equal:", torch.equal(v[1], ref[1]), "loss equal:", torch.equal(v[2], ref[2]),
          "max |dU| %.3e" % float((v[0] - ref[0]).abs().max()))
