"""Host cost of the per-batch Python loop of the lazy Adam step at the reference's default batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
nU = nI = 1_000_000; D = 64
for B, NB in ((2048, 512), (65536, 64)):
    U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
    u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    tabs = hip_ops.BprmfTables(U, I)
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
    st = hip_ops.LazyOptimizerState(tabs, "Adam", 1e-3, 0.0)
    losses = torch.empty(NB, device=dev)
    for k in range(NB): st.step(plan, k, loss_out=losses[k])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(NB): st.step(plan, k, loss_out=losses[k])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d python loop: host enqueue %.1f us/step, wall %.1f us/step" % (B, (t1 - t0) / NB * 1e6, (t2 - t0) / NB * 1e6))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.run(plan, 0, NB, losses)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=%d native loop: host enqueue %.1f us/step, wall %.1f us/step" % (B, (t1 - t0) / NB * 1e6, (t2 - t0) / NB * 1e6))
