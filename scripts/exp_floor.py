import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
for (nU, B, NB) in ((1000, 256, 400), (100000, 4096, 400), (1000000, 16384, 200), (1000000, 65536, 60), (1000000, 262144, 20), (1000000, 1048576, 8)):
    nI = nU; D = 64
    g = torch.Generator(device=dev); g.manual_seed(1)
    U = torch.randn(nU, D, generator=g, device=dev) * 0.01
    I = torch.randn(nI, D, generator=g, device=dev) * 0.01
    u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    tabs = hip_ops.BprmfTables(U, I)
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
    tabs.run_sgd(plan, 0, NB, 0.05); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    tabs.run_sgd(plan, 0, NB, 0.05)
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / NB * 1e3
    print("rows %8d B %8d: %.2f us/step GPU (host enqueue %.2f us/step) -> %.3f G triplets/s, %.2f TB/s algorithmic" % (nU, B, t, (t1 - t0) / NB * 1e6, B / t / 1e3, 1548 * B / t / 1e6))
