"""PipelinedSgd over an epoch handed over as int64 columns (the reference's layout) against int32 columns: both on group plans"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
nU = nI = 1_000_000
D, B, NB = 64, 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 305
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nU, D, generator=g, device=dev) * 0.01
I = torch.randn(nI, D, generator=g, device=dev) * 0.01
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev)
pipe = hip_ops.PipelinedSgd(chunk=64)        # kept across epochs, as HipRunner keeps it: arenas are made once
for dt in (torch.int32, torch.int64, torch.int32, torch.int64):
    cols = [c.to(dt) for c in (u, p, n)]
    losses = torch.empty(NB, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = pipe.plan(U, [(I, *cols)], B)
    t1 = time.perf_counter()
    pipe.run(h, 0, 0.05, losses)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print("%s: %.2f ms per epoch of %d steps = %.2f G triplets/s (plan() %.2f ms, run() returned after %.2f ms), %s" %
          (dt, dt_s * 1e3, NB, NB * B / dt_s / 1e9, (t1 - t0) * 1e3, (t2 - t0) * 1e3, pipe.stats), flush=True)
