import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
nI = 2_000_000; D = 64
g = torch.Generator(device=dev); g.manual_seed(1)
I = torch.randn(nI, D, generator=g, device=dev)
for n in (196608, 786432):
    idxs = [torch.randint(0, nI, (n,), generator=g, device=dev) for _ in range(20)]
    sidx = [torch.sort(i)[0] for i in idxs]
    for name, L in (("random", idxs), ("sorted", sidx)):
        hip_ops.gather_rows(I, L[0]); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in L:
            hip_ops.gather_rows(I, i)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / len(L) * 1e3
        print("gather %s n=%d: %.2f us -> read %.2f TB/s (+ equal sequential write)" % (name, n, t, n * 256 / t / 1e6))
