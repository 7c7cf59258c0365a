"""C5 gather alone, many calls (for rocprofv3 --kernel-trace): is the 12-15 us per call of bench_configs.py kernel time or host time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
for n_rows in (3706, 1_000_000):
    idx = torch.randint(0, n_rows, (45056,), generator=g, device=dev)
    tab = torch.randn(n_rows, 64, device=dev)
    for _ in range(20):
        hip_ops.gather_rows(tab, idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        hip_ops.gather_rows(tab, idx)
    torch.cuda.synchronize()
    print("%d rows: %.1f us per call (wall, back to back)" % (n_rows, (time.perf_counter() - t0) / 300 * 1e6))
