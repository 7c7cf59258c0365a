#!/usr/bin/env python3
"""K9 at the C5 shape (SASRec item-embedding slice, B=2,048, T=20: 45 K gathered rows of 256 B, 3,706-row table): gather and
scatter-add times (HIP events over back-to-back calls)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
for n_rows in (3706, 16383, 100000):
    n, D = 2048 * 22, 64
    rng = np.random.RandomState(0)
    idx = torch.from_numpy(rng.randint(1, n_rows, n)).to(dev)
    src = torch.randn(n, D, device=dev)
    W = torch.randn(n_rows, D, device=dev)
    G = torch.zeros(n_rows, D, device=dev)
    def timeit(fn, reps=200):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tg = timeit(lambda: hip_ops.gather_rows(W, idx))
    ts = timeit(lambda: hip_ops.scatter_add_rows(G, idx, src, padding_idx=0))
    print(json.dumps({"table_rows": n_rows, "rows": n, "gather_us": tg, "scatter_add_us": ts,
                      "scatter_GBs": (2 * n * D * 4) / ts / 1e3}))
