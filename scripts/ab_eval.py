#!/usr/bin/env python3
"""A/B of builds on the full-ranking evaluation kernel. usage: ab_eval.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from whisprrec_amd import abi
abi.LIB_PATH = sys.argv[1]
import torch, numpy as np
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
out = {}
for n, nI in ((100000, 100000), (20000, 1000000)):
    D = 64; nU = 100000
    U = torch.randn(nU, D, generator=g, device=dev); I = torch.randn(nI, D, generator=g, device=dev)
    eu = torch.randint(0, nU, (n,), generator=g, device=dev); et = torch.randint(0, nI, (n,), generator=g, device=dev)
    per = 50
    idx = torch.sort(torch.randint(0, nI, (nU, per), generator=g, device=dev, dtype=torch.int32), dim=1)[0].reshape(-1).contiguous()
    ptr = torch.arange(0, (nU + 1) * per, per, device=dev, dtype=torch.int64)
    hip_ops.rank_eval(U, I, eu, et, ptr, idx); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): hip_ops.rank_eval(U, I, eu, et, ptr, idx)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    out["%%dx%%d" %% (n, nI)] = round(2.0 * n * nI * D / t / 1e12, 1)
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib)], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(os.path.basename(lib), line[-1] if line else r.stderr[-600:], flush=True)
