#!/usr/bin/env python3
"""A/B of builds of libwhisprrec_hip.so on the plan build of the headline shape (run on the GPU box): GPU time from the
first plan kernel to the read-back of the plan's meta tensor (HIP events around BatchPlan), several repetitions interleaved.
usage: ab_plan.py [--rows N] libA.so libB.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from whisprrec_amd import abi
abi.LIB_PATH = sys.argv[1]
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3407)
nU = nI = int(sys.argv[2]); B = 65536; NB = 64
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
ts = []
for rep in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, validate=False); e1.record(); torch.cuda.synchronize()
    assert plan.builder == "fast"
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = sorted(ts[2:])
print(json.dumps({"plan_gpu_us_per_chunk_median": ts[len(ts) // 2], "min": ts[0], "per_step": ts[len(ts) // 2] / NB}))
''' % ROOT
argv, rows = sys.argv[1:], "1000000"
if argv and argv[0] == "--rows":
    rows, argv = argv[1], argv[2:]
for rep in range(2):
    for lib in argv:
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), rows], capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
