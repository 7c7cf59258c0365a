#!/usr/bin/env python3
"""A/B of two builds of libwhisprrec_hip.so on the headline step (run on the GPU box): per-kernel durations from HIP
events around the kernels, several repetitions interleaved.  usage: ab_step.py [--shape USERS,ITEMS[,D]]... libA.so libB.so"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
from whisprrec_amd import abi
abi.LIB_PATH = sys.argv[1]
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3407)
nU, nI = int(sys.argv[2]), int(sys.argv[3]); D = int(sys.argv[4]); B = 65536; NB = 64
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
tabs = hip_ops.BprmfTables(U, I)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
tabs.run_sgd(plan, 0, NB, 0.05); torch.cuda.synchronize()
out = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tabs.run_sgd(plan, 0, NB, 0.05); e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / NB * 1e3)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * NB)]
tabs.run_sgd(plan, 0, NB, 0.05, phase_events=ev); torch.cuda.synchronize()
ua = sum(ev[4 * k].elapsed_time(ev[4 * k + 1]) for k in range(NB)) / NB * 1e3
ia = sum(ev[4 * k + 2].elapsed_time(ev[4 * k + 3]) for k in range(NB)) / NB * 1e3
import time
pt = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl = hip_ops.BatchPlan(u, p, n, B, nU, nI); torch.cuda.synchronize()
    pt.append((time.perf_counter() - t0) / NB * 1e6)
print(json.dumps({"step_us": sorted(out)[len(out) // 2], "user_us": ua, "item_us": ia, "plan_us_per_step": sorted(pt)[2],
                  "builder": pl.builder}))
''' % ROOT
argv, shapes = sys.argv[1:], []
while argv and argv[0] == "--shape":
    shapes.append(tuple(argv[1].split(","))); argv = argv[2:]
shapes = shapes or [("1000000", "1000000")]
libs = argv
for shape in shapes:
    for rnd in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), shape[0], shape[1], shape[2] if len(shape) > 2 else "64"], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(shape, os.path.basename(lib), line[-1] if line else r.stderr[-400:], flush=True)
