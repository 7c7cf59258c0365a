"""durations of the plan-build kernels of ONE 64-batch plan built alone on the GPU (rocprofv3 --kernel-trace this script)"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from whisprrec_amd import hip_ops as ops  # noqa: E402
dev = torch.device("cuda:0")
nU = nI = 1_000_000
B, nb = 65536, 64
g = torch.Generator(device=dev).manual_seed(1)
u = torch.randint(0, nU, (nb * B,), device=dev, generator=g, dtype=torch.int32)
p = torch.randint(0, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
n = torch.randint(1, nI, (nb * B,), device=dev, generator=g, dtype=torch.int32)
arena = ops.PlanArena(dev, nb * B, B, overlap_items=nI)
for rep in range(4):
    plan = ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
    torch.cuda.synchronize()
