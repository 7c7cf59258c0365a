"""duration of the single-workgroup plan kernel (one batch = one workgroup), HIP events around single-batch builds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import abi
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
from whisprrec_amd import hip_ops as ops
dev = torch.device("cuda:0")
nU, nI = 6040, 3706
g = torch.Generator(device=dev).manual_seed(1)
for B in (256, 1024, 2048, 4096):
    u = torch.randint(0, nU, (B,), device=dev, generator=g)
    p = torch.randint(0, nI, (B,), device=dev, generator=g)
    n = torch.randint(1, nI, (B,), device=dev, generator=g)
    for _ in range(3):
        ops.BatchPlan(u, p, n, B, nU, nI, builder="small", hot=False, validate=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        ops.BatchPlan(u, p, n, B, nU, nI, builder="small", hot=False, validate=False, defer=True)
    e1.record(); torch.cuda.synchronize()
    print("B=%d: %.1f us per single-batch plan (launch-to-launch, incl. host)" % (B, e0.elapsed_time(e1) / reps * 1e3))
