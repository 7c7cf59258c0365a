"""host timeline of the driver's 20-step region on the group path: when the native call is entered / returns, when run_steps
returns, when the device is done (us after t0)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
import bench
args = bench.parse(["--steps", "20", "--warmup", "5"])
dev = torch.device("cuda:0")
B, D, K, W = args.batch, args.emb, args.steps, args.warmup
g = torch.Generator(device=dev); g.manual_seed(3407)
U = torch.randn(args.users, D, generator=g, device=dev) * 0.001
I = torch.randn(args.items, D, generator=g, device=dev) * 0.001
u, p, n = bench.synth_triplets((K + W + 20) * B, args.users, args.items, dev, 3407)
T = time.perf_counter
marks = {}
orig = hip_ops.BprmfTables.run_sgd_group
def wrapped(self, *a, **k):
    marks["enter"] = T()
    r = orig(self, *a, **k)
    marks["leave"] = T()
    return r
hip_ops.BprmfTables.run_sgd_group = wrapped
import gc
for rep in range(8):
    spin = rep % 2 == 1
    pipe = hip_ops.PipelinedSgd(chunk=20, min_triplets=1)
    lw = torch.empty(W, device=dev); l = torch.empty(K, device=dev)
    gc.disable()
    h = pipe.plan(U, [(I, u, p, n)], B, first_chunk=[3, 2])
    pipe.run_steps(h, W, 0.05, lw)
    torch.cuda.synchronize()
    t0 = T()
    pipe.run_steps(h, K, 0.05, l)
    t1 = T()
    if spin:                                   # poll an event instead of sleeping in hipDeviceSynchronize
        ev = torch.cuda.Event(); ev.record()
        while not ev.query():
            pass
    torch.cuda.synchronize()
    t2 = T()
    gc.enable()
    print("spin" if spin else "sync", end=": ")
    print("native call entered at %.0f us, returned at %.0f; run_steps returned at %.0f; device done at %.0f (%.1f us/step)" %
          ((marks["enter"] - t0) * 1e6, (marks["leave"] - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t2 - t0) / K * 1e6), flush=True)
