"""PipelinedSgd at the reference's default batch (2,048): how many batches per plan?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
nU = nI = 1_000_000; D = 64
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
for B, n_inter in ((2048, 8_000_000), (16384, 32_000_000)):
    nb = (n_inter + B - 1) // B
    u = torch.randint(0, nU, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    for chunk in (64, 256, 1024, 4096):
        pipe = hip_ops.PipelinedSgd(chunk)
        for rep in range(2):
            losses = torch.empty(nb, dtype=torch.float32, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.run(pipe.plan(U, [(I, u, p, n)], B), 0, 0.05, losses)
            torch.cuda.synchronize(); t = time.perf_counter() - t0
        print("B=%d chunk=%d: %.2f us/step, %.3f G triplets/s" % (B, chunk, t / nb * 1e6, n_inter / t / 1e9), flush=True)
