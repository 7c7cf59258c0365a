"""Why is the whole-epoch pace (plans + steps through PipelinedSgd) slower per step than bench.py's loop?  Same tables and
batch size; (a) fresh random int32 indices, (b) indices produced by index_select of a permutation as in the epoch case."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
nU = nI = 1_000_000; D = 64; B = 65536
U = torch.randn(nU, D, generator=g, device=dev) * 0.01; I = torch.randn(nI, D, generator=g, device=dev) * 0.01
for n_inter in (50_000_000, 100_000_000):
    nb = (n_inter + B - 1) // B
    users = torch.randint(0, nU, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    neg = torch.randint(1, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    pipe = hip_ops.PipelinedSgd(64)
    for mode in ("direct", "direct", "permuted", "permuted"):
        if mode == "permuted":
            order = torch.randperm(n_inter, device=dev, generator=g)
            u, p, n = users[order], items[order], neg[order]
            del order
        else:
            u, p, n = users, items, neg
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h = pipe.plan(U, [(I, u, p, n)], B)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        pipe.run(h, 0, 0.05, losses)
        t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print("%d interactions, %s: first plan %.2f ms, host loop %.2f ms, total %.2f ms = %.2f us/step" %
              (n_inter, mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t0) * 1e3, (t3 - t0) / nb * 1e6), flush=True)
    del users, items, neg
