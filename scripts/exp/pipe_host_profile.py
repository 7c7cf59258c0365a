"""cProfile of the host side of PipelinedSgd.run_steps for the driver's 20-step shape (which Python calls cost what)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import hip_ops
import bench
args = bench.parse(["--steps", "20", "--warmup", "5"])
dev = torch.device("cuda:0")
B, D, K, W = args.batch, args.emb, args.steps, args.warmup
C = bench.plan_chunk(args)
g = torch.Generator(device=dev); g.manual_seed(3407)
U = torch.randn(args.users, D, generator=g, device=dev) * 0.001
I = torch.randn(args.items, D, generator=g, device=dev) * 0.001
u, p, n = bench.synth_triplets((K + W + C) * B, args.users, args.items, dev, 3407)
pr = cProfile.Profile()
for rep in range(4):
    pipe = hip_ops.PipelinedSgd(chunk=C, min_triplets=1)
    lw = torch.empty(W, device=dev); l = torch.empty(K, device=dev)
    h = pipe.plan(U, [(I, u, p, n)], B, first_chunk=[W - W // 2, W // 2])
    pipe.run_steps(h, W, args.lr, lw)
    torch.cuda.synchronize()
    if rep >= 1:
        pr.enable()
    pipe.run_steps(h, K, args.lr, l)
    if rep >= 1:
        pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue())
