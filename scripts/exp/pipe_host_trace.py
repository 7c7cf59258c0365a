"""Where does the host spend its time in PipelinedSgd at the driver's short bench shape (--steps 20 --warmup 5)?
Wraps the pipeline's three host phases with perf_counter and prints a timeline relative to the start of the timed region."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from whisprrec_amd import hip_ops
import bench

args = bench.parse(["--steps", sys.argv[1] if len(sys.argv) > 1 else "20", "--warmup", "5"])
dev = torch.device("cuda:0")
B, D, K, W = args.batch, args.emb, args.steps, args.warmup
C = bench.plan_chunk(args)
g = torch.Generator(device=dev); g.manual_seed(3407)
U = torch.randn(args.users, D, generator=g, device=dev) * 0.001
I = torch.randn(args.items, D, generator=g, device=dev) * 0.001
u, p, n = bench.synth_triplets((K + W + C) * B, args.users, args.items, dev, 3407)
log = []
T0 = [0.0]
def wrap(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, (t - T0[0]) * 1e6, (time.perf_counter() - t) * 1e6)); return r
    setattr(obj, name, w)
for rep in range(6):
    pipe = hip_ops.PipelinedSgd(chunk=C, min_triplets=1, chain=(rep % 2 == 0))
    for nm in ("_prefetch", "_take_next"):
        wrap(pipe, nm)
    lw = torch.empty(W, device=dev); l = torch.empty(K, device=dev)
    torch.cuda.synchronize()
    h = pipe.plan(U, [(I, u, p, n)], B, first_chunk=[W - W // 2, W // 2])
    wrap(h["segs"][0]["tabs"], "run_sgd")
    wrap(h["segs"][0]["tabs"], "run_sgd_chain")
    pipe.run_steps(h, W, args.lr, lw)
    torch.cuda.synchronize()
    log.clear()
    T0[0] = time.perf_counter()
    pipe.run_steps(h, K, args.lr, l)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("rep %d: host done %.0f us, gpu done %.0f us (%.1f us/step)" % (rep, (t1 - T0[0]) * 1e6, (t2 - T0[0]) * 1e6, (t2 - T0[0]) * 1e6 / K))
    for nm, at, dur in log:
        print("   %-12s at %7.0f us  took %7.0f us" % (nm, at, dur))
