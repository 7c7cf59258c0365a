"""where the host time of the first step launch after a device sync goes (PipelinedSgd._take_next + run_sgd_chain prelude)"""
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
for rep in range(4):
    pipe = hip_ops.PipelinedSgd(chunk=20, min_triplets=1)
    lw = torch.empty(W, device=dev); l = torch.empty(K, device=dev)
    h = pipe.plan(U, [(I, u, p, n)], B, first_chunk=[3, 2])
    pipe.run_steps(h, W, 0.05, lw)
    torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    if os.environ.get("WR_GC_COLLECT"):
        import gc; gc.collect()
    t0 = T()
    cur = h["next"]; plan = cur[1]
    t1 = T(); plan.ready.synchronize()
    t2 = T(); plan.finish()
    t3 = T(); plan.validate()
    t4 = T(); main.wait_event(plan.ready)
    t5 = T()
    pipe._take_next(h, h["pos"], main)          # now nearly free (finished already): bookkeeping only
    t6 = T()
    tabs = h["segs"][0]["tabs"]
    ws = tabs.overlap_workspace(plan.batch_size)
    t7 = T(); sync = tabs._chain_sync(20)
    t8 = T()
    tabs.run_sgd_chain(plan, 0, 20, 0.05, losses=l)
    t9 = T()
    torch.cuda.synchronize()
    t10 = T()
    print("rep %d us: ready.sync %.1f finish %.1f validate %.1f wait_event %.1f take_next-rest %.1f workspace %.1f sync-buf %.1f "
          "run_sgd_chain(20 launches) %.1f gpu-drain %.1f" % (rep, *[(b - a) * 1e6 for a, b in
          ((t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7), (t7, t8), (t8, t9), (t9, t10))]))
