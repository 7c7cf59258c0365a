"""host timeline of one timed piece of the rotate-mode bench with one rank (world = 1), --steps 20"""
import os, sys, time
sys.path.insert(0, ".")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import bench
from whisprrec_amd import rotating, hip_ops
args = bench.parse(["--steps", "20", "--warmup", "5", "--users", "125000", "--items", "125000", "--interactions", "1572864"])
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
log = []; T0 = [0.0]
def wrap(obj, name):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, (t - T0[0]) * 1e6, (time.perf_counter() - t) * 1e6)); return r
    setattr(obj, name, w)
B, D, K, W = args.batch, args.emb, args.steps, args.warmup
pipe = hip_ops.PipelinedSgd(20, min_triplets=1)
for nm in ("_prefetch", "_take_next"): wrap(pipe, nm)
model = rotating.RotatingBprmf(args.users, args.items, D, dev, parts=2, local=pipe)
model.init_xavier(1)
g = torch.Generator(device=dev); g.manual_seed(1)
def sched(count):
    u = torch.randint(0, 125000, (count * B,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, 62500, (count * B,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(0, 62500, (count * B,), generator=g, device=dev, dtype=torch.int32)
    return [(u, p, n, [count - count // 2, count // 2])]
for rep in range(3):
    model.run_strata(sched(W), B, 0.05, part_relative=True, defer_last=True)
    prepared = model.prepare(sched(K) + sched(20), B, part_relative=True)
    for sg in prepared["handle"]["segs"]:
        if sg["tabs"] is not None and not hasattr(sg["tabs"], "_wrapped"):
            wrap(sg["tabs"], "run_sgd"); sg["tabs"]._wrapped = True
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    log.clear(); T0[0] = time.perf_counter()
    model.run_prepared(prepared, 0.05, n_strata=1, defer_last=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d host done %.0f us, gpu done %.0f us (%.1f us/step)" % (rep, (t1 - T0[0]) * 1e6, (t2 - T0[0]) * 1e6, (t2 - T0[0]) * 1e6 / K))
    for nm, at, dur in log: print("   %-12s at %7.0f took %7.0f" % (nm, at, dur))
dist.destroy_process_group()
