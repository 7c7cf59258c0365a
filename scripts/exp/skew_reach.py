"""Which plan builder does "auto" end up with as the item ids get more skewed?  (headline shape, Zipf exponent sweep)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
nU = nI = 1_000_000; B = 65536; NB = 16
for z in (0.0, 0.3, 0.5, 0.6, 0.7, 0.8, 1.0):
    u, p, n = bench.synth_triplets(NB * B, nU, nI, dev, 3407, z)
    hip_ops._FAST_BACKOFF.clear()
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, validate=False)
    top = int(torch.bincount(p[:B].long(), minlength=1).max())
    hip_ops._FAST_BACKOFF.clear()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bmap = hip_ops.BucketMap(u, p, nU, nI, B)
    torch.cuda.synchronize(); t_map = time.perf_counter() - t0
    mapped = hip_ops.BatchPlan(u, p, n, B, nU, nI, validate=False, bucket_map=bmap)
    ts = {}
    for name, kw in (("generic", dict(builder="generic")), ("mapped", dict(bucket_map=bmap))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            hip_ops.BatchPlan(u, p, n, B, nU, nI, validate=False, **kw)
        torch.cuda.synchronize(); ts[name] = (time.perf_counter() - t0) / 5 / NB * 1e6
    print("zipf %.1f: plain auto -> %s; with a bucket map (%s + %s buckets, built in %.1f ms) -> %s; plan us/step generic %.1f, "
          "mapped %.1f; most frequent positive item of batch 0: %d occurrences" %
          (z, plan.builder, bmap.users and bmap.users["n_buckets"], bmap.items and bmap.items["n_buckets"], t_map * 1e3,
           mapped.builder, ts["generic"], ts["mapped"], top), flush=True)
