"""Time of hip_ops.BucketMap on a 40 M-row power-law epoch (row shares from a strided sample of 2 M rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
u, p, n = bench.synth_triplets(40_000_000, 1_000_000, 1_000_000, dev, 3407, 1.0)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = hip_ops.BucketMap(u, p, 1_000_000, 1_000_000, 65536)
    torch.cuda.synchronize()
    print("BucketMap of 40 M rows: %.2f ms (%d + %d buckets)" % ((time.perf_counter() - t0) * 1e3, m.users["n_buckets"], m.items["n_buckets"]), flush=True)
plan = hip_ops.BatchPlan(u[:64 * 65536], p[:64 * 65536], n[:64 * 65536], 65536, 1_000_000, 1_000_000, bucket_map=m)
print("plan of the first 64 batches:", plan.builder)
