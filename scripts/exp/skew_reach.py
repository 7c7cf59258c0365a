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
    print("zipf %.1f: builder %s, hot runs %s, most frequent positive item of batch 0: %d occurrences" %
          (z, plan.builder, plan.hot is not None, top), flush=True)
