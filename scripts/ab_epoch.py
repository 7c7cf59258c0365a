#!/usr/bin/env python3
"""Whole epoch on the device at BASELINE.json configs[1] (1M x 1M, D=64, B=65,536, 100 M interactions): negative sampling
against the per-user clicked lists + shuffle + plans + steps.  (a) preparation up front (wr_sample_negatives, wr_epoch_shuffle,
then the step stream), (b) preparation fused and pipelined (wr_epoch_prepare_range per plan chunk, beside the steps)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
n_inter = int(os.environ.get("WR_INTER", "100000000")); nU = nI = 1_000_000; D = 64; B = 65536
g = torch.Generator(device=dev); g.manual_seed(3407)
users = torch.randint(0, nU, (n_inter,), generator=g, device=dev, dtype=torch.int32)
items = torch.randint(0, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
ptr, idx = hip_ops.clicked_csr_from_pairs(users, items, nU, nI)
U = torch.randn(nU, D, generator=g, device=dev) * 0.001; I = torch.randn(nI, D, generator=g, device=dev) * 0.001
pipe = hip_ops.PipelinedSgd(64)
nb = (n_inter + B - 1) // B
t0 = time.perf_counter()
pairs = hip_ops.pair_set(ptr, idx, nU)
torch.cuda.synchronize()
print(json.dumps({"pair_set_build_ms": (time.perf_counter() - t0) * 1e3, "entries": pairs.numel(), "pairs": idx.numel()}))
for epoch in (1, 2):            # the preparation alone: lists against the hash set
    for name, pr in (("lists", None), ("hash set", pairs)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        neg, err = hip_ops.sample_negatives(users, nU, nI, ptr, idx, 3407, epoch, pairs=pr)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        prep = hip_ops.EpochPrep(users, items, nU, nI, ptr, idx, 3407, epoch, pairs=pr)
        prep.fill(0, n_inter)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        res = {"membership": name, "sample_negatives_ms": (t1 - t0) * 1e3, "fused_prepare_ms": (t2 - t1) * 1e3}
        if pr is not None:                                       # source rows as one word each
            packed = hip_ops.pack_rows(users, items)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            prep2 = hip_ops.EpochPrep(users, items, nU, nI, ptr, idx, 3407, epoch, pairs=pr, packed=packed)
            prep2.fill(0, n_inter)
            torch.cuda.synchronize(); t4 = time.perf_counter()
            res["fused_prepare_packed_rows_ms"] = (t4 - t3) * 1e3
            del prep2, packed
        print(json.dumps(res))
        del prep, neg
packed_rows = hip_ops.pack_rows(users, items)
for mode in ("upfront", "pipelined", "upfront+set", "pipelined+set", "pipelined+set+packed"):
    for epoch in (1, 2, 3):
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pr = pairs if "+set" in mode else None
        if mode.startswith("upfront"):
            neg, err = hip_ops.sample_negatives(users, nU, nI, ptr, idx, 3407, epoch, pairs=pr)
            u, p, n = hip_ops.epoch_shuffle([users, items, neg], 3407, epoch)
            pipe.run(pipe.plan(U, [(I, u, p, n)], B, lr=0.05), 0, 0.05, losses)
        else:
            prep = hip_ops.EpochPrep(users, items, nU, nI, ptr, idx, 3407, epoch, pairs=pr,
                                     packed=packed_rows if mode.endswith("+packed") else None)
            pipe.run(pipe.plan(U, [(I, prep.cols[0], prep.cols[1], prep.cols[2])], B, lr=0.05, prep=prep), 0, 0.05, losses)
            prep.check()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"mode": mode, "epoch_ms": dt * 1e3, "triplets_per_s_epoch": n_inter / dt, "steps": nb,
                      "loss_mean": float(losses.mean())}))
