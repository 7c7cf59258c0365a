#!/usr/bin/env python3
"""Lazy Adam at BASELINE.json configs[1] shapes (1M x 1M, D=64): catch-up in a pass of its own (wr_adam_rows_lazy +
wr_bprmf_step_adam: 12 row transfers per touched row) against the catch-up folded into the step kernels' row loads
(wr_bprmf_step_adam_folded: 6).  us per optimizer step, native multi-batch loops, plans prebuilt."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisprrec_amd import abi
if os.environ.get("WR_LIB"):            # A/B builds (build_ab/*.so)
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
from whisprrec_amd import hip_ops

dev = torch.device("cuda:0")
nU = nI = 1_000_000; D = int(os.environ.get("WR_D", "64"))
CASES = [tuple(int(x) for x in c.split(":")) for c in os.environ["WR_BATCHES"].split(",")] if os.environ.get("WR_BATCHES") \
    else [(65536, 48), (2048, 512)]
for B, NB in CASES[:int(os.environ.get("WR_CASES", "99"))]:
    g = torch.Generator(device=dev); g.manual_seed(1)
    u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
    plan = hip_ops.BatchPlan(u, p, n, B, nU, nI)
    for l2 in ((0.0, 1e-6) if not os.environ.get("WR_BATCHES") else (0.0,)):
        res = {}
        for fold in (False, True):
            U = torch.randn(nU, D, generator=g, device=dev) * 0.01
            I = torch.randn(nI, D, generator=g, device=dev) * 0.01
            st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, l2, fold=fold)
            st.run(plan, 0, NB // 2)                       # warm-up: rows get realistic step gaps
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.run(plan, NB // 2, NB - NB // 2); e1.record(); torch.cuda.synchronize()
            res["folded" if fold else "separate"] = e0.elapsed_time(e1) / (NB - NB // 2) * 1e3
        print(json.dumps({"B": B, "D": D, "l2": l2, "us_per_step": res, "G_triplets_per_s": {k: B / v / 1e3 for k, v in res.items()}}))
