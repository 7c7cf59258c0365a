"""Folded Adam step, two launches per step against one (chained): us/step at 1M x 1M x 64, B = 65,536, plans prebuilt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import abi, hip_ops
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
dev = torch.device("cuda:0")
nU = nI = 1_000_000
D, B, NB = 64, 65536, 64
g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
arena = hip_ops.PlanArena(dev, NB * B, B, overlap_items=nI)
plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, arena=arena, overlap=True)
assert plan.overlap is not None
for rep in range(2):
    for chain in (False, True):
        U = torch.randn(nU, D, generator=g, device=dev) * 0.01
        I = torch.randn(nI, D, generator=g, device=dev) * 0.01
        st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, 0.0, fold=True)
        st.chain = chain
        st.run(plan, 0, NB // 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); st.run(plan, NB // 2, NB - NB // 2); e1.record(); torch.cuda.synchronize()
        st.tabs.check_chain()
        print("chain=%s: %.1f us/step (%d chained calls)" % (chain, e0.elapsed_time(e1) / (NB - NB // 2) * 1e3, st.chain_calls), flush=True)
