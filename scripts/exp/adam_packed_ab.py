"""(needs scripts/exp/adam_interleaved_records.patch applied: row_stride argument of the folded Adam entry points)"""
"""A/B on the GPU box: the folded Adam step on plain tables against interleaved records (row_stride 3), 1M x 1M x 64,
B = 65,536, plans prebuilt; and bitwise equality of the two."""
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from whisprrec_amd import abi, hip_ops
from whisprrec_amd.hip_ops import _p, _stream
dev = torch.device("cuda:0")
nU = nI = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, B, warm, K = 64, 65536, 96, 64
g = torch.Generator(device=dev); g.manual_seed(3407)
U0 = torch.randn(nU, D, generator=g, device=dev) * 0.01
I0 = torch.randn(nI, D, generator=g, device=dev) * 0.01
n = (warm + K) * B
u = torch.randint(0, nU, (n,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (n,), generator=g, device=dev, dtype=torch.int32)
ng = torch.randint(1, nI, (n,), generator=g, device=dev, dtype=torch.int32)
plan = hip_ops.BatchPlan(u, p, ng, B, nU, nI)
L = abi.lib()
res = {}
for mode in ("plain", "packed", "plain", "packed"):
    if mode == "plain":
        U, I = U0.clone(), I0.clone()
        mu, vu, mi, vi = (torch.zeros_like(t) for t in (U, U, I, I))
        ptrs = (_p(U), nU, _p(I), nI, D, _p(mu), _p(vu), _p(mi), _p(vi)); rs = 1
    else:
        PU = torch.zeros(nU, 3, D, device=dev); PU[:, 0] = U0
        PI = torch.zeros(nI, 3, D, device=dev); PI[:, 0] = I0
        ptrs = (PU.data_ptr(), nU, PI.data_ptr(), nI, D, PU.data_ptr() + 4 * D, PU.data_ptr() + 8 * D, PI.data_ptr() + 4 * D, PI.data_ptr() + 8 * D); rs = 3
    lu = torch.zeros(nU, dtype=torch.int32, device=dev); li = torch.zeros(nI, dtype=torch.int32, device=dev)
    host = torch.empty(2 * 4096, dtype=torch.float32)
    abi.check(L.wr_adam_consts(0, 4096, 1e-3, 0.9, 0.999, host.data_ptr()), "consts")
    consts = host.to(dev)
    losses = torch.empty(warm + K, dtype=torch.float32, device=dev)
    ws = torch.empty(int(L.wr_bprmf_step_workspace_bytes(B, D)), dtype=torch.uint8, device=dev)
    def run(first, count, t0):
        abi.check(L.wr_bprmf_run_adam_folded(*ptrs, _p(lu), _p(li), _p(plan.tu), _p(plan.tp), _p(plan.tn), _p(plan.oc_item), _p(plan.oc_src),
                                             plan.n_triplets, B, first, count, t0, 1e-3, _p(consts), 4096, 0.0, 0.9, 0.999, 1e-8,
                                             _p(losses), None, rs, _p(ws), ws.numel(), _stream()), "run")
    run(0, warm, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(warm, K, warm + 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.2f us/step" % (mode, dt / K * 1e6))
    res[mode] = (U.clone(), mu.clone()) if mode == "plain" else (PU[:, 0].contiguous(), PU[:, 1].contiguous())
print("bitwise equal:", torch.equal(res["plain"][0], res["packed"][0]), torch.equal(res["plain"][1], res["packed"][1]))
