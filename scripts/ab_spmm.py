#!/usr/bin/env python3
"""A/B of the LightGCN adjacency product on the ml-1m-shaped graph (BASELINE.json configs[2]): chunked CSR gather kernels
(VALU) against the hybrid — dense head of the item popularity on the matrix cores (v_mfma_f32_32x32x2_f32), rest on CSR.
Prints per variant the time of one product (mean of many back-to-back launches, HIP events) and the error against the
oracle's CSR product.  Run under rocprofv3 --kernel-trace --stats / --pmc for the per-kernel split and the MFMA / VALU
counters (profiles/r02_spmm_*)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle
from whisprrec_amd import abi
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])
from whisprrec_amd import hip_ops
from test_hip_config_shapes import ml1m_shaped_pairs

dev = torch.device("cuda:0")
nU, nI, D = 6040, 3706, int(os.environ.get("WR_D", "64"))
uu, ii = ml1m_shaped_pairs()
ptr = np.zeros(nU + 1, np.int64); np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
rp, col, val = oracle.lightgcn_build_adj(nU, nI, ptr, ii.astype(np.int32))
N = nU + nI
rng = np.random.RandomState(0)
X = (rng.standard_normal((N, D)) * 0.1).astype(np.float32)
ref = oracle.spmm_csr(rp, col, val, X)
Xd = torch.from_numpy(X).to(dev)
cptr, crow = hip_ops.spmm_chunks(rp)
cptr, crow = cptr.to(dev), crow.to(dev)
cold, vald = torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev)
partials = torch.empty((crow.numel(), D), device=dev)
Y = torch.empty_like(Xd); acc = torch.zeros_like(Xd)
reps = int(os.environ.get("WR_REPS", "200"))

def timeit(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

def err(y):
    return float(np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max())

out = {"graph": {"users": nU, "items": nI, "nnz": int(rp[-1]), "D": D}}
f_csr = lambda: hip_ops.spmm_csr_chunked(cptr, crow, cold, vald, Xd, Y=Y, acc=acc, partials=partials)
out["csr_chunked"] = {"us": timeit(f_csr), "err": err(Y), "flops": 2.0 * rp[-1] * D}
for mx in [int(x) for x in os.environ.get("WR_CHUNKS", "64,128").split(",")]:
    cp2, cr2 = hip_ops.spmm_chunks(rp, max_nnz=mx)
    cp2, cr2 = cp2.to(dev), cr2.to(dev)
    pt2 = torch.empty((cr2.numel(), D), device=dev)
    f2 = lambda: hip_ops.spmm_csr_chunked(cp2, cr2, cold, vald, Xd, Y=Y, acc=acc, partials=pt2)
    out["csr_chunked_%d" % mx] = {"us": timeit(f2), "err": err(Y), "chunks": int(cr2.numel())}
for md in [float(x) for x in os.environ.get("WR_DENS", "0.25,0.18,0.12,0.08,0.05").split(",")]:
    hy = hip_ops.HybridSpmm(rp, col, val, nU, nI, dev, min_density=md, max_head=1024)
    if not hy.enabled:
        continue
    f_h = lambda: hy.apply(Xd, Y=Y, acc=acc)
    t = timeit(f_h)
    dense_flops = 2.0 * 2 * (hy.u_tiles * 32) * hy.H * D
    out["hybrid_%.2f" % md] = {"us": t, "err": err(Y), "head_items": hy.n_head, "H": hy.H, "block_density": hy.density,
                               "head_share_of_nnz": 2.0 * hy.head_nnz / rp[-1], "dense_flops": dense_flops,
                               "csr_nnz_left": int(hy.col.numel())}
for k, v in out.items():
    print(k, json.dumps(v))
