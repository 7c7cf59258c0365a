"""one adjacency product on the ml-1m-shaped graph: fused combine (one launch) against chunk + combine launches; chunk sizes"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle
from whisprrec_amd import hip_ops
from test_hip_config_shapes import ml1m_shaped_pairs
dev = torch.device("cuda:0")
nU, nI, D = 6040, 3706, 64
uu, ii = ml1m_shaped_pairs()
ptr = np.zeros(nU + 1, np.int64); np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
rp, col, val = oracle.lightgcn_build_adj(nU, nI, ptr, ii.astype(np.int32))
N = nU + nI
X = torch.from_numpy((np.random.RandomState(0).standard_normal((N, D)) * 0.1).astype(np.float32)).to(dev)
cold, vald = torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev)
Y = torch.empty_like(X); acc = torch.zeros_like(X)
def timeit(fn, reps=300):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mx in (64, 96, 128, 192, 256):
    cp, cr = hip_ops.spmm_chunks(rp, max_nnz=mx)
    cp, cr = cp.to(dev), cr.to(dev)
    pt = torch.empty((cr.numel(), D), device=dev)
    f = lambda: hip_ops.spmm_csr_chunked(cp, cr, cold, vald, X, Y=Y, acc=acc, partials=pt, levels=1)
    res = {}
    for fused in (True, False):
        hip_ops.SPMM_FUSED_COMBINE = fused
        res[fused] = timeit(f)
    print("chunk %3d nnz (%5d chunks): fused %.1f us, two launches %.1f us" % (mx, cr.numel(), res[True], res[False]))
