import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
for n in (1, 1):
    hip_ops.axpy(y, x, 1.0)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
N = 2000
t0 = time.perf_counter()
e0.record()
for _ in range(N):
    hip_ops.axpy(y, x, 1.0)
e1.record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("trivial kernel chain: GPU %.2f us/kernel, host enqueue %.2f us/kernel, total wall %.2f us/kernel" % (e0.elapsed_time(e1) / N * 1e3, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# big streaming kernel chain: axpy over 64 MB
X = torch.zeros(16 * 1024 * 1024, device=dev); Y = torch.zeros(16 * 1024 * 1024, device=dev)
hip_ops.axpy(Y, X, 1.0); torch.cuda.synchronize()
e0.record()
for _ in range(200):
    hip_ops.axpy(Y, X, 1.0)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 200 * 1e3
print("axpy 64MB x3 traffic: %.2f us -> %.2f TB/s" % (t, 3 * 64e6 * 1.048576 / t / 1e6))
