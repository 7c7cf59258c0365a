"""Practical ceiling of the user phase's access pattern on this memory system: a kernel that does NOTHING but the traffic
(per 16-lane team: read three random 256-B rows of 1M-row tables, write three; rows change every launch so nothing stays in
the 256 MB infinity cache) — scripts/exp/fence_exp.hip, mode 0.  The product kernel moves the same bytes plus the arithmetic."""
import ctypes, json, os, subprocess, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "fence_exp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "fence_exp.hip")])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
nU = nI = 1_000_000; B = 65536; NL = 64
g = torch.Generator(device=dev); g.manual_seed(1)
U = torch.randn(nU, 64, generator=g, device=dev) * 0.01; I = torch.randn(nI, 64, generator=g, device=dev) * 0.01
Z = torch.empty(B, 64, device=dev); cnt = torch.zeros(B, dtype=torch.int32, device=dev)
rng = np.random.RandomState(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
batches = []
for k in range(NL):   # distinct rows inside a launch (no write races), fresh rows from launch to launch
    u = np.sort(rng.permutation(nU)[:B]).astype(np.int32)
    items = rng.permutation(nI)[:2 * B].astype(np.int32)
    batches.append(tuple(torch.from_numpy(x).to(dev) for x in (u, items[:B], items[B:], np.full(B, -1, np.int32))))
dummy = torch.zeros(2, dtype=torch.int32, device=dev)
def launch(b):
    u, p, n, partner = b
    lib.run(0, P(U), P(I), P(Z), P(cnt), P(u), P(p), P(n), P(partner), B, P(dummy), 0, st)
for b in batches[:4]: launch(b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for rep in range(4):
    for b in batches: launch(b)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / (4 * NL) * 1e-3
byts = B * (6 * 256 + 12)
print(json.dumps({"case": "ceiling: traffic-only kernel, 3 random 256-B rows in + 3 out per team, 1M-row tables, fresh rows every launch",
                  "us_per_launch": t * 1e6, "GBs": byts / t / 1e9, "frac_of_8TBs": byts / t / 8e12}))
