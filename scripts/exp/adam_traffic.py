"""Practical ceiling of the folded Adam step's access pattern (scripts/exp/adam_traffic.hip): nine random 256-B rows in and
out per triplet over six 1M-row tables, with and without the three 4-byte step stamps; fresh rows every launch."""
import ctypes, json, os, subprocess
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "adam_traffic.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "adam_traffic.hip")])
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0")
nU = nI = 1_000_000; B = 65536; NL = 48
tabs = [torch.randn(nU, 64, device=dev) * 0.01 for _ in range(6)]
lastU = torch.zeros(nU, dtype=torch.int32, device=dev); lastI = torch.zeros(nI, dtype=torch.int32, device=dev)
rng = np.random.RandomState(0)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
batches = []
for k in range(NL):
    u = np.sort(rng.permutation(nU)[:B]).astype(np.int32)
    items = rng.permutation(nI)[:2 * B].astype(np.int32)
    batches.append(tuple(torch.from_numpy(x).to(dev) for x in (u, items[:B], items[B:])))
for stamps, lds in ((0, 0), (1, 0), (1, 20 * 1024), (1, 32 * 1024), (1, 40 * 1024)):   # LDS per workgroup limits the waves per SIMD: 8 / 8 / 8 / 5 / 4
    def launch(b, t):
        lib.run(*[P(x) for x in tabs], P(lastU), P(lastI), P(b[0]), P(b[1]), P(b[2]), B, stamps, t, lds, st)
    for b in batches[:4]: launch(b, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(3):
        for k, b in enumerate(batches): launch(b, k)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (3 * NL) * 1e-3
    byts = B * (18 * 256 + (24 if stamps else 0) + 12)
    print(json.dumps({"case": "traffic-only: 9 random 256-B rows in + 9 out per triplet over six 1M-row tables%s" % (", + 3 stamps read and written" if stamps else ""),
                      "lds_per_workgroup": lds, "us_per_launch": round(t * 1e6, 2), "GBs": round(byts / t / 1e9, 1), "frac_of_8TBs": round(byts / t / 8e12, 3)}))
