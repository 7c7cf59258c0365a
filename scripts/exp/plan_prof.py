"""Plan build of the headline shape a few times, for rocprofv3 --kernel-trace --stats (per-kernel durations of the builder):
  rocprofv3 --kernel-trace --stats --output-format csv -d out -o run -- python3 scripts/exp/plan_prof.py [lib.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from whisprrec_amd import abi
if len(sys.argv) > 1:
    abi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from whisprrec_amd import hip_ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3407)
nU = nI = 1_000_000; B = int(os.environ.get("PLAN_B", "65536")); NB = max(1, (64 * 65536) // B)
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev, dtype=torch.int32)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev, dtype=torch.int32)
for rep in range(6):
    try:
        plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, validate=False, builder="fast")
    except Exception as e:      # timing-only variants of the library may produce an invalid plan
        print("plan:", str(e)[:80])
    torch.cuda.synchronize()
