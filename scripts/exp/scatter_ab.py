"""K9 timing: scatter_add_rows per call (plan + apply) and the apply alone with a plan built ahead (ScatterPlan)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from whisprrec_amd import abi, hip_ops
if os.environ.get("WR_LIB"):
    abi.LIB_PATH = os.path.abspath(os.environ["WR_LIB"])      # timing-only variant builds (scripts/exp/build_variant.sh)

dev = torch.device("cuda:0")


def timeit(f, reps=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n_rows, n, D in ((3706, 45056, 64), (100_000, 45056, 64), (1_000_000, 45056, 64), (125_000, 114_688, 64),
                     (1_250_000, 114_688, 128)):
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    idx = torch.randint(0, n_rows, (n,), generator=g, device=dev)
    src = torch.randn(n, D, device=dev)
    tab = torch.zeros(n_rows, D, device=dev)
    t_call = timeit(lambda: hip_ops.scatter_add_rows(tab, idx, src, padding_idx=0))
    line = "%9d rows %7d positions D=%3d: scatter_add_rows %6.1f us" % (n_rows, n, D, t_call)
    if n_rows > 16383:
        plan = hip_ops.ScatterPlan(idx.view(1, -1), n_rows, padding_idx=0)
        t_apply = timeit(lambda: plan.apply(tab, 0, n, src))
        t_plan = timeit(lambda: hip_ops.ScatterPlan(idx.view(1, -1), n_rows, padding_idx=0), reps=50)
        line += "; apply with a plan built ahead %6.1f us, plan build (incl. allocation) %6.1f us, slow=%s" % (t_apply, t_plan, plan.slow)
    print(line, flush=True)
