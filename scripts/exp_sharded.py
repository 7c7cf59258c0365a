import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
from whisprrec_amd.sharded import ShardedBprmf
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nU = nI = 1_000_000; D = 64; B = 65536; NB = 32
m = ShardedBprmf(nU, nI, D, dev); m.init_xavier(1)
g = torch.Generator(device=dev); g.manual_seed(1)
u = torch.randint(0, nU, (NB * B,), generator=g, device=dev)
p = torch.randint(0, nI, (NB * B,), generator=g, device=dev)
n = torch.randint(1, nI, (NB * B,), generator=g, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cp = m.plan_chunk(u, p, n, B)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    m.run_chunk(cp, 0.05)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("plan_chunk %.1f us/step   run_chunk %.1f us/step" % ((t1 - t0) / NB * 1e6, (t2 - t1) / NB * 1e6))
dist.destroy_process_group()
