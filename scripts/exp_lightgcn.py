import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops, host
from whisprrec_amd.lightgcn import LightGCN
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
nU, nI, B = 6040, 3706, 2048
sets = {}
for uu in range(nU):
    k = int(min(nI - 1, max(16, rng.pareto(1.2) * 40)))
    sets[uu] = set(np.unique(np.minimum((rng.pareto(0.8, k) * 30).astype(np.int64), nI - 1)).tolist())
corpus = host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []}, "test": {"user_id": [], "item_id": []}}, sets, {})
args = argparse.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, gcn_layers=2, reg_weight=1e-5, optimizer="Adam", lr=2e-3, l2=0.0)
m = LightGCN(args, corpus).to(dev); m.train()
g = torch.Generator(device=dev); g.manual_seed(1)
batch = {"user_id": torch.randint(0, nU, (B,), generator=g, device=dev), "pos_item": torch.randint(0, nI, (B,), generator=g, device=dev), "neg_items": torch.randint(1, nI, (B,), generator=g, device=dev)}
u, p, n = m._batch(batch)
def wall(fn, it=30):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
saved = {}
def fwd(): saved["x"] = m._forward_loss(u, p, n)
def bwd(): m._backward((u, p, n), saved["x"][1])
def step():
    m.optimizer.zero_grad(); loss = m.predict(batch); loss.backward(); m.optimizer.step()
E0 = torch.cat([m.user_embedding.weight.data, m.item_embedding.weight.data])
print("forward_loss %.0f us | backward %.0f us | propagate %.0f us | full step %.0f us" % (wall(fwd), wall(bwd), wall(lambda: m._propagate(E0)), wall(step)))
print("plan %.0f us | bpr_fwd %.0f us | embloss %.0f us" % (wall(lambda: hip_ops.BatchPlan(u, p, n, B, nU, nI)), wall(lambda: hip_ops.bpr_fwd(E0[:nU], E0[nU:], u, p, n, scores=False)), wall(lambda: hip_ops.embloss_sumsq(E0[:nU], E0[nU:], u, p, n))))
