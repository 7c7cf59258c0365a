"""whole-epoch time of BPRMF.train_epoch with Adam (exact lazy rows) at 1M x 1M x 64: steps + plans, us per step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import argparse
import torch
from whisprrec_amd import bprmf, host, hip_ops
if os.environ.get("WR_LAG_MIN_GAP"):
    hip_ops.LazyOptimizerState.LAG_MIN_GAP = int(os.environ["WR_LAG_MIN_GAP"])
if os.environ.get("WR_FOLD_MAX_GAP"):
    hip_ops.LazyOptimizerState.FOLD_MAX_GAP = int(os.environ["WR_FOLD_MAX_GAP"])
if os.environ.get("WR_MAX_LAG"):
    hip_ops.LazyOptimizerState.MAX_LAG = int(os.environ["WR_MAX_LAG"])
dev = torch.device("cuda:0")
nU = nI = 1_000_000
ap = argparse.ArgumentParser()
ap = bprmf.BPRMF.parse_model_args(ap)
args = ap.parse_args(["--emb_size", "64"] if False else [])
args.device = dev
args.model_path = "/tmp/m.pt"
corpus = host.Corpus.__new__(host.Corpus)
corpus.n_users, corpus.n_items = nU, nI
corpus.train_clicked_set = {}
corpus.residual_clicked_set = {}
model = bprmf.BPRMF(args, corpus).to(dev)
g = torch.Generator(device=dev).manual_seed(1)
for B, nb in [tuple(int(x) for x in c.split(":")) for c in os.environ.get("WR_CASES", "65536:192,2048:4096").split(",")]:
    N = nb * B
    u = torch.randint(0, nU, (N,), device=dev, generator=g)
    p = torch.randint(0, nI, (N,), device=dev, generator=g)
    n = torch.randint(1, nI, (N,), device=dev, generator=g)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        losses = model.train_epoch(u, p, n, B, 1e-3, float(os.environ.get("WR_L2", "0")), os.environ.get("WR_OPT", "Adam"))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("B=%d: %d steps in %.1f ms = %.1f us/step, loss %.4f" % (B, nb, dt * 1e3, dt / nb * 1e6, float(losses.mean())))
