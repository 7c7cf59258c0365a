#!/usr/bin/env python3
"""C3 (BASELINE.json configs[2]): LightGCN gcn_layers=2, emb_size=64 on the ml-1m-shaped graph, B=2,048, Adam — wall time of
one eager zero_grad / predict / backward / step (reference loop, src/helpers/BaseRunner.py:196-199), host-bound or not."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from whisprrec_amd import host
from whisprrec_amd.lightgcn import LightGCN
from test_hip_config_shapes import ml1m_shaped_pairs

dev = torch.device("cuda:0")
nU, nI, D, B, L = 6040, 3706, 64, 2048, 2
uu, ii = ml1m_shaped_pairs()
ptr = np.zeros(nU + 1, np.int64); np.cumsum(np.bincount(uu, minlength=nU), out=ptr[1:])
tcs = {u: set(ii[ptr[u]:ptr[u + 1]].tolist()) for u in range(nU)}
corpus = host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []},
                              "test": {"user_id": [], "item_id": []}}, tcs, {})
for mfma in (0, 1):
    args = argparse.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=D, gcn_layers=L,
                              reg_weight=1e-5, optimizer="Adam", lr=1e-3, l2=0.0, spmm_mfma=mfma)
    m = LightGCN(args, corpus).to(dev)
    m.train()
    m._trusted_indices = True
    rng = np.random.RandomState(0)
    batches = []
    for _ in range(8):
        rows = rng.randint(0, uu.size, B)
        batches.append({"user_id": torch.from_numpy(uu[rows]).to(dev), "pos_item": torch.from_numpy(ii[rows]).to(dev),
                        "neg_items": torch.from_numpy(rng.randint(1, nI, B)).to(dev).unsqueeze(1), "batch_size": B, "phase": "train"})
    def step(k):
        m.optimizer.zero_grad()
        loss = m.predict(batches[k % 8])
        loss.backward()
        m.optimizer.step()
        return loss
    for k in range(10): step(k)
    torch.cuda.synchronize()
    n = int(os.environ.get("WR_STEPS", "100"))
    t0 = time.perf_counter()
    for k in range(n): last = step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"spmm_mfma": mfma, "eager_step_ms": dt * 1e3, "loss": float(last)}))
