"""Where does a C1 (ml-100k, B=2048) HipRunner SGD epoch spend its host time?  cProfile of fit()."""
import sys, os, time, cProfile, pstats, argparse as ap
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from whisprrec_amd import host, runner
from whisprrec_amd.bprmf import BPRMF
dev = torch.device("cuda:0")
g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_ml100k_curve.npz"))
ptr, idx = g2["clicked_ptr"], g2["clicked_idx"]
tcs = {u: set(idx[ptr[u]:ptr[u + 1]].tolist()) for u in range(943)}
data = {"train": {"user_id": g2["train_user"].astype(np.int64), "item_id": g2["train_item"].astype(np.int64)},
        "dev": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)},
        "test": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)}}
corpus = host.Corpus(943, 1574, data, tcs, {u: set() for u in range(943)})
args = ap.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, fused=1,
                    epoch=1, check_epoch=1, test_epoch=-1, early_stop=10, lr=1e-3, l2=0.0, batch_size=2048,
                    eval_batch_size=2048, optimizer=sys.argv[1] if len(sys.argv) > 1 else "SGD", num_workers=0, pin_memory=0,
                    topk="10,20", metric="NDCG, HR", device_epoch_prep=0, random_seed=3407)
model = BPRMF(args, corpus).to(dev)
ds = BPRMF.Dataset(model, corpus, "train")
r = runner.HipRunner(args)
r.fit(ds, 1)
for e in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r.fit(ds, e + 2); torch.cuda.synchronize()
    print("epoch %d: %.1f ms" % (e + 2, (time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); r.fit(ds, 5); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
