#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, CPU torch):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's Python never travels to the GPU box; only the .npz vectors written here do.
Every fixture is pure data: seeded inputs and the outputs the reference computed from them.

Reference entry points exercised (paths relative to /root/reference):
  G1  src/models/general/BPRMF.py:69-80 (predict) + loss.backward() + torch.optim.{SGD,Adam}.step()
      exactly as src/helpers/BaseRunner.py:196-199 issues them
  G2  src/helpers/BaseReader.py + src/models/BaseModel.py:167-177 (sampler) + BaseRunner.fit loop
      (BaseRunner.py:180-201) on ml-100k, seed 3407
  G3  src/models/BaseModel.py:167-177 (negative sampler, NumPy MT19937 stream)
  G4  src/models/general/LightGCN.py:54-175 (adjacency, forward, predict, grads)
  G5  src/models/sequential/SASRec.py:84,105-106 (item-embedding gather / scatter with padding_idx=0)
  G6  src/helpers/BaseRunner.py:50-92 (evaluate_method)
  G8  src/helpers/BaseReader.py + src/utils/sample.py on data/ml-100k/ml-100k.inter: filtered ids, both split rules
  G9  the whole src/main.py flow for BPRMF on the G8 subset: reader -> model -> BaseRunner.train (fit + evaluate every epoch)
      -> test metrics, with the README's optimizer settings (Adam, --lr 1e-3 --l2 1e-6) and with SGD
  G7  src/models/general/SGL.py:67-79 + src/utils/augmentor.py:33-111 (graph views from Python's `random` stream),
      SGL.py:148-246 (three propagations, sum-BPR + EmbLoss + InfoNCE, grads), per view type ED / ND / RW
"""
import argparse
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
import numpy as np

np.float_ = np.float64  # reference utils.format_metric uses np.float_ (removed in NumPy 2)
import torch

REF = os.environ.get("WR_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "src"))
HERE = os.path.dirname(os.path.abspath(__file__))

from models.general.BPRMF import BPRMF  # noqa: E402
from models.general.LightGCN import LightGCN  # noqa: E402
from models.sequential.SASRec import SASRec  # noqa: E402
from models.BaseModel import GeneralModel  # noqa: E402
from helpers.BaseRunner import BaseRunner  # noqa: E402
from helpers.BaseReader import BaseReader  # noqa: E402
from utils import utils as ref_utils  # noqa: E402
from utils.loss import BPRLoss, EmbLoss  # noqa: E402


class _Corpus:
    def __init__(self, n_users, n_items, train_clicked_set=None):
        self.n_users = n_users
        self.n_items = n_items
        self.train_clicked_set = train_clicked_set or {}
        self.residual_clicked_set = {}


def _args(**kw):
    base = dict(device=torch.device("cpu"), model_path="/tmp/wr_golden_model.pt", buffer=1,
                num_neg=1, test_all=1, embedding_size=64)
    base.update(kw)
    return argparse.Namespace(**base)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in arrays.items() if hasattr(v, "shape")})


# --------------------------------------------------------------------------------------------- G1
def g1_bprmf_step():
    torch.manual_seed(3407)
    rng = np.random.RandomState(3407)
    nU, nI, D, B = 97, 131, 64, 512
    corpus = _Corpus(nU, nI)
    model = BPRMF(_args(embedding_size=D), corpus)
    U0 = model.user_embeddings.weight.detach().clone()
    I0 = model.item_embeddings.weight.detach().clone()
    # scale up so scores are O(1) and the sigmoid is exercised away from 0.5
    U0 = U0 * 6.0
    I0 = I0 * 6.0

    def batch(seed_off):
        r = np.random.RandomState(3407 + seed_off)
        u = r.randint(0, nU, size=B).astype(np.int64)
        p = r.randint(0, nI, size=B).astype(np.int64)
        n = r.randint(1, nI, size=B).astype(np.int64)
        # deliberate structure: heavy duplicates, pos/neg collisions on the same item row,
        # same item as pos of one triplet and neg of another, p == n inside one triplet
        u[:40] = 5
        p[:24] = 7
        n[24:48] = 7
        p[60:64] = n[60:64]
        u[-1] = nU - 1
        p[-1] = nI - 1
        n[-2] = nI - 1
        u[100] = 0
        p[101] = 0
        return u, p, n

    batches = [batch(k) for k in range(5)]
    out = dict(U0=U0.numpy(), I0=I0.numpy())
    for k, (u, p, n) in enumerate(batches):
        out[f"u{k}"], out[f"p{k}"], out[f"n{k}"] = u, p, n

    def fresh():
        m = BPRMF(_args(embedding_size=D), corpus)
        with torch.no_grad():
            m.user_embeddings.weight.copy_(U0)
            m.item_embeddings.weight.copy_(I0)
        return m

    def feed(k):
        u, p, n = batches[k]
        return {"user_id": torch.from_numpy(u), "pos_item": torch.from_numpy(p),
                "neg_items": torch.from_numpy(n), "batch_size": B, "phase": "train"}

    # forward pieces + dense grads on batch 0
    m = fresh()
    fd = feed(0)
    ue, pe = m.forward(fd["user_id"], fd["pos_item"])
    ne = m.get_item_embedding(fd["neg_items"])
    out["pos_score"] = torch.mul(ue, pe).sum(dim=1).detach().numpy()
    out["neg_score"] = torch.mul(ue, ne).sum(dim=1).detach().numpy()
    loss = m.predict(fd)
    loss.backward()
    out["loss0"] = loss.detach().numpy().reshape(1)
    out["gU"] = m.user_embeddings.weight.grad.numpy().copy()
    out["gI"] = m.item_embeddings.weight.grad.numpy().copy()

    # optimizer trajectories: same call order as BaseRunner.fit (:196-199)
    def run(opt_name, lr, l2, nsteps, tag):
        m = fresh()
        opt = eval("torch.optim." + opt_name)(m.parameters(), lr=lr, weight_decay=l2)
        losses = []
        for k in range(nsteps):
            opt.zero_grad()
            loss = m.predict(feed(k))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            out[f"{tag}_U{k + 1}"] = m.user_embeddings.weight.detach().numpy().copy()
            out[f"{tag}_I{k + 1}"] = m.item_embeddings.weight.detach().numpy().copy()
        out[f"{tag}_loss"] = np.asarray(losses, dtype=np.float32)
        out[f"{tag}_hp"] = np.asarray([lr, l2], dtype=np.float64)

    run("SGD", 0.05, 0.0, 5, "sgd")
    run("SGD", 0.05, 1e-3, 3, "sgdl2")
    run("Adam", 1e-2, 0.0, 3, "adam")
    run("Adam", 1e-2, 1e-3, 3, "adaml2")
    save("g1_bprmf_step", **out)
    # G10: the other two optimizers of the reference's flag surface (BaseRunner.py:34-37: --optimizer SGD / Adam / Adagrad /
    # Adadelta, eval'ed into torch.optim.<name>(params, lr=lr, weight_decay=l2) at :120-124), same tables and batches; kept
    # in a file of its own so that g1 stays byte-identical
    out10 = {}
    keep, out = out, out10
    run("Adagrad", 5e-2, 0.0, 5, "adagrad")
    run("Adadelta", 2.0, 0.0, 5, "adadelta")
    out = keep
    save("g10_optimizers", **out10)


# --------------------------------------------------------------------------------------------- G2
def g2_ml100k_curve():
    tmp = tempfile.mkdtemp(prefix="wr_golden_")
    try:
        shutil.copytree(os.path.join(REF, "data", "ml-100k"), os.path.join(tmp, "ml-100k"))
        rargs = argparse.Namespace(path=tmp, dataset="ml-100k", sep="\t", sample="random")
        corpus = BaseReader(rargs)
    finally:
        pass
    train = corpus.data_df["train"]
    out = dict(train_user=train["user_id"].to_numpy().astype(np.int32),
               train_item=train["item_id"].to_numpy().astype(np.int32),
               n_users=np.asarray([corpus.n_users], dtype=np.int64),
               n_items=np.asarray([corpus.n_items], dtype=np.int64))

    def run(opt_name, lr, l2, epochs, tag):
        ref_utils.init_seed(3407)
        args = _args(embedding_size=64)
        model = BPRMF(args, corpus)
        out[f"{tag}_U0"] = model.user_embeddings.weight.detach().numpy().copy()
        out[f"{tag}_I0"] = model.item_embeddings.weight.detach().numpy().copy()
        dataset = BPRMF.Dataset(model, corpus, "train")
        runner_args = argparse.Namespace(epoch=epochs, check_epoch=1, test_epoch=-1, early_stop=10, lr=lr, l2=l2,
                                         batch_size=2048, eval_batch_size=2048, optimizer=opt_name, num_workers=0,
                                         pin_memory=0, topk="10,20", metric="NDCG,HR")
        runner = BaseRunner(runner_args)
        # same body as BaseRunner.fit (:180-201), instrumented to record batches and per-batch loss
        from torch.utils.data import DataLoader
        losses, bu, bp, bn, bsz, epoch_mean = [], [], [], [], [], []
        for ep in range(epochs):
            if model.optimizer is None:
                model.optimizer = runner._build_optimizer(model)
            dataset.actions_before_epoch()
            model.train()
            dl = DataLoader(dataset, batch_size=runner.batch_size, shuffle=True, num_workers=0,
                            collate_fn=dataset.collate_batch, pin_memory=False)
            lst = []
            for batch in dl:
                batch = ref_utils.batch_to_gpu(batch, model.device)
                bu.append(batch["user_id"].numpy().astype(np.int32))
                bp.append(batch["pos_item"].numpy().astype(np.int32))
                bn.append(batch["neg_items"].numpy().astype(np.int32))
                bsz.append(len(batch["user_id"]))
                model.optimizer.zero_grad()
                loss = model.predict(batch)
                loss.backward()
                model.optimizer.step()
                lst.append(loss.detach().cpu().data.numpy())
            losses += [float(x) for x in lst]
            epoch_mean.append(np.mean(lst).item())
        out[f"{tag}_loss"] = np.asarray(losses, dtype=np.float32)
        out[f"{tag}_epoch_mean"] = np.asarray(epoch_mean, dtype=np.float64)
        out[f"{tag}_bu"] = np.concatenate(bu)
        out[f"{tag}_bp"] = np.concatenate(bp)
        out[f"{tag}_bn"] = np.concatenate(bn)
        out[f"{tag}_bsz"] = np.asarray(bsz, dtype=np.int32)
        out[f"{tag}_Uend"] = model.user_embeddings.weight.detach().numpy().copy()
        out[f"{tag}_Iend"] = model.item_embeddings.weight.detach().numpy().copy()
        out[f"{tag}_hp"] = np.asarray([lr, l2], dtype=np.float64)

    run("SGD", 0.5, 0.0, 2, "sgd")      # large lr so 66 SGD steps move the loss visibly
    run("Adam", 1e-3, 0.0, 1, "adam")   # README.md:40 learning rate, reference default optimizer
    # clicked sets for the sampler parity test (CSR of train_clicked_set)
    users = sorted(corpus.train_clicked_set.keys())
    ptr, idx = [0], []
    for uid in range(int(corpus.n_users)):
        s = sorted(corpus.train_clicked_set.get(uid, ()))
        idx += s
        ptr.append(len(idx))
    out["clicked_ptr"] = np.asarray(ptr, dtype=np.int32)
    out["clicked_idx"] = np.asarray(idx, dtype=np.int32)
    shutil.rmtree(tmp, ignore_errors=True)
    save("g2_ml100k_curve", **out)


# --------------------------------------------------------------------------------------------- G3
def g3_sampler():
    nU, nI, nrow = 23, 57, 400
    rng = np.random.RandomState(11)
    users = rng.randint(0, nU, size=nrow)
    items = rng.randint(0, nI, size=nrow)
    clicked = {}
    for u, i in zip(users, items):
        clicked.setdefault(int(u), set()).add(int(i))
    # user 3 clicks almost everything -> long redraw chains
    clicked[3] = set(range(0, nI - 2))
    corpus = _Corpus(nU, nI, clicked)

    class _DF:  # minimal stand-in for corpus.data_df[phase] consumed by utils.df_to_dict
        def __init__(self, d):
            self.d = d

        def to_dict(self, kind):
            return {k: list(v) for k, v in self.d.items()}

    corpus.data_df = {"train": _DF({"user_id": users, "item_id": items})}
    model = BPRMF(_args(), corpus)
    ds = BPRMF.Dataset(model, corpus, "train")
    np.random.seed(3407)
    ds.actions_before_epoch()
    neg1 = np.asarray(ds.data["neg_items"]).astype(np.int64).copy()
    ds.actions_before_epoch()  # second epoch continues the same MT19937 stream
    neg2 = np.asarray(ds.data["neg_items"]).astype(np.int64).copy()
    ptr, idx = [0], []
    for uid in range(nU):
        idx += sorted(clicked.get(uid, ()))
        ptr.append(len(idx))
    save("g3_sampler", users=users.astype(np.int64), items=items.astype(np.int64), n_items=np.asarray([nI]),
         clicked_ptr=np.asarray(ptr, dtype=np.int32), clicked_idx=np.asarray(idx, dtype=np.int32),
         neg_epoch1=neg1, neg_epoch2=neg2, seed=np.asarray([3407]))


# --------------------------------------------------------------------------------------------- G4
def g4_lightgcn():
    torch.manual_seed(3407)
    rng = np.random.RandomState(5)
    nU, nI, D, B, L = 37, 53, 64, 256, 2
    clicked = {}
    for u in range(nU):
        if u == 4:
            continue  # isolated user: degree 0 row (rowsum + 1e-10 path, LightGCN.py:89)
        k = rng.randint(1, 12)
        clicked[u] = set(int(x) for x in rng.choice(np.arange(1, nI), size=k, replace=False))
    clicked[0] = set(range(1, nI))  # dense row
    corpus = _Corpus(nU, nI, clicked)
    model = LightGCN(_args(embedding_size=D, gcn_layers=L, reg_weight=1e-5), corpus)
    with torch.no_grad():
        model.user_embedding.weight.mul_(8.0)
        model.item_embedding.weight.mul_(8.0)
    adj = model.norm_adj  # dense at runtime (LightGCN.py:114-121)
    nz = adj.nonzero(as_tuple=False)
    out = dict(adj_row=nz[:, 0].numpy().astype(np.int32), adj_col=nz[:, 1].numpy().astype(np.int32),
               adj_val=adj[nz[:, 0], nz[:, 1]].numpy().astype(np.float32),
               U0=model.user_embedding.weight.detach().numpy().copy(),
               I0=model.item_embedding.weight.detach().numpy().copy(),
               hp=np.asarray([L, 1e-5], dtype=np.float64))
    ptr, idx = [0], []
    for uid in range(nU):
        idx += sorted(clicked.get(uid, ()))
        ptr.append(len(idx))
    out["clicked_ptr"] = np.asarray(ptr, dtype=np.int32)
    out["clicked_idx"] = np.asarray(idx, dtype=np.int32)
    ua, ia = model.forward()
    out["user_all"] = ua.detach().numpy().copy()
    out["item_all"] = ia.detach().numpy().copy()
    u = rng.randint(0, nU, size=B).astype(np.int64)
    p = rng.randint(0, nI, size=B).astype(np.int64)
    n = rng.randint(1, nI, size=B).astype(np.int64)
    u[:30] = 0
    p[:10] = 3
    n[10:20] = 3
    out["u"], out["p"], out["n"] = u, p, n
    fd = {"user_id": torch.from_numpy(u), "pos_item": torch.from_numpy(p), "neg_items": torch.from_numpy(n)}
    loss = model.predict(fd)
    loss.backward()
    out["loss"] = loss.detach().numpy().reshape(1).astype(np.float32)
    out["gU"] = model.user_embedding.weight.grad.numpy().copy()
    out["gI"] = model.item_embedding.weight.grad.numpy().copy()
    # 3 steps of Adam / SGD as BaseRunner.fit would do
    for opt_name, lr, tag in (("SGD", 0.05, "sgd"), ("Adam", 2e-3, "adam")):
        torch.manual_seed(3407)
        m = LightGCN(_args(embedding_size=D, gcn_layers=L, reg_weight=1e-5), corpus)
        with torch.no_grad():
            m.user_embedding.weight.copy_(torch.from_numpy(out["U0"]))
            m.item_embedding.weight.copy_(torch.from_numpy(out["I0"]))
        opt = eval("torch.optim." + opt_name)(m.parameters(), lr=lr, weight_decay=0)
        ls = []
        for k in range(3):
            opt.zero_grad()
            loss = m.predict(fd)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        out[f"{tag}_loss"] = np.asarray(ls, dtype=np.float32)
        out[f"{tag}_U3"] = m.user_embedding.weight.detach().numpy().copy()
        out[f"{tag}_I3"] = m.item_embedding.weight.detach().numpy().copy()
        out[f"{tag}_lr"] = np.asarray([lr])
    save("g4_lightgcn", **out)


# --------------------------------------------------------------------------------------------- G5
def g5_sasrec_emb():
    torch.manual_seed(3407)
    rng = np.random.RandomState(9)
    nI, D, B, T = 71, 64, 96, 20
    corpus = _Corpus(13, nI)
    args = _args(emb_size=D, num_layers=1, num_heads=4, dropout=0.0, history_max=T)
    model = SASRec(args, corpus)
    model.train()
    lengths = rng.randint(1, T + 1, size=B).astype(np.int64)
    lengths[0] = T
    hist = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        hist[b, :lengths[b]] = rng.randint(1, nI, size=lengths[b])
    pos = rng.randint(0, nI, size=B).astype(np.int64)   # includes id 0 -> padding row, grad dropped
    neg = rng.randint(1, nI, size=B).astype(np.int64)
    pos[:4] = 0
    fd = {"history_items": torch.from_numpy(hist), "lengths": torch.from_numpy(lengths),
          "pos_item": torch.from_numpy(pos), "neg_items": torch.from_numpy(neg)}
    W0 = model.item_embedding.weight.detach().numpy().copy()
    # the embedding-layer slice in isolation: gather -> arbitrary upstream grads -> scatter
    his_vec = model.item_embedding(fd["history_items"])
    pos_e = model.item_embedding(fd["pos_item"])
    g_his = torch.from_numpy(rng.standard_normal(his_vec.shape).astype(np.float32))
    g_pos = torch.from_numpy(rng.standard_normal(pos_e.shape).astype(np.float32))
    (his_vec * g_his).sum().add((pos_e * g_pos).sum()).backward()
    out = dict(W0=W0, hist=hist, lengths=lengths, pos=pos, neg=neg,
               gather_hist=his_vec.detach().numpy().copy(), gather_pos=pos_e.detach().numpy().copy(),
               g_his=g_his.numpy(), g_pos=g_pos.numpy(),
               gW_slice=model.item_embedding.weight.grad.numpy().copy())
    model.zero_grad()
    loss = model.predict(fd)
    loss.backward()
    out["loss"] = loss.detach().numpy().reshape(1).astype(np.float32)
    out["gW_full"] = model.item_embedding.weight.grad.numpy().copy()
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        out["sd__" + k] = v
    save("g5_sasrec_emb", **out)


# --------------------------------------------------------------------------------------------- G6
def g6_eval():
    rng = np.random.RandomState(21)
    pred = rng.standard_normal((64, 40)).astype(np.float32)
    pred[5, 7] = -np.inf
    pred[6, 1:] = -np.inf          # ground truth ranks first
    pred[7, 0] = pred[7].min() - 1  # ground truth ranks last
    res = BaseRunner.evaluate_method(pred, [5, 10, 20], ["NDCG", "HR", "RECALL", "PRECISION"])
    keys = sorted(res.keys())
    save("g6_eval", predictions=pred, keys=np.asarray(keys), values=np.asarray([res[k] for k in keys], dtype=np.float64))


def g7_sgl():
    import random
    from models.general.SGL import SGL
    rng = np.random.RandomState(9)
    nU, nI, D, B, L = 41, 59, 32, 192, 2
    clicked = {}
    for u in range(nU):
        if u == 6:
            continue  # isolated user
        k = rng.randint(2, 14)
        clicked[u] = set(int(x) for x in rng.choice(np.arange(1, nI), size=k, replace=False))
    clicked[1] = set(range(1, nI, 2))  # a heavy row
    corpus = _Corpus(nU, nI, clicked)
    ptr, idx = [0], []
    for uid in range(nU):
        idx += sorted(clicked.get(uid, ()))
        ptr.append(len(idx))
    u = rng.randint(0, nU, size=B).astype(np.int64)
    p = rng.randint(0, nI, size=B).astype(np.int64)
    n = rng.randint(1, nI, size=B).astype(np.int64)
    u[:20] = 1
    p[:8] = 3
    n[8:16] = 3
    out = dict(clicked_ptr=np.asarray(ptr, dtype=np.int32), clicked_idx=np.asarray(idx, dtype=np.int32), u=u, p=p, n=n,
               shape=np.asarray([nU, nI, D, L], dtype=np.int64))
    fd = {"user_id": torch.from_numpy(u), "pos_item": torch.from_numpy(p), "neg_items": torch.from_numpy(n)}
    for vtype, ratio in (("ED", 0.25), ("ND", 0.2), ("RW", 0.1)):
        torch.manual_seed(3407)
        hp = dict(embedding_size=D, gcn_layers=L, type=vtype, reg_weight=1e-4, ssl_tau=0.2, ssl_weight=0.05, drop_ratio=ratio)
        model = SGL(_args(**hp), corpus)
        with torch.no_grad():
            model.user_embedding.weight.mul_(6.0)
            model.item_embedding.weight.mul_(6.0)
        random.seed(2024)
        model.graph_construction()          # SGL.Dataset.actions_before_epoch (SGL.py:262)
        t = vtype.lower()
        out[t + "_hp"] = np.asarray([1e-4, 0.2, 0.05, ratio], dtype=np.float64)
        out[t + "_U0"] = model.user_embedding.weight.detach().numpy().copy()
        out[t + "_I0"] = model.item_embedding.weight.detach().numpy().copy()
        for k, g in ((0, model.train_graph), (1, model.sub_graph1), (2, model.sub_graph2)):
            if isinstance(g, list):
                continue                    # RW: graph_construction leaves [] (no propagation at all)
            nz = g.nonzero(as_tuple=False)
            out[f"{t}_g{k}_row"] = nz[:, 0].numpy().astype(np.int32)
            out[f"{t}_g{k}_col"] = nz[:, 1].numpy().astype(np.int32)
            out[f"{t}_g{k}_val"] = g[nz[:, 0], nz[:, 1]].numpy().astype(np.float32)
        loss = model.predict(fd)
        loss.backward()
        out[t + "_loss"] = loss.detach().numpy().reshape(1).astype(np.float32)
        out[t + "_gU"] = model.user_embedding.weight.grad.numpy().copy()
        out[t + "_gI"] = model.item_embedding.weight.grad.numpy().copy()
        with torch.no_grad():
            out[t + "_full"] = model.full_predict({"user_id": torch.from_numpy(u[:16])}).numpy().copy()
    save("g7_sgl", **out)


def g8_reader():
    """first 25,000 rows of ml-100k.inter (kept under the dataset name ml-100k so the rating rule applies) through
    BaseReader under both --sample rules: the input columns and the three splits, in the reference's row order"""
    src = os.path.join(REF, "data", "ml-100k", "ml-100k.inter")
    with open(src) as f:
        lines = f.readlines()[:25001]
    raw = np.asarray([[int(float(x)) for x in ln.rstrip("\n").split("\t")] for ln in lines[1:]], dtype=np.int64)
    out = {"in_user": raw[:, 0].astype(np.int32), "in_item": raw[:, 1].astype(np.int32), "in_rating": raw[:, 2].astype(np.int8),
           "in_time": raw[:, 3]}
    for rule in ("random", "loo"):
        work = tempfile.mkdtemp(prefix="wr_golden_")
        os.makedirs(os.path.join(work, "ml-100k"))
        with open(os.path.join(work, "ml-100k", "ml-100k.inter"), "w") as f:
            f.writelines(lines)
        r = BaseReader(argparse.Namespace(sep="\t", path=work + "/", dataset="ml-100k", sample=rule))
        out[rule + "_shape"] = np.asarray([r.n_users, r.n_items, len(r.all_df)], dtype=np.int64)
        for ph in ("train", "dev", "test"):
            df = r.data_df[ph]
            out[f"{rule}_{ph}_user"] = df["user_id"].to_numpy().astype(np.int32)
            out[f"{rule}_{ph}_item"] = df["item_id"].to_numpy().astype(np.int32)
            out[f"{rule}_{ph}_time"] = df["timestamp"].to_numpy().astype(np.int64)
        if rule == "random":   # SeqReader on the same file: the position column of every split and two users' histories
            from helpers.SeqReader import SeqReader
            sr = SeqReader(argparse.Namespace(sep="\t", path=work + "/", dataset="ml-100k", sample=rule))
            for ph in ("train", "dev", "test"):
                out[f"seq_{ph}_position"] = sr.data_df[ph]["position"].to_numpy().astype(np.int32)
                out[f"seq_{ph}_user"] = sr.data_df[ph]["user_id"].to_numpy().astype(np.int32)
            for uid in (0, 5):
                out[f"seq_his_{uid}"] = np.asarray(sr.user_his[uid], dtype=np.int64)
        shutil.rmtree(work, ignore_errors=True)
    save("g8_reader", **out)


def g9_end_to_end():
    """per-epoch training loss, dev metrics and the final test metrics of the reference's own train loop (seed 3407)"""
    src = os.path.join(REF, "data", "ml-100k", "ml-100k.inter")
    with open(src) as f:
        lines = f.readlines()[:25001]
    out = {}
    from models.general.SGL import SGL
    cases = (("adam", BPRMF, "Adam", 1e-2, 1e-6, 6, {}), ("sgd", BPRMF, "SGD", 8.0, 0.0, 6, {}),
             ("lightgcn", LightGCN, "Adam", 5e-3, 0.0, 3, dict(gcn_layers=2, reg_weight=1e-5)),
             ("sgl", SGL, "Adam", 5e-3, 0.0, 3, dict(gcn_layers=2, reg_weight=1e-4, type="ED", ssl_tau=0.2, ssl_weight=0.05,
                                                  drop_ratio=0.1)),
             ("sasrec", SASRec, "Adam", 2e-3, 0.0, 2, dict(emb_size=32, num_layers=1, num_heads=2, dropout=0.0, history_max=20)))
    for tag, cls, opt, lr, l2, epochs, extra in cases:
        work = tempfile.mkdtemp(prefix="wr_golden_")
        os.makedirs(os.path.join(work, "ml-100k"))
        with open(os.path.join(work, "ml-100k", "ml-100k.inter"), "w") as f:
            f.writelines(lines)
        ref_utils.init_seed(3407)
        rargs = argparse.Namespace(sep="\t", path=work + "/", dataset="ml-100k", sample="random")
        if cls is SASRec:
            from helpers.SeqReader import SeqReader
            corpus = SeqReader(rargs)
        else:
            corpus = BaseReader(rargs)
        args = _args(embedding_size=64, model_path=os.path.join(work, "m.pt"), epoch=epochs, check_epoch=1, test_epoch=-1,
                     early_stop=10, lr=lr, l2=l2, batch_size=1024, eval_batch_size=2048, optimizer=opt, num_workers=0,
                     pin_memory=0, topk="10,20", metric="NDCG, HR", **extra)
        model = cls(args, corpus).to(args.device)
        data = {ph: cls.Dataset(model, corpus, ph) for ph in ("train", "dev", "test")}
        runner = BaseRunner(args)
        losses, devs = [], []
        for epoch in range(args.epoch):           # BaseRunner.train's loop body (BaseRunner.py:131-146), results captured
            losses.append(runner.fit(data["train"], epoch=epoch + 1))
            devs.append(runner.evaluate(data["dev"], runner.topk[:1], runner.metrics))
        test = runner.evaluate(data["test"], runner.topk, runner.metrics)
        out[tag + "_hp"] = np.asarray([lr, l2, epochs], dtype=np.float64)
        out[tag + "_loss"] = np.asarray(losses, dtype=np.float64)
        keys = sorted(devs[0])
        out[tag + "_dev_keys"] = np.asarray(keys)
        out[tag + "_dev"] = np.asarray([[d[k] for k in keys] for d in devs], dtype=np.float64)
        tkeys = sorted(test)
        out[tag + "_test_keys"] = np.asarray(tkeys)
        out[tag + "_test"] = np.asarray([test[k] for k in tkeys], dtype=np.float64)
        shutil.rmtree(work, ignore_errors=True)
    save("g9_end_to_end", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9"]
    torch.set_num_threads(4)
    for w in which:
        {"g1": g1_bprmf_step, "g2": g2_ml100k_curve, "g3": g3_sampler, "g4": g4_lightgcn,
         "g5": g5_sasrec_emb, "g6": g6_eval, "g7": g7_sgl, "g8": g8_reader, "g9": g9_end_to_end}[w]()
