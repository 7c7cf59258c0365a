#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs other than the bench.py headline (run on the GPU box).
Prints one JSON object per line; copy the output to profiles/."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from whisprrec_amd import hip_ops, host

dev = torch.device("cuda:0")


def ev_time(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(iters):
        fn(k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def emit(**kw):
    print(json.dumps(kw), flush=True)


def synth(nU, nI, n, zipf=0.0, seed=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    u = torch.randint(0, nU, (n,), generator=g, device=dev, dtype=torch.int32)
    if zipf > 0:
        r = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
        p = (torch.exp(r * np.log(nI)).to(torch.int64) - 1).clamp_(0, nI - 1).to(torch.int32)
    else:
        p = torch.randint(0, nI, (n,), generator=g, device=dev, dtype=torch.int32)
    ng = torch.randint(1, nI, (n,), generator=g, device=dev, dtype=torch.int32)
    return u, p, ng


def tables(nU, nI, D, seed=1):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    return (torch.randn(nU, D, generator=g, device=dev) * 0.01, torch.randn(nI, D, generator=g, device=dev) * 0.01)


def bprmf_case(name, nU, nI, D, B, NB, opt="SGD", l2=0.0, zipf=0.0, lazy=False):
    U, I = tables(nU, nI, D)
    u, p, n = synth(nU, nI, NB * B, zipf)
    tabs = hip_ops.BprmfTables(U, I)
    t0 = time.perf_counter(); plan = hip_ops.BatchPlan(u, p, n, B, nU, nI); torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
    bmap = hip_ops.BucketMap(u, p, nU, nI, B) if plan.fast_overflowed else None    # skewed ids: load-balanced buckets
    hip_ops._FAST_BACKOFF.clear()
    t0 = time.perf_counter(); plan = hip_ops.BatchPlan(u, p, n, B, nU, nI, bucket_map=bmap); torch.cuda.synchronize(); t_plan = time.perf_counter() - t0
    if lazy:
        st = hip_ops.LazyOptimizerState(tabs, opt, 1e-3 if opt == "Adam" else 0.05, l2)
        for k in range(NB):                                   # first pass: rows reach their steady-state replay lengths
            st.step(plan, k)
        t = ev_time(lambda k: st.run(plan, 0, NB), 1) / NB        # native multi-batch loop
        t_flush = ev_time(lambda k: st.flush(), 1)
        name += " [exact lazy rows; flush of all rows after %d steps: %.0f us]" % (2 * NB, t_flush * 1e6)
    elif opt == "SGD" and l2 == 0.0:
        tabs.run_sgd(plan, 0, min(NB, 4), 0.05)
        t = ev_time(lambda k: tabs.run_sgd(plan, 0, NB, 0.05), 1) / NB
        # round 3: the step stream without a per-batch sort (group plans), where it applies
        if zipf == 0.0 and 4096 <= B <= 81920 and min(nU, nI) >= 12 * B and tabs.group_supported():
            t0 = time.perf_counter(); gplan = hip_ops.GroupPlan(u, p, n, B, nU, nI); torch.cuda.synchronize()
            tg_plan = time.perf_counter() - t0
            t0 = time.perf_counter(); gplan = hip_ops.GroupPlan(u, p, n, B, nU, nI); torch.cuda.synchronize()
            tg_plan = time.perf_counter() - t0
            if not gplan.overflow:
                tabs.run_sgd_group(gplan, 0, min(NB, 4), 0.05)
                tg = ev_time(lambda k: tabs.run_sgd_group(gplan, 0, NB, 0.05), 1) / NB
                emit(case=name + " [group plan: no per-batch sort, one launch per step]", users=nU, items=nI, D=D, batch=B,
                     optimizer=opt, step_us=tg * 1e6, plan_us_per_step=tg_plan / NB * 1e6, triplets_per_s_steps_only=B / tg,
                     triplets_per_s_with_plan=B / (tg + tg_plan / NB), algorithmic_GBs=(6 * D * 4 + 12) * B / tg / 1e9)
            del gplan
    elif opt == "SGD":
        tabs.step_sgd(plan, 0, 0.05, l2)
        t = ev_time(lambda k: tabs.step_sgd(plan, k, 0.05, l2), NB)
    else:
        gU, gI = torch.zeros_like(U), torch.zeros_like(I)
        st = [torch.zeros_like(U), torch.zeros_like(U), torch.zeros_like(I), torch.zeros_like(I)]
        def step(k):
            _, sid = tabs.grads(plan, k, gU, gI)
            hip_ops.adam_dense(tabs.U, st[0], st[1], gU, k + 1, 1e-3, l2, stamp=tabs.stamp_u, step_id=sid)
            hip_ops.adam_dense(tabs.I, st[2], st[3], gI, k + 1, 1e-3, l2, stamp=tabs.stamp_i, step_id=sid)
        step(0)
        t = ev_time(step, NB)
    emit(case=name, users=nU, items=nI, D=D, batch=B, optimizer=opt, l2=l2, zipf=zipf, plan_builder=plan.builder,
         step_us=t * 1e6, plan_us_per_step=t_plan / NB * 1e6, triplets_per_s_steps_only=B / t,
         triplets_per_s_with_plan=B / (t + t_plan / NB), algorithmic_GBs=(6 * D * 4 + 12) * B / t / 1e9)
    del U, I, tabs, plan
    torch.cuda.empty_cache()


def c2_epoch_case(n_inter=100_000_000, nU=1_000_000, nI=1_000_000, D=64, B=65536):
    """whole epoch at C2 scale, everything on the device: negatives (wr_sample_negatives against the per-user clicked
    lists), shuffle, batch plans, steps ('epoch-level throughput incl. shuffle + negative sampling')"""
    g = torch.Generator(device=dev); g.manual_seed(3407)
    users = torch.randint(0, nU, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, nI, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    t0 = time.perf_counter(); ptr, idx = hip_ops.clicked_csr_from_pairs(users, items, nU, nI); torch.cuda.synchronize()
    t_csr = time.perf_counter() - t0
    t0 = time.perf_counter(); pairs = hip_ops.pair_set(ptr, idx, nU); torch.cuda.synchronize()
    t_set = time.perf_counter() - t0
    U, I = tables(nU, nI, D)
    tabs = hip_ops.BprmfTables(U, I)
    pipe = hip_ops.PipelinedSgd(64)
    nb = (n_inter + B - 1) // B
    res = {}
    for epoch in (1, 2, 3):   # the last one is reported (allocator and plan workspaces warm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        neg, err = hip_ops.sample_negatives(users, nU, nI, ptr, idx, 3407, epoch, pairs=pairs)   # membership: hash set of the pairs
        torch.cuda.synchronize(); t1 = time.perf_counter()
        u, p, n = hip_ops.epoch_shuffle([users, items, neg], 3407, epoch)     # keyed bijection per row (wr_epoch_shuffle)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(pipe.plan(U, [(I, u, p, n)], B), 0, 0.05, losses)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        res = dict(sample_ms=(t1 - t0) * 1e3, shuffle_ms=(t2 - t1) * 1e3, plan_and_steps_ms=(t3 - t2) * 1e3,
                   epoch_ms=(t3 - t0) * 1e3, loss_mean=float(losses.mean()))
        del u, p, n, neg
    # the product's form (HipRunner --device_epoch_prep 1): sampler + shuffle fused, produced chunk by chunk beside the steps,
    # membership through the pair set, source rows packed into one word each
    packed = hip_ops.pack_rows(users, items)
    for epoch in (4, 5, 6):
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prep = hip_ops.EpochPrep(users, items, nU, nI, ptr, idx, 3407, epoch, pairs=pairs, packed=packed)
        pipe.run(pipe.plan(U, [(I, prep.cols[0], prep.cols[1], prep.cols[2])], B, prep=prep), 0, 0.05, losses)
        prep.check()
        torch.cuda.synchronize()
        res["epoch_ms_fused_pipelined"] = (time.perf_counter() - t0) * 1e3
        del prep
    res["triplets_per_s_epoch_fused_pipelined"] = n_inter / (res["epoch_ms_fused_pipelined"] * 1e-3)
    emit(case="C2 whole epoch on the device: sampler + shuffle + plans + %d steps (1Mx1M, D=64, B=65536, %d interactions)" % (nb, n_inter),
         clicked_csr_build_once_ms=t_csr * 1e3, pair_set_build_once_ms=t_set * 1e3, triplets_per_s_epoch=n_inter / (res["epoch_ms"] * 1e-3), **res)


def c2_pcie_case(n_inter=50_000_000, nU=1_000_000, nI=1_000_000, D=64, B=65536):
    """the boundary as the reference hands it over: the epoch's (u, p, n) as int64 HOST arrays (BaseModel.py:96-127 builds
    int64 batches on the host); PCIe copy + plans + steps.  Reported beside the HBM-resident headline, never as it."""
    U, I = tables(nU, nI, D)
    pipe = hip_ops.PipelinedSgd(64)
    nb = (n_inter + B - 1) // B
    host = [torch.randint(0, hi, (n_inter,), dtype=torch.int64) for hi in (nU, nI, nI)]
    for kind in ("pageable", "pinned"):
        hs = [h.pin_memory() for h in host] if kind == "pinned" else host
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            d = [h.to(dev, non_blocking=True) for h in hs]
            torch.cuda.synchronize(); t1 = time.perf_counter()
            losses = torch.empty(nb, dtype=torch.float32, device=dev)
            pipe.run(pipe.plan(U, [(I, d[0], d[1], d[2])], B), 0, 0.05, losses)
            torch.cuda.synchronize(); t2 = time.perf_counter()
        emit(case="C2 epoch from HOST int64 index arrays (%s): PCIe copy, then plans + %d steps" % (kind, nb), interactions=n_inter,
             h2d_ms=(t1 - t0) * 1e3, h2d_GBs=3 * 8 * n_inter / (t1 - t0) / 1e9, plan_and_steps_ms=(t2 - t1) * 1e3,
             triplets_per_s_pcie_inclusive=n_inter / (t2 - t0), triplets_per_s_hbm_resident=n_inter / (t2 - t1))
        del d


def lightgcn_case():
    from whisprrec_amd.lightgcn import LightGCN
    rng = np.random.RandomState(0)
    nU, nI, B = 6040, 3706, 2048
    sets = {}
    n_pairs = 0
    for uu in range(nU):   # ml-1m-shaped: >= 16 train items per user, power-law item popularity
        k = int(min(nI - 1, max(16, rng.pareto(1.2) * 40)))
        items = np.unique(np.minimum((rng.pareto(0.8, k) * 30).astype(np.int64), nI - 1))
        sets[uu] = set(items.tolist()); n_pairs += len(items)
    corpus = host.Corpus(nU, nI, {"train": {"user_id": [], "item_id": []}, "dev": {"user_id": [], "item_id": []},
                                  "test": {"user_id": [], "item_id": []}}, sets, {})
    args = argparse.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64,
                              gcn_layers=2, reg_weight=1e-5, optimizer="Adam", lr=2e-3, l2=0.0)
    m = LightGCN(args, corpus).to(dev); m.train()
    g = torch.Generator(device=dev); g.manual_seed(1)
    batch = {"user_id": torch.randint(0, nU, (B,), generator=g, device=dev), "pos_item": torch.randint(0, nI, (B,), generator=g, device=dev),
             "neg_items": torch.randint(1, nI, (B,), generator=g, device=dev)}
    def step(k):
        m.optimizer.zero_grad(); loss = m.predict(batch); loss.backward(); m.optimizer.step()
    for _ in range(3): step(0)
    t = ev_time(step, 20)
    cptr, crow, col, val = m._csr()
    E = torch.cat([m.user_embedding.weight.data, m.item_embedding.weight.data])
    part = torch.empty((crow.numel(), 64), device=dev)
    t_spmm = ev_time(lambda k: hip_ops.spmm_csr_chunked(cptr, crow, col, val, E, partials=part), 50)
    nnz = int(col.numel())
    emit(case="C3 LightGCN L=2 D=64 ml-1m-shaped synthetic graph, Adam, B=2048 (zero_grad/predict/backward/step)", nodes=nU + nI,
         nnz=nnz, step_ms=t * 1e3, triplets_per_s=B / t, spmm_us=t_spmm * 1e6,
         spmm_GBs=(nnz * 8 + 2 * (nU + nI) * 64 * 4) / t_spmm / 1e9, spmm_GFLOPs=2 * nnz * 64 / t_spmm / 1e9)


def lightgcn_epoch_case():
    """C3 end to end: one fit() epoch of LightGCN on the ml-1m-shaped synthetic graph (sampler + batching + steps) through the
    reference loop (per-sample DataLoader collation) and through HipRunner (batches sliced from device columns)"""
    import random
    from whisprrec_amd import runner
    from whisprrec_amd.lightgcn import LightGCN
    rng = np.random.RandomState(0)
    nU, nI, B = 6040, 3706, 2048
    sets, tu, ti = {}, [], []
    for uu in range(nU):
        k = int(min(nI - 1, max(16, rng.pareto(1.2) * 40)))
        items = np.unique(np.minimum((rng.pareto(0.8, k) * 30).astype(np.int64), nI - 1))
        items = items[items < nI - 64]                       # leave every user some items to draw negatives from
        sets[uu] = set(items.tolist()); tu += [uu] * len(items); ti += items.tolist()
    frames = {"train": {"user_id": np.asarray(tu), "item_id": np.asarray(ti)}, "dev": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)},
              "test": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)}}
    corpus = host.Corpus(nU, nI, frames, sets, {u: set() for u in sets})
    for name, cls, prep in (("BaseRunner (reference loop)", runner.BaseRunner, 0), ("HipRunner", runner.HipRunner, 0),
                            ("HipRunner --device_epoch_prep 1", runner.HipRunner, 1)):
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        args = argparse.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, gcn_layers=2,
                                  reg_weight=1e-5, optimizer="Adam", lr=2e-3, l2=0.0, epoch=1, check_epoch=1, test_epoch=-1,
                                  early_stop=10, batch_size=B, eval_batch_size=B, num_workers=0, pin_memory=0, topk="10",
                                  metric="NDCG", device_epoch_prep=prep, random_seed=1)
        m = LightGCN(args, corpus).to(dev)
        ds = LightGCN.Dataset(m, corpus, "train")
        r = cls(args)
        r.fit(ds, epoch=1); torch.cuda.synchronize()
        t0 = time.perf_counter(); loss = r.fit(ds, epoch=2); torch.cuda.synchronize(); t = time.perf_counter() - t0
        emit(case="C3 LightGCN L=2 D=64 ml-1m-shaped, whole fit() epoch (%d train rows, B=2048, Adam)" % len(tu), runner=name,
             epoch_s=t, triplets_per_s=len(tu) / t, loss=loss)


def sasrec_embedding_case():
    nI, D, B, T = 3706, 64, 2048, 20
    g = torch.Generator(device=dev); g.manual_seed(1)
    W = torch.randn(nI, D, generator=g, device=dev)
    hist = torch.randint(0, nI, (B, T), generator=g, device=dev)
    pn = torch.randint(0, nI, (2 * B,), generator=g, device=dev)
    idx = torch.cat([hist.reshape(-1), pn])
    grad_out = torch.randn(idx.numel(), D, generator=g, device=dev)
    G = torch.zeros_like(W)
    t_g = ev_time(lambda k: hip_ops.gather_rows(W, idx), 50)
    t_s = ev_time(lambda k: hip_ops.scatter_add_rows(G, idx, grad_out, padding_idx=0), 50)
    emit(case="C5 SASRec item-embedding slice: gather + sorted scatter-add of B*(T+2) rows, D=64", rows=int(idx.numel()),
         gather_us=t_g * 1e6, scatter_us=t_s * 1e6, gather_GBs=idx.numel() * 512 / t_g / 1e9)


def ml100k_dropin_case():
    import argparse as ap
    from whisprrec_amd.bprmf import BPRMF
    from whisprrec_amd import runner
    g2 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g2_ml100k_curve.npz"))
    ptr, idx = g2["clicked_ptr"], g2["clicked_idx"]
    tcs = {u: set(idx[ptr[u]:ptr[u + 1]].tolist()) for u in range(943)}
    data = {"train": {"user_id": g2["train_user"].astype(np.int64), "item_id": g2["train_item"].astype(np.int64)},
            "dev": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)},
            "test": {"user_id": np.zeros(0, np.int64), "item_id": np.zeros(0, np.int64)}}
    corpus = host.Corpus(943, 1574, data, tcs, {u: set() for u in range(943)})
    for opt, rname in (("Adam", "BaseRunner"), ("SGD", "BaseRunner"), ("Adam", "HipRunner"), ("SGD", "HipRunner"), ("SGD", "HipRunner+device_epoch_prep")):
        args = ap.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, fused=1,
                            epoch=1, check_epoch=1, test_epoch=-1, early_stop=10, lr=1e-3, l2=0.0, batch_size=2048,
                            eval_batch_size=2048, optimizer=opt, num_workers=0, pin_memory=0, topk="10,20", metric="NDCG, HR",
                            device_epoch_prep=1 if "device" in rname else 0, random_seed=3407)
        model = BPRMF(args, corpus).to(dev)
        ds = BPRMF.Dataset(model, corpus, "train")
        r = (runner.BaseRunner if rname == "BaseRunner" else runner.HipRunner)(args)
        r.fit(ds, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for e in range(3): r.fit(ds, e + 2)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 3
        emit(case="C1 BPRMF ml-100k (943x1574, 66,016 train rows, B=2048) whole fit() epoch incl. sampler + batching",
             runner=rname, optimizer=opt, epoch_s=t, triplets_per_s=66016 / t)


def eval_case():
    for nU, nI, n, D in ((943, 1574, 8252, 64), (100_000, 100_000, 100_000, 64), (200_000, 1_000_000, 20_000, 64)):
        g = torch.Generator(device=dev); g.manual_seed(1)
        U = torch.randn(nU, D, generator=g, device=dev); I = torch.randn(nI, D, generator=g, device=dev)
        eu = torch.randint(0, nU, (n,), generator=g, device=dev); et = torch.randint(0, nI, (n,), generator=g, device=dev)
        per = 50
        ptr = torch.arange(0, (nU + 1) * per, per, device=dev, dtype=torch.int64)
        idx = torch.sort(torch.randint(0, nI, (nU, per), generator=g, device=dev, dtype=torch.int32), dim=1)[0].reshape(-1).contiguous()
        hip_ops.rank_eval(U, I, eu, et, ptr, idx)
        t = ev_time(lambda k: hip_ops.rank_eval(U, I, eu, et, ptr, idx), 5)
        flops = 2.0 * n * nI * D
        emit(case="K10 full-ranking evaluation (wr_rank_eval, MFMA f32 32x32x2 score tiles + mask + count)", eval_rows=n,
             items=nI, D=D, ms=t * 1e3, TFLOPs=flops / t / 1e12, mfma_f32_peak_frac=flops / t / 157.3e12)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c2big", "lazy", "c2epoch", "c2pcie", "c4", "c3", "c3epoch", "c5", "c1", "eval"]
    if "c2" in which:
        for B, NB in ((2048, 256), (16384, 128), (65536, 64), (262144, 16)):
            bprmf_case("C2 BPRMF 1Mx1M D=64 SGD l2=0", 1_000_000, 1_000_000, 64, B, NB)
        bprmf_case("C2 SGD l2=1e-6 (dense weight decay)", 1_000_000, 1_000_000, 64, 65536, 16, l2=1e-6)
        bprmf_case("C2 Adam (dense, reference default optimizer)", 1_000_000, 1_000_000, 64, 65536, 16, opt="Adam")
        bprmf_case("C2 Zipf(1.0) items", 1_000_000, 1_000_000, 64, 65536, 64, zipf=1.0)
    if "c2big" in which:
        bprmf_case("C2 BPRMF 1Mx1M D=64 SGD l2=0", 1_000_000, 1_000_000, 64, 1048576, 8)
    if "c2epoch" in which:
        c2_epoch_case()
    if "c2pcie" in which:
        c2_pcie_case()
    if "lazy" in which:
        bprmf_case("C2 SGD l2=1e-6", 1_000_000, 1_000_000, 64, 65536, 64, l2=1e-6, lazy=True)
        bprmf_case("C2 Adam", 1_000_000, 1_000_000, 64, 65536, 64, opt="Adam", lazy=True)
        bprmf_case("C2 Adam l2=1e-6 (the reference README's command line)", 1_000_000, 1_000_000, 64, 65536, 64, opt="Adam", l2=1e-6, lazy=True)
        bprmf_case("C2 Adam l2=1e-6 B=2048 (README command line, default batch)", 1_000_000, 1_000_000, 64, 2048, 256, opt="Adam", l2=1e-6, lazy=True)
        bprmf_case("C2 Adam Zipf(1.0) items", 1_000_000, 1_000_000, 64, 65536, 64, opt="Adam", zipf=1.0, lazy=True)
        bprmf_case("C2 Adam B=2048", 1_000_000, 1_000_000, 64, 2048, 256, opt="Adam", lazy=True)
    if "c4" in which:
        bprmf_case("C4 shapes on ONE GPU: 10Mx10M D=128 SGD l2=0", 10_000_000, 10_000_000, 128, 65536, 32)
    if "c3" in which:
        lightgcn_case()
    if "c3epoch" in which:
        lightgcn_epoch_case()
    if "c5" in which:
        sasrec_embedding_case()
    if "c1" in which:
        ml100k_dropin_case()
    if "eval" in which:
        eval_case()
