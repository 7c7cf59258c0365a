"""GPU integration tests: the BPRMF drop-in model driven (a) by the reference runner loop restated in
whisprrec_amd.runner.BaseRunner.fit (predict / backward / optimizer.step per batch) and (b) by HipRunner.fit (device epoch),
both starting from the reference's seed only — sampler, shuffle, initial tables, loss curve and final tables must
reproduce what the reference produced on ml-100k (tests/golden/g2_ml100k_curve.npz) within 1e-5 relative."""
import argparse

import numpy as np
import pytest
import torch

from conftest import rel_err
from test_host_contract import ml100k_corpus, seed_all

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _args(**kw):
    base = dict(device=torch.device("cuda:0"), model_path="/tmp/wr_test_model_gpu.pt", buffer=1, num_neg=1, test_all=1,
                embedding_size=64, fused=1, epoch=2, check_epoch=1, test_epoch=-1, early_stop=10, lr=0.5, l2=0.0,
                batch_size=2048, eval_batch_size=2048, optimizer="SGD", num_workers=0, pin_memory=0, topk="10,20",
                metric="NDCG, HR")
    base.update(kw)
    return argparse.Namespace(**base)


def _setup(g2, **kw):
    from whisprrec_amd.bprmf import BPRMF
    args = _args(**kw)
    seed_all(3407)
    corpus = ml100k_corpus(g2)
    model = BPRMF(args, corpus).to(args.device)     # main.py:66
    return args, corpus, model, BPRMF.Dataset(model, corpus, "train")


@pytest.mark.parametrize("fused", [1, 0])
def test_reference_runner_loop_sgd(g2, fused):
    from whisprrec_amd import runner
    args, corpus, model, ds = _setup(g2, fused=fused)
    r = runner.BaseRunner(args)
    means = [r.fit(ds, epoch=1), r.fit(ds, epoch=2)]
    assert abs(means[0] - g2["sgd_epoch_mean"][0]) / g2["sgd_epoch_mean"][0] < TOL
    assert abs(means[1] - g2["sgd_epoch_mean"][1]) / g2["sgd_epoch_mean"][1] < TOL
    assert rel_err(model.user_embeddings.weight.detach().cpu().numpy(), g2["sgd_Uend"]) < TOL
    assert rel_err(model.item_embeddings.weight.detach().cpu().numpy(), g2["sgd_Iend"]) < TOL


def test_hip_runner_sgd(g2):
    from whisprrec_amd import runner
    args, corpus, model, ds = _setup(g2)
    r = runner.HipRunner(args)
    means = [r.fit(ds, epoch=1), r.fit(ds, epoch=2)]
    assert abs(means[0] - g2["sgd_epoch_mean"][0]) / g2["sgd_epoch_mean"][0] < TOL
    assert abs(means[1] - g2["sgd_epoch_mean"][1]) / g2["sgd_epoch_mean"][1] < TOL
    assert rel_err(model.user_embeddings.weight.detach().cpu().numpy(), g2["sgd_Uend"]) < TOL
    assert rel_err(model.item_embeddings.weight.detach().cpu().numpy(), g2["sgd_Iend"]) < TOL


def test_hip_runner_device_epoch_prep(g2):
    """--device_epoch_prep 1: negatives and shuffle produced on the device.  Not the NumPy stream, so no golden curve; the
    sampler rule must hold and training must make progress from the same initial tables."""
    from whisprrec_amd import runner, hip_ops
    args, corpus, model, ds = _setup(g2, device_epoch_prep=1, random_seed=3407, lr=2.0)
    r = runner.HipRunner(args)
    u, p, n = r._device_epoch(ds, args.device, 1)
    assert sorted(zip(u.cpu().tolist(), p.cpu().tolist())) == sorted(zip(g2["train_user"].tolist(), g2["train_item"].tolist()))
    ptr, idx = g2["clicked_ptr"], g2["clicked_idx"]
    un, nn = u.cpu().numpy(), n.cpu().numpy()
    assert nn.min() >= 1 and nn.max() < 1574
    assert not np.isin(un.astype(np.int64) * 1574 + nn, g2["train_user"].astype(np.int64) * 1574 + g2["train_item"]).any()
    losses = [r.fit(ds, epoch=e) for e in range(1, 9)]
    assert losses[-1] < losses[0] - 1e-4 and all(np.isfinite(losses))


def test_hip_runner_device_epoch_prep_with_pair_set_and_packed_rows(g2, monkeypatch):
    """big training frames take the hash set of the pairs and the packed source rows (hip_ops.PAIR_SET_MIN_PAIRS): the same
    epochs, bit for bit, as the clicked lists and the two index columns give"""
    from whisprrec_amd import runner, hip_ops
    res = []
    for min_pairs in (1, 1 << 40):
        monkeypatch.setattr(hip_ops, "PAIR_SET_MIN_PAIRS", min_pairs)
        args, corpus, model, ds = _setup(g2, device_epoch_prep=1, random_seed=3407, lr=2.0)
        r = runner.HipRunner(args)
        losses = [r.fit(ds, epoch=e) for e in range(1, 4)]
        cache = r._epoch_cache
        assert (cache[5] is not None and cache[6] is not None) == (min_pairs == 1)       # pair set, packed rows
        res.append((losses, model.user_embeddings.weight.detach().clone(), model.item_embeddings.weight.detach().clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("lazy", [0, 1])
def test_hip_runner_adam_default_optimizer(g2, lazy):
    """reference default optimizer (BaseRunner.py:36), README learning rate; --lazy_optimizer 1 = exact lazy rows, read
    through state_dict() (which brings all rows up to date)"""
    from whisprrec_amd import runner
    args, corpus, model, ds = _setup(g2, optimizer="Adam", lr=1e-3, epoch=1, lazy_optimizer=lazy)
    r = runner.HipRunner(args)
    mean = r.fit(ds, epoch=1)
    assert (model.optimizer.lazy_state is not None) == bool(lazy)
    sd = model.state_dict()
    assert abs(mean - g2["adam_epoch_mean"][0]) / g2["adam_epoch_mean"][0] < TOL
    assert rel_err(sd["user_embeddings.weight"].cpu().numpy(), g2["adam_Uend"]) < 1e-4
    assert rel_err(sd["item_embeddings.weight"].cpu().numpy(), g2["adam_Iend"]) < 1e-4


@pytest.mark.parametrize("lazy", [0, 1])
def test_reference_runner_loop_adam(g2, lazy):
    from whisprrec_amd import runner
    args, corpus, model, ds = _setup(g2, optimizer="Adam", lr=1e-3, epoch=1, lazy_optimizer=lazy)
    r = runner.BaseRunner(args)
    mean = r.fit(ds, epoch=1)
    model.eval()                                     # what evaluate() does first (BaseRunner.py:229): flushes lazy rows
    assert abs(mean - g2["adam_epoch_mean"][0]) / g2["adam_epoch_mean"][0] < TOL
    assert rel_err(model.user_embeddings.weight.detach().cpu().numpy(), g2["adam_Uend"]) < 1e-4


def test_lazy_and_dense_runs_are_bit_identical(g2):
    """two epochs with evaluation in between, SGD with weight decay and Adam with weight decay: same losses, same tables"""
    from whisprrec_amd import runner
    for opt, lr in (("SGD", 0.5), ("Adam", 1e-3)):
        out = []
        for lazy in (0, 1):
            args, corpus, model, ds = _setup(g2, optimizer=opt, lr=lr, l2=1e-4, lazy_optimizer=lazy)
            r = runner.HipRunner(args)
            means = []
            for ep in (1, 2):
                means.append(r.fit(ds, epoch=ep))
                model.eval(); model.train()
            sd = model.state_dict()
            out.append((means, sd["user_embeddings.weight"].clone(), sd["item_embeddings.weight"].clone()))
        assert out[0][0] == out[1][0]
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])


def test_torch_optimizer_on_dense_grads(g1):
    """--optimizer Adagrad with --l2 (BaseRunner.py:36 lists it; weight decay moves every row, so it is not fused): torch.optim
    on the dense gradients our backward emits must equal torch.optim on autograd's dense gradients, which we hold as golden
    gU/gI.  (Without --l2 Adagrad / Adadelta are fused: tests/test_hip_optimizers.py.)"""
    from whisprrec_amd.bprmf import BPRMF
    from whisprrec_amd import host
    args = _args(optimizer="Adagrad", lr=0.05, l2=1e-4)
    corpus = host.Corpus(97, 131, {"train": {"user_id": g1["u0"], "item_id": g1["p0"]},
                                   "dev": {"user_id": [], "item_id": []}, "test": {"user_id": [], "item_id": []}})
    model = BPRMF(args, corpus).to(args.device)
    assert model.optimizer is None
    with torch.no_grad():
        model.user_embeddings.weight.copy_(torch.from_numpy(g1["U0"]))
        model.item_embeddings.weight.copy_(torch.from_numpy(g1["I0"]))
    model.train()
    dev = args.device
    batch = {"user_id": torch.from_numpy(g1["u0"]).to(dev), "pos_item": torch.from_numpy(g1["p0"]).to(dev),
             "neg_items": torch.from_numpy(g1["n0"]).to(dev)}
    loss = model.predict(batch)
    loss.backward()
    assert abs(float(loss.detach()) - float(g1["loss0"][0])) / float(g1["loss0"][0]) < TOL
    assert rel_err(model.user_embeddings.weight.grad.cpu().numpy(), g1["gU"]) < TOL
    assert rel_err(model.item_embeddings.weight.grad.cpu().numpy(), g1["gI"]) < TOL


def test_full_predict_and_eval(g2):
    from whisprrec_amd import runner, host
    args, corpus, model, ds = _setup(g2)
    model.eval()
    users = torch.tensor([0, 5, 942], device=args.device)
    s = model.full_predict({"user_id": users})
    ref = model.user_embeddings.weight.detach()[users] @ model.item_embeddings.weight.detach().t()
    assert s.shape == (3, 1574) and torch.allclose(s, ref, rtol=1e-5, atol=1e-7)
    # evaluation pipeline end to end on a small dev frame
    corpus.data_df["dev"] = {"user_id": np.array([0, 1, 2]), "item_id": np.array([10, 20, 30])}
    for u, i in ((0, 10), (1, 20), (2, 30)):
        corpus.residual_clicked_set[u].add(i)
    dev_ds = type(ds)(model, corpus, "dev")
    r = runner.HipRunner(args)
    res = r.evaluate(dev_ds, [10], ["NDCG", "HR"])
    assert set(res) == {"NDCG@10", "HR@10"} and 0.0 <= res["HR@10"] <= 1.0


def test_save_load_roundtrip(g2, tmp_path):
    args, corpus, model, ds = _setup(g2, model_path=str(tmp_path / "m" / "bprmf.pt"))
    model.save_model()
    sd = torch.load(args.model_path)
    assert list(sd.keys()) == ["user_embeddings.weight", "item_embeddings.weight"]   # reference checkpoint layout
    with torch.no_grad():
        model.user_embeddings.weight.zero_()
    model.load_model()
    assert np.array_equal(model.user_embeddings.weight.detach().cpu().numpy(), g2["sgd_U0"])


@pytest.mark.parametrize("model_name", ["BPRMF", "LightGCN"])
def test_full_train_loop_with_eval_and_checkpoint(g2, tmp_path, model_name, caplog):
    """The whole main.py flow after the reader (main.py:66-84): model -> datasets -> runner.train (fit, evaluate on dev, save best,
    early-stop bookkeeping, reload best) -> print_res on test, with HipRunner and its device evaluation."""
    import logging
    from whisprrec_amd import runner
    from whisprrec_amd.bprmf import BPRMF
    from whisprrec_amd.lightgcn import LightGCN
    cls = {"BPRMF": BPRMF, "LightGCN": LightGCN}[model_name]
    args = _args(model_path=str(tmp_path / "ckpt" / "m.pt"), epoch=3, lr=0.5 if model_name == "BPRMF" else 0.05, gcn_layers=2,
                 reg_weight=1e-5, device_epoch_prep=0, random_seed=3407)
    seed_all(3407)
    corpus = ml100k_corpus(g2)
    rng = np.random.RandomState(0)
    for phase in ("dev", "test"):
        uu = rng.randint(0, 943, 800); ii = rng.randint(0, 1574, 800)
        corpus.data_df[phase] = {"user_id": uu, "item_id": ii}
        for a, b in zip(uu.tolist(), ii.tolist()):
            corpus.residual_clicked_set[a].add(b)
    model = cls(args, corpus).to(args.device)
    data = {ph: cls.Dataset(model, corpus, ph) for ph in ("train", "dev", "test")}
    r = runner.HipRunner(args)
    with caplog.at_level(logging.INFO):
        r.train(data)
    lines = [rec.getMessage() for rec in caplog.records if rec.getMessage().startswith("Epoch")]
    assert len(lines) == 3 and all("loss=" in l and "dev=(" in l for l in lines) and any(l.rstrip().endswith("*") for l in lines)
    assert (tmp_path / "ckpt" / "m.pt").exists()
    res = r.print_res(data["test"])
    assert res.startswith("(") and "NDCG@10" in res and "HR@20" in res
