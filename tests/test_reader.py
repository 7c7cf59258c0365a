"""whisprrec_amd/reader.py against what the reference's BaseReader produced (tests/golden/g8_reader.npz: 25,000 rows of
ml-100k through reference src/helpers/BaseReader.py + src/utils/sample.py under both --sample rules) — ids, split membership
and ROW ORDER are index work: bit-exact."""
import argparse
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g8():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_reader.npz"))


def _write_inter(g8, root):
    d = root / "ml-100k"
    d.mkdir()
    with open(d / "ml-100k.inter", "w") as f:
        f.write("user_id:token\titem_id:token\trating:float\ttimestamp:float\n")
        for a, b, c, t in zip(g8["in_user"].tolist(), g8["in_item"].tolist(), g8["in_rating"].tolist(), g8["in_time"].tolist()):
            f.write("%d\t%d\t%d\t%d\n" % (a, b, c, t))
    return str(root) + "/"


@pytest.mark.parametrize("rule", ["random", "loo"])
def test_reader_matches_reference(g8, tmp_path, rule):
    from whisprrec_amd.reader import BaseReader
    path = _write_inter(g8, tmp_path)
    r = BaseReader(argparse.Namespace(sep="\t", path=path, dataset="ml-100k", sample=rule))
    assert [r.n_users, r.n_items, r.all_df["user_id"].size] == g8[rule + "_shape"].tolist()
    for ph in ("train", "dev", "test"):
        assert np.array_equal(r.data_df[ph]["user_id"], g8[f"{rule}_{ph}_user"]), ph
        assert np.array_equal(r.data_df[ph]["item_id"], g8[f"{rule}_{ph}_item"]), ph
        assert np.array_equal(r.data_df[ph]["timestamp"], g8[f"{rule}_{ph}_time"]), ph
    c = r.corpus()
    tu, ti = g8[f"{rule}_train_user"], g8[f"{rule}_train_item"]
    assert sum(len(s) for s in c.train_clicked_set.values()) == len(set(zip(tu.tolist(), ti.tolist())))
    for ph in ("dev", "test"):
        for a, b in zip(g8[f"{rule}_{ph}_user"][:200].tolist(), g8[f"{rule}_{ph}_item"][:200].tolist()):
            assert b in c.residual_clicked_set[a]


def test_split_restatement_equals_sklearn():
    """random_split restates train_test_split(random_state=42) (sample.py:139-140)"""
    sk = pytest.importorskip("sklearn.model_selection")
    from whisprrec_amd.reader import random_split
    for n in (10, 101, 82520):
        a, rest = sk.train_test_split(np.arange(n), train_size=0.8, random_state=42, shuffle=True)
        d, t = sk.train_test_split(rest, train_size=0.5, random_state=42, shuffle=True)
        tr, dv, te = random_split(n)
        assert np.array_equal(tr, a) and np.array_equal(dv, d) and np.array_equal(te, t)


@pytest.mark.skipif(not os.path.exists("/root/reference/data/ml-100k/ml-100k.inter"), reason="reference data not on this machine")
def test_full_ml100k_matches_the_training_golden(tmp_path):
    """the whole file: the train split must be the one the reference trained on in tests/golden/g2_ml100k_curve.npz"""
    import shutil
    from whisprrec_amd.reader import BaseReader
    shutil.copytree("/root/reference/data/ml-100k", tmp_path / "ml-100k")
    r = BaseReader(argparse.Namespace(sep="\t", path=str(tmp_path) + "/", dataset="ml-100k", sample="random"))
    g2 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g2_ml100k_curve.npz"))
    assert (r.n_users, r.n_items) == (943, 1574) and r.all_df["user_id"].size == 82520
    assert np.array_equal(r.data_df["train"]["user_id"], g2["train_user"])
    assert np.array_equal(r.data_df["train"]["item_id"], g2["train_item"])


@pytest.mark.gpu
@pytest.mark.parametrize("model,extra", [("BPRMF", ["--runner_name", "HipRunner", "--device_epoch_prep", "1", "--optimizer", "SGD", "--lr", "0.5"]),
                                         ("BPRMF", ["--lr", "1e-3", "--l2", "1e-6", "--lazy_optimizer", "1"]),
                                         ("LightGCN", ["--lr", "1e-3", "--gcn_layers", "2"]),
                                         ("SGL", ["--lr", "1e-3", "--type", "ED"]),
                                         ("SASRec", ["--lr", "1e-3", "--emb_size", "32", "--num_layers", "1", "--num_heads", "2"]),
                                         # device-side sampler + shuffle: the per-row histories must follow the device's order
                                         ("SASRec", ["--lr", "1e-3", "--emb_size", "32", "--num_layers", "1", "--num_heads", "2",
                                                     "--runner_name", "HipRunner", "--device_epoch_prep", "1"]),
                                         ("LightGCN", ["--lr", "1e-3", "--gcn_layers", "2", "--runner_name", "HipRunner",
                                                       "--device_epoch_prep", "1"])])
def test_standalone_launcher_trains_from_an_inter_file(g8, tmp_path, model, extra):
    """python -m whisprrec_amd.main: .inter file -> reader -> model -> runner.train (dev evaluation, best checkpoint) ->
    test metrics, with the reference's command line (src/main.py)"""
    import re
    from whisprrec_amd import main as launcher
    path = _write_inter(g8, tmp_path)
    argv = ["--model_name", model, "--dataset", "ml-100k", "--path", path, "--epoch", "3", "--batch_size", "1024",
            "--log_file", str(tmp_path / "log.txt"), "--model_path", str(tmp_path / "m.pt"), "--num_workers", "0"] + extra
    res = launcher.main(argv)
    vals = dict(re.findall(r"(\w+@\d+):([0-9.]+)", res))
    assert {"HR@10", "NDCG@10", "HR@20", "NDCG@20"} <= set(vals)
    assert 0.0 < float(vals["HR@20"]) <= 1.0
    assert (tmp_path / "m.pt").exists()
    log = (tmp_path / "log.txt").read_text()
    assert "Best Iter(dev)" in log and "Epoch 3" in log


def test_seq_reader_matches_reference(g8, tmp_path):
    from whisprrec_amd.reader import SeqReader
    path = _write_inter(g8, tmp_path)
    r = SeqReader(argparse.Namespace(sep="\t", path=path, dataset="ml-100k", sample="random"))
    for ph in ("train", "dev", "test"):
        assert np.array_equal(r.data_df[ph]["position"], g8[f"seq_{ph}_position"]), ph
        assert np.array_equal(r.data_df[ph]["user_id"], g8[f"seq_{ph}_user"]), ph
    for uid in (0, 5):
        assert np.array_equal(np.asarray(r.user_his[uid], dtype=np.int64), g8[f"seq_his_{uid}"])
    assert r.corpus().user_his is r.user_his


@pytest.mark.gpu
@pytest.mark.parametrize("tag,runner_name", [("adam", "BaseRunner"), ("adam", "HipRunner"), ("sgd", "HipRunner"),
                                             ("lightgcn", "BaseRunner"), ("lightgcn", "HipRunner"), ("sgl", "HipRunner"),
                                             ("sasrec", "BaseRunner"), ("sasrec", "HipRunner")])
def test_end_to_end_run_matches_the_reference_train_loop(g8, tmp_path, tag, runner_name):
    """the same command line, the same seed, the same data file: per-epoch training loss, dev metrics after every epoch and
    the final test metrics of the reference's BaseRunner.train on CPU (tests/golden/g9_end_to_end.npz) — reader, sampler and
    shuffle streams, initial tables, six epochs of training (README optimizer settings for 'adam') and full-ranking evaluation."""
    from whisprrec_amd import main as launcher
    g9 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_end_to_end.npz"))
    path = _write_inter(g8, tmp_path)
    lr, l2, epochs = g9[tag + "_hp"]
    name, extra = {"adam": ("BPRMF", []), "sgd": ("BPRMF", []), "lightgcn": ("LightGCN", ["--gcn_layers", "2", "--reg_weight", "1e-5"]),
                   "sgl": ("SGL", ["--gcn_layers", "2", "--reg_weight", "1e-4", "--type", "ED", "--ssl_tau", "0.2", "--ssl_weight", "0.05",
                                   "--drop_ratio", "0.1"]),
                   "sasrec": ("SASRec", ["--emb_size", "32", "--num_layers", "1", "--num_heads", "2", "--dropout", "0.0", "--history_max", "20"])}[tag]
    argv = extra + ["--model_name", name, "--runner_name", runner_name, "--dataset", "ml-100k", "--path", path, "--epoch", str(int(epochs)),
            "--batch_size", "1024", "--eval_batch_size", "2048", "--optimizer", "SGD" if tag == "sgd" else "Adam", "--lr", repr(float(lr)),
            "--l2", repr(float(l2)), "--log_file", str(tmp_path / "log.txt"), "--model_path", str(tmp_path / "m.pt"),
            "--num_workers", "0", "--topk", "10,20", "--metric", "NDCG, HR", "--random_seed", "3407"]
    args, model_class, reader_class, runner_class = launcher.build_args(argv)
    launcher.init_seed(args.random_seed)
    import torch
    args.device = torch.device("cuda")
    corpus = reader_class(args).corpus()
    model = model_class(args, corpus).to(args.device)
    data = {ph: model_class.Dataset(model, corpus, ph) for ph in ("train", "dev", "test")}
    run = runner_class(args)
    losses, devs = [], []
    for epoch in range(args.epoch):
        losses.append(run.fit(data["train"], epoch=epoch + 1))
        devs.append(run.evaluate(data["dev"], run.topk[:1], run.metrics))
    test = run.evaluate(data["test"], run.topk, run.metrics)
    assert np.allclose(losses, g9[tag + "_loss"], rtol=5e-5 if tag in ("lightgcn", "sgl", "sasrec") else 2e-5, atol=0)
    n_eval = len(data["dev"])
    flips = 6.0 / n_eval                              # a handful of near-tied ranks may fall on the other side of the cut
    dev = np.asarray([[d[k] for k in g9[tag + "_dev_keys"]] for d in devs])
    assert np.abs(dev - g9[tag + "_dev"]).max() <= flips, np.abs(dev - g9[tag + "_dev"]).max()
    tst = np.asarray([test[k] for k in g9[tag + "_test_keys"]])
    assert np.abs(tst - g9[tag + "_test"]).max() <= flips


def test_token_ids_need_not_be_numbers(tmp_path):
    """alphanumeric tokens (yelp / food style) and integers beyond 2^53 are ids like any other — the corpus
    equals the one read from the same file with the tokens replaced by small integers"""
    from whisprrec_amd import reader
    rng = np.random.RandomState(3)
    n_u, n_i, rows = 30, 50, 1500
    u = rng.randint(0, n_u, rows)
    i = rng.randint(0, n_i, rows)
    r = rng.randint(1, 6, rows)
    t = np.arange(rows) + 1_000_000
    big = 2 ** 60
    for tag, fu, fi in (("num", lambda x: str(x + 7), lambda x: str(x + 3)),
                        ("tok", lambda x: "u_%dx" % (x * 13), lambda x: str(big + x))):
        d = tmp_path / tag / "toy"
        d.mkdir(parents=True)
        with open(d / "toy.inter", "w") as f:
            f.write("user_id:token\titem_id:token\trating:float\ttimestamp:float\n")
            for a, b, c, e in zip(u, i, r, t):
                f.write("%s\t%s\t%d\t%d\n" % (fu(a), fi(b), c, e))
    cols_num = reader.count_statics(reader.read_inter(str(tmp_path / "num" / "toy" / "toy.inter")), "toy")
    cols_tok = reader.count_statics(reader.read_inter(str(tmp_path / "tok" / "toy" / "toy.inter")), "toy")
    for k in cols_num:
        assert np.array_equal(cols_num[k], cols_tok[k]), k
