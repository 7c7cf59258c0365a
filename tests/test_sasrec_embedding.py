"""SASRec's item-embedding slice (K9) on the HIP gather / scatter kernels against what the reference produced
(tests/golden/g5_sasrec_emb.npz): gathers are bit-exact, the scatter-add drops the padding row, and the whole
predict() — HIP embedding + stock PyTorch transformer block — reproduces the reference loss and embedding gradient."""
import argparse

import numpy as np
import pytest
import torch

from conftest import rel_err
from whisprrec_amd import host

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_gather_and_scatter_match_reference(g5):
    from whisprrec_amd.sasrec import HipEmbedding
    dev = torch.device("cuda:0")
    emb = HipEmbedding(71, 64, padding_idx=0).to(dev)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(g5["W0"]))
    hist, pos = torch.from_numpy(g5["hist"]).to(dev), torch.from_numpy(g5["pos"]).to(dev)
    gh, gp = emb(hist), emb(pos)
    assert np.array_equal(gh.detach().cpu().numpy(), g5["gather_hist"])      # a gather is a copy: bit-exact
    assert np.array_equal(gp.detach().cpu().numpy(), g5["gather_pos"])
    ((gh * torch.from_numpy(g5["g_his"]).to(dev)).sum() + (gp * torch.from_numpy(g5["g_pos"]).to(dev)).sum()).backward()
    g = emb.weight.grad.cpu().numpy()
    assert rel_err(g, g5["gW_slice"]) < TOL
    assert not g[0].any()                                                      # padding_idx=0 row gets no gradient


def test_scatter_is_bitwise_reproducible():
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    idx = torch.randint(0, 50, (20000,), generator=g).to(dev)
    src = torch.randn(20000, 64, generator=g).to(dev)
    a = hip_ops.scatter_add_rows(torch.zeros(50, 64, device=dev), idx, src, padding_idx=0)
    b = hip_ops.scatter_add_rows(torch.zeros(50, 64, device=dev), idx, src, padding_idx=0)
    assert torch.equal(a, b) and not a[0].any()
    ref = torch.zeros(50, 64, dtype=torch.float64).index_add_(0, idx.cpu(), src.cpu().double())
    ref[0] = 0
    assert rel_err(a.cpu().numpy(), ref.numpy()) < TOL


def test_full_predict_loss_and_embedding_grad(g5):
    from whisprrec_amd.sasrec import SASRec
    dev = torch.device("cuda:0")
    args = argparse.Namespace(device=dev, model_path="/tmp/wr_sas.pt", buffer=1, num_neg=1, test_all=1, emb_size=64,
                              num_layers=1, num_heads=4, dropout=0.0, history_max=20)
    corpus = host.Corpus(13, 71, {})
    m = SASRec(args, corpus).to(dev)
    sd = {k[4:]: torch.from_numpy(g5[k]) for k in g5.files if k.startswith("sd__")}
    assert set(sd) == set(m.state_dict().keys())                                # reference checkpoint keys
    m.load_state_dict(sd)
    m.train()
    fd = {"history_items": torch.from_numpy(g5["hist"]).to(dev), "lengths": torch.from_numpy(g5["lengths"]).to(dev),
          "pos_item": torch.from_numpy(g5["pos"]).to(dev), "neg_items": torch.from_numpy(g5["neg"]).to(dev)}
    loss = m.predict(fd)
    loss.backward()
    assert abs(float(loss.detach()) - float(g5["loss"][0])) / float(g5["loss"][0]) < 1e-4   # rocBLAS GEMMs in the block
    g = m.item_embedding.weight.grad.cpu().numpy()
    assert rel_err(g, g5["gW_full"]) < 1e-4
    assert not g[0].any()


@pytest.mark.parametrize("n_rows,n,pad", [(3706, 45056, 0), (16383, 70001, 0), (16384, 30000, 5), (40000, 50000, -1), (7, 5000, 0),
                                          (3706, 1023, 0), (3706, 1025, 0)])
def test_scatter_add_both_sort_paths_match_oracle(n_rows, n, pad):
    """tables of up to 16,383 rows take the counting sort in LDS tiles, larger ones the radix sort: same result (the order
    inside a destination row is the original position order either way), padding row dropped"""
    import oracle
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(n_rows + n)
    idx = rng.randint(0, n_rows, n).astype(np.int64)
    idx[::7] = min(3, n_rows - 1)                      # a popular row: long run
    if pad >= 0:
        idx[::5] = pad
    src = rng.standard_normal((n, 64)).astype(np.float32)
    got = hip_ops.scatter_add_rows(torch.zeros(n_rows, 64, device=dev), torch.from_numpy(idx).to(dev), torch.from_numpy(src).to(dev),
                                   padding_idx=pad)
    ref = oracle.scatter_add_rows(n_rows, idx, src, padding_idx=pad)
    assert rel_err(got.cpu().numpy(), ref) < TOL
    if pad >= 0:
        assert not got[pad].any()
    again = hip_ops.scatter_add_rows(torch.zeros(n_rows, 64, device=dev), torch.from_numpy(idx).to(dev),
                                     torch.from_numpy(src).to(dev), padding_idx=pad)
    assert torch.equal(got, again)
