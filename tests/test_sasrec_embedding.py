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


def _sorted_reference(table, idx, src, alpha, pad):
    """the sorted path's bits: stable sort of the positions by row, runs summed in position order (wr_apply_rows_sorted)"""
    from whisprrec_amd import abi
    keys = torch.where((idx == pad) | (idx < 0) | (idx >= table.shape[0]), torch.full_like(idx, table.shape[0]), idx)
    srt, perm = torch.sort(keys, stable=True)
    srt, perm = srt.to(torch.int32), perm.to(torch.int32)
    abi.check(abi.lib().wr_apply_rows_sorted(table.data_ptr(), table.shape[0], table.shape[1], srt.data_ptr(), perm.data_ptr(),
                                             src.data_ptr(), idx.numel(), alpha, torch.cuda.current_stream().cuda_stream),
              "wr_apply_rows_sorted")
    return table


@pytest.mark.parametrize("n_rows,n,D,hot", [(100_000, 45_056, 64, 0), (1_250_000, 114_688, 128, 0), (3_000_000, 200_000, 64, 0),
                                            (50_000, 60_000, 64, 20_000), (20_000, 262_144, 32, 300), (1_000_000, 300_001, 64, 0),
                                            (16_384, 17, 64, 0), (1_000_000, 45_056, 64, -12_000)])
def test_scatter_add_large_tables_without_a_sort(n_rows, n, D, hot):
    """tables beyond the LDS counting sort take the row plan of wr_scatter.hip (no sort of the positions): same BITS as the
    sorted path — a hot row with more recurring positions than a range's list holds goes the brute-force way (hot = 20,000),
    a long bin is ranked exactly (hot = 300), a range that draws more positions than its bucket holds scans the segment itself
    (hot < 0), more than 2^18 positions still take the radix sort"""
    import oracle
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(n_rows % 1000 + n)
    idx = rng.randint(0, n_rows, n).astype(np.int64)
    if hot > 0:
        idx[rng.permutation(n)[:hot]] = n_rows // 3
    elif hot < 0:                                                    # many positions in ONE range of rows, few of them shared: the
        idx[rng.permutation(n)[:-hot]] = rng.randint(0, 40_000, -hot)    # range's bucket overflows, its list does not
    idx[::11] = 0                                                    # the padding row
    idx[5] = -1                                                      # out of range: skipped like padding
    src = rng.standard_normal((n, D)).astype(np.float32)
    base = rng.standard_normal((n_rows, D)).astype(np.float32) if n_rows <= 100_000 else None
    t0 = torch.from_numpy(base).to(dev) if base is not None else torch.zeros(n_rows, D, device=dev)
    idx_d, src_d = torch.from_numpy(idx).to(dev), torch.from_numpy(src).to(dev)
    got = hip_ops.scatter_add_rows(t0.clone(), idx_d, src_d, padding_idx=0, alpha=-0.5)
    ref_bits = _sorted_reference(t0.clone(), idx_d, src_d, -0.5, 0)
    assert torch.equal(got, ref_bits)
    if n_rows <= 100_000:
        ref = base + (-0.5) * oracle.scatter_add_rows(n_rows, np.where(idx < 0, 0, idx), src, padding_idx=0)
        assert rel_err(got.cpu().numpy(), ref) < TOL


def test_scatter_plan_of_several_segments():
    """one row plan for a chunk of calls (the row-sharded step's gradient rows): ragged segments, applied one by one"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)
    n_rows, D, S, stride = 125_000, 64, 5, 40_000
    idx = torch.from_numpy(rng.randint(0, n_rows, (S, stride)).astype(np.int64)).to(dev)
    lens = [40_000, 1, 0, 39_999, 12_345]
    plan = hip_ops.ScatterPlan(idx, n_rows, seg_len=torch.tensor(lens, dtype=torch.int32, device=dev))
    assert not plan.slow
    tab, ref = torch.zeros(n_rows, D, device=dev), torch.zeros(n_rows, D, device=dev)
    for s in range(S):
        src = torch.from_numpy(rng.standard_normal((stride, D)).astype(np.float32)).to(dev)
        plan.apply(tab, s, lens[s], src, alpha=0.25)
        if lens[s]:
            _sorted_reference(ref, idx[s, :lens[s]].contiguous(), src[:lens[s]].contiguous(), 0.25, -1)
    assert torch.equal(tab, ref)
