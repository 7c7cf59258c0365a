"""Full-ranking evaluation on the matrix cores (wr_rank_eval) against the oracle's restatement of
BaseRunner.interface + evaluate_method, and the end-to-end metrics against the host path of the runner."""
import argparse

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _mask_csr(rng, nU, nI, max_len):
    ptr = np.zeros(nU + 1, np.int64); chunks = []
    for u in range(nU):
        k = rng.randint(0, max_len)
        it = np.sort(rng.choice(nI, k, replace=False)).astype(np.int32)
        chunks.append(it); ptr[u + 1] = ptr[u] + k
    return ptr, np.concatenate(chunks) if chunks else np.zeros(1, np.int32)


def test_embedding_sizes_beyond_the_lds_are_refused_not_launched():
    """the LDS-operand kernel needs (128 + 32) * (D + 1) * 4 + 512 B: 164,992 B at D = 256, more than gfx950's 163,840 B per
    workgroup — wr_rank_eval must say so (WR_E_RANGE) instead of failing at launch; HipRunner.evaluate then takes the host path"""
    from whisprrec_amd import abi, hip_ops
    dev = torch.device("cuda:0")
    assert hip_ops.rank_eval_supports(252) and not hip_ops.rank_eval_supports(256) and hip_ops.rank_eval_supports(64)
    z = torch.zeros(8, 256, device=dev)
    i = torch.zeros(4, dtype=torch.int64, device=dev)
    with pytest.raises(abi.WhisprRecHipError, match="D <= 252"):
        hip_ops.rank_eval(z, z, i, i)


@pytest.mark.parametrize("D,nI,n", [(64, 1574, 700), (32, 5000, 300), (128, 333, 200), (64, 40, 130), (16, 2100, 257), (8, 777, 129),
                                    (24, 500, 100), (64, 4133, 1000), (192, 700, 150), (252, 600, 140)])
def test_ranks_match_oracle(D, nI, n):
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(D + nI)
    nU = 400
    U = rng.standard_normal((nU, D)).astype(np.float32)
    I = rng.standard_normal((nI, D)).astype(np.float32)
    ptr, idx = _mask_csr(rng, nU, nI, min(60, nI // 2))
    eu = rng.randint(0, nU, n); et = rng.randint(0, nI, n)
    t = lambda a: torch.from_numpy(a).to(dev)
    rank, tsc = hip_ops.rank_eval(t(U), t(I), t(eu), t(et), t(ptr), t(idx))
    ref, margin = oracle.eval_ranks(U, I, eu, et, ptr, idx)
    ok = margin > 1e-4                                   # rows with a near-tie are decided by rounding, skip them
    assert ok.mean() > 0.9
    assert np.array_equal(rank.cpu().numpy()[ok], ref[ok])
    assert np.allclose(tsc.cpu().numpy(), (U[eu] * I[et]).sum(1), rtol=1e-5, atol=1e-5)
    # no mask (--test_all 0)
    rank2, _ = hip_ops.rank_eval(t(U), t(I), t(eu), t(et))
    ref2, margin2 = oracle.eval_ranks(U, I, eu, et)
    ok2 = margin2 > 1e-4
    assert np.array_equal(rank2.cpu().numpy()[ok2], ref2[ok2])
    # masked target never beats itself, everything masked -> rank 1
    full_ptr = np.arange(0, (nU + 1) * nI, nI, dtype=np.int64)
    full_idx = np.tile(np.arange(nI, dtype=np.int32), nU)
    rank3, _ = hip_ops.rank_eval(t(U), t(I), t(eu), t(et), t(full_ptr), t(full_idx))
    assert int(rank3.min()) == 1 and int(rank3.max()) == 1


def test_runner_metrics_device_vs_host(g2):
    """HipRunner.evaluate through wr_rank_eval equals the reference-style host evaluation (interface + evaluate_method)."""
    from whisprrec_amd import runner, host
    from whisprrec_amd.bprmf import BPRMF
    from test_host_contract import ml100k_corpus, seed_all
    dev = torch.device("cuda:0")
    args = argparse.Namespace(device=dev, model_path="/tmp/x.pt", buffer=1, num_neg=1, test_all=1, embedding_size=64, fused=1,
                              epoch=1, check_epoch=1, test_epoch=-1, early_stop=10, lr=1e-3, l2=0.0, batch_size=2048,
                              eval_batch_size=512, optimizer="SGD", num_workers=0, pin_memory=0, topk="5,10,20",
                              metric="NDCG, HR", device_epoch_prep=0, random_seed=3407)
    seed_all(7)
    corpus = ml100k_corpus(g2)
    rng = np.random.RandomState(3)
    dev_u = rng.randint(0, 943, 1500); dev_i = rng.randint(0, 1574, 1500)
    corpus.data_df["dev"] = {"user_id": dev_u, "item_id": dev_i}
    for a, b in zip(dev_u.tolist(), dev_i.tolist()):
        corpus.residual_clicked_set[a].add(b)
    model = BPRMF(args, corpus).to(dev)
    with torch.no_grad():
        model.user_embeddings.weight.mul_(30); model.item_embeddings.weight.mul_(30)
    ds = BPRMF.Dataset(model, corpus, "dev")
    host_res = runner.BaseRunner(args).evaluate(ds, [5, 10, 20], ["NDCG", "HR"])
    dev_res = runner.HipRunner(args).evaluate(ds, [5, 10, 20], ["NDCG", "HR"])
    for k in host_res:
        assert abs(host_res[k] - dev_res[k]) < 2e-3, (k, host_res[k], dev_res[k])   # a near-tie may move one rank in 1,500
