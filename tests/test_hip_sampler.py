"""Device negative sampler (wr_sample_negatives) vs its NumPy restatement in the oracle: bit-exact indices, the
reference's rule (range [1, n_items), never an item of the user's train set — src/models/BaseModel.py:167-177), and
reproducibility."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _csr(sets, nU):
    ptr = np.zeros(nU + 1, np.int64); chunks = []
    for u in range(nU):
        it = sorted(sets.get(u, ()))
        ptr[u + 1] = ptr[u] + len(it)
        chunks.append(np.asarray(it, np.int32))
    idx = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
    return ptr, (idx if len(idx) else np.zeros(1, np.int32))


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_matches_oracle_bit_exact(g3, dtype):
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    nI = int(g3["n_items"][0])
    ptr = g3["clicked_ptr"].astype(np.int64); idx = g3["clicked_idx"].astype(np.int32)
    users = g3["users"]
    nU = len(ptr) - 1
    for epoch in (0, 1, 7):
        neg, err = hip_ops.sample_negatives(torch.from_numpy(users).to(dev).to(dtype), nU, nI, torch.from_numpy(ptr).to(dev),
                                            torch.from_numpy(idx).to(dev), 3407, epoch)
        ref = oracle.sample_negatives_counter(users, nI, ptr, idx, 3407, epoch)
        got = neg.cpu().numpy().astype(np.int64)
        assert np.array_equal(got, ref)                      # indices bit-exact
        assert int(err.item()) == 0
        assert got.min() >= 1 and got.max() < nI              # item 0 is never a negative (BaseModel.py:168)
        for uu, nn in zip(users, got):
            assert nn not in idx[ptr[uu]:ptr[uu + 1]]         # user 3 clicked 55 of 57 items: long redraw chains
    a, _ = hip_ops.sample_negatives(torch.from_numpy(users).to(dev), nU, nI, torch.from_numpy(ptr).to(dev),
                                    torch.from_numpy(idx).to(dev), 3407, 0)
    b, _ = hip_ops.sample_negatives(torch.from_numpy(users).to(dev), nU, nI, torch.from_numpy(ptr).to(dev),
                                    torch.from_numpy(idx).to(dev), 3408, 0)
    assert not torch.equal(a, b)


def test_large_uniformity_and_exclusion():
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    nU, nI, n = 2000, 5000, 1_000_000
    sets = {u: set(rng.randint(0, nI, rng.randint(1, 60)).tolist()) for u in range(nU)}
    ptr, idx = _csr(sets, nU)
    users = rng.randint(0, nU, n)
    neg, err = hip_ops.sample_negatives(torch.from_numpy(users).to(dev), nU, nI, torch.from_numpy(ptr).to(dev),
                                        torch.from_numpy(idx).to(dev), 1, 0)
    got = neg.cpu().numpy()
    assert int(err.item()) == 0 and got.min() >= 1 and got.max() < nI
    # exclusion, vectorised: (user, item) pairs of the train set never appear
    train_keys = np.concatenate([np.full(ptr[u + 1] - ptr[u], u, np.int64) * nI + idx[ptr[u]:ptr[u + 1]] for u in range(nU)])
    assert not np.isin(users.astype(np.int64) * nI + got, train_keys).any()
    counts = np.bincount(got, minlength=nI)[1:]
    assert abs(counts.mean() - n / (nI - 1)) < 1e-6 and counts.std() / counts.mean() < 0.15   # flat histogram
    sample = slice(0, 3000)
    assert np.array_equal(got[sample], oracle.sample_negatives_counter(users[sample], nI, ptr, idx, 1, 0))


def test_user_who_clicked_everything_is_flagged():
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    nI = 20
    sets = {0: set(range(nI)), 1: {3}}
    ptr, idx = _csr(sets, 2)
    neg, err = hip_ops.sample_negatives(torch.tensor([0, 1, 1], device=dev), 2, nI, torch.from_numpy(ptr).to(dev),
                                        torch.from_numpy(idx).to(dev), 5, 0)
    assert int(err.item()) == 2
    assert neg[1].item() != 3 and neg[2].item() != 3


# ----------------------------------------------------------------------------------------------- epoch shuffle
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 4097, 65536, 100003])
def test_epoch_shuffle_matches_oracle_bit_exact(dtype, n):
    """wr_epoch_shuffle: out_k[i] = col_k[perm(i)] with perm = oracle.epoch_permutation, a bijection of [0, n) — the
    device's epoch order (reference: DataLoader(shuffle=True), BaseRunner.py:188-193)."""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(n)
    cols = [rng.randint(0, 1 << 30, n) for _ in range(3)]
    perm = oracle.epoch_permutation(n, 3407, 5)
    assert np.array_equal(np.sort(perm), np.arange(n))
    outs, order = hip_ops.epoch_shuffle([torch.from_numpy(c).to(dtype).to(dev) for c in cols], 3407, 5, want_order=True)
    assert np.array_equal(order.cpu().numpy(), perm)
    for c, o in zip(cols, outs):
        assert o.dtype == dtype and np.array_equal(o.cpu().numpy(), c[perm])
    # fewer columns, no order; another epoch gives another order; the same call twice the same one
    (o0,) = hip_ops.epoch_shuffle([torch.from_numpy(cols[0]).to(dtype).to(dev)], 3407, 5)
    assert torch.equal(o0, outs[0])
    if n >= 1000:
        (o1,) = hip_ops.epoch_shuffle([torch.from_numpy(cols[0]).to(dtype).to(dev)], 3407, 6)
        assert not torch.equal(o1, o0)


def test_epoch_shuffle_full_size_is_a_permutation():
    """C2 scale (100 M rows, BASELINE.json configs[1]): the order is a bijection (every row exactly once) and the three
    columns travel together; sampled positions equal the oracle's permutation."""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    n = 100_000_000
    base = torch.arange(n, dtype=torch.int32, device=dev)
    outs, order = hip_ops.epoch_shuffle([base, base + 1, base * 2 % 1000003], 11, 3, want_order=True)
    assert int(order.min()) == 0 and int(order.max()) == n - 1
    seen = torch.zeros(n, dtype=torch.uint8, device=dev)
    seen[order] = 1
    assert int(seen.sum(dtype=torch.int64)) == n
    assert torch.equal(outs[0].long(), order) and torch.equal(outs[1].long(), order + 1)
    assert torch.equal(outs[2].long(), order * 2 % 1000003)
    # the network is evaluated per row: the oracle restates a sample of positions
    probe = np.array([0, 1, 2, 12345, 65536, n // 2, n - 2, n - 1])
    got = order.cpu().numpy()[probe]
    ref = oracle.epoch_permutation_at(probe, n, 11, 3)
    assert np.array_equal(got, ref)


def test_epoch_shuffle_rejects_bad_arguments():
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    a = torch.arange(10, device=dev)
    with pytest.raises(ValueError):
        hip_ops.epoch_shuffle([a, a[:5]], 1, 1)
    with pytest.raises(TypeError):
        hip_ops.epoch_shuffle([a.float()], 1, 1)
    with pytest.raises(ValueError):
        hip_ops.epoch_shuffle([], 1, 1)


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_fused_range_preparation_equals_sampler_then_shuffle(dtype):
    """wr_epoch_prepare_range (shuffle + negatives fused, any range of output rows) = wr_sample_negatives followed by
    wr_epoch_shuffle, bit for bit, and — through the oracle restatements of both — the reference's rule
    (src/models/BaseModel.py:167-177, src/helpers/BaseRunner.py:188-193)"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)
    nU, nI, n = 300, 97, 20011
    users = rng.randint(0, nU, n)
    items = rng.randint(0, nI, n)
    sets = {}
    for a, b in zip(users, items):
        sets.setdefault(int(a), set()).add(int(b))
    sets[5] = set(range(1, nI - 2))                                  # a user who clicked almost everything: long redraw chains
    ptr, idx = _csr(sets, nU)
    tu, ti = torch.from_numpy(users).to(dev).to(dtype), torch.from_numpy(items).to(dev).to(dtype)
    tptr, tidx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    for epoch in (0, 3):
        neg, _ = hip_ops.sample_negatives(tu, nU, nI, tptr, tidx, 3407, epoch)
        (su, si, sn), order = hip_ops.epoch_shuffle([tu, ti, neg], 3407, epoch, want_order=True)
        prep = hip_ops.EpochPrep(tu, ti, nU, nI, tptr, tidx, 3407, epoch, want_order=True)
        for lo, hi in ((7000, 20011), (0, 1), (1, 7000)):           # ranges in any order, any cut
            prep.fill(lo, hi)
        # membership through the hash set of the pairs (wr_pairset_build) instead of the lists: the same columns
        pairs = hip_ops.pair_set(tptr, tidx, nU)
        neg_set, _ = hip_ops.sample_negatives(tu, nU, nI, None, None, 3407, epoch, pairs=pairs)
        prep_set = hip_ops.EpochPrep(tu, ti, nU, nI, None, None, 3407, epoch, want_order=True, pairs=pairs)
        prep_set.fill(0, n)
        # ... and with the source rows packed into one word each (wr_epoch_prepare_range_packed): the same columns again
        prep_pk = hip_ops.EpochPrep(tu, ti, nU, nI, None, None, 3407, epoch, want_order=True, pairs=pairs,
                                    packed=hip_ops.pack_rows(tu, ti))
        for lo, hi in ((100, n), (0, 100)):
            prep_pk.fill(lo, hi)
        torch.cuda.synchronize()
        assert torch.equal(neg_set, neg)
        assert all(torch.equal(a, b) for a, b in zip(prep_set.cols, prep.cols)) and torch.equal(prep_set.order, prep.order)
        assert prep_pk.packed is not None
        assert all(torch.equal(a, b) for a, b in zip(prep_pk.cols, prep.cols)) and torch.equal(prep_pk.order, prep.order)
        assert torch.equal(prep.cols[0], su) and torch.equal(prep.cols[1], si) and torch.equal(prep.cols[2], sn)
        assert torch.equal(prep.order, order)
        prep.check()
        ref_neg = oracle.sample_negatives_counter(users, nI, ptr, idx, 3407, epoch)
        perm = oracle.epoch_permutation(n, 3407, epoch)
        assert np.array_equal(prep.cols[2].cpu().numpy().astype(np.int64), ref_neg[perm])
        assert np.array_equal(prep.cols[0].cpu().numpy().astype(np.int64), users[perm])
    bad = tu.clone(); bad[17] = nU
    p2 = hip_ops.EpochPrep(bad, ti, nU, nI, tptr, tidx, 1, 0)
    p2.fill(0, n)
    with pytest.raises(IndexError):
        p2.check()


def test_pipelined_preparation_trains_the_same_epoch():
    """PipelinedSgd with an EpochPrep (rows of plan chunk c+1 produced beside the steps of chunk c) = the same epoch with the
    columns prepared up front: tables and losses bitwise equal"""
    from whisprrec_amd import hip_ops
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(4)
    nU, nI, D, B, nb = 5000, 4000, 64, 2048, 13
    n = nb * B - 77
    users = rng.randint(0, nU, n).astype(np.int32); items = rng.randint(0, nI, n).astype(np.int32)
    ptr, idx = hip_ops.clicked_csr_from_pairs(torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev), nU, nI)
    U = (rng.standard_normal((nU, D)) * 0.1).astype(np.float32); I = (rng.standard_normal((nI, D)) * 0.1).astype(np.float32)
    tu, ti = torch.from_numpy(users).to(dev), torch.from_numpy(items).to(dev)
    outs = []
    for pipelined in (False, True):
        Ud, Id = torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev)
        prep = hip_ops.EpochPrep(tu, ti, nU, nI, ptr, idx, 11, 2)
        if not pipelined:
            prep.fill(0, n)
        pipe = hip_ops.PipelinedSgd(chunk=3, min_triplets=1)
        h = pipe.plan(Ud, [(Id, prep.cols[0], prep.cols[1], prep.cols[2])], B, prep=prep if pipelined else None)
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        pipe.run(h, 0, 0.05, losses)
        torch.cuda.synchronize()
        prep.check()
        assert prep.filled == n
        outs.append((Ud, Id, losses))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
