"""hip_ops.BucketMap (struct wr_bucket_side, include/whisprrec_hip.h) — the load-balanced row ranges the hand-written plan
builder takes on skewed ids.  The map is plain array arithmetic (bincount / cumsum), so its invariants are checked on the
CPU: buckets ascend with the rows and tile the row space, heavy rows own their sub-buckets, expected loads stay within
the builder's fixed bucket capacity.  (The builder itself with a map: tests/test_hip_bprmf.py, GPU.)"""
import numpy as np
import pytest
import torch

from whisprrec_amd import hip_ops


def _epoch(rng, nU, nI, N, a_u, a_i):
    u = np.minimum((rng.pareto(a_u, N) * 20).astype(np.int64), nU - 1)
    p = np.minimum((rng.pareto(a_i, N) * 5).astype(np.int64), nI - 1)
    return u, p


@pytest.mark.parametrize("case", [(5000, 3000, 2048, 200000, 1.1, 0.9), (100000, 100000, 16384, 400000, 1.5, 1.0),
                                  (943, 1574, 2048, 66016, 3.0, 1.2), (1 << 18, 1 << 18, 65536, 1 << 20, 50.0, 0.8)])
def test_bucket_map_invariants(case):
    nU, nI, B, N, a_u, a_i = case
    rng = np.random.RandomState(N)
    u, p = _epoch(rng, nU, nI, N, a_u, a_i)
    m = hip_ops.BucketMap(torch.from_numpy(u), torch.from_numpy(p), nU, nI, B)
    lam = {"users": np.bincount(u, minlength=nU) * (B / N),
           "items": np.bincount(p, minlength=nI) * (B / N) + np.r_[0.0, np.full(nI - 1, B / (nI - 1))]}
    for name, side, n_rows, npos in (("users", m.users, nU, B), ("items", m.items, nI, 2 * B)):
        assert side is not None, name
        rb, start, rows, sub = (a.numpy().astype(np.int64) for a in side["arrays"])
        nb = side["n_buckets"]
        assert 1 <= nb <= 1024 and len(rb) == n_rows and len(start) == len(rows) == len(sub) == nb
        first, nsub = rb & 0xffff, rb >> 16
        assert (np.diff(first) >= 0).all() and first[0] == 0                     # buckets ascend with the rows
        mean = npos / 256
        cap = 2 * mean + 64                                                        # the builder's fixed capacity
        covered = np.zeros(n_rows, np.int64)
        for k in range(nb):
            if sub[k] == 0:                                                        # ordinary bucket: a run of consecutive rows
                r = np.arange(start[k], start[k] + rows[k])
                assert (first[r] == k).all() and (nsub[r] == 0).all()
                covered[r] += 1
                load = lam[name][r].sum()
                assert load <= 1.5 * mean + 1e-6 and load + 5 * np.sqrt(load) <= cap
                assert lam[name][r].max() <= 96 + 1e-9                             # no row outgrows a bin of the bucket sort
            else:                                                                  # sub-bucket q of the heavy row start[k]
                q, msub = sub[k] & 0xffff, sub[k] >> 16
                h = start[k]
                assert rows[k] == 1 and nsub[h] == msub and first[h] + q == k and q < msub
                covered[h] += (q == 0)
                share = lam[name][h] / msub
                assert share <= mean + 1e-6 and share + 5 * np.sqrt(share) <= cap
        assert (covered == 1).all()                                                # every row in exactly one (first) bucket
        assert nb == first[-1] + max(nsub[-1], 1)


def test_bucket_map_gives_up_beyond_1024_buckets():
    """tables smaller than the batch: most rows are heavy, more than 1024 buckets would be needed -> no map (radix-sort builder)"""
    rng = np.random.RandomState(0)
    nU = nI = 3000
    B, N = 1 << 20, 1 << 22
    u = rng.randint(0, nU, N); p = rng.randint(0, nI, N)
    m = hip_ops.BucketMap(torch.from_numpy(u), torch.from_numpy(p), nU, nI, B)
    assert m.users is None and m.items is None


def test_bucket_map_uniform_ids_reduce_to_equal_ranges():
    rng = np.random.RandomState(1)
    nU = nI = 1 << 16
    B, N = 4096, 1 << 21
    m = hip_ops.BucketMap(torch.from_numpy(rng.randint(0, nU, N)), torch.from_numpy(rng.randint(0, nI, N)), nU, nI, B)
    for side in (m.users, m.items):
        rows = side["arrays"][2].numpy()
        assert 250 <= side["n_buckets"] <= 262 and rows.max() <= 1.2 * nU / 256 and (side["arrays"][3].numpy() == 0).all()
