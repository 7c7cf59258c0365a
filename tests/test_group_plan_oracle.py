"""CPU checks of oracle.group_plan — the NumPy restatement the GPU tests compare the device's group plan with
(tests/test_hip_group.py): its flags and lists against a brute-force count of the rows' occurrences per batch."""
from collections import Counter

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("nU,nI,B,N", [(50, 40, 64, 200), (300_000, 600_000, 128, 300), (3_000_000, 5_000_000, 256, 512)])
def test_flags_and_lists_against_brute_force(nU, nI, B, N):
    rng = np.random.RandomState(nU % 97)
    u = rng.randint(0, min(nU, 60), N) * (nU // min(nU, 60))           # few distinct rows: plenty of sharing, ids spread over the ranges
    p = rng.randint(0, min(nI, 50), N) * (nI // min(nI, 50))
    n = rng.randint(0, min(nI, 50), N) * (nI // min(nI, 50))
    g = oracle.group_plan(u, p, n, B, nU, nI)
    nb = (N + B - 1) // B
    bit = lambda b, kind, t: (int(g["flags"][b, t >> 5, kind]) >> (t & 31)) & 1
    prev_rows = None
    for b in range(nb):
        sl = slice(b * B, min(N, (b + 1) * B))
        ub, pb, nbk = u[sl], p[sl], n[sl]
        cu, ci = Counter(ub.tolist()), Counter(pb.tolist() + nbk.tolist())
        for t in range(ub.size):
            assert bit(b, 0, t) == (cu[int(ub[t])] > 1)
            assert bit(b, 1, t) == (ci[int(pb[t])] > 1) and bit(b, 2, t) == (ci[int(nbk[t])] > 1)
            if prev_rows is not None:
                want = int(ub[t]) in prev_rows[0] or int(pb[t]) in prev_rows[1] or int(nbk[t]) in prev_rows[1]
                assert bit(b, 3, t) == want
            else:
                assert bit(b, 3, t) == 0
        prev_rows = ({r for r, c in cu.items() if c > 1}, {r for r, c in ci.items() if c > 1})
        # lists: exactly the shared occurrences, equal rows adjacent and in source order
        rows = np.concatenate([g["users"][(b, r)][0] for r in range(g["R_u"])])
        src = np.concatenate([g["users"][(b, r)][1] for r in range(g["R_u"])])
        assert sorted(zip(rows.tolist(), src.tolist())) == sorted((int(ub[t]), t << 1) for t in range(ub.size) if cu[int(ub[t])] > 1)
        for r in range(g["R_i"]):
            rows, src = g["items"][(b, r)]
            for a in range(1, rows.size):
                assert rows[a] != rows[a - 1] or src[a] > src[a - 1]
            seen = set()
            for a in range(rows.size):
                if a and rows[a] != rows[a - 1]:
                    assert int(rows[a]) not in seen          # a row's occurrences form ONE run
                seen.add(int(rows[a]))
        irows = np.concatenate([g["items"][(b, r)][0] for r in range(g["R_i"])])
        isrc = np.concatenate([g["items"][(b, r)][1] for r in range(g["R_i"])])
        want = sorted([(int(pb[t]), t << 1) for t in range(pb.size) if ci[int(pb[t])] > 1] +
                      [(int(nbk[t]), (t << 1) | 1) for t in range(pb.size) if ci[int(nbk[t])] > 1])
        assert sorted(zip(irows.tolist(), isrc.tolist())) == want
