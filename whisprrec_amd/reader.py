"""Reader for the reference's interaction files — the data format on the input side of the path (reference
src/helpers/BaseReader.py, src/utils/sample.py).  Host-side, one-off per run; plain NumPy (no DataFrame round trips).

``<path>/<dataset>/<dataset>.inter``: TSV with a header naming ``user_id:token item_id:token rating:float timestamp:float``
(any order).  Processing = the reference's, quirks included:
  * users with fewer than 20 rows are dropped BEFORE the rating filter (sample.py:34-36);
  * ml-* datasets keep rating >= 3, yelp / food keep rating >= 4, others keep everything (sample.py:38-45);
  * user and item ids are renumbered in order of first appearance in the filtered file (sample.py:47-51);
  * ``--sample random`` (default) = two ``sklearn.model_selection.train_test_split`` calls with ``random_state=42``
    (0.8 / then 0.5, sample.py:116-141) — independent of ``--random_seed``; restated here with the same NumPy stream;
  * anything else = leave-one-out in FILE order: a user's first row goes to train, the last to test, the last but one to
    dev (sample.py:85-102);
  * ``n_users`` / ``n_items`` = max id + 1 of the filtered file; per-user clicked sets as BaseReader.py:35-46.
Row order inside each split is the reference's (it fixes the sampler and shuffle streams of BaseRunner.fit).
"""
import logging
import os

import numpy as np

from .host import Corpus

_COLUMNS = {"user_id:token": "user_id", "item_id:token": "item_id", "rating:float": "rating", "timestamp:float": "timestamp"}


def read_inter(path, sep="\t"):
    """-> dict column -> float64/int64 array, in file order.  The id columns are TOKENS (the reference reads them with
    pandas and renumbers whatever they are, sample.py:47-51): integer literals are taken as they stand, anything else
    (alphanumeric ids of the yelp / food files, integers beyond 2^63) is numbered by first appearance here — the
    renumbering after the filters (count_statics) then yields the same ids as renumbering the strings would."""
    with open(path, "r") as f:
        header = f.readline().rstrip("\n").split(sep)
    names = [_COLUMNS.get(h, h) for h in header]
    id_cols = [j for j, nm in enumerate(names) if nm in ("user_id", "item_id")]
    num_cols = [j for j in range(len(names)) if j not in id_cols]
    out = {}
    if id_cols:
        tok = np.loadtxt(path, delimiter=sep, skiprows=1, dtype=str, usecols=id_cols, ndmin=2, comments=None)
        for k, j in enumerate(id_cols):
            col = tok[:, k]
            try:
                out[names[j]] = col.astype(np.int64)
            except (ValueError, OverflowError):
                out[names[j]] = _first_appearance_ids(col)
    if num_cols:
        raw = np.loadtxt(path, delimiter=sep, skiprows=1, dtype=np.float64, usecols=num_cols, ndmin=2)
        for k, j in enumerate(num_cols):
            col = raw[:, k]
            out[names[j]] = col.astype(np.int64) if np.all(col == np.floor(col)) else col
    return {nm: out[nm] for nm in names}


def _first_appearance_ids(a):
    """pandas ``unique()`` order: ids numbered by first appearance"""
    uniq, first = np.unique(a, return_index=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty(uniq.size, np.int64)
    rank[order] = np.arange(uniq.size)
    return rank[np.searchsorted(uniq, a)]


def count_statics(cols, dataset):
    """sample.count_statics (sample.py:17-72): filters + id renumbering; returns the filtered columns"""
    users = cols["user_id"]
    uniq, cnt = np.unique(users, return_counts=True)
    keep = np.isin(users, uniq[cnt >= 20])
    if dataset in ("ml-1m", "ml-100k", "ml-10m"):
        keep &= cols["rating"] >= 3
    elif dataset in ("yelp", "food"):
        keep &= cols["rating"] >= 4
    out = {k: v[keep] for k, v in cols.items()}
    if dataset in ("ml-1m", "ml-100k", "ml-10m", "yelp", "food"):
        out.pop("rating", None)
    out["user_id"] = _first_appearance_ids(out["user_id"])
    out["item_id"] = _first_appearance_ids(out["item_id"])
    n = len(out["user_id"])
    logging.info("# Users: %d  # Items: %d  # Interactions: %d", out["user_id"].max() + 1 if n else 0,
                 out["item_id"].max() + 1 if n else 0, n)
    return out


def _shuffle_split(n, train_size, seed):
    """sklearn ShuffleSplit as train_test_split(train_size=<float>, random_state=seed, shuffle=True) draws it:
    n_train = floor(train_size * n), the rest is the test side; test = first n_test of one permutation, train = the next."""
    n_train = int(np.floor(train_size * n))
    n_test = n - n_train
    perm = np.random.RandomState(seed).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]


def random_split(n_rows, ratios=(0.8, 0.1, 0.1)):
    """row indices (train, dev, test) of sample.random_split (sample.py:116-151)"""
    train, rest = _shuffle_split(n_rows, ratios[0], 42)
    dev_of_rest, test_of_rest = _shuffle_split(rest.size, ratios[1] / (ratios[1] + ratios[2]), 42)
    return train, rest[dev_of_rest], rest[test_of_rest]


def leave_one_out_split(users):
    """row indices (train, dev, test) of sample.leave_one_out_split (sample.py:75-113), file order"""
    n = users.size
    idx = np.arange(n)
    order = np.argsort(users, kind="stable")
    su = users[order]
    starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]])
    ends = np.r_[starts[1:], n]
    first = order[starts]                                   # head(1) per user
    rem = np.ones(n, bool); rem[first] = False
    # tail(1) per user among the remaining rows, twice
    def tails(mask):
        o = order[mask[order]]
        u = users[o]
        last = np.flatnonzero(np.r_[u[1:] != u[:-1], True]) if o.size else np.zeros(0, np.int64)
        return o[last]
    test = tails(rem); rem[test] = False
    dev = tails(rem); rem[dev] = False
    train = np.sort(np.r_[first, idx[rem]])
    return train, np.sort(dev), np.sort(test)


class BaseReader:
    """Same flags and attributes as the reference class (BaseReader.py:14-86): ``data_df`` (phase -> dict of arrays),
    ``n_users``, ``n_items``, ``train_clicked_set``, ``residual_clicked_set``; ``corpus()`` wraps them for the models."""

    @staticmethod
    def parse_reader_args(parser):
        parser.add_argument("--path", type=str, default="../data/", help="Input data dir.")
        parser.add_argument("--dataset", type=str, default="ml-100k", help="Choose a dataset.")
        parser.add_argument("--sep", type=str, default="\t", help="sep of csv file.")
        parser.add_argument("--sample", type=str, default="random", help="random or leave one out")
        return parser

    def __init__(self, args):
        self.sep, self.prefix, self.dataset, self.sample = args.sep, args.path, args.dataset, args.sample
        path = os.path.join(self.prefix, self.dataset, self.dataset + ".inter")
        if not os.path.exists(path):
            raise FileNotFoundError("Interactions file not found: %s" % path)
        self.all_df = count_statics(read_inter(path, "\t"), self.dataset)       # the reference reads with '\\t' whatever --sep says
        n = self.all_df["user_id"].size
        tr, dv, te = random_split(n) if self.sample == "random" else leave_one_out_split(self.all_df["user_id"])
        self.data_df = {ph: {k: v[ix] for k, v in self.all_df.items()} for ph, ix in (("train", tr), ("dev", dv), ("test", te))}
        logging.info("Dataset has been split. Train dataset length: %d, Dev dataset length: %d, Test dataset length: %d",
                     tr.size, dv.size, te.size)
        self.n_users = int(self.all_df["user_id"].max()) + 1
        self.n_items = int(self.all_df["item_id"].max()) + 1
        self.train_clicked_set, self.residual_clicked_set = {}, {}
        for key in ("train", "dev", "test"):
            df = self.data_df[key]
            for uid, iid in zip(df["user_id"].tolist(), df["item_id"].tolist()):
                if uid not in self.train_clicked_set:
                    self.train_clicked_set[uid] = set()
                    self.residual_clicked_set[uid] = set()
                (self.train_clicked_set if key == "train" else self.residual_clicked_set)[uid].add(iid)

    def corpus(self):
        return Corpus(self.n_users, self.n_items, self.data_df, self.train_clicked_set, self.residual_clicked_set)


class SeqReader(BaseReader):
    """reference src/helpers/SeqReader.py: every interaction's position in its user's time-ordered history (stable sort by
    user, then timestamp) and ``user_his`` = user -> [(item, time), ...]; the splits gain a ``position`` column.  The
    reference attaches it with a left merge on (user, item, timestamp): a split row whose key occurs k times in the file
    becomes k rows — reproduced."""

    def __init__(self, args):
        super().__init__(args)
        users, items, times = self.all_df["user_id"], self.all_df["item_id"], self.all_df["timestamp"]
        order = np.lexsort((np.arange(users.size), times, users))          # mergesort by (user_id, timestamp): stable
        su, si, st = users[order], items[order], times[order]
        starts = np.flatnonzero(np.r_[True, su[1:] != su[:-1]]) if su.size else np.zeros(0, np.int64)
        pos = np.arange(su.size) - np.repeat(starts, np.diff(np.r_[starts, su.size]))
        self.user_his = {}
        for a, b, t in zip(su.tolist(), si.tolist(), st.tolist()):
            self.user_his.setdefault(a, []).append((b, t))
        by_key = {}
        for a, b, t, q in zip(su.tolist(), si.tolist(), st.tolist(), pos.tolist()):
            by_key.setdefault((a, b, t), []).append(q)
        for key in ("train", "dev", "test"):
            df = self.data_df[key]
            rows, positions = [], []
            for r, k in enumerate(zip(df["user_id"].tolist(), df["item_id"].tolist(), df["timestamp"].tolist())):
                for q in by_key[k]:
                    rows.append(r); positions.append(q)
            rows = np.asarray(rows, np.int64)
            self.data_df[key] = {c: v[rows] for c, v in df.items()}
            self.data_df[key]["position"] = np.asarray(positions, np.int64)

    def corpus(self):
        c = super().corpus()
        c.user_his = self.user_his
        return c
