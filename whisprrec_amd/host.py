"""Host-side mirror of the reference's model/dataset contract for the embedding-CF path.

The reference resolves Reader / Model / Runner classes by name and duck-types them (reference src/main.py:104-122,
SURVEY.md §8b).  This module restates, from the contract, the pieces a model on this path needs so that the
package is usable without the reference tree (the GPU box has no copy of it):

  BaseModel      <- src/models/BaseModel.py:18-66     flags, device/model_path/buffer, save/load, count_variables
  BaseModel.Dataset <- :68-127                        per-sample feed dicts, collate_batch
  GeneralModel   <- :130-177                          --num_neg/--test_all, (user, pos, neg) feed dict, sampler
  Corpus         <- what models read from a Reader    n_users, n_items, data_df, train/residual clicked sets

When the package is dropped into the reference tree the reference's own ``GeneralModel`` is used instead (see
``bprmf.bind`` and INTEGRATION.md), so these classes and the reference's never coexist in one hierarchy.
"""
import logging
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset as TorchDataset


class Corpus:
    """What a model consumes from a Reader (reference src/helpers/BaseReader.py:24-46,85-86): interaction frames per
    phase plus id ranges and per-user clicked sets.  ``from_arrays`` builds one from plain arrays."""

    def __init__(self, n_users, n_items, data, train_clicked_set=None, residual_clicked_set=None):
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.data_df = data  # phase -> dict of equal-length arrays (user_id, item_id, ...), or DataFrames
        self.train_clicked_set = train_clicked_set if train_clicked_set is not None else {}
        self.residual_clicked_set = residual_clicked_set if residual_clicked_set is not None else {}

    @classmethod
    def from_arrays(cls, n_users, n_items, train, dev=None, test=None):
        """train/dev/test: (user_ids, item_ids) pairs.  Clicked sets follow BaseReader.py:35-46."""
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int64))
        frames, tcs, rcs = {}, {}, {}
        for phase, pair in (("train", train), ("dev", dev or empty), ("test", test or empty)):
            uu, ii = np.asarray(pair[0]), np.asarray(pair[1])
            frames[phase] = {"user_id": uu, "item_id": ii}
            for a, b in zip(uu.tolist(), ii.tolist()):
                if a not in tcs:
                    tcs[a], rcs[a] = set(), set()
                (tcs if phase == "train" else rcs)[a].add(b)
        return cls(n_users, n_items, frames, tcs, rcs)


def _frame_to_arrays(frame):
    """DataFrame or dict -> dict of NumPy arrays (reference utils.df_to_dict, src/utils/utils.py:26-30)."""
    cols = frame.to_dict("list") if hasattr(frame, "to_dict") else dict(frame)
    return {k: np.asarray(v) for k, v in cols.items()}


class BaseModel(nn.Module):
    reader, runner = None, None
    extra_log_args = []

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument("--model_path", type=str, default="", help="Model save path.")
        parser.add_argument("--buffer", type=int, default=1, help="Whether to buffer feed dicts for dev/test")
        return parser

    def __init__(self, args, corpus):
        super().__init__()
        self.device = args.device
        self.model_path = args.model_path
        self.buffer = args.buffer
        self.optimizer = None  # a runner builds one only while this is None (reference BaseRunner.py:182-183)
        self.check_list = []

    def save_model(self, model_path=None):
        path = self.model_path if model_path is None else model_path
        folder = os.path.dirname(path)
        if folder and not os.path.exists(folder):
            os.makedirs(folder)
        torch.save(self.state_dict(), path)

    def load_model(self, model_path=None):
        path = self.model_path if model_path is None else model_path
        self.load_state_dict(torch.load(path, map_location=self.device))
        logging.info("Load model from " + path)

    def count_variables(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def actions_after_train(self):
        pass

    class Dataset(TorchDataset):
        def __init__(self, model, corpus, phase):
            self.model, self.corpus, self.phase = model, corpus, phase
            self.buffer_dict = {}
            self.data = _frame_to_arrays(corpus.data_df[phase])

        def __len__(self):
            for v in self.data.values():
                return len(v)
            return 0

        def __getitem__(self, index):
            return self._get_feed_dict(index)

        def _get_feed_dict(self, index):
            raise NotImplementedError

        def actions_before_epoch(self):
            pass

        def collate_batch(self, feed_dicts):
            """List of per-sample dicts -> one dict of tensors (int64 for ids, reference BaseModel.py:96-127):
            equal-length values are stacked, ragged NumPy values are right-padded with 0."""
            out = {}
            for key in feed_dicts[0]:
                vals = [d[key] for d in feed_dicts]
                sizes = {len(v) if isinstance(v, (list, np.ndarray)) else 1 for v in vals}
                if len(sizes) > 1:
                    out[key] = pad_sequence([torch.from_numpy(np.asarray(v)) for v in vals], batch_first=True)
                else:
                    out[key] = torch.from_numpy(np.array(vals))
            out["batch_size"] = len(feed_dicts)
            out["phase"] = self.phase
            return out


def clicked_keys(clicked_sets, n_items):
    """sorted int64 keys user * n_items + item of every (user, train item) pair: one vectorised membership test instead of a
    Python ``in`` per row"""
    parts = [np.fromiter(items, dtype=np.int64, count=len(items)) + int(uu) * int(n_items)
             for uu, items in clicked_sets.items() if len(items)]
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.int64)


def sample_negatives(user_ids, n_items, clicked_sets, num_neg=1, rng=np.random, keys=None):
    """One negative per training row, NumPy legacy global stream, bit-identical to the reference
    (src/models/BaseModel.py:167-177): a single vectorised randint(1, n_items) draw, then row by row scalar redraws
    while the candidate is in the user's train set.  Item 0 is never drawn.

    With ``keys`` (clicked_keys) and one negative per row, the rows whose first draw is rejected are found in one vectorised
    pass and only those are visited — in ascending order, with the same redraw loop, so the stream is consumed exactly as
    the row-by-row loop consumes it."""
    neg = rng.randint(1, n_items, size=(len(user_ids), num_neg))
    if keys is not None and num_neg == 1:
        cand = np.asarray(user_ids, dtype=np.int64) * int(n_items) + neg[:, 0]
        pos = np.searchsorted(keys, cand)
        hit = (pos < keys.size) & (keys[np.minimum(pos, max(keys.size - 1, 0))] == cand) if keys.size else np.zeros(len(cand), bool)
        for i in np.flatnonzero(hit).tolist():
            clicked = clicked_sets[user_ids[i]]
            while neg[i][0] in clicked:
                neg[i][0] = rng.randint(1, n_items)
        return neg.reshape(-1)
    for i, uu in enumerate(user_ids):
        clicked = clicked_sets[uu]
        for j in range(num_neg):
            while neg[i][j] in clicked:
                neg[i][j] = rng.randint(1, n_items)
    return neg.reshape(-1)


class GeneralModel(BaseModel):
    reader, runner = "BaseReader", "BaseRunner"

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument("--num_neg", type=int, default=1, help="The number of negative items during training.")
        parser.add_argument("--test_all", type=int, default=1, help="Whether testing on all the items.")
        return BaseModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.user_num = int(corpus.n_users)
        self.item_num = int(corpus.n_items)
        self.num_neg = args.num_neg
        self.test_all = args.test_all

    class Dataset(BaseModel.Dataset):
        def _get_feed_dict(self, index):
            if self.phase != "train" and self.model.test_all:
                neg = np.arange(1, self.corpus.n_items)
            else:
                neg = self.data["neg_items"][index]
            return {"user_id": self.data["user_id"][index], "pos_item": self.data["item_id"][index], "neg_items": neg}

        def actions_before_epoch(self):
            if getattr(self, "_clicked_keys", None) is None:
                self._clicked_keys = clicked_keys(self.corpus.train_clicked_set, self.corpus.n_items)
            self.data["neg_items"] = sample_negatives(self.data["user_id"], self.corpus.n_items,
                                                      self.corpus.train_clicked_set, self.model.num_neg,
                                                      keys=self._clicked_keys)


class SequentialModel(GeneralModel):
    """History-aware variant (reference src/models/BaseModel.py:181-213): ``--history_max``; samples with an empty
    history are dropped; each feed dict carries the user's last ``history_max`` (item, time) pairs before the target."""
    reader = "SeqReader"

    @staticmethod
    def parse_model_args(parser):
        parser.add_argument("--history_max", type=int, default=20, help="Maximum length of history.")
        return GeneralModel.parse_model_args(parser)

    def __init__(self, args, corpus):
        super().__init__(args, corpus)
        self.history_max = args.history_max

    class Dataset(GeneralModel.Dataset):
        def __init__(self, model, corpus, phase):
            super().__init__(model, corpus, phase)
            keep = np.asarray(self.data["position"]) > 0
            self.data = {k: np.asarray(v)[keep] for k, v in self.data.items()}

        def _get_feed_dict(self, index):
            fd = super()._get_feed_dict(index)
            seq = self.corpus.user_his[fd["user_id"]][:self.data["position"][index]]
            if self.model.history_max > 0:
                seq = seq[-self.model.history_max:]
            fd["history_items"] = np.array([x[0] for x in seq])
            fd["history_times"] = np.array([x[1] for x in seq])
            fd["lengths"] = len(fd["history_items"])
            return fd
