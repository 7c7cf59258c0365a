"""Runner for the HIP models behind the reference's runner contract (reference src/helpers/BaseRunner.py).

``BaseRunner`` restates the reference runner's surface (flags, ``train``/``fit``/``evaluate``/``print_res``,
``evaluate_method``) so the package runs stand-alone; ``HipRunner`` replaces the inner loop of ``fit``
(BaseRunner.py:180-201) by device-side epoch preparation plus the native multi-step driver, keeping:
  * negative sampling = the dataset's own ``actions_before_epoch`` (bit-exact NumPy stream);
  * batch composition = exactly what ``DataLoader(shuffle=True)`` would yield (same torch RNG draws);
  * return value = unweighted mean of per-batch mean losses, short last batch included (BaseRunner.py:200-201).
Inside the reference tree use ``bind_runner(BaseRunner)`` to inherit everything else from it.
"""
import gc
import logging
import os
from time import time

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import abi


def format_metric(result_dict):
    """'NDCG@10:0.1085,HR@10:0.2254'-style line (reference src/utils/utils.py:58-71, without its NumPy-2 breakage)."""
    names = sorted({k.split("@")[0] for k in result_dict})
    topks = sorted({int(k.split("@")[1]) for k in result_dict})
    parts = []
    for k in topks:
        for m in names:
            key = "{}@{}".format(m, k)
            v = result_dict[key]
            parts.append("{}:{:<.4f}".format(key, v) if isinstance(v, (float, np.floating)) else "{}:{}".format(key, v))
    return ",".join(parts)


def _epoch_order_via_loader(n, batch_size):
    dl = DataLoader(_Rows(n), batch_size=batch_size, shuffle=True, num_workers=0)
    it = iter(dl)
    sampler_iter = getattr(it, "_sampler_iter", None)
    if sampler_iter is None:                                   # unknown DataLoader internals: fetch through the loader
        return torch.cat([b for b in it])
    return torch.tensor([i for batch in sampler_iter for i in batch], dtype=torch.int64)


def _epoch_order_direct(n, batch_size):
    """what the loader machinery does to the global torch RNG, without its per-index Python loops: the iterator draws a base
    seed, RandomSampler draws its own seed and permutes with a private generator (torch/utils/data/{dataloader,sampler}.py)"""
    torch.empty((), dtype=torch.int64).random_()               # _BaseDataLoaderIter: base seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


_DIRECT_OK = None


def epoch_order(n, batch_size, num_workers=0):
    """The row order DataLoader(dataset, batch_size, shuffle=True) visits (reference BaseRunner.py:188-193), with the
    global torch RNG consumed identically.  The direct restatement is checked once per process against the loader
    machinery itself (order AND resulting RNG state); if this torch version behaves differently the machinery is used."""
    global _DIRECT_OK
    if _DIRECT_OK is None:
        state = torch.get_rng_state()
        a = _epoch_order_via_loader(257, 32); sa = torch.get_rng_state()
        torch.set_rng_state(state)
        b = _epoch_order_direct(257, 32); sb = torch.get_rng_state()
        torch.set_rng_state(state)
        _DIRECT_OK = bool(torch.equal(a, b) and torch.equal(sa, sb))
    return _epoch_order_direct(n, batch_size) if _DIRECT_OK else _epoch_order_via_loader(n, batch_size)


def consume_loader_seed():
    """Every DataLoader iterator draws a base seed from the global torch RNG when it is created — the reference's evaluation
    loop too (BaseRunner.py:229-235), between two training epochs.  Code that replaces such a loop by array work draws the
    same number, so the next epoch's shuffle sees the stream where the reference's would."""
    torch.empty((), dtype=torch.int64).random_()


class _Rows:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class BaseRunner(object):
    @staticmethod
    def parse_runner_args(parser):
        parser.add_argument("--epoch", type=int, default=200, help="Number of epochs.")
        parser.add_argument("--check_epoch", type=int, default=1, help="Check some tensors every check_epoch.")
        parser.add_argument("--test_epoch", type=int, default=-1, help="Print test results every test_epoch (-1: never).")
        parser.add_argument("--early_stop", type=int, default=10, help="Epochs of non-improving dev results before stopping.")
        parser.add_argument("--lr", type=float, default=5e-4, help="Learning rate.")
        parser.add_argument("--l2", type=float, default=0, help="Weight decay in optimizer.")
        parser.add_argument("--batch_size", type=int, default=2048, help="Batch size during training.")
        parser.add_argument("--eval_batch_size", type=int, default=2048, help="Batch size during testing.")
        parser.add_argument("--optimizer", type=str, default="Adam", help="optimizer: SGD, Adam, Adagrad, Adadelta")
        parser.add_argument("--num_workers", type=int, default=5, help="DataLoader workers (unused by HipRunner.fit).")
        parser.add_argument("--pin_memory", type=int, default=0, help="pin_memory in DataLoader")
        parser.add_argument("--topk", type=str, default="10,20", help="The number of items recommended to each user.")
        parser.add_argument("--metric", type=str, default="NDCG, HR", help="metrics: NDCG, HR, RECALL, PRECISION")
        return parser

    @staticmethod
    def evaluate_method(predictions, topk, metrics):
        """predictions[:, 0] is the ground-truth score; rank = its position in the descending order of the row
        (reference BaseRunner.py:50-92)."""
        order = (-predictions).argsort(axis=1)
        gt_rank = np.argwhere(order == 0)[:, 1] + 1
        return BaseRunner.metrics_from_ranks(gt_rank, topk, metrics)

    @staticmethod
    def metrics_from_ranks(gt_rank, topk, metrics):
        out = {}
        for k in topk:
            hit = gt_rank <= k
            for m in metrics:
                key = "{}@{}".format(m, k)
                name = m.lower()
                if name in ("hr", "recall"):
                    out[key] = hit.mean()
                elif name == "ndcg":
                    out[key] = np.mean(hit / np.log2(gt_rank + 1))
                elif name == "precision":
                    out[key] = hit.sum() / (hit.shape[0] * k)
                else:
                    raise ValueError("Undefined evaluation metric: {}.".format(m))
        return out

    def __init__(self, args):
        self.epoch = args.epoch
        self.check_epoch = args.check_epoch
        self.test_epoch = args.test_epoch
        self.early_stop = args.early_stop
        self.learning_rate = float(args.lr)
        self.batch_size = args.batch_size
        self.eval_batch_size = args.eval_batch_size
        self.l2 = args.l2
        self.optimizer_name = args.optimizer
        self.num_workers = args.num_workers
        self.pin_memory = args.pin_memory
        self.topk = [int(x) for x in args.topk.split(",")]
        self.metrics = [m.strip().upper() for m in args.metric.split(",")]
        self.main_metric = "{}@{}".format(self.metrics[0], self.topk[0])
        self.time = None

    def _check_time(self, start=False):
        if self.time is None or start:
            self.time = [time()] * 2
            return self.time[0]
        last = self.time[1]
        self.time[1] = time()
        return self.time[1] - last

    def _build_optimizer(self, model):
        logging.info("Optimizer: " + self.optimizer_name)
        return getattr(torch.optim, self.optimizer_name)(model.parameters(), lr=self.learning_rate, weight_decay=self.l2)

    def eval_termination(self, criterion):
        recent = criterion[-self.early_stop:]
        if len(criterion) > self.early_stop and all(a >= b for a, b in zip(recent, recent[1:])):
            return True
        return len(criterion) - criterion.index(max(criterion)) > self.early_stop

    def train(self, data_dict):
        model = data_dict["train"].model
        main_results, dev_results = [], []
        self._check_time(start=True)
        for epoch in range(self.epoch):
            self._check_time()
            gc.collect()
            loss = self.fit(data_dict["train"], epoch=epoch + 1)
            train_t = self._check_time()
            dev = self.evaluate(data_dict["dev"], self.topk[:1], self.metrics)
            dev_results.append(dev)
            main_results.append(dev[self.main_metric])
            line = "Epoch {:<5} loss={:<.4f} [{:<3.1f} s]    dev=({})".format(epoch + 1, loss, train_t, format_metric(dev))
            if self.test_epoch > 0 and epoch % self.test_epoch == 0:
                line += " test=({})".format(format_metric(self.evaluate(data_dict["test"], self.topk[:1], self.metrics)))
            line += " [{:<.1f} s]".format(self._check_time())
            if max(main_results) == main_results[-1]:
                model.save_model()
                line += " *"
            logging.info(line)
            if self.early_stop > 0 and self.eval_termination(main_results):
                logging.info("Early stop at %d based on dev result." % (epoch + 1))
                break
        best = main_results.index(max(main_results))
        logging.info(os.linesep + "Best Iter(dev)={:>5}\t dev=({}) [{:<.1f} s] ".format(
            best + 1, format_metric(dev_results[best]), self.time[1] - self.time[0]))
        model.load_model()

    def fit(self, dataset, epoch=-1):
        """The reference loop (BaseRunner.py:180-201): DataLoader batches -> predict/backward/step."""
        model = dataset.model
        if model.optimizer is None:
            model.optimizer = self._build_optimizer(model)
        dataset.actions_before_epoch()
        model.train()
        losses = []
        dl = DataLoader(dataset, batch_size=self.batch_size, shuffle=True, num_workers=0, collate_fn=dataset.collate_batch)
        for batch in dl:
            batch = {k: (v.to(model.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            model.optimizer.zero_grad()
            loss = model.predict(batch)
            loss.backward()
            model.optimizer.step()
            losses.append(loss.detach().cpu().data.numpy())
        return np.mean(losses).item()

    def evaluate(self, dataset, topks, metrics):
        return self.evaluate_method(self.interface(dataset), topks, metrics)

    def interface(self, dataset):
        """[n_eval, 1 + n_items] matrix: column 0 = score of the ground-truth item, then all item scores with the
        user's train/dev/test items masked to -inf (reference BaseRunner.py:218-258)."""
        model = dataset.model
        model.eval()
        users = np.asarray(dataset.data["user_id"])
        items = np.asarray(dataset.data["item_id"])
        scores, targets = [], []
        with torch.no_grad():
            if "position" in dataset.data:   # sequential models: the batch needs the history fields the Dataset collates
                dl = DataLoader(dataset, batch_size=self.eval_batch_size, shuffle=False, num_workers=0,
                                collate_fn=dataset.collate_batch)
                for batch in dl:
                    batch = {k: (v.to(model.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
                    s = model.full_predict(batch)
                    pb = batch["pos_item"]
                    targets.append(s[torch.arange(len(pb), device=s.device), pb])
                    scores.append(s)
            else:
                consume_loader_seed()
                for lo in range(0, len(users), self.eval_batch_size):
                    ub = torch.from_numpy(users[lo:lo + self.eval_batch_size]).to(model.device)
                    pb = torch.from_numpy(items[lo:lo + self.eval_batch_size]).to(model.device)
                    s = model.full_predict({"user_id": ub, "pos_item": pb})
                    targets.append(s[torch.arange(len(pb), device=s.device), pb])
                    scores.append(s)
        target = torch.cat(targets).cpu().numpy()
        score = torch.cat(scores).cpu().numpy()
        if model.test_all:
            corpus = dataset.corpus
            for row, uid in enumerate(users):
                clicked = corpus.train_clicked_set[uid] | corpus.residual_clicked_set[uid]
                score[row, list(clicked)] = -np.inf
        return np.concatenate([target[:, None], score], axis=1)

    def print_res(self, dataset):
        return "(" + format_metric(self.evaluate(dataset, self.topk, self.metrics)) + ")"


def _metrics_from_ranks(gt_rank, topk, metrics):
    return BaseRunner.metrics_from_ranks(gt_rank, topk, metrics)


def make_hip_runner(base_runner_cls):
    class HipRunner(base_runner_cls):
        metrics_from_ranks = staticmethod(_metrics_from_ranks)   # the reference's BaseRunner has no such helper

        @staticmethod
        def parse_runner_args(parser):
            parser.add_argument("--device_epoch_prep", type=int, default=0,
                                help="1: sample negatives and shuffle on the device (counter-based generator; same rule "
                                     "as the reference sampler but not its NumPy stream). 0: reference streams, bit-exact.")
            parser.add_argument("--hip_graphs", type=int, default=1,
                                help="1: capture the training step of graph-capturable models (LightGCN) in a hipGraph.")
            return base_runner_cls.parse_runner_args(parser)

        def __init__(self, args):
            super().__init__(args)
            self.device_epoch_prep = int(getattr(args, "device_epoch_prep", 0))
            self.hip_graphs = int(getattr(args, "hip_graphs", 1))
            self.seed = int(getattr(args, "random_seed", 3407))
            self._epoch_cache = None

        def _device_epoch(self, dataset, dev, epoch, pipelined=False):
            """negatives (wr_sample_negatives) + shuffle, all on the device: no Python loop over rows"""
            from . import hip_ops
            model, corpus = dataset.model, dataset.corpus
            if self._epoch_cache is None or self._epoch_cache[0] is not dataset:
                users = torch.from_numpy(np.ascontiguousarray(dataset.data["user_id"])).to(torch.int64).to(dev)
                items = torch.from_numpy(np.ascontiguousarray(dataset.data["item_id"])).to(torch.int64).to(dev)
                ptr, idx = hip_ops.clicked_csr(corpus.train_clicked_set, model.user_num, dev)
                # big frames: membership tests through a hash set of the pairs (one sector per test instead of a binary search)
                pairs = hip_ops.pair_set(ptr, idx, model.user_num) if idx.numel() >= hip_ops.PAIR_SET_MIN_PAIRS else None
                # ... and the source rows as one word each: one random sector per output row instead of two
                packed = hip_ops.pack_rows(users, items) if pairs is not None else None
                self._epoch_cache = (dataset, users, items, ptr, idx, pairs, packed)
            _, users, items, ptr, idx, pairs, packed = self._epoch_cache
            # the epoch's rows in batch order: a keyed bijection of the rows (no 100 M-key sort as in torch.randperm) and the
            # negative of every row, fused and produced range by range (wr_epoch_prepare_range) — a model with a native epoch
            # loop gets the EpochPrep itself and fills each plan chunk's rows beside the previous chunk's steps
            sequential = "position" in dataset.data
            prep = hip_ops.EpochPrep(users, items, model.user_num, model.item_num, ptr, idx, self.seed, max(epoch, 0),
                                     want_order=sequential, pairs=pairs, packed=packed)
            self._epoch_prep = prep
            if pipelined:
                return prep
            prep.fill(0, prep.n)
            if sequential:
                self._last_order = prep.order
            prep.check()
            return prep.cols

        def evaluate(self, dataset, topks, metrics):
            """Full-ranking evaluation on the device when the model exposes its factor matrices (``eval_factors``): ranks
            from wr_rank_eval (MFMA score tiles + on-the-fly masking + counting), no [n_eval, n_items] matrix, no Python
            loop over rows (reference BaseRunner.py:218-258).  Otherwise the inherited host path."""
            model = dataset.model
            from . import hip_ops
            if not hasattr(model, "eval_factors") or not hip_ops.rank_eval_supports(model.eval_factors()[0].shape[1]):
                return base_runner_cls.evaluate(self, dataset, topks, metrics)
            from . import hip_ops
            consume_loader_seed()                       # the evaluation DataLoader this replaces would draw its base seed
            model.eval()
            user_mat, item_mat = model.eval_factors()
            dev = user_mat.device
            cache = getattr(self, "_mask_cache", None)
            if cache is None or cache[0] is not dataset.corpus or cache[1] != bool(model.test_all):
                if model.test_all:
                    corpus = dataset.corpus
                    merged = {u: corpus.train_clicked_set.get(u, set()) | corpus.residual_clicked_set.get(u, set())
                              for u in set(corpus.train_clicked_set) | set(corpus.residual_clicked_set)}
                    ptr, idx = hip_ops.clicked_csr(merged, user_mat.shape[0], dev)
                else:
                    ptr = idx = None
                cache = self._mask_cache = (dataset.corpus, bool(model.test_all), ptr, idx)
            eu = torch.from_numpy(np.ascontiguousarray(dataset.data["user_id"])).to(dev)
            et = torch.from_numpy(np.ascontiguousarray(dataset.data["item_id"])).to(dev)
            rank, _ = hip_ops.rank_eval(user_mat.contiguous(), item_mat.contiguous(), eu, et, cache[2], cache[3])
            return self.metrics_from_ranks(rank.cpu().numpy().astype(np.int64), topks, metrics)

        def _step_graph(self, model, cols, B, n, eager_step):
            """hipGraph of one whole training step (zero_grad / predict / backward / optimizer.step) of a model that declares
            itself ``graph_capturable`` — its step is ~60 small launches and host-bound, nothing in it depends on the host
            from step to step (indices arrive in static buffers, Adam's step number lives on the device).  The first three
            batches run eagerly on a side stream (they are ordinary training steps and size every scratch buffer), then the
            step is captured once per (model, batch size, graph view) and replayed.  -> (graph, static index tensors, static
            loss, first row not yet trained) or None."""
            if not getattr(model, "graph_capturable", False) or n < 8 * B or not hasattr(model.optimizer, "prepare"):
                return None
            key = (id(model), B, getattr(model, "graph_key", lambda: 0)())
            cached = getattr(self, "_graph_cache", None)
            if cached is not None and cached[0] == key:
                return cached[1], cached[2], cached[3], 0
            # the three eager batches are ordinary training steps of this epoch: outside the try, so that a failed capture
            # neither hides their errors nor lets the eager loop train them a second time
            dev = cols[0].device
            model.optimizer.prepare()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for i in range(3):
                    eager_step(i * B)
            torch.cuda.current_stream(dev).wait_stream(side)
            t_before = getattr(model.optimizer, "t", None)
            try:
                # one (3, B) buffer: a replay's indices arrive in one strided copy from the stacked epoch columns
                static3 = torch.empty((3, B), dtype=cols[0].dtype, device=dev)
                static = [static3[0], static3[1], static3[2]]
                batch = {"user_id": static[0], "pos_item": static[1], "neg_items": static[2].unsqueeze(1), "batch_size": B,
                         "phase": "train"}
                # the root gradient of backward(): made once here (autograd would fill a fresh one inside every replay)
                shape = getattr(self, "_loss_shape", None)
                root = torch.ones(shape, dtype=torch.float32, device=dev) if shape is not None else None
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g):
                        model.optimizer.zero_grad()
                        static_loss = model.predict(batch)
                        if root is not None and static_loss.shape == root.shape and static_loss.dtype == root.dtype:
                            model._unit_root = True      # the model's backward may skip its scale by the root gradient (ones)
                            static_loss.backward(root)
                        else:
                            static_loss.backward()
                        model.optimizer.step()
                finally:
                    model._unit_root = False
            except (RuntimeError, abi.WhisprRecHipError) as e:
                # capture is an optimisation: the epoch goes on eagerly FROM THE FIRST UNTRAINED ROW.  A capture that died
                # after the captured optimizer.step() call has bumped the host-side step count without training anything.
                logging.warning("hipGraph capture of the training step failed (%r); continuing without graphs", e)
                self.hip_graphs = 0
                torch.cuda.synchronize()
                if t_before is not None:
                    model.optimizer.t = t_before
                return None, None, None, 3 * B
            if hasattr(model.optimizer, "sync_step_count"):
                model.optimizer.t -= 1               # the captured step() call itself trained nothing
            self._graph_cache = (key, g, static, static_loss, root)      # root: kept alive for the replays
            return g, static, static_loss, 3 * B

        def _epoch_columns(self, dataset, dev, epoch):
            """the epoch's (user, positive, negative) columns in batch order on the device"""
            if self.device_epoch_prep:
                if hasattr(dataset.model, "graph_construction"):
                    dataset.model.graph_construction()  # SGL draws its views in actions_before_epoch (SGL.py:258-262)
                return self._device_epoch(dataset, dev, epoch)
            dataset.actions_before_epoch()              # must happen before the shuffle draws, as in the reference
            order = epoch_order(len(dataset), self.batch_size)
            self._last_order = order
            # --num_neg > 1: the reference flattens neg_items (BaseModel.py:176-177) and _get_feed_dict reads flat[i] (:157), so
            # row i trains on flat[i], not on its own first draw; one value per row either way
            n_rows = len(dataset)
            return [torch.from_numpy(np.ascontiguousarray(dataset.data[k])).to(torch.int64).reshape(-1)[:n_rows][order].to(dev)
                    for k in ("user_id", "item_id", "neg_items")]

        def _history_columns(self, dataset, dev):
            """[n, history_max] item histories (left-aligned, zero-padded like collate_batch's pad_sequence) and their lengths
            for every training row of a sequential dataset, built once with array operations: row i takes the last
            history_max entries of its user's history before position[i] (SequentialModel.Dataset._get_feed_dict)."""
            cached = getattr(self, "_hist_cache", None)
            if cached is not None and cached[0] is dataset:
                return cached[1], cached[2]
            T = int(dataset.model.history_max)
            users = np.asarray(dataset.data["user_id"]).astype(np.int64)
            pos = np.asarray(dataset.data["position"]).astype(np.int64)
            his = dataset.corpus.user_his
            uniq = np.unique(users)
            counts = np.asarray([len(his[int(uu)]) for uu in uniq], dtype=np.int64)
            start_of = np.zeros(int(uniq.max()) + 2, np.int64)
            start_of[uniq] = np.concatenate([[0], np.cumsum(counts)[:-1]])
            flat = np.concatenate([np.fromiter((x[0] for x in his[int(uu)]), dtype=np.int64, count=len(his[int(uu)])) for uu in uniq])
            length = np.minimum(pos, T) if T > 0 else pos
            width = int(length.max()) if length.size else 1
            first = start_of[users] + pos - length
            idx = first[:, None] + np.arange(width)[None, :]
            valid = np.arange(width)[None, :] < length[:, None]
            hist = np.where(valid, flat[np.minimum(idx, flat.size - 1)], 0)
            out = (torch.from_numpy(hist).to(dev), torch.from_numpy(length).to(dev))
            self._hist_cache = (dataset, out[0], out[1])
            return out

        def fit(self, dataset, epoch=-1):
            model = dataset.model
            sequential = "position" in dataset.data
            if not hasattr(model, "user_num") or (sequential and not hasattr(model, "history_max")):
                return base_runner_cls.fit(self, dataset, epoch)
            dev = next(model.parameters()).device
            model.train()
            if hasattr(model, "train_epoch") and (self.optimizer_name in ("SGD", "Adam") or (
                    self.optimizer_name in ("Adagrad", "Adadelta") and float(self.l2) == 0.0)):
                if self.device_epoch_prep and not sequential:
                    prep = self._device_epoch(dataset, dev, epoch, pipelined=True)
                    losses = model.train_epoch(prep.cols[0], prep.cols[1], prep.cols[2], self.batch_size, self.learning_rate,
                                               float(self.l2), self.optimizer_name, prep=prep)
                    out = float(np.mean(losses.cpu().numpy()))
                    prep.check()
                    model.check_step_stream()
                    return out
                cols = self._epoch_columns(dataset, dev, epoch)
                losses = model.train_epoch(cols[0], cols[1], cols[2], self.batch_size, self.learning_rate, float(self.l2),
                                           self.optimizer_name)
                out = float(np.mean(losses.cpu().numpy()))   # one sync per epoch instead of one per batch (:200)
                model.check_step_stream()
                return out
            # any other triplet model (LightGCN, SGL, ...): the reference loop (BaseRunner.py:196-200) over batches that are
            # slices of the epoch's device columns — the same batches the DataLoader would collate sample by sample in Python
            if model.optimizer is None:
                model.optimizer = self._build_optimizer(model)
            cols = self._epoch_columns(dataset, dev, epoch)
            n, B = cols[0].numel(), self.batch_size
            # one range check for the whole epoch (nn.Embedding would raise IndexError batch by batch); the models then skip
            # their per-batch check and its device-to-host read-back
            lo_ok = min(int(c.min()) for c in cols) >= 0
            if not lo_ok or int(cols[0].max()) >= model.user_num or max(int(cols[1].max()), int(cols[2].max())) >= model.item_num:
                raise IndexError("index out of range in the training frame")
            model._trusted_indices = True
            losses = []
            hist = None
            if sequential:   # the per-row histories travel with the shuffle: same batches as the per-sample collate builds
                hist_all, len_all = self._history_columns(dataset, dev)
                order = self._last_order.to(dev)
                hist, hlen = hist_all[order], len_all[order]

            def eager_step(lo):
                batch = {"user_id": cols[0][lo:lo + B], "pos_item": cols[1][lo:lo + B], "neg_items": cols[2][lo:lo + B].unsqueeze(1),
                         "batch_size": min(B, n - lo), "phase": "train"}
                if hist is not None:
                    batch["history_items"], batch["lengths"] = hist[lo:lo + B], hlen[lo:lo + B]
                model.optimizer.zero_grad()
                loss = model.predict(batch)
                self._loss_shape = tuple(loss.shape)
                loss.backward()
                model.optimizer.step()
                losses.append(loss.detach().reshape(-1)[0])

            try:
                lo = 0
                graph = self._step_graph(model, cols, B, n, eager_step) if self.hip_graphs else None
                if graph is not None:
                    g, static, static_loss, lo = graph  # lo: first row not trained yet (a failed capture: g is None)
                    if g is not None:
                        static3 = static[0]._base           # the (3, B) buffer behind the three static index rows
                        cols3 = torch.stack(cols)
                    while g is not None and lo + B <= n:    # full batches: copy the indices in, replay the captured step
                        static3.copy_(cols3[:, lo:lo + B])
                        g.replay()
                        losses.append(static_loss.detach().reshape(-1)[0].clone())
                        lo += B
                    if g is not None and hasattr(model.optimizer, "sync_step_count"):
                        model.optimizer.sync_step_count()
                while lo < n:                               # no graph, or the short last batch
                    eager_step(lo)
                    lo += B
            finally:
                model._trusted_indices = False
            return float(torch.stack(losses).mean().cpu())

    HipRunner.__qualname__ = "HipRunner"
    return HipRunner


HipRunner = make_hip_runner(BaseRunner)


def bind_runner(reference_base_runner_cls):
    return make_hip_runner(reference_base_runner_cls)
