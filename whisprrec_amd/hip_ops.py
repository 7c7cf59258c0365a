"""Tensor-level wrappers over the C-ABI (whisprrec_amd.abi).

PyTorch-ROCm is used here only as the owner of device memory and of the HIP stream; every computation is a
call into libwhisprrec_hip.so on ``torch.cuda.current_stream()``.  Nothing in this module has a CPU path:
tensors must live on a ROCm device and the library must be built.
"""
import ctypes

import torch

from . import abi


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _req(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.device.type != "cuda":
        raise abi.WhisprRecHipError("%s must live on a ROCm device (got %s); there is no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims (got %d)" % (name, ndim, t.dim()))
    return t


def _idx64(t, name):
    """Reference batches are int64 tensors (src/models/BaseModel.py:121)."""
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return _req(t.contiguous(), torch.int64, name)


class Workspace:
    """A growable byte buffer handed to the library as caller-owned scratch."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        nbytes = int(nbytes)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self.buf


_WS = {}


def workspace(device, tag):
    """scratch buffer per (device, tag, current stream): two streams never share scratch, whoever created them"""
    sid = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (str(device), tag, sid)
    if key not in _WS:
        _WS[key] = Workspace(device)
    return _WS[key]


def side_stream(device):
    """A stream for work that must OVERLAP the step kernels (plan builds, ring transfers).  HIP spreads the streams of
    one priority round-robin over a few hardware queues, and two streams that land on the same queue run one after the
    other (seen in a kernel trace: plan stream and main stream both on queue 4, every plan build waiting for the step
    kernels to drain).  A high-priority stream has its own queue, and the small kernels it carries should not wait behind
    the whole-GPU step kernels anyway."""
    return torch.cuda.Stream(device=device, priority=-1)


def device_info():
    n_cu, wave = ctypes.c_int32(0), ctypes.c_int32(0)
    arch = ctypes.create_string_buffer(64)
    abi.check(abi.lib().wr_device_info(ctypes.addressof(n_cu), ctypes.addressof(wave), ctypes.addressof(arch), 64),
              "wr_device_info")
    return {"n_cu": n_cu.value, "wave_size": wave.value, "arch": arch.value.decode()}


# ----------------------------------------------------------------------------------------------- forward
def bpr_fwd(user_tab, item_tab, u, p, n, scores=True, coef=False):
    """BPRMF.predict forward (reference src/models/general/BPRMF.py:69-80, src/utils/loss.py:37-39).

    Returns dict(loss=0-d tensor, pos_score, neg_score, coef) — the optional entries are None when not asked."""
    L = abi.lib()
    _req(user_tab, torch.float32, "user_tab", 2)
    _req(item_tab, torch.float32, "item_tab", 2)
    u, p, n = _idx64(u, "user_id"), _idx64(p, "pos_item"), _idx64(n, "neg_items")
    B, D = u.numel(), user_tab.shape[1]
    if not (p.numel() == B and n.numel() == B):
        raise ValueError("user_id/pos_item/neg_items must have the same length")
    dev = user_tab.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    pos = torch.empty(B, dtype=torch.float32, device=dev) if scores else None
    neg = torch.empty(B, dtype=torch.float32, device=dev) if scores else None
    cf = torch.empty(B, dtype=torch.float32, device=dev) if coef else None
    nbytes = abi.check_size(L.wr_bpr_fwd_workspace_bytes(B), "wr_bpr_fwd_workspace_bytes")
    ws = workspace(dev, "fwd").get(nbytes)
    abi.check(L.wr_bpr_fwd(_p(user_tab), user_tab.shape[0], _p(item_tab), item_tab.shape[0], D, _p(u), _p(p), _p(n), B,
                           _p(pos), _p(neg), _p(cf), _p(loss), _p(ws), ws.numel(), _stream()), "wr_bpr_fwd")
    return {"loss": loss, "pos_score": pos, "neg_score": neg, "coef": cf}


# ----------------------------------------------------------------------------------------------- plan
_FAST_BACKOFF = {}


class BucketMap:
    """Load-balanced row ranges for the hand-written plan builder on skewed ids (struct wr_bucket_side of
    include/whisprrec_hip.h).  Built once per epoch from the id columns with array operations on the device: the expected
    number of triplets (item occurrences) a row contributes to a batch is its share of the epoch's rows times the batch size
    (+ the uniform negatives for items, reference BaseModel.py:167-177); consecutive rows are grouped until a bucket holds
    the mean load, and a row that holds more than half of it on its own becomes one or more buckets of its own (its
    occurrences split by position in the batch).  The buckets ascend with the rows, so the plan is the one every other
    builder emits.  ``users`` / ``items`` are None when that side needs more than 1024 buckets (then: generic builder)."""

    NOMINAL = 256        # the builder's bucket count (and capacity = 2 x mean + 64) in the regime maps are made for
    SAMPLE = 1 << 21     # rows of the epoch the shares are estimated from

    def __init__(self, u, p, n_users, n_items, batch_size):
        B, n = int(batch_size), u.numel()
        dev = u.device
        self.batch_size = B
        # Row shares from a strided sample of at most SAMPLE rows: the buckets need the heavy rows' shares to a few percent
        # and the light rows' only in sums over whole ranges, and torch.bincount's global atomics on 40 M power-law ids
        # take 18 ms (one hot counter) against <1 ms on the sample.  (Out-of-range ids are the plan's business — flags[0],
        # IndexError at validate(); here they only must not break bincount.)
        step = max(1, n // self.SAMPLE)
        us, ps = u[::step].long().clamp_(0, n_users - 1), p[::step].long().clamp_(0, n_items - 1)
        lam_u = torch.bincount(us, minlength=n_users).double() * (B / us.numel())
        lam_i = torch.bincount(ps, minlength=n_items).double() * (B / ps.numel())
        if n_items > 1:                                      # negatives: uniform over [1, n_items)
            lam_i[1:] += B / (n_items - 1)
        self.users = self._side(lam_u, B / self.NOMINAL, dev)
        self.items = self._side(lam_i, 2 * B / self.NOMINAL, dev)

    @staticmethod
    def _side(lam, mean, dev):
        n_rows = lam.numel()
        # heavy: more than half a bucket's mean load — or so many occurrences per batch that one bin of the bucket sort (bins
        # separate ROWS in an ordinary bucket, at most 256 composites each) could not hold them; heavy rows get position bins
        heavy = lam > min(mean / 2, 96.0)
        w = torch.where(heavy, torch.zeros_like(lam), lam)
        before = torch.cumsum(w, 0) - w                                   # load of the ordinary rows in front of each row
        group = torch.floor(before / mean).long()
        nsub = torch.where(heavy, torch.ceil(lam / mean), torch.zeros_like(lam)).long()
        start = torch.ones(n_rows, dtype=torch.bool, device=dev)
        if n_rows > 1:
            start[1:] = (group[1:] != group[:-1]) | heavy[1:] | heavy[:-1]
        extra = torch.zeros(n_rows, dtype=torch.long, device=dev)        # sub-buckets of the heavy row in front
        if n_rows > 1:
            extra[1:] = torch.clamp(nsub[:-1] - 1, min=0)
        first = torch.cumsum(start.long() + extra, 0) - 1                 # id of the row's (first) bucket
        n_buckets = int(first[-1].item()) + max(int(nsub[-1].item()), 1)
        if n_buckets > 1024 or int(nsub.max().item()) >= 65536:
            return None
        row_bucket = (first | (nsub << 16)).to(torch.int32)
        rows_at = torch.nonzero(start).flatten()                          # first row of every run of rows sharing a bucket
        b_start = torch.zeros(n_buckets, dtype=torch.long, device=dev)
        b_rows = torch.ones(n_buckets, dtype=torch.long, device=dev)
        b_sub = torch.zeros(n_buckets, dtype=torch.long, device=dev)
        ids = first[rows_at]
        b_start[ids] = rows_at
        ends = torch.cat([rows_at[1:], torch.tensor([n_rows], device=dev)])
        b_rows[ids] = ends - rows_at
        hrows = torch.nonzero(heavy).flatten()
        if hrows.numel():
            m = nsub[hrows]
            rep = torch.repeat_interleave(torch.arange(hrows.numel(), device=dev), m)      # one entry per sub-bucket
            q = torch.arange(rep.numel(), device=dev) - torch.repeat_interleave(torch.cumsum(m, 0) - m, m)
            ids_h = first[hrows][rep] + q
            b_start[ids_h] = hrows[rep]
            b_rows[ids_h] = 1
            b_sub[ids_h] = q | (m[rep] << 16)
        arrs = [t.to(torch.int32).contiguous() for t in (row_bucket, b_start, b_rows, b_sub)]
        return {"n_buckets": n_buckets, "arrays": arrs,
                "struct": abi.BucketSide(n_buckets, *[a.data_ptr() for a in arrs])}

    def record_stream(self, stream):
        for side in (self.users, self.items):
            if side is not None:
                for a in side["arrays"]:
                    a.record_stream(stream)


class PlanArena:
    """Pre-sized home of ONE plan's arrays (sorted triplets, occurrence lists, hot-run lists, flags + counts and their
    pinned host mirror), allocated once before a step stream starts.  A BatchPlan built into an arena makes no allocation
    (`torch.empty` of a new size ends in hipMalloc, which stalls the host long enough to drain the queue) and its small
    read-back is an asynchronous copy into pinned memory that is consumed when the plan is first used.  PipelinedSgd
    keeps two arenas and alternates them: chunk c+1 is built while chunk c trains."""

    def __init__(self, device, max_triplets, batch_size, hot=True, overlap_items=0):
        N, B = int(max_triplets), int(batch_size)
        self.device, self.max_triplets, self.batch_size = device, N, B
        self.max_batches = (N + B - 1) // B
        i32 = dict(dtype=torch.int32, device=device)
        self.idx = torch.empty(7 * N, **i32)                       # tu | tp | tn | oc_item (2N) | oc_src (2N)
        n_meta = BatchPlan.meta_len(self.max_batches)
        self.meta = torch.zeros(n_meta, **i32)
        self.meta_host = torch.zeros(n_meta, dtype=torch.int32, pin_memory=True)
        self.meta_np = self.meta_host.numpy()      # the same pinned words as a NumPy view: cheap scalar reads on the step path
        self.views = {}                            # BatchPlan.finish: slices of the words above, by plan length
        self.hot_sides = None
        if hot:
            self.hot_sides = []
            for kind in (0, 1):
                cp, cr = hot_caps(B, kind)
                buf = torch.empty(self.max_batches * (2 * cp + 3 * cr), **i32)
                self.hot_sides.append((buf, cp, cr))
        # overlap marks (wr_bprmf_plan_overlap_marks): per batch a bitmap of the item rows with several occurrences, a
        # bitmask of the deferred run heads and their list
        self.overlap = None
        if overlap_items > 0:
            words = (int(overlap_items) + 31) // 32
            dwords = (B + 31) // 32
            self.overlap = {"words": words, "dwords": dwords, "cap": overlap_def_cap(B),
                            "bitmap": torch.empty(self.max_batches * words, **i32),
                            "tdef": torch.empty(self.max_batches * dwords, **i32),
                            "def_q": torch.empty(self.max_batches * overlap_def_cap(B), **i32)}
        self.free = None      # event on the consuming stream: every step that reads this arena's arrays has run

    def release_after(self, stream):
        """the arena's arrays are read by work queued on `stream` up to here; the next build into it waits for that"""
        self.free = torch.cuda.Event()
        self.free.record(stream)


def hot_caps(batch_size, kind):
    cp, cr = ctypes.c_int64(0), ctypes.c_int64(0)
    abi.lib().wr_bprmf_hot_caps(int(batch_size), kind, ctypes.addressof(cp), ctypes.addressof(cr))
    return cp.value, cr.value


def overlap_def_cap(batch_size):
    """capacity of a batch's list of deferred user runs (overlapped step stream)"""
    return max(256, int(batch_size) // 8)


class BatchPlan:
    """Sorted batches for the fused step (see include/whisprrec_hip.h, "Batch plan").

    ``u, p, n`` are the epoch's triplets in batch order (int64 as in the reference, or int32); batch k is
    ``[k*batch_size, (k+1)*batch_size)``, the last one may be short (no drop_last, BaseRunner.py:201).

    The small device-to-host read-back a plan needs (index-range flag, bucket-overflow flag of the hand-written builder,
    hot-run counts) is either taken at once (default: the constructor returns a finished plan) or, with ``defer=True``,
    queued as an asynchronous copy into pinned memory behind the build and consumed by ``finish()`` — which every consumer
    of the plan's host-side state calls, and which costs nothing once the build has run."""

    META_HEAD = 2

    @staticmethod
    def meta_len(n_batches):
        """[0] index out of range, [1] fast-builder bucket overflow, then per batch the number of hot-run pieces and hot
        runs (items, users: wr_bprmf_plan_hot_runs), then per batch the number of deferred user runs (overlap marks)"""
        return BatchPlan.META_HEAD + 5 * int(n_batches)

    def __init__(self, u, p, n, batch_size, n_users, n_items, keep_orig=False, validate=True, ws_tag="plan", builder="auto",
                 hot=True, bucket_map=None, arena=None, defer=False, overlap=False):
        """builder: "auto" (default) = hand-written bucket/LDS-sort builder when applicable, generic radix-sort builder
        otherwise or when a bucket overflowed (skewed ids) — both emit identical arrays; "fast" / "generic" force one;
        "small" = one workgroup per batch in LDS (batch_size <= 4,096), one launch, same arrays.
        bucket_map: a BucketMap of the epoch these batches come from — load-balanced buckets for the hand-written builder
        (skewed ids); ``fast_overflowed`` tells afterwards whether the hand-written builder was tried and gave up.
        arena: a PlanArena to build into (no allocation); defer: do not wait for the read-back (see finish());
        overlap: also compute the marks of the overlapped step stream (wr_bprmf_plan_overlap_marks; needs an arena created
        with overlap_items) — ``self.overlap`` is then set by finish() when the plan qualifies (no hot rows, lists fit)."""
        if u.dtype not in (torch.int64, torch.int32):
            raise TypeError("indices must be int64 or int32")
        dt = u.dtype
        u, p, n = (_req(t.contiguous(), dt, nm, 1) for t, nm in ((u, "u"), (p, "p"), (n, "n")))
        N = u.numel()
        if not (p.numel() == N and n.numel() == N):
            raise ValueError("u/p/n must have the same length")
        if N == 0:
            raise ValueError("empty epoch")
        dev = u.device
        self.n_triplets, self.batch_size = N, int(batch_size)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.n_batches = (N + self.batch_size - 1) // self.batch_size
        self._src = (u, p, n)
        self._ws_tag, self._want_hot, self._bucket_map = ws_tag, bool(hot), bucket_map
        self._want_overlap = bool(overlap)
        if overlap and (arena is None or arena.overlap is None or not hot):
            raise ValueError("overlap marks need an arena with overlap_items and the hot-run scan")
        self._stream = torch.cuda.current_stream(dev)
        self.arena = arena
        i32 = dict(dtype=torch.int32, device=dev)
        if arena is not None:
            if arena.batch_size != self.batch_size or N > arena.max_triplets or (hot and arena.hot_sides is None):
                raise ValueError("plan does not fit its arena")
            if arena.free is not None:                       # steps that still read the arena's previous plan
                self._stream.wait_event(arena.free)
                arena.free = None
            a, M = arena.idx, arena.max_triplets
            self.tu, self.tp, self.tn = a[:N], a[M:M + N], a[2 * M:2 * M + N]
            self.oc_item, self.oc_src = a[3 * M:3 * M + 2 * N], a[5 * M:5 * M + 2 * N]
            self._cap_batches = arena.max_batches            # the layout of the counts does not depend on this plan's size
            self.meta = arena.meta
            self.meta.zero_()
            self._pinned = arena.meta_host
            self._unchecked = False
        else:
            self.tu, self.tp, self.tn = torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N, **i32)
            self.oc_item, self.oc_src = torch.empty(2 * N, **i32), torch.empty(2 * N, **i32)
            self._cap_batches = self.n_batches
            # a single-workgroup plan nobody will ask for its flags (indices range-checked by the caller, no hot-run scan):
            # no flag words to clear — one launch less in a step that builds a plan of its own (LightGCN)
            self._unchecked = builder == "small" and not validate and not hot and not overlap
            self.meta = (torch.empty if self._unchecked else torch.zeros)(self.meta_len(self.n_batches), **i32)
            self._pinned = None
        self.torig = torch.empty(N, **i32) if keep_orig else None
        self.flags = self.meta[:2]
        self.err = self.flags[:1]
        self.hot = None
        self.builder = None
        self.overlap = None
        self.meta_host = None
        self.ready = None
        self.fast_overflowed = False
        self._finished = False
        self._sides = None
        self._tried = None           # "fast" / "fast+map": the hand-written builder ran and its overflow flag is pending
        # "auto": ids skewed enough to overflow a bucket keep doing so chunk after chunk — after an overflow the next 32
        # plans of the same shape go straight to the radix-sort builder instead of paying for a failed attempt each
        self._backoff_key = (str(dev), self.batch_size, self.n_users, self.n_items)
        mapped = self._mapped()
        if builder == "auto" and not mapped and _FAST_BACKOFF.get(self._backoff_key, 0) > 0:
            _FAST_BACKOFF[self._backoff_key] -= 1
            builder = "generic"
            self.fast_overflowed = True
        self._builder_arg = builder
        if builder == "small":
            self._build_small()
        if builder in ("auto", "fast") and not self._build_fast():
            if builder == "fast":
                raise abi.WhisprRecHipError("fast plan builder not applicable to this batch size")
        if self._tried is None:
            self._build_generic()
        self._queue_readback()
        if not defer:
            self.finish()
            if validate:
                self.validate()

    # ------------------------------------------------------------------ builders (enqueue only)
    def _mapped(self):
        bm = self._bucket_map
        return bm is not None and (bm.users is not None or bm.items is not None) and bm.batch_size == self.batch_size

    def _build_fast(self):
        """enqueue the hand-written builder (+ hot-run scan); False when it does not apply to this shape"""
        L = abi.lib()
        u, p, n = self._src
        dt, dev = u.dtype, u.device
        args = (self.n_triplets, self.batch_size, self.n_users, self.n_items)
        mapped = self._mapped()
        if mapped:
            mu, mi = self._bucket_map.users, self._bucket_map.items      # a side without a map keeps its equal-width buckets
            nbytes = abi.check_size(L.wr_bprmf_plan_fast_mapped_workspace_bytes(*args, mu["n_buckets"] if mu else 0,
                                                                                mi["n_buckets"] if mi else 0),
                                    "wr_bprmf_plan_fast_mapped_workspace_bytes")
        else:
            nbytes = abi.check_size(L.wr_bprmf_plan_fast_workspace_bytes(*args), "wr_bprmf_plan_fast_workspace_bytes")
        if nbytes <= 0:
            return False
        ws = workspace(dev, self._ws_tag + ("_fastmap" if mapped else "_fast")).get(nbytes)
        if self._want_hot:
            self._sides = self._hot_arrays(dev)      # before the plan kernels: no host-side allocation between build and scan
        outs = (_p(self.tu), _p(self.tp), _p(self.tn), _p(self.torig), _p(self.oc_item), _p(self.oc_src), _p(self.flags),
                _p(ws), ws.numel(), _stream())
        self._bitmap_ready = False
        if self._want_overlap and not mapped and L.wr_bprmf_plan_fast_marks_supported(*args):
            # the item-sort workgroups write the batch's "several occurrences" bitmap themselves (no global atomics)
            fn = L.wr_bprmf_plan_build_fast_marks_i64 if dt == torch.int64 else L.wr_bprmf_plan_build_fast_marks_i32
            abi.check(fn(_p(u), _p(p), _p(n), *args, *outs[:-1], _p(self.arena.overlap["bitmap"]), _stream()),
                      "wr_bprmf_plan_build_fast_marks")
            self._bitmap_ready = True
        elif mapped:
            fn = L.wr_bprmf_plan_build_fast_mapped_i64 if dt == torch.int64 else L.wr_bprmf_plan_build_fast_mapped_i32
            abi.check(fn(_p(u), _p(p), _p(n), *args, ctypes.addressof(mu["struct"]) if mu else None,
                         ctypes.addressof(mi["struct"]) if mi else None, *outs), "wr_bprmf_plan_build_fast_mapped")
        else:
            fn = L.wr_bprmf_plan_build_fast_i64 if dt == torch.int64 else L.wr_bprmf_plan_build_fast_i32
            abi.check(fn(_p(u), _p(p), _p(n), *args, *outs), "wr_bprmf_plan_build_fast")
        self._tried = "fast+map" if mapped else "fast"
        # the hot-run scan is bounds-safe for any key values; after an overflow its output is thrown away with the plan
        if self._want_hot:
            self._enqueue_hot_runs()
        if self._want_overlap:
            self._enqueue_overlap_marks()
        return True

    def _build_small(self):
        """one launch, one workgroup per batch (batch_size <= 4,096): nothing to read back but the index-range flag"""
        L = abi.lib()
        u, p, n = self._src
        if self.batch_size > int(L.wr_bprmf_plan_small_max_batch()):
            raise abi.WhisprRecHipError("small plan builder: batch size %d > %d" % (self.batch_size,
                                                                                    L.wr_bprmf_plan_small_max_batch()))
        fn = L.wr_bprmf_plan_build_small_i64 if u.dtype == torch.int64 else L.wr_bprmf_plan_build_small_i32
        abi.check(fn(_p(u), _p(p), _p(n), self.n_triplets, self.batch_size, self.n_users, self.n_items, _p(self.tu), _p(self.tp),
                     _p(self.tn), _p(self.torig), _p(self.oc_item), _p(self.oc_src), None if self._unchecked else _p(self.err),
                     _stream()), "wr_bprmf_plan_build_small")
        self.builder = self._tried = "small"
        self._bitmap_ready = False
        if self._want_hot:
            self._sides = self._hot_arrays(u.device)
            self._enqueue_hot_runs()
        if self._want_overlap:
            self._enqueue_overlap_marks()

    def _build_generic(self):
        L = abi.lib()
        u, p, n = self._src
        dev = u.device
        args = (self.n_triplets, self.batch_size, self.n_users, self.n_items)
        nbytes = abi.check_size(L.wr_bprmf_plan_workspace_bytes(*args), "wr_bprmf_plan_workspace_bytes")
        ws = workspace(dev, self._ws_tag).get(nbytes)  # distinct tags allow plan builds in flight on different streams
        fn = L.wr_bprmf_plan_build_i64 if u.dtype == torch.int64 else L.wr_bprmf_plan_build_i32
        abi.check(fn(_p(u), _p(p), _p(n), *args, _p(self.tu), _p(self.tp), _p(self.tn), _p(self.torig),
                     _p(self.oc_item), _p(self.oc_src), _p(self.err), _p(ws), ws.numel(), _stream()),
                  "wr_bprmf_plan_build")
        self.builder = "generic"
        self._tried = "generic"
        self._bitmap_ready = False
        if self._want_hot:
            if self._sides is None:
                self._sides = self._hot_arrays(dev)
            self._enqueue_hot_runs()
        if self._want_overlap:
            self._enqueue_overlap_marks()

    def _hot_arrays(self, dev):
        """list arrays of the hot-run scan: views of the arena, or fresh tensors"""
        i32 = dict(dtype=torch.int32, device=dev)
        nb = self.n_batches
        sides = []
        for kind in (0, 1):
            if self.arena is not None:
                buf, cp, cr = self.arena.hot_sides[kind]
                m = self.arena.max_batches
                arrs = [buf[:nb * cp], buf[m * cp:m * cp + nb * cp], buf[2 * m * cp:2 * m * cp + nb * cr],
                        buf[2 * m * cp + m * cr:2 * m * cp + m * cr + nb * cr],
                        buf[2 * m * cp + 2 * m * cr:2 * m * cp + 2 * m * cr + nb * cr]]
            else:
                cp, cr = hot_caps(self.batch_size, kind)
                arrs = [torch.empty(nb * cp, **i32), torch.empty(nb * cp, **i32), torch.empty(nb * cr, **i32),
                        torch.empty(nb * cr, **i32), torch.empty(nb * cr, **i32)]
            sides.append((arrs, cp, cr))
        return sides

    def _enqueue_hot_runs(self):
        """Cuts table rows with more than 32 occurrences in a batch (item rows: runs of oc_item; user rows: runs of tu) into
        pieces for the many-workgroup path (power-law ids); the per-batch counts travel with the plan's flags."""
        L = abi.lib()
        counts = self.meta[self.META_HEAD:self.META_HEAD + 4 * self.n_batches]
        for (kind, keys), (arrs, _, _) in zip(((0, self.oc_item), (1, self.tu)), self._sides):
            abi.check(L.wr_bprmf_plan_hot_runs(_p(keys), kind, self.n_triplets, self.batch_size, *[_p(a) for a in arrs],
                                               _p(counts), _stream()), "wr_bprmf_plan_hot_runs")

    def _enqueue_overlap_marks(self):
        """marks of the overlapped step stream: which user runs of batch k+1 read an item row that step k's item phase
        rewrites (index work only).  The first batch of a plan defers nothing: the stream joins at plan boundaries."""
        o, cb = self.arena.overlap, self._cap_batches
        counts = self.meta[self.META_HEAD + 4 * cb:self.META_HEAD + 5 * cb]
        L = abi.lib()
        fn = L.wr_bprmf_plan_overlap_deferred if getattr(self, "_bitmap_ready", False) else L.wr_bprmf_plan_overlap_marks
        abi.check(fn(_p(self.tu), _p(self.tp), _p(self.tn), self.n_triplets, self.batch_size, self.n_items, None,
                     _p(o["bitmap"]), _p(o["tdef"]), _p(o["def_q"]), o["cap"], _p(counts), _stream()),
                  "wr_bprmf_plan_overlap_marks")

    def _queue_readback(self):
        """ONE small read-back per plan: flags + hot-run counts (+ deferred-run counts).  With an arena it is an asynchronous
        copy into pinned memory and an event; finish() waits for the event."""
        if self._pinned is not None:
            self._pinned.copy_(self.meta, non_blocking=True)
            self.ready = torch.cuda.Event()
            self.ready.record(self._stream)

    # ------------------------------------------------------------------ host-side completion
    def _read_meta(self):
        if self._pinned is not None:
            self.ready.synchronize()
            return self._pinned
        return self.meta.cpu()

    def finish(self):
        """Makes the plan's host-side state final: waits for the read-back (no wait at all once the build has run, which is
        the normal case one chunk ahead of the steps), falls back to the radix-sort builder after a bucket overflow, and
        publishes the hot-run counts.  Idempotent.  A plan that has nothing to report (radix-sort builder, no hot-run scan)
        is finished without any device-to-host traffic; its index-range flag is read by validate() on demand."""
        if self._finished:
            return self
        mh = None
        while self._tried in ("fast", "fast+map") or self._want_hot:
            mh = self._read_meta()
            overflow = int(self.arena.meta_np[1]) if self._pinned is not None else int(mh[1])
            if self._tried in ("fast", "fast+map") and overflow != 0:
                if self._builder_arg == "fast":
                    raise abi.WhisprRecHipError("fast plan builder: bucket overflow (skewed ids)")
                self.fast_overflowed = True
                if self._tried == "fast":
                    _FAST_BACKOFF[self._backoff_key] = 32
                with torch.cuda.stream(self._stream):        # rebuild where the plan was built (rare: skewed ids without a map)
                    self.meta.zero_()
                    self._build_generic()
                    self._queue_readback()
                mh = None                                    # stale: the rebuilt plan reports again
                continue
            break
        self.meta_host = mh
        if self.builder is None:
            self.builder = self._tried
        if self._want_hot:
            if self._pinned is not None:
                any_hot = bool(self.arena.meta_np[self.META_HEAD:self.META_HEAD + 4 * self.n_batches].any())
            else:
                any_hot = int(mh[self.META_HEAD:self.META_HEAD + 4 * self.n_batches].sum().item()) > 0
            if any_hot:
                self.hot = {"sides": self._sides, "counts_host": mh[self.META_HEAD:self.META_HEAD + 4 * self.n_batches]}
        if self._want_overlap and self.hot is None:
            o, nb, cb = self.arena.overlap, self.n_batches, self._cap_batches
            # views of the arena's words, made once per (arena, plan length): this runs on the step path, between a plan's
            # read-back and its first launch
            views = self.arena.views.get((nb, cb))
            if views is None:
                lo = self.META_HEAD + 4 * cb
                views = self.arena.views[(nb, cb)] = (mh[lo:lo + nb], self.arena.meta_np[lo:lo + nb], self.meta[lo:lo + cb])
            dc, dcn, dcd = views
            lo_n, hi_n = (int(dcn.min()), int(dcn.max())) if nb >= 2 else (-1, -1)
            if nb >= 2 and lo_n >= 0:
                # "fits": every batch's list of deferred runs is within capacity (the two-stream form needs that; the
                # chained launch decides per step)
                self.overlap = {"tdef": o["tdef"], "def_q": o["def_q"], "def_count_host": dc, "cap": o["cap"],
                                "def_count_np": dcn, "fits": hi_n <= o["cap"], "def_count_dev": dcd}
        self._finished = True
        return self

    def hot_struct(self, batch=None):
        """ctypes wr_hot_runs for the whole plan (batch=None) or for one batch; None when the plan has no hot runs."""
        self.finish()
        if self.hot is None:
            return None
        k = 0 if batch is None else batch
        vals = []
        for arrs, cp, cr in self.hot["sides"]:
            vals += [arrs[0].data_ptr() + 4 * k * cp, arrs[1].data_ptr() + 4 * k * cp, arrs[2].data_ptr() + 4 * k * cr,
                     arrs[3].data_ptr() + 4 * k * cr, arrs[4].data_ptr() + 4 * k * cr]
        (_, cp0, cr0), (_, cp1, cr1) = self.hot["sides"]
        return abi.HotRuns(*vals, self.hot["counts_host"].data_ptr() + 16 * k, cp0, cr0, cp1, cr1)

    def validate(self):
        """nn.Embedding raises IndexError for out-of-range ids; so does the plan."""
        if self._unchecked:
            raise abi.WhisprRecHipError("this plan was built with validate=False and keeps no index-range flag")
        self.finish()
        if self.meta_host is None:
            self.meta_host = self._read_meta()
        bad = int(self.arena.meta_np[0]) if self._pinned is not None else int(self.meta_host[0].item())
        if bad != 0:
            raise IndexError("index out of range in batch (user_id >= n_users or item id >= n_items)")

    def batch_len(self, k):
        return min(self.batch_size, self.n_triplets - k * self.batch_size)

    def record_stream(self, stream):
        """A plan built on a side stream and consumed on `stream`: tell the caching allocator, so that dropping the plan
        while its steps are still queued does not hand the arrays to the next build.  (Arena plans: PlanArena.release_after.)"""
        if self.arena is not None:
            return
        ts = [self.tu, self.tp, self.tn, self.oc_item, self.oc_src, self.meta]
        if self.torig is not None:
            ts.append(self.torig)
        if self._sides is not None:
            for arrs, _, _ in self._sides:
                ts += arrs
        for t in ts:
            t.record_stream(stream)


class GroupArena:
    """Pre-sized home of ONE group plan (GroupPlan): the plan words and the pinned mirror of its meta words.  Like
    PlanArena: nothing is allocated while steps run, the read-back is an asynchronous copy consumed at first use."""

    def __init__(self, device, max_triplets, batch_size, n_users, n_items):
        words = int(abi.lib().wr_group_plan_words(int(max_triplets), int(batch_size), int(n_users), int(n_items)))
        if words <= 0:
            raise abi.WhisprRecHipError("group plan: shape not supported (batch size %d)" % batch_size)
        self.device, self.max_triplets, self.batch_size = device, int(max_triplets), int(batch_size)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.buf = torch.empty(words, dtype=torch.int32, device=device)
        self.meta_host = torch.zeros(GroupPlan.META, dtype=torch.int32, pin_memory=True)
        self.meta_np = self.meta_host.numpy()
        self.free = None

    def release_after(self, stream):
        self.free = torch.cuda.Event()
        self.free.record(stream)


class GroupPlan:
    """Batches for the step stream WITHOUT a per-batch sort (include/whisprrec_hip.h, wr_group_plan_build): the triplets stay
    where they are (``u, p, n`` int32, batch order — kept by the plan, the step kernels read them); per batch the plan holds
    four flag bits per triplet and the shared occurrences only, as sorted lists.  One launch per build, ~0.1 B of plan per
    triplet.  ``finish()`` publishes ``bad_index`` / ``overflow`` (the plan is unusable: fall back to BatchPlan) /
    ``long_run`` (usable, but a row has more than 64 occurrences in a batch)."""

    META = 16

    def __init__(self, u, p, n, batch_size, n_users, n_items, arena=None, defer=False):
        u, p, n = (_req(t, torch.int32, nm, 1) for t, nm in ((u, "u"), (p, "p"), (n, "n")))
        N = u.numel()
        if not (p.numel() == N and n.numel() == N) or N == 0:
            raise ValueError("u/p/n must have the same non-zero length")
        L = abi.lib()
        self.u, self.p, self.n = u, p, n
        self.n_triplets, self.batch_size = N, int(batch_size)
        self.n_users, self.n_items = int(n_users), int(n_items)
        self.n_batches = (N + self.batch_size - 1) // self.batch_size
        self.words = int(L.wr_group_plan_words(N, self.batch_size, self.n_users, self.n_items))
        if self.words <= 0:
            raise abi.WhisprRecHipError("group plan: shape not supported (batch size %d)" % self.batch_size)
        self.arena = arena
        self._stream = torch.cuda.current_stream(u.device)
        if arena is not None:
            if arena.buf.numel() < self.words or arena.batch_size != self.batch_size or \
                    (arena.n_users, arena.n_items) != (self.n_users, self.n_items):
                raise ValueError("plan does not fit its arena")
            if arena.free is not None:
                self._stream.wait_event(arena.free)
                arena.free = None
            self.buf = arena.buf
            self._pinned = arena.meta_host
        else:
            self.buf = torch.empty(self.words, dtype=torch.int32, device=u.device)
            self._pinned = None
        abi.check(L.wr_group_plan_build(_p(u), _p(p), _p(n), N, self.batch_size, self.n_users, self.n_items, _p(self.buf),
                                        self.buf.numel(), _stream()), "wr_group_plan_build")
        self.ready = None
        if self._pinned is not None:
            self._pinned.copy_(self.buf[:self.META], non_blocking=True)
            self.ready = torch.cuda.Event()
            self.ready.record(self._stream)
        self._finished = False
        self.bad_index = self.overflow = self.long_run = False
        if not defer:
            self.finish()

    def finish(self):
        if self._finished:
            return self
        if self._pinned is not None:
            self.ready.synchronize()
            m = self.arena.meta_np
        else:
            m = self.buf[:self.META].cpu().numpy()
        self.bad_index, self.overflow, self.long_run = bool(m[0]), bool(m[1]), bool(m[2])
        self._finished = True
        return self

    def finish_from(self, meta):
        """the plan's first three words, read back by the caller together with words of its own"""
        self.bad_index, self.overflow, self.long_run = bool(meta[0]), bool(meta[1]), bool(meta[2])
        self._finished = True
        return self

    def validate(self):
        """nn.Embedding raises IndexError for out-of-range ids; so does the plan."""
        self.finish()
        if self.bad_index:
            raise IndexError("index out of range in batch (user_id >= n_users or item id >= n_items)")

    def batch_len(self, k):
        return min(self.batch_size, self.n_triplets - k * self.batch_size)

    def layout(self):
        out = (ctypes.c_int64 * 16)()
        abi.check(abi.lib().wr_group_plan_layout(self.n_triplets, self.batch_size, self.n_users, self.n_items,
                                                 ctypes.addressof(out)), "wr_group_plan_layout")
        keys = ("nb", "fw", "R_u", "R_i", "mask_u", "mask_i", "flags", "ucnt", "icnt", "ul_row", "ul_src", "il_row", "il_src",
                "total", "cap_u", "cap_i")
        return dict(zip(keys, [int(v) for v in out]))

    def decode(self):
        """the plan as NumPy arrays (tests, statistics): flags [nb, fw, 4] uint32 and, per (batch, range), the user and item
        lists as (rows, sources)"""
        import numpy as np
        self.finish()
        Ly = self.layout()
        w = self.buf[:Ly["total"]].cpu().numpy()
        nb, fw = Ly["nb"], Ly["fw"]
        flags = w[Ly["flags"]:Ly["flags"] + nb * fw * 4].view(np.uint32).reshape(nb, fw, 4).copy()
        out = {"flags": flags, "users": {}, "items": {}, "R_u": Ly["R_u"], "R_i": Ly["R_i"], "meta": w[:self.META].copy()}
        for side, R, cap, co, ro, so in (("users", Ly["R_u"], Ly["cap_u"], "ucnt", "ul_row", "ul_src"),
                                         ("items", Ly["R_i"], Ly["cap_i"], "icnt", "il_row", "il_src")):
            for b in range(nb):
                for r in range(R):
                    c = int(w[Ly[co] + b * R + r])
                    at = (b * R + r) * cap
                    out[side][(b, r)] = (w[Ly[ro] + at:Ly[ro] + at + c].astype(np.int64),
                                         w[Ly[so] + at:Ly[so] + at + c].astype(np.int64))
        return out


_CHAIN_SYNC = {}      # (device, stream) -> [int32 tensors]: hand-off counters of the chained step launches (BprmfTables._chain_sync)
_OVERLAP_WS_BYTES = {}       # (batch size, D) -> bytes of the two-slot step workspace


class BprmfTables:
    """The two embedding tables plus the scratch the step kernels need.  Tables are plain fp32 tensors
    (row-major [n_rows, D]) that stay valid PyTorch tensors between steps."""

    def __init__(self, user_tab, item_tab):
        self.U = _req(user_tab, torch.float32, "user_tab", 2)
        self.I = _req(item_tab, torch.float32, "item_tab", 2)
        if self.U.shape[1] != self.I.shape[1]:
            raise ValueError("user and item tables must have the same embedding size")
        self.D = self.U.shape[1]
        self.dev = self.U.device
        self.stamp_u = None
        self.stamp_i = None
        self.step_id = 0

    def _stamps(self):
        if self.stamp_u is None:
            self.stamp_u = torch.full((self.U.shape[0],), -1, dtype=torch.int32, device=self.dev)
            self.stamp_i = torch.full((self.I.shape[0],), -1, dtype=torch.int32, device=self.dev)
        return self.stamp_u, self.stamp_i

    def _ws(self, B):
        nbytes = abi.check_size(abi.lib().wr_bprmf_step_workspace_bytes(B, self.D), "wr_bprmf_step_workspace_bytes")
        return workspace(self.dev, "step").get(nbytes)

    def _plan_ptrs(self, plan, k):
        off = k * plan.batch_size
        return (plan.tu.data_ptr() + 4 * off, plan.tp.data_ptr() + 4 * off, plan.tn.data_ptr() + 4 * off,
                plan.oc_item.data_ptr() + 8 * off, plan.oc_src.data_ptr() + 8 * off, plan.batch_len(k))

    def step_sgd(self, plan, k, lr, l2=0.0, loss_out=None, decay_untouched=True):
        """One BaseRunner.fit iteration (zero_grad/predict/backward/SGD.step, BaseRunner.py:196-199) on batch k.
        decay_untouched=False leaves the weight decay of the rows outside the batch to LazyOptimizerState."""
        L = abi.lib()
        tu, tp, tn, oi, os_, B = self._plan_ptrs(plan, k)
        ws = self._ws(plan.batch_size)
        if loss_out is None:
            loss_out = torch.empty((), dtype=torch.float32, device=self.dev)
        su = si = None
        self.step_id += 1
        if l2 != 0.0:
            su, si = self._stamps()
        hot = plan.hot_struct(k)
        abi.check(L.wr_bprmf_step_sgd(_p(self.U), self.U.shape[0], _p(self.I), self.I.shape[0], self.D, tu, tp, tn, oi,
                                      os_, B, lr, l2, _p(su), _p(si), self.step_id, _p(loss_out),
                                      ctypes.addressof(hot) if hot is not None else None, _p(ws), ws.numel(),
                                      _stream()), "wr_bprmf_step_sgd")
        if l2 != 0.0 and decay_untouched:  # dense weight decay on the rows the batch did not touch (torch.optim.SGD semantics)
            abi.check(L.wr_sgd_decay_untouched(_p(self.U), self.U.shape[0], self.D, _p(su), self.step_id, lr, l2,
                                               _stream()), "wr_sgd_decay_untouched")
            abi.check(L.wr_sgd_decay_untouched(_p(self.I), self.I.shape[0], self.D, _p(si), self.step_id, lr, l2,
                                               _stream()), "wr_sgd_decay_untouched")
        return loss_out

    def run_sgd(self, plan, first, count, lr, losses=None, phase_events=None):
        """`count` consecutive steps starting at batch `first` (native inner loop of BaseRunner.fit, l2 = 0).

        phase_events: optional list of 4*count ``torch.cuda.Event(enable_timing=True)`` or None entries: per step the
        start / stop events of its two phases (include/whisprrec_hip.h; bench.py's per-kernel timing)."""
        L = abi.lib()
        ws = self._ws(plan.batch_size)
        if losses is None:
            losses = torch.empty(count, dtype=torch.float32, device=self.dev)
        hot = plan.hot_struct()
        ev = None
        if phase_events is not None:
            if len(phase_events) != 4 * count:
                raise ValueError("phase_events must hold 4 events per step")
            handles = []
            for e in phase_events:
                if e is None:
                    handles.append(None)
                    continue
                if not e.cuda_event:  # torch creates the hipEvent_t lazily on first record
                    e.record()
                handles.append(e.cuda_event)
            ev = (ctypes.c_void_p * len(handles))(*handles)
        abi.check(L.wr_bprmf_run_sgd(_p(self.U), self.U.shape[0], _p(self.I), self.I.shape[0], self.D, _p(plan.tu),
                                     _p(plan.tp), _p(plan.tn), _p(plan.oc_item), _p(plan.oc_src), plan.n_triplets,
                                     plan.batch_size, first, count, lr, _p(losses),
                                     ctypes.addressof(ev) if ev is not None else None,
                                     ctypes.addressof(hot) if hot is not None else None, _p(ws), ws.numel(), _stream()),
                  "wr_bprmf_run_sgd")
        self.step_id += count
        return losses

    def overlap_workspace(self, batch_size):
        nbytes = _OVERLAP_WS_BYTES.get((batch_size, self.D))
        if nbytes is None:
            nbytes = _OVERLAP_WS_BYTES[(batch_size, self.D)] = 2 * abi.check_size(
                abi.lib().wr_bprmf_step_workspace_bytes(batch_size, self.D), "wr_bprmf_step_workspace_bytes")
        return workspace(self.dev, "step_overlap").get(nbytes)

    def chain_supported(self):
        """rows are whole 128-byte lines (D % 32 == 0, tables 128-byte aligned): the chained step launch applies"""
        return bool(abi.lib().wr_bprmf_chain_supported(_p(self.U), _p(self.I), self.D))

    CHAIN_SYNC_STEPS = 256      # the hand-off counters are sized for calls of this many steps up front (1 MB): growing
                                # them means an allocation in the middle of a step stream

    def _chain_sync(self, count):
        """hand-off counters + sticky timeout word of the chained launches: ONE buffer per (device, stream) shared by every
        tables object (the stratified schedule makes one per segment) — calls on a stream are ordered, each zeroes the
        counters it uses"""
        key = (str(self.dev), torch.cuda.current_stream(self.dev).cuda_stream)
        bufs = _CHAIN_SYNC.setdefault(key, [])
        if bufs and count <= self.CHAIN_SYNC_STEPS:
            return bufs[-1]                              # sized for CHAIN_SYNC_STEPS steps or more
        if not bufs or bufs[-1].numel() < int(abi.lib().wr_bprmf_chain_sync_words(count)):
            words = int(abi.lib().wr_bprmf_chain_sync_words(max(count, self.CHAIN_SYNC_STEPS)))
            bufs.append(torch.zeros(words, dtype=torch.int32, device=self.dev))      # older ones stay for check_chain
        return bufs[-1]

    def sticky_words(self):
        """the sticky "a bounded wait expired" words of the step launches issued on the current stream of this device (one
        int32 view per hand-off buffer) — PipelinedSgd copies them to pinned memory behind every chunk"""
        sid = torch.cuda.current_stream(self.dev).cuda_stream
        return [buf[-4:-3] for key, bufs in _CHAIN_SYNC.items() if key[0] == str(self.dev) and key[1] == sid for buf in bufs]

    def check_chain(self):
        """raises if a bounded wait inside a chained step launch ever expired on this device (synchronises; call it where
        the caller waits for the device anyway: end of an epoch, end of a benchmark)"""
        for key, bufs in _CHAIN_SYNC.items():
            if key[0] != str(self.dev):
                continue
            for buf in bufs:
                if int(buf[-4].item()) != 0:
                    raise abi.WhisprRecHipError("chained step launch: a wait for the item tiles expired — the tables are "
                                                "not valid")

    def run_sgd_chain(self, plan, first, count, lr, losses=None, phase_events=None, ws=None, def_limit=None):
        """`count` consecutive steps like run_sgd, one launch per step (wr_bprmf_run_sgd_chain): the item phase of step k-1
        rides in the launch of step k's user phase.  plan.overlap must be set (BatchPlan built with overlap=True that
        qualified: no hot rows, deferred-run lists within capacity).  Same tables, bit for bit, as run_sgd."""
        L = abi.lib()
        o = plan.overlap
        if ws is None:
            ws = self.overlap_workspace(plan.batch_size)
        if losses is None:
            losses = torch.empty(count, dtype=torch.float32, device=self.dev)
        ev = None
        if phase_events is not None:
            if len(phase_events) != 4 * count:
                raise ValueError("phase_events must hold 4 events per step")
            handles = []
            for e in phase_events:
                if e is None:
                    handles.append(None)
                    continue
                if not e.cuda_event:
                    e.record()
                handles.append(e.cuda_event)
            ev = (ctypes.c_void_p * len(handles))(*handles)
        sync = self._chain_sync(count)
        abi.check(L.wr_bprmf_run_sgd_chain(_p(self.U), self.U.shape[0], _p(self.I), self.I.shape[0], self.D, _p(plan.tu),
                                           _p(plan.tp), _p(plan.tn), _p(plan.oc_item), _p(plan.oc_src), plan.n_triplets,
                                           plan.batch_size, first, count, lr, _p(losses), _p(o["tdef"]), _p(o["def_q"]),
                                           o["def_count_host"].data_ptr(), o["cap"],
                                           o["cap"] if def_limit is None else int(def_limit),
                                           ctypes.addressof(ev) if ev is not None else None, _p(ws), ws.numel(), _p(sync),
                                           sync.numel(), _stream()), "wr_bprmf_run_sgd_chain")
        self.step_id += count
        return losses

    def group_supported(self):
        """rows are whole 128-byte lines: the step stream without a per-batch sort applies (wr_bprmf_run_sgd_group)"""
        return bool(abi.lib().wr_bprmf_group_supported(_p(self.U), _p(self.I), self.D))

    def _group_sync(self, count):
        """hand-off counters + sticky timeout word of wr_bprmf_run_sgd_group: kept with the chained launches' (check_chain
        reads both)"""
        key = (str(self.dev), torch.cuda.current_stream(self.dev).cuda_stream, "group")
        bufs = _CHAIN_SYNC.setdefault(key, [])
        need = int(abi.lib().wr_bprmf_group_sync_words(count))
        if not bufs or bufs[-1].numel() < need:
            words = int(abi.lib().wr_bprmf_group_sync_words(max(count, self.CHAIN_SYNC_STEPS)))
            bufs.append(torch.zeros(words, dtype=torch.int32, device=self.dev))
        return bufs[-1]

    def run_sgd_group(self, gplan, first, count, lr, losses=None, events=None):
        """`count` consecutive steps of a GroupPlan from batch `first` (wr_bprmf_run_sgd_group): one launch per step, no
        sorted plan.  events: optional 2 * count ``torch.cuda.Event(enable_timing=True)`` — start / stop of each step's launch."""
        L = abi.lib()
        nbytes = abi.check_size(L.wr_bprmf_group_workspace_bytes(gplan.batch_size, self.D), "wr_bprmf_group_workspace_bytes")
        ws = workspace(self.dev, "group_step").get(nbytes)
        if losses is None:
            losses = torch.empty(count, dtype=torch.float32, device=self.dev)
        ev = None
        if events is not None:
            if len(events) != 2 * count:
                raise ValueError("events must hold 2 events per step")
            handles = []
            for e in events:
                if not e.cuda_event:
                    e.record()
                handles.append(e.cuda_event)
            ev = (ctypes.c_void_p * len(handles))(*handles)
        sync = self._group_sync(count)
        abi.check(L.wr_bprmf_run_sgd_group(_p(self.U), self.U.shape[0], _p(self.I), self.I.shape[0], self.D, _p(gplan.u),
                                           _p(gplan.p), _p(gplan.n), gplan.n_triplets, gplan.batch_size, _p(gplan.buf),
                                           gplan.buf.numel(), first, count, lr, _p(losses),
                                           ctypes.addressof(ev) if ev is not None else None, _p(ws), ws.numel(), _p(sync),
                                           sync.numel(), _stream()), "wr_bprmf_run_sgd_group")
        self.step_id += count
        return losses

    def grads(self, plan, k, grad_u, grad_i, loss_out=None, stamps=True):
        """embedding_dense_backward of BaseRunner.py:198 for batch k: writes the gradient rows of the rows in the
        batch into grad_u / grad_i and stamps them with the returned step id (other rows are not written).
        stamps=False: no stamp arrays at all — for a caller that zero-filled grad_u / grad_i and reads them densely."""
        L = abi.lib()
        tu, tp, tn, oi, os_, B = self._plan_ptrs(plan, k)
        ws = self._ws(plan.batch_size)
        su, si = self._stamps() if stamps else (None, None)
        if loss_out is None:
            loss_out = torch.empty((), dtype=torch.float32, device=self.dev)
        self.step_id += 1
        hot = plan.hot_struct(k)
        abi.check(L.wr_bprmf_grads(_p(self.U), self.U.shape[0], _p(self.I), self.I.shape[0], self.D, tu, tp, tn, oi, os_,
                                   B, _p(grad_u), _p(grad_i), _p(su), _p(si), self.step_id, _p(loss_out),
                                   ctypes.addressof(hot) if hot is not None else None, _p(ws), ws.numel(), _stream()),
                  "wr_bprmf_grads")
        return loss_out, self.step_id


# ----------------------------------------------------------------------------------------------- epoch prep
PAIR_SET_MIN_PAIRS = 1 << 20      # from this many (user, item) pairs on the samplers test membership in a hash set


def pair_set(clicked_ptr, clicked_idx, n_users):
    """Hash set of the (user, item) pairs of the clicked lists (wr_pairset_build): int64 tensor [capacity] for the `pairs`
    argument of sample_negatives / EpochPrep.  One membership test = one random 64-byte sector instead of a binary search
    in the user's list (~7 sectors): same answers, same negatives.  Built once per training frame."""
    _req(clicked_ptr, torch.int64, "clicked_ptr", 1)
    _req(clicked_idx, torch.int32, "clicked_idx", 1)
    L = abi.lib()
    cap = abi.check_size(L.wr_pairset_capacity(int(clicked_idx.numel())), "wr_pairset_capacity")
    table = torch.empty(cap, dtype=torch.int64, device=clicked_ptr.device)
    err = torch.zeros(1, dtype=torch.int32, device=clicked_ptr.device)
    abi.check(L.wr_pairset_build(_p(clicked_ptr), _p(clicked_idx), int(n_users), _p(table), cap, _p(err), _stream()),
              "wr_pairset_build")
    if int(err.item()) != 0:
        raise abi.WhisprRecHipError("pair set: table too small")
    return table


def sample_negatives(users, n_users, n_items, clicked_ptr, clicked_idx, seed, epoch, pairs=None):
    """Device negative sampler (reference src/models/BaseModel.py:167-177 semantics, counter-based generator).
    pairs: pair_set(clicked_ptr, clicked_idx, n_users) — membership through the hash set instead of the lists."""
    if users.dtype not in (torch.int64, torch.int32):
        raise TypeError("users must be int64 or int32")
    users = _req(users.contiguous(), users.dtype, "users", 1)
    if pairs is None:
        _req(clicked_ptr, torch.int64, "clicked_ptr", 1)
        _req(clicked_idx, torch.int32, "clicked_idx", 1)
    neg = torch.empty_like(users)
    err = torch.zeros(1, dtype=torch.int32, device=users.device)
    L = abi.lib()
    if pairs is not None:
        fn = L.wr_sample_negatives_set_i64 if users.dtype == torch.int64 else L.wr_sample_negatives_set_i32
        abi.check(fn(_p(users), users.numel(), n_users, n_items, _p(_req(pairs, torch.int64, "pairs", 1)), pairs.numel(), seed,
                     epoch, _p(neg), _p(err), _stream()), "wr_sample_negatives_set")
        return neg, err
    fn = L.wr_sample_negatives_i64 if users.dtype == torch.int64 else L.wr_sample_negatives_i32
    abi.check(fn(_p(users), users.numel(), n_users, n_items, _p(clicked_ptr), _p(clicked_idx), seed, epoch, _p(neg), _p(err),
                 _stream()), "wr_sample_negatives")
    return neg, err


def epoch_shuffle(cols, seed, epoch, want_order=False):
    """Device epoch shuffle (wr_epoch_shuffle): up to three 1-D index columns of one dtype (int64 or int32) and one length,
    permuted by the same keyed bijection of the rows.  Returns the permuted columns (and the int64 order if asked)."""
    cols = list(cols)
    if not 1 <= len(cols) <= 3:
        raise ValueError("one to three columns")
    dt = cols[0].dtype
    if dt not in (torch.int64, torch.int32):
        raise TypeError("columns must be int64 or int32")
    n = cols[0].numel()
    cols = [_req(c.contiguous(), dt, "column", 1) for c in cols]
    if any(c.numel() != n for c in cols):
        raise ValueError("columns must have the same length")
    outs = [torch.empty_like(c) for c in cols]
    order = torch.empty(n, dtype=torch.int64, device=cols[0].device) if want_order else None
    pad = [None] * (3 - len(cols))
    fn = abi.lib().wr_epoch_shuffle_i64 if dt == torch.int64 else abi.lib().wr_epoch_shuffle_i32
    abi.check(fn(*[_p(c) for c in cols + pad], n, int(seed), int(epoch), *[_p(o) for o in outs + pad], _p(order), _stream()),
              "wr_epoch_shuffle")
    return (outs, order) if want_order else outs


class EpochPrep:
    """The epoch's (user, positive, negative) columns in batch order, produced RANGE BY RANGE on the device
    (wr_epoch_prepare_range: keyed shuffle + negative sampling fused; same columns as sample_negatives + epoch_shuffle).
    ``cols`` are the epoch-sized output arrays; ``fill(lo, hi)`` enqueues rows [lo, hi) on the current stream.  PipelinedSgd
    calls it for each plan chunk right before that chunk's plan build, on the plan stream: the preparation of chunk c+1 runs
    beside the steps of chunk c."""

    def __init__(self, users, items, n_users, n_items, clicked_ptr, clicked_idx, seed, epoch, want_order=False, pairs=None,
                 packed=None):
        """pairs: pair_set(...) — membership through the hash set; packed: pack_rows(users, items) — with a pair set, the
        source rows are read as one 8-byte word each (one random sector per output row instead of two; same columns)"""
        if users.dtype not in (torch.int64, torch.int32) or items.dtype != users.dtype:
            raise TypeError("users / items must both be int64 or int32")
        self.users = _req(users.contiguous(), users.dtype, "users", 1)
        self.items = _req(items.contiguous(), users.dtype, "items", 1)
        if self.items.numel() != self.users.numel():
            raise ValueError("users / items must have the same length")
        if pairs is None:
            _req(clicked_ptr, torch.int64, "clicked_ptr", 1)
            _req(clicked_idx, torch.int32, "clicked_idx", 1)
        self.ptr, self.idx = clicked_ptr, clicked_idx
        self.pairs = None if pairs is None else _req(pairs, torch.int64, "pairs", 1)     # pair_set(...): hash-set membership
        self.packed = None
        if packed is not None and pairs is not None:
            self.packed = _req(packed, torch.int64, "packed", 1)
            if self.packed.numel() != self.users.numel():
                raise ValueError("packed rows: one word per interaction")
        self.n, self.n_users, self.n_items = users.numel(), int(n_users), int(n_items)
        self.seed, self.epoch = int(seed), int(epoch)
        self.cols = [torch.empty_like(self.users) for _ in range(3)]
        self.order = torch.empty(self.n, dtype=torch.int64, device=users.device) if want_order else None
        self.err = torch.zeros(1, dtype=torch.int32, device=users.device)
        self.filled = 0

    def fill(self, lo, hi):
        lo, hi = int(lo), min(int(hi), self.n)
        if hi <= lo:
            return
        es = self.users.element_size()
        L = abi.lib()
        tail = (self.seed, self.epoch, lo, hi - lo, self.cols[0].data_ptr() + es * lo, self.cols[1].data_ptr() + es * lo,
                self.cols[2].data_ptr() + es * lo, None if self.order is None else self.order.data_ptr() + 8 * lo,
                _p(self.err), _stream())
        if self.packed is not None:
            fn = L.wr_epoch_prepare_range_packed_i64 if self.users.dtype == torch.int64 else L.wr_epoch_prepare_range_packed_i32
            abi.check(fn(_p(self.packed), self.n, self.n_users, self.n_items, _p(self.pairs), self.pairs.numel(), *tail),
                      "wr_epoch_prepare_range_packed")
        elif self.pairs is not None:
            fn = L.wr_epoch_prepare_range_set_i64 if self.users.dtype == torch.int64 else L.wr_epoch_prepare_range_set_i32
            abi.check(fn(_p(self.users), _p(self.items), self.n, self.n_users, self.n_items, _p(self.pairs), self.pairs.numel(),
                         *tail), "wr_epoch_prepare_range_set")
        else:
            fn = L.wr_epoch_prepare_range_i64 if self.users.dtype == torch.int64 else L.wr_epoch_prepare_range_i32
            abi.check(fn(_p(self.users), _p(self.items), self.n, self.n_users, self.n_items, _p(self.ptr), _p(self.idx), *tail),
                      "wr_epoch_prepare_range")
        self.filled = max(self.filled, hi)

    def check(self):
        """after the epoch (one read-back): nn.Embedding would have raised IndexError for an out-of-range user id"""
        if int(self.err.item()) == 1:
            raise IndexError("user id out of range in the training frame")


def pack_rows(users, items):
    """the training frame's (user, item) rows as one int64 word each, (user << 32) | item (EpochPrep(packed=...)); ids < 2^31"""
    return ((users.to(torch.int64) << 32) | items.to(torch.int64)).contiguous()


def clicked_csr_from_pairs(users, items, n_users, n_items):
    """device (user, item) interaction pairs -> device CSR of each user's distinct items, ascending (the layout
    wr_sample_negatives and wr_rank_eval take).  One-off setup: torch sort/unique glue."""
    key = torch.unique(users.to(torch.int64) * int(n_items) + items.to(torch.int64))      # sorted, distinct
    uu = torch.div(key, int(n_items), rounding_mode="floor")
    ptr = torch.zeros(n_users + 1, dtype=torch.int64, device=users.device)
    ptr[1:] = torch.cumsum(torch.bincount(uu, minlength=n_users), 0)
    return ptr, (key - uu * int(n_items)).to(torch.int32)


def clicked_csr(train_clicked_set, n_users, device):
    """train_clicked_set (dict user -> set of items, reference BaseReader.py:35-46) -> device CSR with ascending items."""
    import numpy as np
    ptr = np.zeros(n_users + 1, np.int64)
    chunks = []
    for uu in range(n_users):
        items = train_clicked_set.get(uu, ())
        ptr[uu + 1] = ptr[uu] + len(items)
        if len(items):
            chunks.append(np.sort(np.fromiter(items, dtype=np.int32, count=len(items))))
    idx = np.concatenate(chunks) if chunks else np.zeros(1, np.int32)
    return torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device)


# ----------------------------------------------------------------------------------------------- evaluation
LDS_PER_WORKGROUP = 160 * 1024      # gfx950: 163,840 B per workgroup


def rank_eval_supports(D):
    """embedding sizes wr_rank_eval takes: the register-operand kernel for D in {8, 16, 32, 64}, otherwise the LDS-operand
    kernel, which stages (128 + 32) rows of D + 1 floats and 128 counters — D <= 252 on gfx950's 160 KiB of LDS"""
    D = int(D)
    if D % 4 != 0 or D < 4:
        return False
    return D in (8, 16, 32, 64) or ((128 + 32) * (D + 1) + 128) * 4 <= LDS_PER_WORKGROUP


def rank_eval(user_mat, item_tab, eval_user, eval_target, mask_ptr=None, mask_idx=None):
    """Rank of the ground-truth item among all (unmasked) items for every evaluation row (reference
    BaseRunner.interface + evaluate_method); returns (rank int32 [n], target_score fp32 [n])."""
    _req(user_mat, torch.float32, "user_mat", 2)
    _req(item_tab, torch.float32, "item_tab", 2)
    eu, et = _idx64(eval_user.reshape(-1), "eval_user"), _idx64(eval_target.reshape(-1), "eval_target")
    if mask_ptr is not None:
        _req(mask_ptr, torch.int64, "mask_ptr", 1)
        _req(mask_idx, torch.int32, "mask_idx", 1)
    n = eu.numel()
    rank = torch.empty(n, dtype=torch.int32, device=user_mat.device)
    tsc = torch.empty(n, dtype=torch.float32, device=user_mat.device)
    abi.check(abi.lib().wr_rank_eval(_p(user_mat), user_mat.shape[0], _p(item_tab), item_tab.shape[0], user_mat.shape[1],
                                     _p(eu), _p(et), n, _p(mask_ptr), _p(mask_idx), _p(rank), _p(tsc), _stream()),
              "wr_rank_eval")
    return rank, tsc


# ----------------------------------------------------------------------------------------------- optimizers
def sgd_dense(tab, grad, lr, l2=0.0, stamp=None, step_id=0):
    abi.check(abi.lib().wr_sgd_dense(_p(_req(tab, torch.float32, "tab", 2)), tab.shape[0], tab.shape[1],
                                     _p(_req(grad, torch.float32, "grad", 2)), _p(stamp), step_id, lr, l2, _stream()),
              "wr_sgd_dense")


def adam_dense(tab, exp_avg, exp_avg_sq, grad, adam_step, lr, l2=0.0, beta1=0.9, beta2=0.999, eps=1e-8, stamp=None,
               step_id=0):
    for t, nm in ((tab, "tab"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (grad, "grad")):
        _req(t, torch.float32, nm, 2)
    abi.check(abi.lib().wr_adam_dense(_p(tab), _p(exp_avg), _p(exp_avg_sq), tab.shape[0], tab.shape[1], _p(grad),
                                      _p(stamp), step_id, adam_step, lr, l2, beta1, beta2, eps, _stream()),
              "wr_adam_dense")


def _join_views(ts):
    """concatenation of 1-D tensors; free when they are adjacent views of the same storage"""
    if len(ts) == 1:
        return ts[0]
    a = ts[0]
    adjacent, end = a.is_contiguous(), a.storage_offset() + a.numel()
    for b in ts[1:]:
        adjacent = adjacent and b.is_contiguous() and b.dtype == a.dtype and \
            b.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and b.storage_offset() == end
        end += b.numel()
    if adjacent:
        return torch.as_strided(a, (end - a.storage_offset(),), (1,), a.storage_offset())
    return torch.cat(ts)


class PipelinedSgd:
    """Plain-SGD training over pre-ordered triplets with the plan build off the critical path.  The work is a list of
    segments (item table view + indices; one segment for a whole epoch on one GPU, one per part of the held block in the
    stratified multi-GPU schedule).  Batches are planned a chunk at a time on a side stream, one chunk ahead of the steps
    (a plan depends only on the indices, never on the tables) — across segment boundaries too — and a segment's steps are
    issued from native code (wr_bprmf_run_sgd).  A chunk is at least ``chunk`` batches and at least PLAN_TRIPLETS triplets
    (64 small batches do not amortise a plan's fixed host cost: scripts/exp/small_batch_pace.py, us/step with 64 batches per
    plan -> with 4 M triplets per plan: B = 2,048 11.1 -> 8.8, B = 16,384 17.8 -> 12.9; B = 65,536 is 64 batches either way).

    Nothing in the step stream waits on the host or allocates: plans are built into two pre-sized arenas (PlanArena) that
    alternate, and a plan's flags and hot-run counts come back through an asynchronous copy into pinned memory that is read
    when the plan's first step is queued — by then the build has long run, one chunk ahead (BatchPlan.finish)."""

    # Queue the build of the next plan behind the first N triplets' worth of steps of the current plan instead of behind all
    # of them (0 = off, the default).  It removes a bubble when queueing launches is slow — under rocprofv3 the host needs as
    # long to queue a 64-step chunk of the two-launch stream as the GPU to run it, and the build then starts when the steps end
    # (330 us of idle step stream per chunk in such a trace) — but on an unburdened host it costs: the driver's 20-step
    # command, same box, four runs each: off 30.3-32.8 us/step, 8 steps first 33.1-35.7, 16 steps first 31.5-37.3 (the build
    # then runs beside more of the steps, and the chunk's steps go out in two calls).
    PREFETCH_AFTER_TRIPLETS = 0
    PLAN_TRIPLETS = 1 << 22
    GROUP_MIN_ROWS_PER_TRIPLET = 12   # group plans (no per-batch sort): both tables hold at least this many rows per triplet of a
                                      # batch — with fewer, the lists of shared rows outgrow their capacity (uniform ids: a
                                      # share 1 - exp(-2 B / rows) of the item occurrences is shared: 15 % at 12 rows per triplet)
    GROUP_MIN_BATCH = 4096            # below this the step is launch-bound either way and the sorted plan's two tiny launches win

    CHAIN_MIN_BATCH = 8192        # below this the step is launch-bound: nothing to hide the item phase behind
    CHAIN_MIN_ITEMS_PER_TRIPLET = 6   # item rows per triplet of a batch: with fewer, too many runs are deferred (uniform ids:
                                      # 2 * (1 - exp(-x)(1 + x)), x = 2 B / rows, of the runs: 1/20 at 6 rows per triplet)

    def __init__(self, chunk=64, min_triplets=None, chain=True, inline_plan=False, group=True):
        """group (default): batches that qualify (int32 ids; 4,096 <= B <= 131,072; rows = whole 128-B lines; both tables large
        against the batch; no popularity-skewed ids) are NOT sorted at all: the plan only groups the rows that recur in a batch
        (GroupPlan, wr_group_plan_build: one launch per chunk, LDS bitmaps, ~0.1 B of plan per triplet) and the step is one
        launch (wr_bprmf_run_sgd_group).  A chunk whose lists overflow is re-planned with the sorted builder, and the stream
        stays with sorted plans from there on; so it does after a chunk that reports a long run.
        chain: steps of plans that qualify (no hot rows; B >= CHAIN_MIN_BATCH; rows = whole 128-B lines; item
        table large against the batch) go out as ONE launch per step — the item phase of step k-1 inside the launch of step
        k's user phase (wr_bprmf_run_sgd_chain; same tables bit for bit; MI355X, 1M x 1M x 64, B = 65,536, steps only:
        27.3 -> 24.7 us/step).
        inline_plan: build the plan of the NEXT chunk on the step stream itself, between the two halves of the current
        chunk's steps, instead of on a side stream beside them.  OFF by default — measured, not assumed: in a kernel trace of
        1M x 1M x 64, B = 65,536 the ~11 step kernels per chunk that overlap a plan kernel take 49 us instead of 24 (~280 us
        per chunk), which looks like a reason to serialise; but the plan of a chunk takes ~330 us when it runs alone (it is
        latency-bound: 32 K short workgroups), so in-stream costs MORE (29.9 against 29.4 us/step)."""
        import sys
        self.ops = sys.modules[__name__]
        self.chunk = int(chunk)
        self.chain = bool(chain)
        self.group = bool(group)
        self.inline_plan = inline_plan
        self.stats = {"plain_calls": 0, "chain_calls": 0, "group_calls": 0, "group_fallbacks": 0}
        self._garenas = {}
        if min_triplets is not None:
            self.PLAN_TRIPLETS = int(min_triplets)
        self.plan_stream = None
        self._arenas = {}

    def chunk_batches(self, batch_size):
        return max(self.chunk, self.PLAN_TRIPLETS // max(int(batch_size), 1))

    def _arena_pair(self, device, B, nb_total, overlap_items):
        """two arenas for plans of up to chunk_batches(B) batches of B triplets (never more than the work at hand), kept
        across epochs"""
        cap = min(self.chunk_batches(B), max(int(nb_total), 1)) * B
        key = (str(device), B, int(overlap_items))
        pair = self._arenas.get(key)
        if pair is None or pair[0].max_triplets < cap:
            pair = [self.ops.PlanArena(device, cap, B, overlap_items=overlap_items) for _ in range(2)]
            self._arenas[key] = pair
        return pair

    def _garena_pair(self, device, B, nb_total, n_users, n_items):
        cap = min(self.chunk_batches(B), max(int(nb_total), 1)) * B
        key = (str(device), B, int(n_users), int(n_items))
        pair = self._garenas.get(key)
        if pair is None or pair[0].max_triplets < cap:
            pair = [self.ops.GroupArena(device, cap, B, n_users, n_items) for _ in range(2)]
            self._garenas[key] = pair
        return pair

    def plan(self, U, segments, batch, first_chunk=None, lr=None, prep=None, runner=None):
        """segments: [(item rows view [rows, D], u, p, n)] — u rows of U, p and n rows of the view, in batch order.
        Only the last segment may end with a short batch.  first_chunk: batches in the first plan, or a list with the sizes
        of the first few plans (default: full chunks) — lets a caller that consumes the stream piecewise (bench.py: warm-up,
        then timed steps) put a plan boundary where its pieces meet.
        runner: callable(plan, first, count, losses) that issues `count` steps of `plan` from batch `first` — for updates
        other than plain SGD (the exact-lazy Adam / weight-decay steps, Adagrad, Adadelta: FusedOptimizer.run_batches); the
        plan pipeline (arenas, builds one chunk ahead on the side stream, asynchronous read-back) is the same."""
        if self.plan_stream is None:
            self.plan_stream = side_stream(U.device)
        B = int(batch)
        segs, first = [], 0
        for k, (rows, u, p, n) in enumerate(segments):
            nb = (u.numel() + B - 1) // B
            if u.numel() % B != 0 and any(s[1].numel() for s in segments[k + 1:]):
                raise ValueError("only the last segment may end with a short batch")
            segs.append({"tabs": self.ops.BprmfTables(U, rows) if nb else None, "nb": nb, "first": first})
            first += nb
        live = [s for s in segments if s[1].numel()]
        # one batch sequence across the segments: plan chunks are cut by size, not at segment ends (a plan holds indices
        # only; with short segments — 8-GPU strata of a few steps — that is 5x fewer plans and host syncs).  Segments that
        # are consecutive views of one array (the usual case: slices of an epoch's arrays) are joined without a copy.
        u_all, p_all, n_all = (_join_views([s[j] for s in live]) for j in (1, 2, 3))
        n_items = max([s[0].shape[0] for s in live] or [1])
        main = torch.cuda.current_stream(U.device)
        use_chain = self.chain and (runner is None or getattr(runner, "wants_chain_marks", False)) and \
            B >= self.CHAIN_MIN_BATCH and first >= 2 and \
            min([s[0].shape[0] for s in live] or [0]) >= self.CHAIN_MIN_ITEMS_PER_TRIPLET * B and \
            all(sg["tabs"].chain_supported() for sg in segs if sg["tabs"] is not None)
        min_rows = min([U.shape[0]] + [s[0].shape[0] for s in live])
        use_group = self.group and runner is None and len(live) == 1 and first >= 2 and \
            self.GROUP_MIN_BATCH <= B <= 131072 and u_all.dtype in (torch.int32, torch.int64) and \
            min_rows >= self.GROUP_MIN_ROWS_PER_TRIPLET * B and \
            all(sg["tabs"].group_supported() for sg in segs if sg["tabs"] is not None)
        garenas = self._garena_pair(U.device, B, first, U.shape[0], n_items) if use_group else None
        # the sorted plans' arenas are made on demand when a group-plan stream has to fall back (skewed ids)
        arenas = None if use_group else self._arena_pair(U.device, B, first, n_items if use_chain else 0)
        for a in (arenas or []) + (garenas or []):     # a previous handle may have left steps queued that read these arrays
            a.release_after(main)
        h = {"segs": segs, "B": B, "u": u_all, "p": p_all, "n": n_all, "nb": first, "n_users": U.shape[0], "n_items": n_items,
             "at": 0, "tag": 0, "next": None, "cur": None, "map": None, "arenas": arenas, "pos": 0,
             "group": use_group, "garenas": garenas, "device": U.device,
             # int64 columns (the reference's batch layout): the group plan and its step read int32 shadows, narrowed chunk by
             # chunk on the plan stream (12 B per triplet, once per epoch; the sorted plans read the int64 columns themselves)
             "ids32": [torch.empty(u_all.numel(), dtype=torch.int32, device=U.device) for _ in range(3)]
                      if use_group and u_all.dtype == torch.int64 else None,
             "chain": use_chain, "prep": prep, "runner": runner, "inline": bool(self.inline_plan),
             "lead": [int(c) for c in (first_chunk if isinstance(first_chunk, (list, tuple)) else [first_chunk or 0]) if c]}
        self.plan_stream.wait_stream(main)   # the index tensors are ready
        self._prefetch(h)
        return h

    def _build_stream(self, h):
        return torch.cuda.current_stream(h["u"].device) if h["inline"] else self.plan_stream

    def _prefetch(self, h):
        """enqueue the plan of the next chunk of batches on the side stream (no host wait, no allocation)"""
        if h["at"] >= h["nb"]:
            h["next"] = None
            return
        first, B = h["at"], h["B"]
        c = min(self.chunk_batches(B), h["nb"] - first)
        if h["lead"]:
            c = min(c, h["lead"].pop(0))
        h["at"] += c
        lo, hi = first * B, min(h["u"].numel(), (first + c) * B)
        with torch.cuda.stream(self._build_stream(h)):
            bmap = h["map"] if h["map"] else None
            if h["prep"] is not None:
                h["prep"].fill(lo, hi)       # this chunk's rows: shuffle + negatives, on the plan stream, before its plan
            if h["group"]:
                cols = (h["u"], h["p"], h["n"])
                if h["ids32"] is not None:
                    abi.check(abi.lib().wr_narrow_ids_i64(*[_p(c[lo:hi]) for c in cols], *[_p(c[lo:hi]) for c in h["ids32"]],
                                                          hi - lo, _stream()), "wr_narrow_ids_i64")
                    cols = h["ids32"]
                plan = self.ops.GroupPlan(cols[0][lo:hi], cols[1][lo:hi], cols[2][lo:hi], B, h["n_users"], h["n_items"],
                                          arena=h["garenas"][h["tag"]], defer=True)
                h["tag"] ^= 1
                h["next"] = (first, plan)
                return
            if h["arenas"] is None:          # first sorted plan of a stream that started with group plans
                h["arenas"] = self._arena_pair(h["device"], B, h["nb"], h["n_items"] if h["chain"] else 0)
            plan = self.ops.BatchPlan(h["u"][lo:hi], h["p"][lo:hi], h["n"][lo:hi], B, h["n_users"], h["n_items"],
                                      validate=False, ws_tag="rot%d" % h["tag"], bucket_map=bmap, arena=h["arenas"][h["tag"]],
                                      defer=True, overlap=h["chain"])
        h["tag"] ^= 1
        h["next"] = (first, plan)

    STICKY_MSG = "a wait inside a step launch expired (hand-off between the workgroups of one launch): the tables are not " \
                 "valid — do not save or evaluate them; restart the run"

    def _queue_sticky(self, h, tabs, main):
        """behind a chunk's last step: the launches' sticky timeout words -> pinned memory, asynchronously; read when the next
        chunk starts (or the run ends), i.e. before anything but ONE more chunk of steps can have used the tables"""
        words = tabs.sticky_words()
        if not words:
            return
        pin = h.get("sticky_pin")
        if pin is None or pin.numel() < 2 * len(words):
            pin = h["sticky_pin"] = torch.zeros(2 * max(len(words), 4), dtype=torch.int32, pin_memory=True)
        half = pin.numel() // 2
        off = half * (h.get("sticky_tag", 0) & 1)
        h["sticky_tag"] = h.get("sticky_tag", 0) + 1
        for j, wd in enumerate(words):
            pin[off + j:off + j + 1].copy_(wd, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(main)
        self._check_sticky(h)                       # at most one chunk's check is ever outstanding
        h["sticky_pending"] = (ev, pin[off:off + len(words)])

    def _check_sticky(self, h):
        pend = h.pop("sticky_pending", None)
        if pend is not None:
            pend[0].synchronize()
            if bool(pend[1].numpy().any()):
                raise abi.WhisprRecHipError(self.STICKY_MSG)

    def _take_next(self, h, pos, main):
        """the prefetched plan becomes the current one: its read-back is consumed here (flags, hot-run counts)"""
        self._check_sticky(h)
        if h["next"] is None:
            self._prefetch(h)
        cur = h["next"]
        assert cur is not None and cur[0] == pos, "steps must be run in order"
        plan = cur[1]
        plan.validate()                      # finish(): waits for the build's event only if the build has not run yet
        if isinstance(plan, GroupPlan):
            if plan.overflow or plan.long_run:
                # ids the group plan is not made for (tables small against the batch, popularity skew): sorted plans from
                # here on; after an overflow this chunk is re-planned too (its lists are incomplete)
                h["group"] = False
                self.stats["group_fallbacks"] += 1
                if h["next"] is not None and h["next"][0] != cur[0]:
                    raise AssertionError("plan pipeline out of step")
                if plan.overflow:
                    first, B = cur[0], h["B"]
                    lo, hi = first * B, first * B + plan.n_triplets
                    plan.arena.release_after(main)
                    if h["arenas"] is None:
                        h["arenas"] = self._arena_pair(h["device"], B, h["nb"], h["n_items"] if h["chain"] else 0)
                    with torch.cuda.stream(self._build_stream(h)):
                        plan = self.ops.BatchPlan(h["u"][lo:hi], h["p"][lo:hi], h["n"][lo:hi], B, h["n_users"], h["n_items"],
                                                  validate=False, ws_tag="rot%d" % h["tag"], arena=h["arenas"][h["tag"]],
                                                  defer=True, overlap=h["chain"])
                    h["tag"] ^= 1
                    plan.validate()
                    cur = (first, plan)
            main.wait_event(plan.ready)
            h["cur"], h["next"] = cur, None
            return cur
        main.wait_event(plan.ready)
        if plan.fast_overflowed:
            # skewed ids: equal-width buckets overflow chunk after chunk.  Balance the buckets by the rows' share of
            # the epoch (one pass over the id columns) and use them from the next chunk on; if even those overflow
            # (batches that are not random samples of the epoch), stay with the radix-sort builder.
            if plan._mapped():
                h["map"] = False
            elif h["map"] is None:
                with torch.cuda.stream(self._build_stream(h)):
                    # row shares from the whole epoch: with a pipelined preparation the batch-order columns are not
                    # complete yet, the source columns have the same rows in another order
                    src = (h["prep"].users, h["prep"].items) if h["prep"] is not None else (h["u"], h["p"])
                    h["map"] = self.ops.BucketMap(src[0], src[1], h["n_users"], h["n_items"], h["B"])
        if h["chain"] and plan.hot is not None:
            h["chain"] = False      # skewed ids stay skewed: the following plans skip the marks (hot batches take the two launches)
        if h["chain"] and plan.hot is None and plan.overlap is not None and not plan.overlap["fits"]:
            # ids that defer most runs (many shared item rows without any hot one): the marks are wasted plan work
            dcn = plan.overlap["def_count_np"]
            if int((dcn[1:] > plan.overlap["cap"]).sum()) * 2 > plan.n_batches - 1:
                h["chain"] = False
        h["cur"], h["next"] = cur, None
        return cur

    def _run_span(self, h, sg, pos, end, lr, losses, loss_off):
        """steps [pos, end) of the handle's batch sequence, all inside segment `sg`; losses[loss_off + k] for step pos + k"""
        main = torch.cuda.current_stream(sg["tabs"].U.device)
        while pos < end:
            cur = h["cur"]
            if cur is None or pos >= cur[0] + cur[1].n_batches:
                cur = self._take_next(h, pos, main)
            base, plan = cur
            c = min(end, base + plan.n_batches) - pos
            if h["inline"] and h["next"] is None and h["at"] < h["nb"]:
                # in-stream plan build: the next chunk's plan goes between the two halves of this chunk's steps
                mid = base + plan.n_batches // 2
                if pos >= mid:
                    self._prefetch(h)
                else:
                    c = min(c, mid - pos)
            elif self.PREFETCH_AFTER_TRIPLETS > 0 and h["next"] is None and h["at"] < h["nb"] and \
                    c > (self.PREFETCH_AFTER_TRIPLETS + h["B"] - 1) // h["B"]:
                c = (self.PREFETCH_AFTER_TRIPLETS + h["B"] - 1) // h["B"]      # see PREFETCH_AFTER_TRIPLETS
            if isinstance(plan, GroupPlan):
                sg["tabs"].run_sgd_group(plan, pos - base, c, lr, losses=losses[loss_off:loss_off + c])
                self.stats["group_calls"] += 1
            elif h["runner"] is not None:
                h["runner"](plan, pos - base, c, losses[loss_off:loss_off + c])
                self.stats["plain_calls"] += 1
            elif h["chain"] and plan.overlap is not None and c >= 2:
                sg["tabs"].run_sgd_chain(plan, pos - base, c, lr, losses=losses[loss_off:loss_off + c])
                self.stats["chain_calls"] += 1
            else:
                sg["tabs"].run_sgd(plan, pos - base, c, lr, losses=losses[loss_off:loss_off + c])
                self.stats["plain_calls"] += 1
            pos += c
            loss_off += c
            if pos >= base + plan.n_batches:
                plan.arena.release_after(main)                       # every step that reads the plan's arrays is queued
                self._queue_sticky(h, sg["tabs"], main)
            if h["next"] is None and not h["inline"]:
                self._prefetch(h)                                    # the next plan is built beside the queued steps
        h["pos"] = pos

    def run(self, handle, seg, lr, losses):
        """all steps of segment `seg` (segments must be run in order); losses: one slot per step"""
        sg = handle["segs"][seg]
        if sg["nb"] == 0:
            return
        self._run_span(handle, sg, sg["first"], sg["first"] + sg["nb"], lr, losses, 0)
        if seg == len(handle["segs"]) - 1:
            self._check_sticky(handle)      # the last chunk's word (waits for the stream: the caller is about to read the losses)

    def run_steps(self, handle, count, lr, losses):
        """the next `count` steps of a single-segment handle (a step stream consumed piecewise: warm-up, then the timed
        steps of bench.py — the plan pipeline runs on across the calls, as it does inside an epoch)"""
        assert len(handle["segs"]) == 1, "run_steps: single-segment handles"
        sg = handle["segs"][0]
        pos = handle["pos"]
        if pos + count > sg["nb"]:
            raise ValueError("run_steps: %d steps asked, %d left" % (count, sg["nb"] - pos))
        self._run_span(handle, sg, pos, pos + count, lr, losses, 0)


class LazyOptimizerState:
    """Exact lazy evaluation of torch.optim.Adam / SGD(weight_decay) on the two BPRMF tables (include/whisprrec_hip.h,
    "K5 (lazy, exact)"): per row the number of the last optimizer step applied to it; rows are replayed when a batch
    needs them and all together in ``flush()`` — call it before anything else reads the tables.  Bit-identical to the
    dense kernels (wr_adam_dense / wr_sgd_dense), without their table passes."""

    FOLD_MAX_GAP = 40      # fold the catch-up into the step kernels while a row misses about this many steps between two uses

    MAX_LAG = 64           # bounded lag (Adam): rows of a rotating window are replayed every step so that none misses more
    LAG_MIN_GAP = 64       # ... used when a row misses more than this many steps between two uses on average (rows / batch)

    def __init__(self, tabs, name, lr, l2, betas=(0.9, 0.999), eps=1e-8, fold=None, max_lag=None):
        """fold (Adam): the catch-up of a batch's rows happens inside the step kernels' row loads (wr_bprmf_step_adam_folded:
        6 instead of 12 row transfers per touched row) instead of in a pass of its own; same bits either way.  The folded
        replay is balanced over the four rows a wave holds (adam_replay_balanced: the wave transposes its data so that all
        64 lanes replay one row at a time — no team waits for the longest gap), but a row shared by several triplets is
        replayed by each of its readers, and with few waves on the GPU the replay is latency-bound, so it pays while replays are
        short: None (default) = fold when a row misses about FOLD_MAX_GAP steps or fewer between two batches that contain it
        (rows / batch size).  MI355X, 1M x 1M x 64, us/step separate -> folded: B = 65,536 (gap ~16) 155 -> 105; B = 32,768
        (~31) 105 -> 84; B = 16,384 (~61) 73 -> 91; B = 2,048 (~490) 55 -> 283 (scripts/ab_adam.py).
        max_lag (Adam, separate catch-up): None (default) = MAX_LAG when a row misses more than LAG_MIN_GAP steps between
        two uses on average, else unbounded; 0 = unbounded; n = before every step a rotating window of rows / n rows per
        table is brought up to date, so that no row ever misses more than n steps (wr_bprmf_run_adam_lazy_bounded).  Missed
        steps are geometrically distributed: at B = 2,048 on 1M-row tables the mean is ~490 and the longest among a batch's
        rows ~3,700 — one wave's serial chain that the whole launch waits for (steady state 206 us/step; the 44-56 us of
        short runs only hold until rows have been idle for long).  The window does the replays every row is owed anyway as
        uniform-length work over thousands of waves.  Same bits.  MI355X, 1M x 1M x 64, whole epochs in steady state
        (scripts/exp/adam_epoch.py), us/step at max_lag = unbounded / 1024 / 512 / 256 / 128 / 64 / 32 / 16: B = 2,048: 206 /
        151 / 96 / 76 / 66 / 61.6 / 62.4 / 78 (the window's row traffic grows as the lag shrinks); B = 8,192: - / 133 / 109 /
        - / 80 / 72.5 / 73.5 / 89; B = 16,384 (a row misses ~61 steps): 88 either way."""
        self.chain, self.chain_calls = True, 0     # folded Adam steps as one launch per step where the plan carries the marks
        if name not in ("SGD", "Adam"):
            raise ValueError(name)
        self.tabs, self.name, self.lr, self.l2, self.betas, self.eps = tabs, name, float(lr), float(l2), betas, float(eps)
        self.fold = fold if fold is None else bool(fold)
        self.max_lag = max_lag if max_lag is None else int(max_lag)
        self._sweep_pos = (ctypes.c_int64 * 2)(0, 0)
        dev = tabs.dev
        self.t = 0                                                       # optimizer steps taken
        self.flushed_at = 0
        self.last_u = torch.zeros(tabs.U.shape[0], dtype=torch.int32, device=dev)
        self.last_i = torch.zeros(tabs.I.shape[0], dtype=torch.int32, device=dev)
        if name == "Adam":
            z = torch.zeros_like
            self.m_u, self.v_u, self.m_i, self.v_i = z(tabs.U), z(tabs.U), z(tabs.I), z(tabs.I)
            self.consts = None
            self._grow_consts(4096)

    def _folds(self, plan):
        if self.fold is None:
            gap = max(self.tabs.U.shape[0], self.tabs.I.shape[0]) / float(max(plan.batch_size, 1))
            return gap <= self.FOLD_MAX_GAP
        return self.fold

    def _lag(self, plan):
        if self.max_lag is None:
            gap = max(self.tabs.U.shape[0], self.tabs.I.shape[0]) / float(max(plan.batch_size, 1))
            return self.MAX_LAG if gap > self.LAG_MIN_GAP else 0
        return self.max_lag

    def _grow_consts(self, n):
        host = torch.empty(2 * n, dtype=torch.float32)
        abi.check(abi.lib().wr_adam_consts(0, n, self.lr, self.betas[0], self.betas[1], host.data_ptr()), "wr_adam_consts")
        self.consts = host.to(self.tabs.dev)
        self.n_consts = n

    def _adam_rows(self, tab, m, v, last, keys_ptr, n_keys, grad):
        abi.check(abi.lib().wr_adam_rows_lazy(_p(tab), _p(m), _p(v), _p(last), tab.shape[0], tab.shape[1], keys_ptr, n_keys,
                                              _p(grad), self.t, _p(self.consts), self.n_consts, self.l2, self.betas[0],
                                              self.betas[1], self.eps, _stream()), "wr_adam_rows_lazy")

    def step(self, plan, k, loss_out=None):
        """optimizer step on batch k of the plan (gradient computation included); returns the loss tensor"""
        tabs = self.tabs
        if self._lag(plan) > 0 and not (self.name == "Adam" and self._folds(plan)):
            out = self.run(plan, k, 1, losses=None if loss_out is None else loss_out.reshape(1))   # keeps the window turning
            return out.reshape(()) if loss_out is None else loss_out
        self.t += 1
        tu, _, _, oi, _, B = tabs._plan_ptrs(plan, k)
        if self.name == "Adam":
            if self.t >= self.n_consts:
                self._grow_consts(2 * self.n_consts)
            if self._folds(plan) and plan.hot_struct(k) is None:
                if loss_out is None:
                    loss_out = torch.empty((), dtype=torch.float32, device=tabs.dev)
                _, tp, tn, _, os_, _ = tabs._plan_ptrs(plan, k)
                ws = tabs._ws(plan.batch_size)
                abi.check(abi.lib().wr_bprmf_step_adam_folded(
                    _p(tabs.U), tabs.U.shape[0], _p(tabs.I), tabs.I.shape[0], tabs.D, _p(self.m_u), _p(self.v_u), _p(self.m_i),
                    _p(self.v_i), _p(self.last_u), _p(self.last_i), tu, tp, tn, oi, os_, B, self.t, self.lr, _p(self.consts),
                    self.n_consts, self.l2, self.betas[0], self.betas[1], self.eps, _p(loss_out), _p(ws), ws.numel(), _stream()),
                    "wr_bprmf_step_adam_folded")
                tabs.step_id += 1
                return loss_out
            self._adam_rows(tabs.U, self.m_u, self.v_u, self.last_u, tu, B, None)       # the batch's rows up to t-1
            self._adam_rows(tabs.I, self.m_i, self.v_i, self.last_i, oi, 2 * B, None)
            # gradients + Adam on the rows the step kernels finish (no gradient table)
            if loss_out is None:
                loss_out = torch.empty((), dtype=torch.float32, device=tabs.dev)
            _, tp, tn, _, os_, _ = tabs._plan_ptrs(plan, k)
            ws = tabs._ws(plan.batch_size)
            hot = plan.hot_struct(k)
            abi.check(abi.lib().wr_bprmf_step_adam(
                _p(tabs.U), tabs.U.shape[0], _p(tabs.I), tabs.I.shape[0], tabs.D, _p(self.m_u), _p(self.v_u), _p(self.m_i),
                _p(self.v_i), _p(self.last_u), _p(self.last_i), tu, tp, tn, oi, os_, B, self.t, self.lr, self.l2,
                self.betas[0], self.betas[1], self.eps, _p(loss_out), ctypes.addressof(hot) if hot is not None else None,
                _p(ws), ws.numel(), _stream()), "wr_bprmf_step_adam")
            tabs.step_id += 1
            return loss_out
        L = abi.lib()
        for tab, last, keys, n in ((tabs.U, self.last_u, tu, B), (tabs.I, self.last_i, oi, 2 * B)):
            abi.check(L.wr_sgd_rows_lazy(_p(tab), _p(last), tab.shape[0], tab.shape[1], keys, n, self.t, self.lr, self.l2,
                                         _stream()), "wr_sgd_rows_lazy")
        return tabs.step_sgd(plan, k, self.lr, self.l2, loss_out=loss_out, decay_untouched=False)

    def run(self, plan, first, count, losses=None):
        """`count` consecutive optimizer steps on batches [first, first + count) of the plan, issued from native code
        (wr_bprmf_run_adam_lazy / wr_bprmf_run_sgd_lazy); same result as `count` calls of step()"""
        tabs, L = self.tabs, abi.lib()
        if losses is None:
            losses = torch.empty(count, dtype=torch.float32, device=tabs.dev)
        ws = tabs._ws(plan.batch_size)
        su, si = tabs._stamps()
        hot = plan.hot_struct()
        hp = ctypes.addressof(hot) if hot is not None else None
        t0 = self.t + 1
        if self.name == "Adam":
            while self.t + count >= self.n_consts:
                self._grow_consts(2 * self.n_consts)
            head = (_p(tabs.U), tabs.U.shape[0], _p(tabs.I), tabs.I.shape[0], tabs.D, _p(self.m_u), _p(self.v_u), _p(self.m_i),
                    _p(self.v_i), _p(self.last_u), _p(self.last_i),
                    _p(plan.tu), _p(plan.tp), _p(plan.tn), _p(plan.oc_item), _p(plan.oc_src), plan.n_triplets, plan.batch_size,
                    first, count, t0, self.lr, _p(self.consts), self.n_consts, self.l2, self.betas[0], self.betas[1], self.eps,
                    _p(losses), hp)
            lag = self._lag(plan)
            o = getattr(plan, "overlap", None)
            if self._folds(plan) and self.chain and o is not None and hot is None and count >= 2 and tabs.chain_supported():
                # one launch per step: the item phase of step k-1 inside the launch of step k's user phase
                ws2 = tabs.overlap_workspace(plan.batch_size)
                sync = tabs._chain_sync(count)
                abi.check(L.wr_bprmf_run_adam_folded_chain(*head[:-1], _p(o["tdef"]), _p(o["def_q"]),
                                                           o["def_count_host"].data_ptr(), o["cap"], o["cap"], _p(ws2),
                                                           ws2.numel(), _p(sync), sync.numel(), _stream()),
                          "wr_bprmf_run_adam_folded_chain")
                self.chain_calls += 1
            elif self._folds(plan):
                abi.check(L.wr_bprmf_run_adam_folded(*head, _p(ws), ws.numel(), _stream()), "wr_bprmf_run_adam_folded")
            elif lag > 0:
                abi.check(L.wr_bprmf_run_adam_lazy_bounded(*head, lag, ctypes.addressof(self._sweep_pos), _p(ws), ws.numel(),
                                                           _stream()), "wr_bprmf_run_adam_lazy_bounded")
            else:
                abi.check(L.wr_bprmf_run_adam_lazy(*head, _p(ws), ws.numel(), _stream()), "wr_bprmf_run_adam_lazy")
        else:
            head = (_p(tabs.U), tabs.U.shape[0], _p(tabs.I), tabs.I.shape[0], tabs.D, _p(self.last_u), _p(self.last_i), _p(su),
                    _p(si), tabs.step_id + 1, _p(plan.tu), _p(plan.tp), _p(plan.tn), _p(plan.oc_item), _p(plan.oc_src),
                    plan.n_triplets, plan.batch_size, first, count, t0, self.lr, self.l2, _p(losses), hp)
            lag = self._lag(plan)
            if lag > 0:
                abi.check(L.wr_bprmf_run_sgd_lazy_bounded(*head, lag, ctypes.addressof(self._sweep_pos), _p(ws), ws.numel(),
                                                          _stream()), "wr_bprmf_run_sgd_lazy_bounded")
            else:
                abi.check(L.wr_bprmf_run_sgd_lazy(*head, _p(ws), ws.numel(), _stream()), "wr_bprmf_run_sgd_lazy")
        self.t += count
        tabs.step_id += count
        return losses

    def flush(self):
        """every row up to the current step: the tables are then what the dense optimizer would hold"""
        if self.flushed_at == self.t:
            return
        L, tabs = abi.lib(), self.tabs
        if self.name == "Adam":
            for tab, m, v, last in ((tabs.U, self.m_u, self.v_u, self.last_u), (tabs.I, self.m_i, self.v_i, self.last_i)):
                abi.check(L.wr_adam_catchup_all(_p(tab), _p(m), _p(v), _p(last), tab.shape[0], tab.shape[1], self.t,
                                                _p(self.consts), self.n_consts, self.l2, self.betas[0], self.betas[1],
                                                self.eps, _stream()), "wr_adam_catchup_all")
        else:
            for tab, last in ((tabs.U, self.last_u), (tabs.I, self.last_i)):
                abi.check(L.wr_sgd_catchup_all(_p(tab), _p(last), tab.shape[0], tab.shape[1], self.t, self.lr, self.l2,
                                               _stream()), "wr_sgd_catchup_all")
        self.flushed_at = self.t


class StatefulSparseState:
    """torch.optim.Adagrad / Adadelta (weight_decay = 0) fused into the step kernels (wr_bprmf_run_stateful): with a zero
    gradient neither moves a weight, so only the rows of a batch are touched and the tables are always current (nothing to
    flush).  Adagrad's state (state_sum) is exactly sparse; Adadelta's two state rows decay by rho at every step — the
    missed decays of a row are replayed when it is next updated (``last`` = step of its last update)."""

    KIND = {"Adagrad": 1, "Adadelta": 2}

    MAX_LAG, LAG_MIN_GAP = 64, 64       # Adadelta: bounded lag as in LazyOptimizerState

    def __init__(self, tabs, name, lr, rho=0.9, eps=None, max_lag=None):
        """max_lag (Adadelta): a rotating window of rows / max_lag rows per table takes its missed decays before every step
        (wr_bprmf_run_stateful_bounded) — None = MAX_LAG when a row misses more than LAG_MIN_GAP steps on average, 0 = never."""
        if name not in self.KIND:
            raise ValueError(name)
        self.max_lag = max_lag if max_lag is None else int(max_lag)
        self._sweep_pos = (ctypes.c_int64 * 2)(0, 0)
        self.tabs, self.name, self.lr, self.rho = tabs, name, float(lr), float(rho)
        self.eps = float(eps) if eps is not None else (1e-10 if name == "Adagrad" else 1e-6)   # torch.optim defaults
        z = torch.zeros_like
        self.s1_u, self.s1_i = z(tabs.U), z(tabs.I)
        self.s2_u = self.s2_i = self.last_u = self.last_i = None
        if name == "Adadelta":
            self.s2_u, self.s2_i = z(tabs.U), z(tabs.I)
            self.last_u = torch.zeros(tabs.U.shape[0], dtype=torch.int32, device=tabs.dev)
            self.last_i = torch.zeros(tabs.I.shape[0], dtype=torch.int32, device=tabs.dev)
        self.t = 0

    def run(self, plan, first, count, losses=None):
        tabs = self.tabs
        if losses is None:
            losses = torch.empty(count, dtype=torch.float32, device=tabs.dev)
        ws = tabs._ws(plan.batch_size)
        hot = plan.hot_struct()
        head = (self.KIND[self.name], _p(tabs.U), tabs.U.shape[0], _p(tabs.I), tabs.I.shape[0], tabs.D, _p(self.s1_u),
                _p(self.s2_u), _p(self.s1_i), _p(self.s2_i), _p(self.last_u), _p(self.last_i), _p(plan.tu), _p(plan.tp), _p(plan.tn),
                _p(plan.oc_item), _p(plan.oc_src), plan.n_triplets, plan.batch_size, first, count, self.t + 1, self.lr, self.rho,
                self.eps, _p(losses), ctypes.addressof(hot) if hot is not None else None)
        lag = self.max_lag
        if lag is None:
            gap = max(tabs.U.shape[0], tabs.I.shape[0]) / float(max(plan.batch_size, 1))
            lag = self.MAX_LAG if gap > self.LAG_MIN_GAP else 0
        if self.name == "Adadelta" and lag > 0:
            abi.check(abi.lib().wr_bprmf_run_stateful_bounded(*head, lag, ctypes.addressof(self._sweep_pos), _p(ws), ws.numel(),
                                                              _stream()), "wr_bprmf_run_stateful_bounded")
        else:
            abi.check(abi.lib().wr_bprmf_run_stateful(*head, _p(ws), ws.numel(), _stream()), "wr_bprmf_run_stateful")
        self.t += count
        tabs.step_id += count
        return losses

    def step(self, plan, k, loss_out=None):
        out = self.run(plan, k, 1, losses=None if loss_out is None else loss_out.reshape(1))
        return out.reshape(()) if loss_out is None else loss_out

    def flush(self):
        """nothing to do: the weights never lag (kept so that callers treat every optimizer state alike)"""


def adam_consts(n, lr, beta1=0.9, beta2=0.999, device=None):
    """device table of the per-step Adam constants for steps 0..n-1 (wr_adam_consts: the host expressions of wr_adam_dense)"""
    host = torch.empty(2 * n, dtype=torch.float32)
    abi.check(abi.lib().wr_adam_consts(0, n, lr, beta1, beta2, host.data_ptr()), "wr_adam_consts")
    return host.to(device) if device is not None else host


def adam_dense_dev(tab, exp_avg, exp_avg_sq, grad, consts, step_dev, l2=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """wr_adam_dense with the step number read from device memory (graph-replayable)"""
    for t, nm in ((tab, "tab"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (grad, "grad")):
        _req(t, torch.float32, nm, 2)
    abi.check(abi.lib().wr_adam_dense_dev(_p(tab), _p(exp_avg), _p(exp_avg_sq), tab.shape[0], tab.shape[1], _p(grad),
                                          _p(consts), consts.numel() // 2, _p(step_dev), l2, beta1, beta2, eps, _stream()),
              "wr_adam_dense_dev")


def adam_dense_dev_pair(a, b, consts, step_dev, l2=0.0, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_dense_dev for two tables of the same width in one launch; a, b: (tab, exp_avg, exp_avg_sq, grad)"""
    for grp in (a, b):
        for t, nm in zip(grp, ("tab", "exp_avg", "exp_avg_sq", "grad")):
            _req(t, torch.float32, nm, 2)
    if a[0].shape[1] != b[0].shape[1]:
        raise ValueError("adam_dense_dev_pair: tables of different width")
    abi.check(abi.lib().wr_adam_dense_dev_pair(_p(a[0]), _p(a[1]), _p(a[2]), a[0].shape[0], _p(a[3]), _p(b[0]), _p(b[1]), _p(b[2]),
                                               b[0].shape[0], _p(b[3]), a[0].shape[1], _p(consts), consts.numel() // 2,
                                               _p(step_dev), l2, beta1, beta2, eps, _stream()), "wr_adam_dense_dev_pair")


def counter_add(counter, delta=1):
    abi.check(abi.lib().wr_counter_add(_p(_req(counter, torch.int32, "counter", 1)), int(delta), _stream()), "wr_counter_add")


# ----------------------------------------------------------------------------------------------- rows
def gather_rows(tab, idx):
    """nn.Embedding forward: out[..., :] = tab[idx[...], :]."""
    _req(tab, torch.float32, "tab", 2)
    flat = _idx64(idx.reshape(-1), "idx")
    out = torch.empty((flat.numel(), tab.shape[1]), dtype=torch.float32, device=tab.device)
    abi.check(abi.lib().wr_gather_rows(_p(tab), tab.shape[0], tab.shape[1], _p(flat), flat.numel(), _p(out), _stream()),
              "wr_gather_rows")
    return out.reshape(tuple(idx.shape) + (tab.shape[1],))


def scatter_add_rows(grad, idx, src, padding_idx=-1, alpha=1.0):
    """embedding_dense_backward: grad[idx[k], :] += alpha * src[k, :] for idx[k] != padding_idx (in place)."""
    _req(grad, torch.float32, "grad", 2)
    flat = _idx64(idx.reshape(-1), "idx")
    src2 = _req(src.reshape(flat.numel(), grad.shape[1]).contiguous(), torch.float32, "src", 2)
    if flat.numel() == 0:
        return grad
    L = abi.lib()
    nbytes = abi.check_size(L.wr_scatter_add_workspace_bytes(flat.numel(), grad.shape[0]),
                            "wr_scatter_add_workspace_bytes")
    ws = workspace(grad.device, "scatter").get(nbytes)
    abi.check(L.wr_scatter_add_rows(_p(grad), grad.shape[0], grad.shape[1], _p(flat), _p(src2), flat.numel(),
                                    padding_idx, alpha, _p(ws), ws.numel(), _stream()), "wr_scatter_add_rows")
    return grad


class ScatterPlan:
    """Row plan of several scatter-adds whose indices are known ahead (wr_scatter.hip): idx [n_segments, seg_stride] int64
    (segment s uses its first seg_len[s] positions; seg_len int32 on the device, or None = all).  ``apply(table, s, n, src,
    alpha)`` is then ONE launch.  ``slow`` (read with the caller's own read-back of ``meta``): some range of rows was not
    listed and is summed by brute force — exact, slow; callers with such ids stay with scatter_add_rows' sorted path."""

    def __init__(self, idx, n_rows, seg_len=None, padding_idx=-1):
        L = abi.lib()
        _req(idx, torch.int64, "idx", 2)
        self.idx = idx.contiguous()
        self.n_segments, self.stride = self.idx.shape
        self.n_rows, self.padding_idx = int(n_rows), int(padding_idx)
        words = int(L.wr_scatter_plan_words(self.n_segments, self.stride, self.n_rows))
        if words <= 0:
            raise ValueError("row plan not applicable: segments of %d positions" % self.stride)
        if seg_len is not None:
            seg_len = _req(seg_len.contiguous(), torch.int32, "seg_len", 1)
            assert seg_len.numel() == self.n_segments
        self.buf = torch.empty(words, dtype=torch.int32, device=idx.device)
        abi.check(L.wr_scatter_plan_build(_p(self.idx), self.n_segments, self.stride, _p(seg_len), self.n_rows,
                                          self.padding_idx, _p(self.buf), words, _stream()), "wr_scatter_plan_build")
        self.meta = self.buf[:4]

    @property
    def slow(self):
        return bool(int(self.meta[1].item()) != 0)

    def apply(self, table, segment, n, src, alpha=1.0):
        _req(table, torch.float32, "table", 2)
        _req(src, torch.float32, "src", 2)
        assert table.shape[0] == self.n_rows and src.shape[0] >= n and src.shape[1] == table.shape[1] and src.is_contiguous()
        abi.check(abi.lib().wr_scatter_add_planned(_p(table), self.n_rows, table.shape[1], _p(self.idx), self.n_segments,
                                                   self.stride, int(segment), int(n), self.padding_idx, _p(src), float(alpha),
                                                   _p(self.buf), self.buf.numel(), _stream()), "wr_scatter_add_planned")
        return table


# ----------------------------------------------------------------------------------------------- LightGCN pieces
def spmm_csr(row_ptr, col, val, X, Y=None, acc=None):
    _req(row_ptr, torch.int64, "row_ptr", 1)
    _req(col, torch.int32, "col", 1)
    _req(val, torch.float32, "val", 1)
    _req(X, torch.float32, "X", 2)
    if Y is None:
        Y = torch.empty_like(X)
    abi.check(abi.lib().wr_spmm_csr(X.shape[0], _p(row_ptr), _p(col), _p(val), _p(X), X.shape[1], _p(Y), _p(acc),
                                    _stream()), "wr_spmm_csr")
    return Y


SPMM_CHUNK_NNZ = 96              # non-zeros per chunk of the load-balanced CSR product
SPMM_ONE_LEVEL_MAX_CHUNKS = 48   # one combine level while no row has more chunks than this
SPMM_FUSED_COMBINE = False       # True: one-level products in ONE launch (wr_spmm_csr_chunked_fused: the last chunk of a cut
                                 # row to finish combines; D % 32 == 0).  Same bits; measured, not adopted: every chunk of a cut
                                 # row waits for its write-through store and for a returning atomic (~3 us of a wave's ~8 us
                                 # life).  ml-1m-shaped graphs, us per product fused / two launches: 1.41 M non-zeros, chunks of
                                 # 64 / 96 / 128 / 192: 63.4 / 38.7, 48.7 / 34.4, 43.2 / 37.7, 42.5 / 48.1; 0.71 M non-zeros
                                 # (C3 epoch): 35.0 against 35.8 ms per epoch (scripts/exp/spmm_fused_ab.py)


def spmm_chunks(row_ptr, max_nnz=SPMM_CHUNK_NNZ):
    """Cut CSR rows into chunks of at most `max_nnz` non-zeros (host, once per graph).  Returns (chunk_ptr int64
    [n_chunks+1], chunk_row int32 [n_chunks]) as CPU tensors; every row gets at least one chunk.  chunk_row carries the
    number of combine levels spmm_csr_chunked should use (``_wr_levels``: 1 while no row has more than
    SPMM_ONE_LEVEL_MAX_CHUNKS chunks).  MI355X, ml-1m-shaped graph (1.41 M non-zeros, hub rows of 3.4 K), us per product
    (scripts/ab_spmm.py), two levels / one level: 32 per chunk 39.3 / 41.2; 64: 37.6 / 34.8; 96: 34.9 / 31.5; 128: 35.8 /
    32.4; 192: 44 / 41; 256: 48 / 45."""
    import numpy as np
    rp = row_ptr.cpu().numpy() if isinstance(row_ptr, torch.Tensor) else np.asarray(row_ptr)
    deg = np.diff(rp)
    per = np.maximum(1, (deg + max_nnz - 1) // max_nnz)
    chunk_row = np.repeat(np.arange(len(deg), dtype=np.int32), per)
    first = np.cumsum(per) - per                       # index of each row's first chunk
    within = np.arange(chunk_row.size) - np.repeat(first, per)
    start = rp[:-1][chunk_row] + within * max_nnz
    chunk_ptr = np.concatenate([start, rp[-1:]]).astype(np.int64)
    crow = torch.from_numpy(chunk_row)
    crow._wr_levels = 1 if int(per.max(initial=1)) <= SPMM_ONE_LEVEL_MAX_CHUNKS else 2
    return torch.from_numpy(chunk_ptr), crow


def spmm_levels_of(chunk_row):
    """combine levels for a chunk_row tensor (the attribute set by spmm_chunks does not survive .to(device): recomputed once
    per tensor from the longest run of equal rows)"""
    lv = getattr(chunk_row, "_wr_levels", None)
    if lv is None:
        cr = chunk_row
        if cr.numel() <= 1:
            lv = 1
        else:
            change = torch.nonzero(cr[1:] != cr[:-1]).flatten() + 1
            edges = torch.cat([change.new_zeros(1), change, change.new_full((1,), cr.numel())])
            lv = 1 if int((edges[1:] - edges[:-1]).max()) <= SPMM_ONE_LEVEL_MAX_CHUNKS else 2
        chunk_row._wr_levels = lv
    return lv


def spmm_csr_chunked(chunk_ptr, chunk_row, col, val, X, Y=None, acc=None, partials=None, levels=None, acc_from_x=False,
                     acc_scale=1.0):
    """Y = A X (load-balanced CSR product); acc: running layer sum, acc = (acc + Y) * acc_scale, or (X + Y) * acc_scale with
    acc_from_x (first layer of a propagation: acc need not be initialised)."""
    _req(chunk_ptr, torch.int64, "chunk_ptr", 1)
    _req(chunk_row, torch.int32, "chunk_row", 1)
    _req(col, torch.int32, "col", 1)
    _req(val, torch.float32, "val", 1)
    _req(X, torch.float32, "X", 2)
    if Y is None:
        Y = torch.empty_like(X)
    if partials is None:
        partials = torch.empty((chunk_row.numel(), X.shape[1]), dtype=torch.float32, device=X.device)
    lv = spmm_levels_of(chunk_row) if levels is None else int(levels)
    if SPMM_FUSED_COMBINE and lv == 1 and X.shape[1] % 32 == 0 and partials.data_ptr() % 128 == 0:
        # one launch: the last chunk of a cut row to finish adds the row's partials (wr_spmm_csr_chunked_fused)
        st = getattr(chunk_row, "_wr_fuse", None)
        if st is None:
            per = torch.bincount(chunk_row.long(), minlength=X.shape[0])
            span = torch.stack([torch.cumsum(per, 0) - per, per], dim=1).to(torch.int32).reshape(-1).contiguous()
            st = chunk_row._wr_fuse = (span, torch.zeros(X.shape[0], dtype=torch.int32, device=X.device))
        abi.check(abi.lib().wr_spmm_csr_chunked_fused(X.shape[0], chunk_row.numel(), _p(chunk_ptr), _p(chunk_row), _p(st[0]),
                                                      _p(col), _p(val), _p(X), X.shape[1], _p(Y), _p(acc), _p(partials), None,
                                                      1 if acc_from_x else 0, float(acc_scale), _p(st[1]), _stream()),
                  "wr_spmm_csr_chunked_fused")
        return Y
    abi.check(abi.lib().wr_spmm_csr_chunked_levels(X.shape[0], chunk_row.numel(), _p(chunk_ptr), _p(chunk_row), _p(col),
                                                   _p(val), _p(X), X.shape[1], _p(Y), _p(acc), _p(partials), None,
                                                   lv, 1 if acc_from_x else 0, float(acc_scale), _stream()),
              "wr_spmm_csr_chunked_levels")
    return Y


class HybridSpmm:
    """Normalised-adjacency product with the dense head of the item popularity on the matrix cores (wr_spmm_mfma.hip) and
    the rest on the chunked CSR kernels.  Built once per graph from the symmetric bipartite CSR of LightGCN
    (reference src/models/general/LightGCN.py:54-121: nodes 0..n_users-1 are users, the others items).  Head = the items
    rated by at least `min_density` of the users (at most `max_head`, rounded up to a multiple of 64: two 32-row tiles).  ``enabled`` is False when the graph
    has no such items, no more users than one K split, or nothing but head items — then use spmm_csr_chunked.  The dense tiles
    multiply their zero entries too: a non-finite value in a user row reaches every head item (0 x inf), which the CSR kernels
    would not do; padding columns and rows are masked in the kernel."""

    def __init__(self, row_ptr, col, val, n_users, n_items, device, min_density=0.12, max_head=512, k_split=128):
        import numpy as np
        rp, col, val = np.asarray(row_ptr, np.int64), np.asarray(col, np.int64), np.asarray(val, np.float32)
        nU, nI = int(n_users), int(n_items)
        N = nU + nI
        deg = np.diff(rp)
        order = np.argsort(-deg[nU:], kind="stable")
        n_head = int((deg[nU:] >= min_density * nU).sum())
        H = min((n_head + 63) // 64 * 64, int(max_head) // 64 * 64, nI // 64 * 64)
        # the head-item tiles split K = all users over several workgroups: with no more users than one split holds there is
        # nothing to split (and nothing to gain on so small a graph): CSR kernels only
        self.enabled = H >= 64 and n_head >= 16 and nU > int(k_split)
        self.n_nodes, self.device = N, device
        if not self.enabled:
            return
        head_nodes = nU + order[:H]
        pos_of = np.full(N, -1, np.int64)
        pos_of[head_nodes] = np.arange(H)
        rows = np.repeat(np.arange(N), deg)
        user_head = (rows < nU) & (pos_of[col] >= 0)
        nU_pad = (nU + 31) // 32 * 32
        Dn = np.zeros((nU_pad, H), np.float32)
        Dn[rows[user_head], pos_of[col[user_head]]] = val[user_head]
        self.head_nnz = int(user_head.sum())
        self.density = self.head_nnz / float(nU * H)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        # user-side tiles: 32 users x K = H head items (one split)
        self.u_tiles = nU_pad // 32
        self.u_A = f32(Dn.reshape(self.u_tiles, 32, H).transpose(0, 2, 1))            # [tile][k][32]
        self.u_K = H
        self.u_cols = i32(head_nodes)
        u_rows = np.arange(nU_pad)
        u_rows[nU:] = -1
        self.u_rows = i32(u_rows)
        # item-side tiles: 32 head items x K = all users, K cut into splits of k_split columns
        self.k_split = int(k_split)
        K2 = (nU + self.k_split - 1) // self.k_split * self.k_split
        Dk = np.zeros((K2, H), np.float32)
        Dk[:nU] = Dn[:nU]                                                               # A[h, u] = A[u, h]
        self.i_tiles = H // 32
        self.i_A = f32(Dk.reshape(K2, self.i_tiles, 32).transpose(1, 0, 2))            # [tile][k = user][32 head items]
        self.i_K = K2
        i_cols = np.arange(K2)
        i_cols[nU:] = -1                                                                # padding columns: masked in the kernel (exact zeros)
        self.i_cols = i32(i_cols)
        self.i_rows = i32(head_nodes)
        # CSR part: everything but the (user, head item) entries and the head items' rows
        keep = ~(user_head | (pos_of[rows] >= 0))
        if not keep.any():
            self.enabled = False        # every item is a head item: no CSR part to carry the user rows' layer sum
            return
        r_deg = np.bincount(rows[keep], minlength=N)
        r_rp = np.zeros(N + 1, np.int64)
        np.cumsum(r_deg, out=r_rp[1:])
        cptr, crow = spmm_chunks(r_rp)
        self.chunk_ptr, self.chunk_row = cptr.to(device), crow.to(device)
        self.col, self.val = i32(col[keep]), f32(val[keep])
        mode = np.zeros(N, np.int8)
        mode[:nU] = 1
        mode[head_nodes] = 2
        self.row_mode = torch.from_numpy(mode).to(device)
        self.n_head, self.H = n_head, H
        self._partials = {}

    def apply(self, X, Y=None, acc=None):
        """Y = A X (and acc += Y); X [n_nodes, D] fp32"""
        L = abi.lib()
        _req(X, torch.float32, "X", 2)
        N, D = X.shape
        if Y is None:
            Y = torch.empty_like(X)
        pt = self._partials.get(D)
        if pt is None:
            nb = abi.check_size(L.wr_spmm_dense_partials_bytes(self.i_tiles, self.i_K, self.k_split, D), "wr_spmm_dense_partials_bytes")
            pt = (torch.empty(max(nb // 4, 4), dtype=torch.float32, device=X.device),
                  torch.empty((self.chunk_row.numel(), D), dtype=torch.float32, device=X.device))
            self._partials[D] = pt
        # one launch for both tile groups: user rows get their dense part in Y (the CSR kernels then add the rest and the
        # layer sum); head item rows are complete on the matrix cores (K = all users, split over workgroups)
        g0 = abi.DenseGroup(_p(self.u_A), _p(self.u_cols), _p(self.u_rows), None, self.u_K, self.u_K, self.u_tiles)
        g1 = abi.DenseGroup(_p(self.i_A), _p(self.i_cols), _p(self.i_rows), _p(pt[0]), self.i_K, self.k_split, self.i_tiles)
        abi.check(L.wr_spmm_dense_tiles(ctypes.addressof(g0), ctypes.addressof(g1), _p(X), N, D, _p(Y), _p(acc), _stream()),
                  "wr_spmm_dense_tiles")
        abi.check(L.wr_spmm_csr_chunked_modes(N, self.chunk_row.numel(), _p(self.chunk_ptr), _p(self.chunk_row), _p(self.col),
                                              _p(self.val), _p(X), D, _p(Y), _p(acc), _p(pt[1]), _p(self.row_mode), _stream()),
                  "wr_spmm_csr_chunked_modes")
        return Y


def axpy(y, x, alpha, overwrite=False):
    abi.check(abi.lib().wr_axpy(_p(_req(y, torch.float32, "y")), _p(_req(x, torch.float32, "x")), y.numel(), alpha,
                                1 if overwrite else 0, _stream()), "wr_axpy")
    return y


def lightgcn_step(user_tab, item_tab, csr, n_layers, u, p, n, reg_weight, trusted=False):
    """LightGCN.predict + backward in one native call (wr_lightgcn_step): -> (loss (1,), gradient [n_users + n_items, D], error
    flag tensor or None).  csr = (chunk_ptr, chunk_row, col, val) of the normalised adjacency on the device."""
    cptr, crow, col, val = csr
    u, p, n = _idx64(u, "u"), _idx64(p, "p"), _idx64(n, "n")
    _req(user_tab, torch.float32, "user_tab", 2)
    _req(item_tab, torch.float32, "item_tab", 2)
    dev, D, B = user_tab.device, user_tab.shape[1], u.numel()
    L = abi.lib()
    nbytes = abi.check_size(L.wr_lightgcn_step_workspace_bytes(user_tab.shape[0], item_tab.shape[0], D, crow.numel(), B),
                            "wr_lightgcn_step_workspace_bytes")
    ws = workspace(dev, "lgcn_step").get(nbytes)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    grad = torch.empty(user_tab.shape[0] + item_tab.shape[0], D, dtype=torch.float32, device=dev)
    err = None if trusted else torch.zeros(1, dtype=torch.int32, device=dev)
    abi.check(L.wr_lightgcn_step(_p(user_tab), _p(item_tab), user_tab.shape[0], item_tab.shape[0], D, crow.numel(), _p(cptr),
                                 _p(crow), _p(col), _p(val), spmm_levels_of(crow), int(n_layers), _p(u), _p(p), _p(n), B,
                                 float(reg_weight), 1 if trusted else 0, _p(loss), _p(grad), _p(err), _p(ws), ws.numel(),
                                 _stream()), "wr_lightgcn_step")
    return loss, grad, err


def embloss_grad(user_tab, item_tab, plan, k, sq3, reg_weight, grad_user, grad_item):
    """EmbLoss backward for batch k of a plan (wr_embloss_grad): adds into grad_user / grad_item, no host sync."""
    off = k * plan.batch_size
    abi.check(abi.lib().wr_embloss_grad(_p(user_tab), _p(item_tab), user_tab.shape[1], plan.tu.data_ptr() + 4 * off,
                                        plan.oc_item.data_ptr() + 8 * off, plan.oc_src.data_ptr() + 8 * off,
                                        plan.batch_len(k), _p(sq3), reg_weight, _p(grad_user), _p(grad_item), _stream()),
              "wr_embloss_grad")


def lightgcn_loss(user_all, item_all, user_ego, item_ego, u, p, n, reg_weight):
    """LightGCN.predict's per-batch tail (wr_lightgcn_loss): -> (loss tensor of shape (1,), sq3 tensor of the three EmbLoss
    sums of squares)"""
    u, p, n = _idx64(u, "u"), _idx64(p, "p"), _idx64(n, "n")
    for t, nm in ((user_all, "user_all"), (item_all, "item_all"), (user_ego, "user_ego"), (item_ego, "item_ego")):
        _req(t, torch.float32, nm, 2)
    B, D = u.numel(), user_all.shape[1]
    dev = user_all.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    sq3 = torch.empty(3, dtype=torch.float32, device=dev)
    L = abi.lib()
    ws = workspace(dev, "lgcn_loss").get(abi.check_size(L.wr_lightgcn_loss_workspace_bytes(B), "wr_lightgcn_loss_workspace_bytes"))
    abi.check(L.wr_lightgcn_loss(_p(user_all), _p(item_all), _p(user_ego), _p(item_ego), user_all.shape[0], item_all.shape[0], D,
                                 _p(u), _p(p), _p(n), B, float(reg_weight), _p(loss), _p(sq3), _p(ws), ws.numel(), _stream()),
              "wr_lightgcn_loss")
    return loss, sq3


def embloss_sumsq(user_tab, item_tab, u, p, n):
    u, p, n = _idx64(u, "u"), _idx64(p, "p"), _idx64(n, "n")
    B, D = u.numel(), user_tab.shape[1]
    out = torch.empty(3, dtype=torch.float32, device=user_tab.device)
    ws = workspace(user_tab.device, "embloss").get(((B + 15) // 16 + 1) * 12 + 256)
    abi.check(abi.lib().wr_embloss_sumsq(_p(user_tab), _p(item_tab), D, _p(u), _p(p), _p(n), B, _p(out), _p(ws),
                                         ws.numel(), _stream()), "wr_embloss_sumsq")
    return out
