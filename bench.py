#!/usr/bin/env python3
"""bench.py — BPR training triplets/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N=1 workload = BASELINE.json configs[1]: BPRMF, emb_size=64, synthetic 1M users x 1M items (uniform ids from the
100M-interaction generator, seed 3407), batch 65,536 triplets, SGD, l2=0.  A "step" is one BaseRunner.fit iteration
(zero_grad / predict / backward / optimizer.step, reference src/helpers/BaseRunner.py:196-199) over one batch: raw
(u, p, n) int32 triplets resident in HBM -> group plan (flags + lists of the rows that recur in the batch; no sort) -> one
step launch (wr_bprmf_run_sgd_group) -> updated tables + loss; `--no-group`: sorted batch plan -> user-phase / item-phase
kernels.  The step stream is the product's own pipeline (whisprrec_amd.hip_ops.PipelinedSgd): plans are built a
chunk ahead on a side stream, INSIDE the timed region (K steps' worth of plan builds run between the two timestamps).

N>1 (`python bench.py --gpus N` starts its own N ranks; under `torch.distributed.run` it uses the ranks it is given):
tables row-sharded over N GPUs, RCCL over xGMI, both sharding modes in one run (see whisprrec_amd/rotating.py and
whisprrec_amd/sharded.py); `--emb 128 --users 10000000 --items 10000000` is BASELINE.json configs[3].

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the 8 TB/s HBM peak using
algorithmic bytes; `cpu_baseline` times the oracle's sparse SGD restatement on one host core.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC only on this pool's hosts: RCCL / cross-process tensor sharing fails without it (set before HIP initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# RCCL prints a version banner on stdout at NCCL_DEBUG=VERSION: keep stdout to the one JSON line unless the caller asks for more
os.environ.setdefault("NCCL_DEBUG", "WARN")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--chunk", type=int, default=0, help="batches per plan build (0: min(64, steps))")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent over items (0 = uniform, headline)")
    ap.add_argument("--plan-stream", choices=["side", "inline"], default="side",
                    help="N=1: where the next chunk's plan is built: on a side stream beside the steps (default), or on the "
                         "step stream between the halves of the current chunk (measured slower; kept for the A/B)")
    ap.add_argument("--no-adam", action="store_true",
                    help="N=1: skip the Adam figure reported under `extra` (the reference's default optimizer on the same shape)")
    ap.add_argument("--no-epoch", action="store_true",
                    help="N=1: skip the whole-epoch figure reported under `extra` (negative sampling + shuffle + plans + steps)")
    ap.add_argument("--epoch-interactions", type=int, default=100_000_000,
                    help="N=1: interactions of the synthetic training frame of the whole-epoch figure")
    ap.add_argument("--no-group", action="store_true",
                    help="N=1: sorted batch plans (two bucket scatters + two LDS sorts per batch) instead of group plans "
                         "(no per-batch sort, whisprrec_amd/csrc/wr_group.hip)")
    ap.add_argument("--no-chain", action="store_true",
                    help="N=1: two launches per step (user phase, item phase) instead of the chained step launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phase-events", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--interactions", type=int, default=100_000_000, help="interactions per epoch (sets the rotation period)")
    ap.add_argument("--shard-mode", default="both", choices=["both", "rotate", "alltoall"],
                    help="N>1: 'rotate' = stratified schedule, item blocks move round the ring (whisprrec_amd/rotating.py); "
                         "'alltoall' = per-step row exchange (whisprrec_amd/sharded.py); 'both' = one after the other")
    ap.add_argument("--parts", type=int, default=2, help="rotate mode: parts per item block (overlap of transfer and compute)")
    ap.add_argument("--loopback-world", type=int, default=0,
                    help="with --force-sharded --gpus 1: also time rank 0's work of a G-rank all-to-all job on this one GPU, every "
                         "exchange replaced by a device copy (reported under modes.alltoall.loopback)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-sharded path even with one rank (testing)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N>1 path (nccl = RCCL; gloo only for the CPU dry run of the launcher)")
    ap.add_argument("--dry-run", action="store_true",
                    help="N>1 launcher rehearsal without GPUs: ranks rendezvous, barrier, and rank 0 prints the line's skeleton")
    return ap.parse_args(argv)


def plan_chunk(args):
    """batches per plan: 64, the training default; a run shorter than that (the driver's --steps 20) is one plan long — its
    plan is built during the warm-up, and the plan of the chunk that would follow is built beside its steps"""
    if args.chunk > 0:
        return args.chunk
    return max(1, min(64, args.steps))


# --------------------------------------------------------------------------------------------------- N>1 launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` as a plain command: start N ranks (one per GPU) as CHILD processes through
    torch.distributed.run and relay their output.  Nothing in this parent process touches the GPU (no HIP call, no
    torch.cuda query), and nothing is exec'ed: the parent waits and exits with the children's code."""
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


# --------------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(args, seconds):
    """Oracle's sparse SGD restatement (oracle/wr_oracle.c:orc_bprmf_step_sgd_sparse) on ONE host core, same table
    shapes, same batch size, same id distribution; bounded to ~`seconds` of CPU work."""
    import numpy as np
    import oracle
    rng = np.random.RandomState(3407)
    U = (rng.standard_normal((args.users, args.emb)) * np.sqrt(2.0 / (args.users + args.emb))).astype(np.float32)
    I = (rng.standard_normal((args.items, args.emb)) * np.sqrt(2.0 / (args.items + args.emb))).astype(np.float32)
    bl = oracle.SparseSgdBaseline(U, I, args.batch)
    batches = []
    for _ in range(8):
        batches.append((rng.randint(0, args.users, args.batch), rng.randint(0, args.items, args.batch),
                        rng.randint(1, args.items, args.batch)))
    bl.step(*batches[0], args.lr)  # warm-up (page faults of the scratch tables)
    t0 = time.perf_counter()
    k = 0
    while True:
        bl.step(*batches[k % len(batches)], args.lr)
        k += 1
        if time.perf_counter() - t0 >= seconds and k >= 3:
            break
    dt = time.perf_counter() - t0
    return {"value": k * args.batch / dt, "unit": "triplets/s", "cores": 1, "kind": "port",
            "sample": "%d steps of B=%d on %dx%d tables, D=%d, SGD l2=0, %.1f s" % (k, args.batch, args.users, args.items,
                                                                                  args.emb, dt),
            "host_cpus": os.cpu_count()}


def cpu_torch_sequence(args, seconds):
    """The aten-op sequence the reference issues on its CPU path — BPRMF.predict (src/models/general/BPRMF.py:69-80) +
    BPRLoss (src/utils/loss.py:38) + loss.backward() + torch.optim.SGD.step() (src/helpers/BaseRunner.py:196-199), dense
    gradients and all — restated with stock PyTorch CPU ops on the box's host cores (the reference's own files do not
    travel to the GPU box).  Same table shapes, batch size and id distribution; bounded to ~`seconds`."""
    import torch
    torch.manual_seed(3407)
    ue = torch.nn.Embedding(args.users, args.emb)
    ie = torch.nn.Embedding(args.items, args.emb)
    torch.nn.init.xavier_normal_(ue.weight.data)
    torch.nn.init.xavier_normal_(ie.weight.data)
    opt = torch.optim.SGD(list(ue.parameters()) + list(ie.parameters()), lr=args.lr, weight_decay=0)
    g = torch.Generator().manual_seed(3407)
    batches = [(torch.randint(0, args.users, (args.batch,), generator=g), torch.randint(0, args.items, (args.batch,), generator=g),
                torch.randint(1, args.items, (args.batch,), generator=g)) for _ in range(4)]

    def step(k):
        u, p, n = batches[k % len(batches)]
        opt.zero_grad()
        user_e, pos_e, neg_e = ue(u), ie(p), ie(n)
        pos = torch.mul(user_e, pos_e).sum(dim=1)
        neg = torch.mul(user_e, neg_e).sum(dim=1)
        loss = -torch.log(1e-10 + torch.sigmoid(pos - neg)).mean()
        loss.backward()
        opt.step()
        return loss.detach().cpu().data.numpy()

    step(0)
    t0 = time.perf_counter()
    k = 0
    while True:
        step(k)
        k += 1
        if time.perf_counter() - t0 >= seconds and k >= 2:
            break
    dt = time.perf_counter() - t0
    return {"value": k * args.batch / dt, "unit": "triplets/s", "cores": torch.get_num_threads(),
            "kind": "torch-cpu aten sequence of the reference step (dense gradients, torch.optim.SGD)",
            "sample": "%d steps of B=%d on %dx%d tables, D=%d, %.1f s" % (k, args.batch, args.users, args.items, args.emb, dt),
            "host_cpus": os.cpu_count()}


# --------------------------------------------------------------------------------------------------- N = 1
def synth_triplets(n, n_users, n_items, dev, seed, zipf=0.0):
    """Synthetic interactions of configs[1]: uniform user / item ids (worst case for caches), negatives uniform in
    [1, n_items) as in GeneralModel.Dataset.actions_before_epoch (reference src/models/BaseModel.py:168)."""
    import numpy as np
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    u = torch.randint(0, n_users, (n,), generator=g, device=dev, dtype=torch.int32)
    if zipf > 0.0:
        # inverse-CDF sampling of a Zipf(alpha) over item ranks (contention variant, not the headline)
        r = torch.rand(n, generator=g, device=dev, dtype=torch.float64)
        if abs(zipf - 1.0) < 1e-9:
            p = torch.exp(r * np.log(n_items)).to(torch.int64) - 1
        else:
            a = 1.0 - zipf
            p = (((n_items ** a - 1.0) * r + 1.0) ** (1.0 / a)).to(torch.int64) - 1
        p = p.clamp_(0, n_items - 1).to(torch.int32)
    else:
        p = torch.randint(0, n_items, (n,), generator=g, device=dev, dtype=torch.int32)
    neg = torch.randint(1, n_items, (n,), generator=g, device=dev, dtype=torch.int32)
    return u, p, neg


def csrc_sha():
    """hash of the sources of every kernel that runs in the timed region (step kernels and plan builders): a committed PMC
    traffic figure is only quoted for the kernels it was taken on"""
    h = hashlib.sha256()
    for f in ("wr_group.hip", "wr_bpr.hip", "wr_common.h", "wr_plan_fast.hip", "wr_plan.hip", "wr_overlap.hip"):
        with open(os.path.join(ROOT, "whisprrec_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def single_gpu(args, local_rank):
    import numpy as np
    import torch
    from whisprrec_amd import hip_ops
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, D, K, W = args.batch, args.emb, args.steps, args.warmup
    C = plan_chunk(args)

    # tables: N(0, 2/(rows+D)) = xavier_normal_ (reference src/models/init.py:25, BPRMF.py:40)
    g = torch.Generator(device=dev)
    g.manual_seed(3407)
    U = torch.randn(args.users, D, generator=g, device=dev) * float(np.sqrt(2.0 / (args.users + D)))
    I = torch.randn(args.items, D, generator=g, device=dev) * float(np.sqrt(2.0 / (args.items + D)))
    # One step stream of W + K + C batches, consumed as warm-up | timed steps | one plan chunk nobody trains on.  Plans: the
    # first covers the warm-up, then C batches each; a plan is built beside the steps of the plan before it.  So the plan of
    # the first timed chunk is built during the warm-up — as in an epoch, where every plan but the first is built beside
    # steps — and the timed region builds the plans of the chunks that FOLLOW its own chunks, the last of which is the spare
    # one: K batches' worth of plan builds between the two timestamps, K steps trained.
    u, p, n = synth_triplets((K + W + C) * B, args.users, args.items, dev, 3407, args.zipf)
    pipe = hip_ops.PipelinedSgd(chunk=C, min_triplets=1, chain=not args.no_chain,
                                inline_plan=args.plan_stream == "inline", group=not args.no_group)
    losses_w = torch.empty(max(W, 1), dtype=torch.float32, device=dev)
    losses = torch.empty(K, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    # the warm-up is two plans long when it can be, so that every host path of the pipeline has run twice before the clock starts
    # No cyclic-GC pass of the interpreter between here and the end of the timed region, which can be as short as 0.6 ms: a
    # pass costs milliseconds and leaves the host path behind it slower (a gc.collect() right before the clock: host time to
    # queue the 20 steps 0.33 -> 0.85 ms, 33 -> 47 us/step, three A/B pairs on one box); one run in ~20 of the driver's
    # command came out at 122 us/step.  A training loop amortises such a pass over its thousands of steps.
    import gc
    gc.disable()
    try:
        handle = pipe.plan(U, [(I, u, p, n)], B, first_chunk=[W - W // 2, W // 2] if W > 0 else None, lr=args.lr)
        if W > 0:
            pipe.run_steps(handle, W, args.lr, losses_w)
        torch.cuda.synchronize()

        t0 = time.perf_counter()
        pipe.run_steps(handle, K, args.lr, losses)
        t_queued = time.perf_counter()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    host_queue_ms = (t_queued - t0) * 1e3   # host time to queue the K steps + the next plan's build (a diagnostic: a host stall
                                            # inside a 20-step region — scheduler, allocator — shows here)

    plan = handle["cur"][1]            # the plan of the last timed chunk: the per-kernel timing pass runs on it
    tabs = handle["segs"][0]["tabs"]
    grouped = isinstance(plan, hip_ops.GroupPlan)
    # per-kernel timing: HIP events that the library attaches to the dispatches of the step kernels (the kernels' own start /
    # end timestamps), on a second pass over already planned batches — outside the throughput measurement, so nothing extra
    # sits in the timed region.
    events, KP, chain_ev, chained, group_ev = None, 0, None, [], None
    if not args.no_phase_events:
        KP = min(K, 64, plan.n_batches)

        def fresh_events(per_step=4):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(per_step * KP)]
            for e in evs:
                e.record()
            torch.cuda.synchronize()
            return evs
        if grouped:
            # what the timed region ran: ONE launch per step (the triplets of batch k + the tiles of batch k-1)
            group_ev = fresh_events(2)
            tabs.run_sgd_group(plan, 0, KP, args.lr, events=group_ev)
            torch.cuda.synchronize()
            lo_p = handle["cur"][0] * B
            sorted_plan = hip_ops.BatchPlan(u[lo_p:lo_p + plan.n_triplets], p[lo_p:lo_p + plan.n_triplets],
                                            n[lo_p:lo_p + plan.n_triplets], B, args.users, args.items)
        else:
            sorted_plan = plan
            if handle["chain"] and plan.overlap is not None and KP >= 2:
                # the chained form first (what the timed region ran): per step ONE launch that carries the user phase of the
                # step and the item phase of the step before (the first step of a call and steps with too many deferred runs
                # go out as two launches)
                chain_ev = fresh_events()
                tabs.run_sgd_chain(plan, 0, KP, args.lr, phase_events=chain_ev)
                torch.cuda.synchronize()
                dc = plan.overlap["def_count_np"]
                chained = [k for k in range(1, KP) if 0 <= int(dc[k]) <= plan.overlap["cap"]]
        events = fresh_events()
        tabs.run_sgd(sorted_plan, 0, KP, args.lr, phase_events=events)     # the two-launch form of the same steps
        torch.cuda.synchronize()
    else:
        sorted_plan = plan if not grouped else None
    tabs.check_chain()                 # a bounded wait inside a launch expired: the run is invalid — always checked

    lv = losses.cpu().numpy()
    assert np.all(np.isfinite(lv)), "non-finite loss"
    value = K * B / dt

    # unique rows per step (for algorithmic bytes with in-batch duplicates counted once): from the ids of up
    # to 8 batches of the last timed chunk
    lo_p = handle["cur"][0] * B
    nbs = max(1, min(8, plan.n_triplets // B))
    uniq_u = uniq_i = single_i = 0.0
    for k in range(nbs):
        sl = slice(lo_p + k * B, lo_p + (k + 1) * B)
        uniq_u += torch.unique(u[sl]).numel() / nbs
        _, ci = torch.unique(torch.cat([p[sl], n[sl]]), return_counts=True)
        uniq_i += ci.numel() / nbs
        single_i += int((ci == 1).sum().item()) / nbs
    row = D * 4
    # Algorithmic bytes: every unique row of the batch read once and written once + 12 B of indices
    # per triplet.  Split by who does it in the two-launch form: the user phase reads all of them, writes the user rows and
    # the single-occurrence item rows; the item phase writes the item rows that have several occurrences (its re-read of
    # those rows and the stash traffic are overhead, not algorithmic).
    bytes_user = row * (2 * uniq_u + uniq_i + single_i) + 12 * B
    bytes_item = row * (uniq_i - single_i)
    bytes_step = bytes_user + bytes_item

    roofline = None
    if events is not None:
        t_user = np.mean([events[4 * k].elapsed_time(events[4 * k + 1]) for k in range(KP)]) * 1e-3
        t_item = np.mean([events[4 * k + 2].elapsed_time(events[4 * k + 3]) for k in range(KP)]) * 1e-3
        if t_item >= t_user:
            name, tk, bk = "bprmf_item_phase", t_item, bytes_item
        else:
            name, tk, bk = "bprmf_user_phase", t_user, bytes_user
        t_chain = None
        if group_ev is not None and KP >= 2:
            # the launch of step k: the triplets of batch k + the tiles of batch k-1 = one step's algorithmic bytes
            t_chain = np.mean([group_ev[2 * k].elapsed_time(group_ev[2 * k + 1]) for k in range(1, KP)]) * 1e-3
            name, tk, bk = "bprmf_group_step", t_chain, bytes_step
        elif chained:
            # the launch of step k: user phase of batch k + item phase of batch k-1 = one step's algorithmic bytes
            t_chain = np.mean([chain_ev[4 * k].elapsed_time(chain_ev[4 * k + 1]) for k in chained]) * 1e-3
            name, tk, bk = "bprmf_chain_step", t_chain, bytes_step
        ach = bk / tk / 1e9
        roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "kernel": name, "kernel_us": tk * 1e6, "algorithmic_bytes_per_launch": bk,
                    "kernel_us_method": "start/stop events attached to the dispatch (hipExtLaunchKernelGGL)",
                    "chain_step_us": None if t_chain is None else t_chain * 1e6,
                    "chained_steps_of_timing_pass": "%d of %d" % (KP - 1 if group_ev is not None else len(chained), KP),
                    # the two-launch form of the same steps (a second pass; what --no-chain runs)
                    "user_phase_us": t_user * 1e6, "item_phase_us": t_item * 1e6,
                    "user_phase_GBs": bytes_user / t_user / 1e9, "item_phase_GBs": bytes_item / t_item / 1e9,
                    "step_algorithmic_bytes": bytes_step, "step_achieved_GBs": bytes_step * K / dt / 1e9,
                    "step_frac": bytes_step * K / dt / 1e9 / HBM_PEAK_GBS,
                    "uniq_users_per_step": uniq_u, "uniq_items_per_step": uniq_i,
                    "single_occurrence_items_per_step": single_i,
                    # the read-only variant SURVEY 8d asks for beside the read+write figure: every unique row read once +
                    # the indices, over the dominant kernel's time, against the same 8 TB/s
                    "read_only_algorithmic_bytes_per_launch": row * (uniq_u + uniq_i) + 12 * B,
                    "read_only_GBs": (row * (uniq_u + uniq_i) + 12 * B) / t_user / 1e9,
                    "read_only_frac": (row * (uniq_u + uniq_i) + 12 * B) / t_user / 1e9 / HBM_PEAK_GBS}
        # HBM bytes per launch of the dominant kernel: rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
        # passes: scripts/pmc_passes.sh) cannot run inside this process, so the figure comes from the committed summary — and
        # only when that summary was taken on THESE kernel sources and this configuration; otherwise null.
        pmc = os.path.join(ROOT, "profiles", "r03_pmc_traffic_configs1.json")
        if os.path.exists(pmc) and (B, D, args.users, args.items, args.zipf) == (65536, 64, 1_000_000, 1_000_000, 0.0):
            rec = json.load(open(pmc))
            if rec.get("csrc_sha") == csrc_sha():
                for k, v in rec.get("kernels", {}).items():
                    if k.startswith(name):
                        roofline["traffic"] = v["hbm_bytes"]
                        roofline["traffic_source"] = "profiles/r03_pmc_traffic_configs1.json (rocprofv3 --pmc FETCH_SIZE / " \
                                                     "WRITE_SIZE on kernel sources %s)" % rec["csrc_sha"]
    out = {"metric": "BPR training triplets/sec", "value": value, "unit": "triplets/s", "n_gpus": 1, "steps": K,
           "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPRMF emb_size=%d, synthetic %d users x %d items (%s ids), batch %d, SGD l2=0, "
                                  "plan build in timed region" % (D, args.users, args.items,
                                                                  "uniform" if args.zipf == 0 else "zipf(%.2f) item" % args.zipf, B),
                      "batch": B, "emb_size": D, "optimizer": "SGD", "l2": 0.0, "lr": args.lr,
                      "plan_chunk_batches": C, "tables": "single GPU", "step_stream": "whisprrec_amd.hip_ops.PipelinedSgd",
                      "plan": "group plan (no per-batch sort: flags + lists of the shared rows, wr_group.hip)" if grouped
                              else "sorted batch plan (bucket scatter + LDS sort per table)",
                      "chained_step_launch": bool(grouped or (plan.overlap is not None and handle["chain"])),
                      "step_stream_calls": dict(pipe.stats),
                      "plan_build": "on the step stream, between the halves of the chunk before" if handle["inline"]
                                    else "on a side stream, beside the steps of the chunk before",
                      "host_queue_ms_of_timed_region": host_queue_ms},
           "loss_first": float(lv[0]), "loss_last": float(lv[-1]),
           "roofline": roofline}
    if not args.no_adam:
        out["extra"] = {"adam": adam_extra(args, hip_ops, U, I, u, p, n, dev)}
    if not args.no_epoch:
        del u, p, n
        out.setdefault("extra", {})["epoch"] = epoch_extra(args, hip_ops, U, I, dev)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        out["cpu_baseline_torch"] = cpu_torch_sequence(args, max(4.0, args.cpu_seconds / 2))
    print(json.dumps(out))
    return 0


def adam_extra(args, hip_ops, U, I, u, p, n, dev):
    """The same tables, batch size and data under torch.optim.Adam semantics — what the reference's main.py runs unless
    --optimizer SGD is passed (src/helpers/BaseRunner.py:36,120-124): every row moves at every step.  Exact lazy rows
    (hip_ops.LazyOptimizerState: bit-identical to the dense optimizer), plans built in the timed region by the same pipeline.
    Not the headline: BASELINE.json's metric is quoted on SGD; reported beside it."""
    import torch
    B = args.batch
    warm = 96                       # rows must have been idle for a realistic number of steps before the clock starts
    steps = max(32, min(args.steps, 128))
    total = warm + steps + 64       # + one spare chunk: its plan is built beside the last timed steps, as in an epoch
    u, p, n = synth_triplets(total * B, args.users, args.items, dev, 3408, args.zipf)   # its own stream of batches
    st = hip_ops.LazyOptimizerState(hip_ops.BprmfTables(U, I), "Adam", 1e-3, 0.0)
    pipe = hip_ops.PipelinedSgd(chunk=64, min_triplets=1)
    nn = total * B
    def runner(plan, first, count, out):
        return st.run(plan, first, count, out)
    runner.wants_chain_marks = not getattr(args, "no_chain", False)   # folded Adam steps as one launch per step
    handle = pipe.plan(U, [(I, u[:nn], p[:nn], n[:nn])], B, first_chunk=[warm], runner=runner)
    lw = torch.empty(max(warm, 1), dtype=torch.float32, device=dev)
    lt = torch.empty(steps, dtype=torch.float32, device=dev)
    if warm > 0:
        pipe.run_steps(handle, warm, 0.0, lw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run_steps(handle, steps, 0.0, lt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st.flush()
    torch.cuda.synchronize()
    return {"metric": "BPR training triplets/sec, Adam (lr 1e-3, l2 0)", "value": steps * B / dt, "unit": "triplets/s",
            "us_per_step": dt / steps * 1e6, "steps": steps, "warmup": warm,
            "how": "exact lazy rows, catch-up %s%s; plan builds inside the timed region" %
                   ("folded into the step kernels' row loads" if st._folds(handle["cur"][1]) else "in a pass of its own",
                    ", one launch per step (%d chained calls)" % st.chain_calls if st.chain_calls else ""),
            "loss_last": float(lt[-1])}


def epoch_extra(args, hip_ops, U, I, dev):
    """SURVEY 8(d)'s epoch-level figure on the same tables: one WHOLE epoch of a synthetic training frame with everything
    the reference does per epoch on the device — a negative per interaction that the user has not clicked
    (BaseModel.py:167-177), the epoch's shuffle (BaseRunner.py:188-193), batch plans, SGD steps — as HipRunner
    --device_epoch_prep 1 runs it: sampler + shuffle fused and produced chunk by chunk beside the steps (hip_ops.EpochPrep),
    membership through the pair set, source rows packed.  The clicked lists / pair set / packed rows are per-frame setup
    (built once, reported); the third epoch is the one timed."""
    import torch
    n_inter, B = int(args.epoch_interactions), args.batch
    g = torch.Generator(device=dev); g.manual_seed(3409)
    users = torch.randint(0, args.users, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    items = torch.randint(0, args.items, (n_inter,), generator=g, device=dev, dtype=torch.int32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ptr, idx = hip_ops.clicked_csr_from_pairs(users, items, args.users, args.items)
    pairs = hip_ops.pair_set(ptr, idx, args.users)
    packed = hip_ops.pack_rows(users, items)
    torch.cuda.synchronize(); setup = time.perf_counter() - t0
    pipe = hip_ops.PipelinedSgd(chunk=64)
    nb = (n_inter + B - 1) // B
    dt = 0.0
    for epoch in (1, 2, 3):
        losses = torch.empty(nb, dtype=torch.float32, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prep = hip_ops.EpochPrep(users, items, args.users, args.items, ptr, idx, 3407, epoch, pairs=pairs, packed=packed)
        pipe.run(pipe.plan(U, [(I, prep.cols[0], prep.cols[1], prep.cols[2])], B, prep=prep), 0, args.lr, losses)
        prep.check()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        del prep
    return {"metric": "BPR training triplets/sec over a whole epoch (negative sampling + shuffle + plans + steps on the device)",
            "value": n_inter / dt, "unit": "triplets/s", "epoch_ms": dt * 1e3, "interactions": n_inter, "steps": nb,
            "per_frame_setup_ms": setup * 1e3, "loss_mean": float(losses.mean())}


# --------------------------------------------------------------------------------------------------- N > 1
def multi_gpu(args, rank, world, local_rank):
    """One rank per GPU (RCCL).  Runs the sharding mode(s) asked for back to back inside one process group and prints one
    line: `value` is the all-to-all mode's (the one that keeps the reference's sampling) unless only 'rotate' was asked for;
    every mode's numbers are under `modes`."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", str(rank))
    os.environ.setdefault("WORLD_SIZE", str(world))
    # stdout carries ONE JSON line: whatever libraries print there while the ranks initialise (RCCL's version banner) goes to
    # stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if args.dry_run:
        dist.init_process_group(args.backend if args.backend == "gloo" else "gloo")
        dist.barrier()
        if rank == 0:
            emit(multi_line(args, world, {}, None, dry_run=True))
        dist.barrier()
        dist.destroy_process_group()
        return 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    modes = ["alltoall", "rotate"] if args.shard_mode == "both" else [args.shard_mode]
    results = {}
    for mode in modes:
        if mode == "rotate":
            from whisprrec_amd import rotating
            res = rotating.bench_run(args, rank, world, dev)
        else:
            from whisprrec_amd import sharded
            res = sharded.bench_run(args, rank, world, dev)
        if rank == 0:
            results[mode] = res
        torch.cuda.synchronize()
        dist.barrier()
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, min(args.cpu_seconds, 8.0))
    if rank == 0:
        emit(multi_line(args, world, results, cpu))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def multi_line(args, world, results, cpu, dry_run=False):
    """the N>1 JSON line from the per-mode results (also the skeleton the CPU dry run of the launcher prints)"""
    B, D = args.batch, args.emb
    # the headline comes from the mode that takes the reference's batches as they are (row exchange by all-to-all, negatives
    # from all items); the stratified rotation — which draws a triplet's negative from its positive's item block — is
    # reported beside it under `modes.rotate` with that waiver
    primary = "rotate" if args.shard_mode == "rotate" else "alltoall"
    head = results.get(primary, {})
    out = {"metric": "BPR training triplets/sec", "value": head.get("value"), "unit": "triplets/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": head.get("ms_per_step"), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPRMF emb_size=%d, synthetic %d users x %d items (uniform ids), batch %d per GPU (global "
                                  "%d), SGD l2=0, tables row-sharded over %d GPUs, plan build and exchange in timed region"
                                  % (D, args.users, args.items, B, B * world, world),
                      "batch_per_gpu": B, "global_batch": B * world, "emb_size": D, "optimizer": "SGD", "l2": 0.0,
                      "lr": args.lr, "rccl_world_size": world, "ranks": "one process per GPU, torch.distributed nccl (=RCCL)",
                      "parallelism": head.get("parallelism", primary), "value_from_mode": primary,
                      "sampling": head.get("sampling")},
           "roofline": head.get("roofline"), "modes": results}
    if dry_run:
        out["dry_run"] = True
    if cpu is not None:
        out["cpu_baseline"] = cpu
    return out


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)          # plain `python bench.py --gpus N`: start the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit("bench.py --gpus %d runs under WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 or args.force_sharded:
        return multi_gpu(args, rank, world, local_rank)
    return single_gpu(args, local_rank)


if __name__ == "__main__":
    sys.exit(main())
