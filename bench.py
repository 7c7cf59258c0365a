#!/usr/bin/env python3
"""bench.py — BPR training triplets/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N=1 workload = BASELINE.json configs[1]: BPRMF, emb_size=64, synthetic 1M users x 1M items (uniform ids from the
100M-interaction generator, seed 3407), batch 65,536 triplets, SGD, l2=0.  A "step" is one BaseRunner.fit iteration
(zero_grad / predict / backward / optimizer.step, reference src/helpers/BaseRunner.py:196-199) over one batch: raw
(u, p, n) int32 triplets resident in HBM -> sorted batch plan -> user-phase kernel -> item-phase kernel -> updated
tables + loss.  Plan building is INSIDE the timed region (done per chunk of batches).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the 8 TB/s HBM peak using
algorithmic bytes; `cpu_baseline` times the oracle's sparse SGD restatement on one host core.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC only on this pool's hosts: RCCL / cross-process tensor sharing fails without it (set before HIP initialises)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--lr", type=float, default=0.05)
    ap.add_argument("--chunk", type=int, default=64, help="batches per plan build")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent over items (0 = uniform, headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phase-events", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--plan-builder", default="auto", choices=["auto", "generic", "fast"])
    ap.add_argument("--interactions", type=int, default=100_000_000, help="interactions per epoch (sets the rotation period)")
    ap.add_argument("--shard-mode", default="rotate", choices=["rotate", "alltoall"],
                    help="N>1: 'rotate' = stratified schedule, item blocks move round the ring (whisprrec_amd/rotating.py); "
                         "'alltoall' = per-step row exchange (whisprrec_amd/sharded.py)")
    ap.add_argument("--parts", type=int, default=2, help="rotate mode: parts per item block (overlap of transfer and compute)")
    ap.add_argument("--force-sharded", action="store_true", help="run the row-sharded path even with one rank (testing)")
    return ap.parse_args()


def synth_triplets(n, n_users, n_items, dev, seed, zipf=0.0):
    """Synthetic interactions of configs[1]: uniform user / item ids (worst case for caches), negatives uniform in
    [1, n_items) as in GeneralModel.Dataset.actions_before_epoch (reference src/models/BaseModel.py:168)."""
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


def cpu_baseline(args, seconds):
    """Oracle's sparse SGD restatement (oracle/wr_oracle.c:orc_bprmf_step_sgd_sparse) on ONE host core, same table
    shapes, same batch size, same id distribution; bounded to ~`seconds` of CPU work."""
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


KEEP_PLANS = 2


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" %
                             (args.gpus, args.gpus))
    if args.gpus > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.shard_mode == "rotate":
            from whisprrec_amd import rotating
            return rotating.bench_main(args, rank, world, local_rank)
        from whisprrec_amd import sharded
        return sharded.bench_main(args, rank, world, local_rank)

    from whisprrec_amd import hip_ops
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B, D, K, W = args.batch, args.emb, args.steps, args.warmup

    # tables: N(0, 2/(rows+D)) = xavier_normal_ (reference src/models/init.py:25, BPRMF.py:40)
    g = torch.Generator(device=dev)
    g.manual_seed(3407)
    U = torch.randn(args.users, D, generator=g, device=dev) * float(np.sqrt(2.0 / (args.users + D)))
    I = torch.randn(args.items, D, generator=g, device=dev) * float(np.sqrt(2.0 / (args.items + D)))
    tabs = hip_ops.BprmfTables(U, I)
    u, p, n = synth_triplets((K + W) * B, args.users, args.items, dev, 3407, args.zipf)
    torch.cuda.synchronize()

    plan_stream = hip_ops.side_stream(dev)   # high priority: its own hardware queue, never serialised behind the steps
    main_stream = torch.cuda.current_stream(dev)

    bucket_map = {"map": None}

    def build_plan(first_step, c, tag):
        """enqueue the plan build of steps [first_step, first_step+c) on the side stream; returns (plan, ready event)"""
        lo = first_step * B
        with torch.cuda.stream(plan_stream):
            plan = hip_ops.BatchPlan(u[lo:lo + c * B], p[lo:lo + c * B], n[lo:lo + c * B], B, args.users, args.items,
                                     validate=False, ws_tag="plan%d" % tag, builder=args.plan_builder,
                                     bucket_map=bucket_map["map"] or None)
            if plan.fast_overflowed and args.plan_builder == "auto":
                # skewed ids (--zipf): balance the builder's buckets by the rows' share of the data, as PipelinedSgd does
                bucket_map["map"] = hip_ops.BucketMap(u, p, args.users, args.items, B) if bucket_map["map"] is None else False
            ready = torch.cuda.Event()
            ready.record(plan_stream)
        return plan, ready

    def run_range(first_step, count):
        """plan + steps for global steps [first_step, first_step+count), chunk by chunk.  The plan of chunk c+1 is built
        on a side stream while the steps of chunk c run (it depends only on the indices, never on the tables)."""
        out = []
        chunks = []
        done = 0
        while done < count:
            c = min(args.chunk, count - done)
            chunks.append((first_step + done, c))
            done += c
        plan_stream.wait_stream(main_stream)
        nxt = build_plan(chunks[0][0], chunks[0][1], 0)
        for i, (fs, c) in enumerate(chunks):
            plan, ready = nxt
            main_stream.wait_event(ready)
            plan.validate()                                             # flags came back with the hot-run counts: no sync
            plan.record_stream(main_stream)
            # plans are kept for the first KEEP_PLANS chunks only (row statistics, per-kernel timing pass): holding every
            # plan alive makes each build a fresh hipMalloc, and that call stalls the host long enough to drain the queue
            out.append((plan if i < KEEP_PLANS else None, tabs.run_sgd(plan, 0, c, args.lr)))   # this chunk's steps first ...
            if i + 1 < len(chunks):                                     # ... then build the next plan beside them
                nxt = build_plan(chunks[i + 1][0], chunks[i + 1][1], (i + 1) % 2)
        return out

    run_range(0, W)
    torch.cuda.synchronize()

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_range(W, K)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # per-kernel timing: HIP events recorded by the library on the launch stream around the two kernels of each
    # step, on a second pass over the same (already planned) batches — outside the throughput measurement, so
    # the event records do not sit in the timed region.
    events = None
    if not args.no_phase_events:
        KP = min(K, 64, res[0][0].n_batches)
        events = [torch.cuda.Event(enable_timing=True) for _ in range(4 * KP)]
        for e in events:
            e.record()
        torch.cuda.synchronize()
        tabs.run_sgd(res[0][0], 0, KP, args.lr, phase_events=events)
        torch.cuda.synchronize()

    losses = torch.cat([r[1] for r in res]).cpu().numpy()
    assert np.all(np.isfinite(losses)), "non-finite loss"
    value = K * B / dt

    # unique rows per step (for algorithmic bytes with in-batch duplicates counted once)
    uniq_u = uniq_i = single_i = 0
    n_stat = 0
    for plan, _ in res:
        if plan is None:
            continue
        nb = plan.n_batches
        n_stat += nb
        tu = plan.tu.view(nb, B)
        oi = plan.oc_item.view(nb, 2 * B)
        uniq_u += int((tu[:, 1:] != tu[:, :-1]).sum().item()) + nb
        uniq_i += int((oi[:, 1:] != oi[:, :-1]).sum().item()) + nb
        single_i += int((plan.tp >= 0).sum().item()) + int((plan.tn >= 0).sum().item())  # rows with one occurrence
    uniq_u /= n_stat                   # averages over the first n_stat steps of the timed region
    uniq_i /= n_stat
    single_i /= n_stat
    row = D * 4
    # Algorithmic bytes: every unique row of the batch read once and written once + 12 B of indices
    # per triplet.  Split by who does it: the user phase reads all of them, writes the user rows and the
    # single-occurrence item rows; the item phase writes the item rows that have several occurrences (its re-read of
    # those rows and the stash traffic are overhead, not algorithmic).
    bytes_user = row * (2 * uniq_u + uniq_i + single_i) + 12 * B
    bytes_item = row * (uniq_i - single_i)
    bytes_step = bytes_user + bytes_item

    roofline = None
    if events is not None:
        # the dispatches' own start / end timestamps (events attached to the kernels by the library: what rocprofv3
        # reports per dispatch; no marker packets in the stream)
        t_user = np.mean([events[4 * k].elapsed_time(events[4 * k + 1]) for k in range(KP)]) * 1e-3
        t_item = np.mean([events[4 * k + 2].elapsed_time(events[4 * k + 3]) for k in range(KP)]) * 1e-3
        if t_item >= t_user:
            name, tk, bk = "bprmf_item_phase<16,1,true,0>", t_item, bytes_item
        else:
            name, tk, bk = "bprmf_user_phase<16,1,true,0>", t_user, bytes_user
        ach = bk / tk / 1e9
        roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None, "kernel": name, "kernel_us": tk * 1e6, "algorithmic_bytes_per_launch": bk,
                    "kernel_us_method": "start/stop events attached to the dispatch (hipExtLaunchKernelGGL)",
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

    if roofline is not None:
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 +
        # WRITE_SIZE, separate passes: scripts/pmc_passes.sh); only valid for the configuration it was collected on.
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_configs1.json")
        if os.path.exists(pmc) and (B, D, args.users, args.items, args.zipf) == (65536, 64, 1_000_000, 1_000_000, 0.0):
            for k, v in json.load(open(pmc)).items():
                if roofline["kernel"].split("<")[0] in k:
                    roofline["traffic"] = v["hbm_bytes"]
                    roofline["traffic_source"] = "profiles/r01_pmc_traffic_configs1.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
    out = {"metric": "BPR training triplets/sec", "value": value, "unit": "triplets/s", "n_gpus": 1, "steps": K,
           "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BPRMF emb_size=%d, synthetic %d users x %d items (%s ids), batch %d, SGD l2=0, "
                                  "plan build in timed region" % (D, args.users, args.items,
                                                                  "uniform" if args.zipf == 0 else "zipf(%.2f) item" % args.zipf, B),
                      "batch": B, "emb_size": D, "optimizer": "SGD", "l2": 0.0, "lr": args.lr,
                      "plan_chunk_batches": args.chunk, "tables": "single GPU"},
           "loss_first": float(losses[0]), "loss_last": float(losses[-1]),
           "roofline": roofline}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        out["cpu_baseline_torch"] = cpu_torch_sequence(args, max(4.0, args.cpu_seconds / 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
