"""Rehearsal of bench.py's N > 1 path with TWO ranks on ONE GPU: the same bench_run functions of both sharding modes, real
kernels, torch.distributed over gloo (RCCL refuses two ranks on one device).  Checks that every rank gets through and that
rank 0 can build the N > 1 JSON line; the numbers mean nothing (two processes share the card) and neither do the tables: gloo's
point-to-point calls read device memory from the host, unordered with the streams (tests/test_hip_rotating_two_ranks.py stages
them through the host and checks the result)."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)


def worker(rank, world, port, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    args = bench.parse(argv)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    results = {}
    for mode in ("rotate", "alltoall"):
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
    if rank == 0:
        line = bench.multi_line(args, world, results, None)
        print(json.dumps(line)[:1500])
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    import bench
    world = int(os.environ.get("WR_RANKS", "2"))          # at most 6 processes may share the card on the GPU pool
    argv = ["--gpus", str(world), "--steps", "20", "--warmup", "5", "--users", "200000", "--items", "200000", "--no-cpu-baseline"] + sys.argv[1:]
    mp.spawn(worker, args=(world, bench.free_port(), argv), nprocs=world, join=True)
    print("%d ranks on one GPU: ok" % world)
