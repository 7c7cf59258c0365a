"""bench.py's N>1 entry on CPU: `python bench.py --gpus 2` as a plain command starts its own two ranks (child processes
through torch.distributed.run, nothing exec'ed, the parent never touches a GPU), the ranks rendezvous on 127.0.0.1 and rank 0
prints ONE JSON line in the driver's format.  `--dry-run --backend gloo` stops after the rendezvous (no GPU here); the line's
content for real results is checked on synthetic per-mode dicts."""
import json
import os
import subprocess
import sys

from conftest import ROOT

import bench


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, text
    return json.loads(lines[0])


def test_plain_command_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--dry-run", "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5 and line["dry_run"] is True
    assert line["config"]["rccl_world_size"] == 2 and line["config"]["global_batch"] == 2 * line["config"]["batch_per_gpu"]
    assert line["metric"] == "BPR training triplets/sec" and line["scaling"] == "weak" and line["unit"] == "triplets/s"


def test_under_torch_distributed_run_it_uses_the_given_ranks():
    """the driver's own launch line: python -m torch.distributed.run ... bench.py --gpus N"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(bench.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--steps", "4", "--warmup", "1", "--dry-run", "--backend", "gloo"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_multi_gpu_line_carries_both_modes_and_the_single_gpu_definitions():
    args = bench.parse(["--gpus", "8", "--steps", "20", "--warmup", "5", "--emb", "128", "--users", "10000000", "--items",
                        "10000000"])
    rot = {"value": 8e9, "ms_per_step": 0.05, "parallelism": "stratified-rotation x8", "sampling": "stratified: ...",
           "roofline": {"bound": "hbm", "achieved": 3000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.375, "traffic": None,
                        "definition": "2*D*4*(unique users + unique items of the local batch) + 12*B, as at N=1"}}
    a2a = {"value": 2e9, "ms_per_step": 0.2, "parallelism": "row-sharded tables x8, RCCL all-to-all",
           "sampling": "reference rule", "roofline": dict(rot["roofline"], achieved=700.0, frac=0.0875)}
    line = bench.multi_line(args, 8, {"rotate": rot, "alltoall": a2a}, {"value": 1e6, "unit": "triplets/s", "cores": 1,
                                                                         "kind": "port", "sample": "x"})
    # the headline is the mode that keeps the reference's sampling (all-to-all); the rotation is reported beside it
    assert line["value"] == 2e9 and line["config"]["value_from_mode"] == "alltoall" and line["n_gpus"] == 8
    assert set(line["modes"]) == {"rotate", "alltoall"} and line["modes"]["alltoall"]["value"] == 2e9
    assert line["config"]["sampling"].startswith("reference rule") and line["config"]["rccl_world_size"] == 8
    assert line["config"]["emb_size"] == 128 and "10000000 users" in line["config"]["workload"]
    assert line["roofline"]["definition"].endswith("as at N=1") and line["cpu_baseline"]["kind"] == "port"
    json.dumps(line)


def test_plan_chunk_of_a_short_run_is_the_run():
    assert bench.plan_chunk(bench.parse(["--steps", "20", "--warmup", "5"])) == 20
    assert bench.plan_chunk(bench.parse([])) == 64
    assert bench.plan_chunk(bench.parse(["--steps", "20", "--chunk", "7"])) == 7
