"""Drop-in check inside a scratch copy of the reference tree (build container only: the reference does not travel to
the GPU box, where this test is skipped).  Adds the two stub files of INTEGRATION.md and runs the reference's own
`main.py`: class discovery, the argparse chain, reader, model construction, dataset and runner wiring must all work;
with no GPU here the run must then stop at the first HIP call with WhisprRecHipError — never fall back to a CPU path."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"

MODEL_STUB = """from models.BaseModel import GeneralModel
from whisprrec_amd.bprmf import bind

BPRMFHip = bind(GeneralModel)
BPRMFHip.__name__ = "BPRMFHip"
"""
RUNNER_STUB = """from helpers.BaseRunner import BaseRunner
from whisprrec_amd.runner import bind_runner

HipRunner = bind_runner(BaseRunner)
HipRunner.__name__ = "HipRunner"
"""


LIGHTGCN_STUB = """from models.BaseModel import GeneralModel
from whisprrec_amd.lightgcn import bind

LightGCNHip = bind(GeneralModel)
LightGCNHip.__name__ = "LightGCNHip"
"""
SGL_STUB = """from models.BaseModel import GeneralModel
from whisprrec_amd.sgl import bind

SGLHip = bind(GeneralModel)
SGLHip.__name__ = "SGLHip"
"""
SASREC_STUB = """from models.BaseModel import SequentialModel
from whisprrec_amd.sasrec import bind

SASRecHip = bind(SequentialModel)
SASRecHip.__name__ = "SASRecHip"
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present on this machine")
@pytest.mark.parametrize("model,stub,where,nparams,extra", [
    ("LightGCNHip", LIGHTGCN_STUB, ("models", "general"), 161088, ["--gcn_layers", "2"]),
    ("SGLHip", SGL_STUB, ("models", "general"), 161088, ["--gcn_layers", "2", "--type", "ED"]),
    ("SASRecHip", SASREC_STUB, ("models", "sequential"), None, ["--emb_size", "64", "--num_layers", "1", "--num_heads", "4"]),
])
def test_other_model_stubs_drop_into_reference_main(tmp_path, model, stub, where, nparams, extra):
    """LightGCN and SASRec stubs: discovered by name, flags chained, reader (SeqReader for SASRec) and model built by the
    reference's main.py; then the first HIP call must refuse the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    shutil.copytree(os.path.join(REF, "src"), tmp_path / "src")
    shutil.copytree(os.path.join(REF, "data", "ml-100k"), tmp_path / "data" / "ml-100k")
    (tmp_path / "src" / where[0] / where[1] / (model + ".py")).write_text(stub)
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "main.py", "--model_name", model, "--optimizer", "SGD", "--lr", "0.01", "--dataset", "ml-100k",
           "--path", str(tmp_path / "data") + "/", "--log_file", str(tmp_path / "log.txt"), "--model_path",
           str(tmp_path / "m.pt"), "--num_workers", "0", "--gpu", "", "--epoch", "1"] + extra
    res = subprocess.run(cmd, cwd=tmp_path / "src", env=env, capture_output=True, text=True, timeout=900)
    out = res.stdout + res.stderr
    assert "#params:" in out
    if nparams is not None:
        assert "#params: %d" % nparams in out
    assert res.returncode != 0 and "WhisprRecHipError" in out and "no CPU fallback" in out


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present on this machine")
@pytest.mark.parametrize("runner", ["HipRunner", "BaseRunner"])
def test_stubs_drop_into_reference_main(tmp_path, runner):
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    shutil.copytree(os.path.join(REF, "src"), tmp_path / "src")
    shutil.copytree(os.path.join(REF, "data", "ml-100k"), tmp_path / "data" / "ml-100k")
    (tmp_path / "src" / "models" / "general" / "BPRMFHip.py").write_text(MODEL_STUB)
    (tmp_path / "src" / "helpers" / "HipRunner.py").write_text(RUNNER_STUB)
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "main.py", "--model_name", "BPRMFHip", "--runner_name", runner, "--optimizer", "SGD", "--lr", "0.5",
           "--dataset", "ml-100k", "--path", str(tmp_path / "data") + "/", "--log_file", str(tmp_path / "log.txt"),
           "--model_path", str(tmp_path / "m.pt"), "--num_workers", "0", "--gpu", "", "--epoch", "1"]
    res = subprocess.run(cmd, cwd=tmp_path / "src", env=env, capture_output=True, text=True, timeout=600)
    out = res.stdout + res.stderr
    assert "#params: 161088" in out                    # (943 + 1574) * 64: the model was built from the reference reader
    assert res.returncode != 0 and "WhisprRecHipError" in out and "no CPU fallback" in out
