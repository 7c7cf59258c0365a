"""whisprrec_amd — MI355X (gfx950) implementation of WhisprRec's embedding-CF training hot path.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI (include/whisprrec_hip.h) -> libwhisprrec_hip.so
  abi.py       ctypes binding of the C-ABI (no CPU fallback: raises if the library is missing)
  hip_ops.py   tensor-level wrappers (PyTorch-ROCm owns memory and streams only)
  host.py      host-side mirror of the reference's BaseModel / GeneralModel / Dataset contract
  bprmf.py     BPRMF drop-in model (reference src/models/general/BPRMF.py)
  lightgcn.py  LightGCN drop-in model (reference src/models/general/LightGCN.py)
  sasrec.py    SASRec with its item-embedding gather/scatter on the HIP kernels (reference src/models/sequential/SASRec.py)
  runner.py    HipRunner: BaseRunner-compatible runner that drives the fused step
  sharded.py   row-sharded multi-GPU step (RCCL all-to-all over xGMI)
"""
__version__ = "0.1.0"
