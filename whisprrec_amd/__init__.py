"""whisprrec_amd — MI355X (gfx950) implementation of WhisprRec's embedding-CF training hot path.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI (include/whisprrec_hip.h) -> libwhisprrec_hip.so
  abi.py       ctypes binding of the C-ABI (no CPU fallback: raises if the library is missing)
  hip_ops.py   tensor-level wrappers (PyTorch-ROCm owns memory and streams only)
  host.py      host-side mirror of the reference's BaseModel / GeneralModel / Dataset contract
  reader.py    .inter reader with the reference's filters and split rules (reference src/helpers/BaseReader.py, utils/sample.py)
  main.py      stand-alone launcher with the reference's command line (reference src/main.py)
  bprmf.py     BPRMF drop-in model (reference src/models/general/BPRMF.py)
  lightgcn.py  LightGCN drop-in model (reference src/models/general/LightGCN.py)
  sgl.py       SGL drop-in model: graph views on CSR + three propagations (reference src/models/general/SGL.py, utils/augmentor.py)
  sasrec.py    SASRec with its item-embedding gather/scatter on the HIP kernels (reference src/models/sequential/SASRec.py)
  runner.py    HipRunner: BaseRunner-compatible runner that drives the fused step
  rotating.py  multi-GPU, stratified schedule: user rows fixed, item blocks rotate round the xGMI ring (default for N > 1)
  sharded.py   multi-GPU, row-sharded step with RCCL all-to-all of item rows and gradient rows
"""
__version__ = "0.1.0"
