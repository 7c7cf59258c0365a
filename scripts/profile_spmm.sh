#!/bin/bash
# LightGCN adjacency product on the ml-1m-shaped graph: CSR gather kernels vs the hybrid with the dense head on the matrix
# cores.  Kernel trace + two counter passes (separate runs; --pmc only ever combined with --kernel-trace).  Run on the GPU box:
#   bash scripts/profile_spmm.sh <out_dir>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/spmm_prof}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd "$R"
WR_DENS=0.12 timeout -k 10 200 python3 scripts/ab_spmm.py > "$OUT/ab.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
export WR_REPS=20 WR_DENS=0.12
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$R/scripts/ab_spmm.py" > "$OUT/trace.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F32 --output-format csv -d "$OUT/pmc_sq" -o run -- python3 "$R/scripts/ab_spmm.py" > "$OUT/pmc_sq.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_tcc" -o run -- python3 "$R/scripts/ab_spmm.py" > "$OUT/pmc_tcc.log" 2>&1
cd "$R"
python3 scripts/spmm_prof_summary.py "$OUT" > "$OUT/summary.json"
find "$OUT" -name '*kernel_trace.csv' -delete
