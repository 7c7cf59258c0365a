#!/bin/bash
# Collect HBM traffic counters for the bench workload in SEPARATE rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass on gfx950; --pmc is never combined with trace domains other than --kernel-trace).  Run on the GPU box:
#   bash scripts/pmc_passes.sh <out_dir>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python3 "$R/scripts/exp_kernels.py" 0 > "$OUT/$C.log" 2>&1
done
python3 "$R/scripts/pmc_summary.py" "$OUT"
