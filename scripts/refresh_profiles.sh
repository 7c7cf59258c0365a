#!/bin/bash
# Refresh the measurements kept under profiles/ (run on the GPU box; outputs under gpurun_out/refresh, copy what is judged):
#   1. bench.py default run                        -> bench_n1.json
#   2. rocprofv3 --kernel-trace --stats of bench   -> kernel_stats.csv, trace_summary.txt
#   0. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE     -> pmc_traffic.json   (separate passes, scripts/pmc_passes.sh; first)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
# the counter passes first: bench.py quotes roofline.traffic from the PMC summary whose source hash matches the tree
timeout -k 10 500 bash scripts/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>&1
cp "$OUT/pmc/pmc_summary.json" "$R/profiles/${ROUND:-r03}_pmc_traffic_configs1.json"
echo "pmc done" >> "$OUT/progress.log"
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_n1_driver_cmd.log" 2>&1     # the driver's command
tail -n 1 "$OUT/bench_n1_driver_cmd.log" > "$OUT/bench_n1_driver_cmd.json"
timeout -k 10 400 python3 bench.py > "$OUT/bench_n1.log" 2>&1                                                # default: 512 steps
tail -n 1 "$OUT/bench_n1.log" > "$OUT/bench_n1.json"
echo "bench done" >> "$OUT/progress.log"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-epoch --no-adam > "$OUT/trace.log" 2>&1
echo "trace done" >> "$OUT/progress.log"
cd "$R"
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -n 1)" "$OUT/kernel_stats.csv"
python3 scripts/trace_summary.py "$OUT/trace" 12 > "$OUT/trace_summary.txt" 2>&1
rm -rf "$OUT/trace"/*kernel_trace.csv "$OUT/pmc"/*/*kernel_trace.csv 2>/dev/null || true
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
