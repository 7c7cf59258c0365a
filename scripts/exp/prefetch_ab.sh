# A/B of PipelinedSgd.PREFETCH_AFTER_TRIPLETS on the driver's command (run on the GPU box)
for rep in 1 2 3 4; do
  for pf in 1048576 524288 1073741824; do
    python - <<PY 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prefetch_after', $pf, round(d['value']/1e9,3), round(d['ms_per_step']*1e3,2))"
import sys
sys.path.insert(0, ".")
import whisprrec_amd.hip_ops as h
h.PipelinedSgd.PREFETCH_AFTER_TRIPLETS = $pf
import bench
bench.main(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-phase-events", "--no-adam"])
PY
  done
done
