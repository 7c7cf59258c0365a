#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: average FETCH_SIZE / WRITE_SIZE per launch of each wr:: kernel.
FETCH_SIZE/WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. HALF the
bytes of a wide (16 B/lane) read stream (MI355X_MICROARCH.md §HBM) — the x2 correction is applied here and checked against
the forward-only kernel, whose read volume is known exactly (3 rows of 256 B + 24 B of int64 indices per triplet)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "wr::" in k:
            res[k.split("(")[0]][c] = (sum(v) / len(v), len(v))
summary = {}
for k, d in sorted(res.items()):
    f = d.get("FETCH_SIZE", (0, 0)); w = d.get("WRITE_SIZE", (0, 0))
    summary[k] = {"launches": f[1] or w[1], "FETCH_SIZE_KiB_raw": f[0], "WRITE_SIZE_KiB": w[0],
                  "read_bytes_corrected_x2": f[0] * 1024 * 2, "write_bytes": w[0] * 1024,
                  "hbm_bytes": f[0] * 1024 * 2 + w[0] * 1024}
# bench.py quotes a kernel's traffic only when this summary was taken on the kernel sources it runs: record their hash
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
rec = {"csrc_sha": bench.csrc_sha(), "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> (separate passes) -- "
       "python3 scripts/exp_kernels.py 0   [configs[1]: 1M x 1M, D=64, B=65,536, uniform ids]",
       "kernels": {k.replace("void wr::", ""): v for k, v in summary.items()}}
print(json.dumps(rec, indent=1))
json.dump(rec, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
