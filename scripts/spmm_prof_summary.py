#!/usr/bin/env python3
"""Summarise scripts/profile_spmm.sh: per kernel of the adjacency product the average duration (kernel trace) and the counter
averages of the two --pmc passes; plus the wall times of scripts/ab_spmm.py."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
res = defaultdict(dict)
f = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if f:
    agg = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        if "spmm" in k:
            res[k.split("(")[0].replace("void wr::", "").replace("wr::", "")]["avg_us"] = sum(v) / len(v) / 1e3
            res[k.split("(")[0].replace("void wr::", "").replace("wr::", "")]["launches"] = len(v)
for d in ("pmc_sq", "pmc_tcc"):
    f = glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "spmm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void wr::", "").replace("wr::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res[k][c] = sum(v) / len(v)
for k, d in res.items():
    if d.get("SQ_BUSY_CYCLES"):
        d["mfma_busy_frac_of_SQ_busy"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d["SQ_BUSY_CYCLES"]
    if d.get("SQ_WAVE_CYCLES"):
        d["valu_active_frac_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0.0) / d["SQ_WAVE_CYCLES"]
    if d.get("TCC_HIT_sum") is not None and (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
ab = {}
p = os.path.join(out, "ab.txt")
if os.path.exists(p):
    for line in open(p):
        parts = line.strip().split(" ", 1)
        if len(parts) == 2 and parts[1].startswith("{"):
            ab[parts[0]] = json.loads(parts[1])
print(json.dumps({"wall_times_scripts_ab_spmm": ab, "kernels": res}, indent=1))
