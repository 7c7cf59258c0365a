#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel average duration, and the timeline (durations + gaps)
of the steady-state BPRMF steps.  Usage: trace_summary.py <dir containing *_kernel_trace.csv> [n_timeline]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    agg = defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print("%-100s %7s %10s %10s %6s" % ("kernel", "calls", "avg_us", "total_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print("%-100s %7d %10.2f %10.1f %6.2f" % (k[:100], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3, 100.0 * sum(v) / tot))
    # the step stream: chained launches where the run used them, else the two-launch form
    idx = [i for i, r in enumerate(rows) if "bprmf_chain_step" in r["Kernel_Name"]]
    if len(idx) <= 20:
        idx = [i for i, r in enumerate(rows) if "bprmf_user_phase" in r["Kernel_Name"]]
    if len(idx) > 16:
        i0 = idx[len(idx) // 2]
        prev = None
        print("\ntimeline (mid-run): kernel, duration us, gap before it us")
        for r in rows[i0:i0 + nshow]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print("  %-70s %8.2f %8.2f" % (r["Kernel_Name"][:70], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
            prev = e
        half = idx[len(idx) // 2:]
        span = (int(rows[half[-1]]["Start_Timestamp"]) - int(rows[half[0]]["Start_Timestamp"])) / 1e3 / (len(half) - 1)
        print("\nsteady-state span per step (2nd half of run, incl. plan builds): %.2f us" % span)


if __name__ == "__main__":
    main()
