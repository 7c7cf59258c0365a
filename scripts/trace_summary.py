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
    # the step stream: chained launches where the run used them, else the two-launch form.  The timed region of bench.py is
    # the FIRST long group of consecutive step kernels (groups are separated by host synchronisations: warm-up pieces before
    # it, the per-kernel timing passes after it)
    count = lambda nm: sum(nm in r["Kernel_Name"] for r in rows)
    name = "bprmf_group_step" if count("bprmf_group_step") > 12 else \
        "bprmf_chain_step" if count("bprmf_chain_step") > 12 else "bprmf_user_phase"
    idx = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
    groups, cur = [], []
    for i in idx:
        if cur and int(rows[i]["Start_Timestamp"]) - int(rows[cur[-1]]["End_Timestamp"]) > 80_000:
            groups.append(cur)
            cur = []
        cur.append(i)
    if cur:
        groups.append(cur)
    # (a region of fewer than 12 chained launches in a row — the stream is cut where the next plan's build is queued — still
    # counts when the cuts are short: groups closer than 400 us are one region)
    merged = []
    for g in groups:
        if merged and int(rows[g[0]]["Start_Timestamp"]) - int(rows[merged[-1][-1]]["End_Timestamp"]) < 400_000:
            merged[-1] = merged[-1] + g
        else:
            merged.append(g)
    timed = next((g for g in merged if len(g) >= 12), None)
    if timed is not None:
        print("\ntimeline (first steps of the timed region; every kernel of every stream): kernel, start us, duration us")
        t0 = int(rows[timed[0]]["Start_Timestamp"])
        for r in rows[timed[0]:timed[min(nshow, len(timed) - 1)] + 1]:
            s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print("  %-70s %9.2f %8.2f" % (r["Kernel_Name"][:70], (s0 - t0) / 1e3, (e0 - s0) / 1e3))
        clean = [g for g in zip(timed[:-1], timed[1:])]
        span = (int(rows[timed[-1]]["Start_Timestamp"]) - t0) / 1e3 / (len(timed) - 1)
        durs = [(int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i in timed]
        print("\n%s kernels of the timed region: %d, start-to-start %.2f us per step (plan builds of the other stream included), "
              "duration min %.2f / median %.2f / max %.2f us" % (name, len(timed), span, min(durs), sorted(durs)[len(durs) // 2],
                                                                 max(durs)))


if __name__ == "__main__":
    main()
