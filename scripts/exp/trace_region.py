"""every kernel around the timed region of a traced bench.py run (the first long group of chained step kernels):
start relative to the region's first step kernel, duration, queue, name.  Usage: trace_region.py <dir> [us before] [us after]"""
import csv, glob, os, sys
d = sys.argv[1]
before = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
after = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bprmf_chain_step" in r["Kernel_Name"]]
groups, cur = [], []
for i in idx:
    if cur and int(rows[i]["Start_Timestamp"]) - int(rows[cur[-1]]["End_Timestamp"]) > 80_000:
        groups.append(cur); cur = []
    cur.append(i)
if cur:
    groups.append(cur)
timed = next(g for g in groups if len(g) >= 12)
# the region's first kernel is the plain user phase right before the first chained launch
t0 = int(rows[timed[0] - 1]["Start_Timestamp"]) if "bprmf_user_phase" in rows[timed[0] - 1]["Kernel_Name"] else int(rows[timed[0]]["Start_Timestamp"])
t1 = int(rows[timed[-1]]["End_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 - before * 1e3 or s > t1 + after * 1e3:
        continue
    name = r["Kernel_Name"]
    if "bprmf_chain_step" in name and timed[1] < rows.index(r) < timed[-2]:
        continue
    print("%9.1f %7.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name[:80]))
print("region: first step kernel to end of last chained launch: %.1f us" % ((t1 - t0) / 1e3))
