import csv, glob, os, sys
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True))[0]
agg = defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][:70]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
agg = {k: [d for _, d in sorted(v)] for k, v in agg.items()}
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[1:] if len(v) > 1 else v
    tail = v[-max(1, len(v) // 5):]
    print("%-72s n=%5d mean(after first) %8.2f us, last fifth mean %8.2f max %8.2f" % (k, len(v), sum(v2) / len(v2),
                                                                                       sum(tail) / len(tail), max(tail)))
