import csv, glob, os, sys
from collections import defaultdict
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True))[0]
agg = defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[1:] if len(v) > 1 else v
    print("%-72s n=%3d mean(after first) %8.2f us  grid?" % (k, len(v), sum(v2) / len(v2)))
