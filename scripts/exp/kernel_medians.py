"""median duration per (kernel, grid) from a rocprofv3 --kernel-trace csv: python kernel_medians.py <dir> [name filter]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
flt = sys.argv[2].split(",") if len(sys.argv) > 2 else None
seen = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if flt and not any(x in n for x in flt):
        continue
    seen[(n[:60], r.get("Grid_Size_X", r.get("Grid_Size")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in seen.items():
    v = sorted(v)
    print("%-62s grid %-9s n=%-5d median %.2f us  min %.2f" % (k[0], k[1], len(v), v[len(v) // 2], v[0]))
