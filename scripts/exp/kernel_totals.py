"""total / count / mean duration per kernel name from a rocprofv3 --kernel-trace csv: python kernel_totals.py <dir> [divide-by]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot, cnt = collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"][:70]
    tot[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[n] += 1
for n, t in tot.most_common(25):
    print("%-72s n=%-6d total %9.1f us  mean %7.2f  per-unit %7.2f" % (n, cnt[n], t, t / cnt[n], t / div))
