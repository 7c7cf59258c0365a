"""last kernels of a rocprofv3 --kernel-trace CSV as a timeline: start (us, relative), duration, queue/stream, name.
Usage: trace_tail.py <dir> <n_last> [substring marking the window's first kernel]"""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2])
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %7.1f q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
