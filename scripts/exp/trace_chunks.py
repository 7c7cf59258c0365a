"""step-stream timeline of a rocprofv3 --kernel-trace CSV of bench.py: per plan chunk, where the time between the first and
the last step kernel goes.  Usage: trace_chunks.py <dir>"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = [r for r in rows if "bprmf_chain_step" in r["Kernel_Name"] or "bprmf_user_phase" in r["Kernel_Name"]]
other = [r for r in rows if r not in step and "bprmf_item_phase" not in r["Kernel_Name"]]
S = lambda r: int(r["Start_Timestamp"]) / 1e3
E = lambda r: int(r["End_Timestamp"]) / 1e3
# the timed region: the last 512 step kernels
step = step[-512:]
t0 = S(step[0])
print("steps %d, span %.1f us -> %.2f us/step" % (len(step), E(step[-1]) - t0, (E(step[-1]) - t0) / len(step)))
busy = [(S(r), E(r)) for r in other if S(r) >= t0 - 5000 and S(r) <= E(step[-1])]
def overlapped(a, b):
    return any(s < b and e > a for s, e in busy)
clean = [E(r) - S(r) for r in step if not overlapped(S(r), E(r))]
dirty = [E(r) - S(r) for r in step if overlapped(S(r), E(r))]
print("step kernels without a concurrent kernel: %d, mean %.2f us; with one: %d, mean %.2f us" %
      (len(clean), sum(clean) / max(len(clean), 1), len(dirty), sum(dirty) / max(len(dirty), 1)))
gaps = [(S(b) - E(a), i) for i, (a, b) in enumerate(zip(step[:-1], step[1:]))]
big = [(g, i) for g, i in gaps if g > 3.0]
print("gaps > 3 us between consecutive step kernels: %d, total %.1f us" % (len(big), sum(g for g, _ in big)))
for g, i in big[:20]:
    print("   after step %d: %.1f us" % (i, g))
print("sum of all gaps %.1f us; sum of step kernel durations %.1f us" % (sum(g for g, _ in gaps), sum(E(r) - S(r) for r in step)))
if len(sys.argv) > 2:
    # timeline around the middle of the region: every kernel, start relative, duration, queue
    mid = S(step[len(step) // 2 + int(sys.argv[2])])
    for r in rows:
        if mid - 20 <= S(r) <= mid + 600:
            print("%9.1f %7.1f q%-3s %s" % (S(r) - mid, E(r) - S(r), r.get("Queue_Id", "?"), r["Kernel_Name"][:80]))
