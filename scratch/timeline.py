"""Print one step of a rocprofv3 --kernel-trace CSV as a per-queue timeline: start offset, duration, gap to the previous kernel
of the same queue.  usage: python scratch/timeline.py <kernel_trace.csv> [step_from_end=1]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
        n = n.replace("_ZN12_GLOBAL__N_1", "")
        g = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:70], r["Queue_Id"], g))
rows.sort()
marks = [i for i, r in enumerate(rows) if "bertadam" in r[2]]
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] - marks[i] > 50]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = ends[-1 - k] + 1, ends[-k] + 1
win = rows[lo:hi]
t0 = win[0][0]
last = {}
qs = sorted({r[3] for r in win})
print("queues:", qs, "kernels:", len(win), "wall ms:", (max(r[1] for r in win) - t0) / 1e6)
for s, e, n, q, g in win:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    print(f"{(s-t0)/1e3:9.1f} {'  ' * qs.index(q)}q{q} {(e-s)/1e3:7.1f}us gap {gap:6.1f} g={g:<5d} {n}")
