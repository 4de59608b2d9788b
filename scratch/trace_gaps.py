"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals over the last `steps` bench steps, GPU-busy union and idle time.
usage: python scratch/trace_gaps.py <kernel_trace.csv> [top]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "<" in n and not n.startswith("at::"):
            n = n.split("(")[0] if n.index("<") < n.index("(") else n.split("<")[0]
            n += f" grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", "?")))
rows.sort()
# steps are delimited by the optimizer's last kernel (mt_bertadam); take the windows between consecutive last-bertadam launches
marks = [i for i, r in enumerate(rows) if "bertadam" in r[2]]
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] - marks[i] > 50]
if len(ends) < 3:
    sys.exit("not enough steps in trace")
lo, hi = ends[-3] + 1, ends[-1] + 1          # two whole steps
win = rows[lo:hi]
t0, t1 = win[0][0], max(r[1] for r in win)
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in win:
    k = n[:100]
    tot[k] += e - s; cnt[k] += 1
print(f"2 steps: wall {(t1-t0)/2e6:.3f} ms/step, busy(union) {busy/2e6:.3f} ms/step, idle {(t1-t0-busy)/2e6:.3f} ms/step, kernels/step {len(win)//2}")
for k, v in tot.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    print(f"{v/2e6:8.3f} ms/step  {cnt[k]//2:5d}x  avg {v/cnt[k]/1e3:7.1f} us  {k}")
