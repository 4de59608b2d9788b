"""Launches per steady-state step of a rocprofv3 kernel trace, split into this library's kernels and torch / runtime glue.
usage: python scratch/glue_count.py <kernel_trace.csv>"""
import csv, sys, collections
OURS = ('gemm_f', 'attn', 'ln_', 'mt_', 'rowstat', 'fold_', 'patchify', 'vit_embed', 'text_embed', 'splitk', 'colreduce', 'colsum', 'multi_col',
        'l2norm', 'infonce', 'temporal', 'tattn', 'add_rowbias', 'moco', 'bn_', 'ce_', 'gelu_erf', 'enqueue', 'rowdot', 'row_axpy', 'eot_index',
        'eval_', 'topk', 'cast_kernel', 'gattn', 'retrieval', 'segment', 'sum_kernel')
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "bertadam" in r[2]]
ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] - marks[i] > 50]
lo, hi = ends[-3] + 1, ends[-1] + 1
win = rows[lo:hi]
glue = [r for r in win if not any(o in r[2] for o in OURS)]
print(f"per step: {len(win) / 2:.0f} launches, {len(glue) / 2:.0f} of them torch / runtime glue ({sum(e - s for s, e, _ in glue) / 2e6:.3f} ms of kernel time)")
cnt = collections.Counter(r[2].replace("void at::native::", "")[:80] for r in glue)
for k, v in cnt.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    print(f"  {v / 2:5.1f}  {k}")
