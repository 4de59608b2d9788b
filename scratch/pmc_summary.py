"""Average the counters of rocprofv3 --pmc counter_collection CSVs per kernel name.  usage: python scratch/pmc_summary.py <csv>..."""
import csv, sys, collections
import os
KEEP = os.environ.get("PMC_KERNELS", "gemm_f16").split(",")      # substrings of the kernel names to summarise
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
            if not any(k in n for k in KEEP): continue
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
