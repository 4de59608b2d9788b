"""usage: python scratch/hbm_kernels_summary.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> <kernel_stats.csv>"""
import csv, sys, collections
T, D = 153600, 768
alg = {"ln_fwd_kernel": 2 * T * D * 2, "ln_bwd_kernel": 4 * T * D * 2, "ln_bwd_fold_kernel": 4 * T * D * 2, "attn_fwd_kernel": 4 * T * D * 2,
       "attn_bwd_kernel": 7 * T * D * 2, "rowstat_kernel": T * D * 2}
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024)
    return acc
fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
dur = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[3]))}
print(f"{'kernel':28s} {'fetch x2 MB':>11s} {'write MB':>9s} {'total MB':>9s} {'algorithmic':>11s} {'x':>5s} {'us':>7s} {'TB/s (alg)':>10s} {'of 8 TB/s':>9s}")
for key, a in alg.items():
    names = [n for n in fe if key in n and "float" not in n]
    for n in names:
        f = 2 * sum(fe[n][1:]) / max(len(fe[n]) - 1, 1)           # first launch = warm-up; FETCH_SIZE doubled (MI355X_MICROARCH.md)
        w = sum(wr[n][1:]) / max(len(wr[n]) - 1, 1)
        us = dur.get(n, float("nan"))
        print(f"{key:28s} {f/1e6:11.0f} {w/1e6:9.0f} {(f+w)/1e6:9.0f} {a/1e6:11.0f} {(f+w)/a:5.2f} {us:7.1f} {a/us/1e6:10.2f} {a/us/1e6/8:9.3f}")
