"""HBM traffic of the twelve GEMMs of a ViT-B/32 layer, one by one: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`scratch/gemm_bench.py 3` (4 launches per shape, in the order of its shape list), grouped by dispatch order.
usage: python scratch/pmc_by_shape.py <fetch counter_collection.csv> <write counter_collection.csv>"""
import csv, sys
T = 153600
# name, M, N, K, extra read bytes (residual / aux operand), extra write bytes (aux output)
shapes = [("kk qkv  +b", T, 2304, 768, 0, 0), ("kk out  +b+r", T, 768, 768, 2 * T * 768, 0), ("kk fc   +b+gelu", T, 3072, 768, 0, 2 * T * 3072),
          ("kk proj +b+r", T, 768, 3072, 2 * T * 768, 0), ("km dfc", T, 768, 3072, 0, 0), ("km dproj *dgelu", T, 3072, 768, 2 * T * 3072, 0),
          ("km dqkv", T, 768, 2304, 0, 0), ("km dout", T, 768, 768, 0, 0), ("mm wqkv", 2304, 768, T, 0, 0), ("mm wfc", 3072, 768, T, 0, 0),
          ("mm wproj", 768, 3072, T, 0, 0), ("mm wout", 768, 768, T, 0, 0)]

def per_dispatch(path, counter):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and "gemm_f16_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024))
    rows.sort()
    return [v for _, v in rows]

fe, wr = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
assert len(fe) == len(wr) >= 4 * len(shapes), (len(fe), len(wr))
fe, wr = fe[-4 * len(shapes):], wr[-4 * len(shapes):]          # gemm_bench.py warms the board up with unmeasured launches first
print(f"{'shape':18s} {'reads MB':>9s} {'algorithmic':>11s} {'x':>5s}   {'writes MB':>9s} {'algorithmic':>11s} {'x':>5s}")
tf = tw = af = aw = 0.0
for i, (name, M, N, K, xr, xw) in enumerate(shapes):
    f = 2 * sum(fe[4 * i + 1:4 * i + 4]) / 3 / 1e6          # FETCH_SIZE doubled (MI355X_MICROARCH.md); first launch = warm-up
    w = sum(wr[4 * i + 1:4 * i + 4]) / 3 / 1e6
    ar, awr = (2 * (M * K + N * K) + xr) / 1e6, (2 * M * N + xw) / 1e6
    tf += f; tw += w; af += ar; aw += awr
    print(f"{name:18s} {f:9.0f} {ar:11.0f} {f / ar:5.2f}   {w:9.0f} {awr:11.0f} {w / awr:5.2f}")
print(f"{'layer':18s} {tf:9.0f} {af:11.0f} {tf / af:5.2f}   {tw:9.0f} {aw:11.0f} {tw / aw:5.2f}")
