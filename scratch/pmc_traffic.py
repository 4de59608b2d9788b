"""Sum the FETCH_SIZE / WRITE_SIZE counters of every gemm_f16_kernel dispatch in two rocprofv3 --pmc passes and write the
per-launch HBM traffic record bench.py reports (profiles/r01_gemm_f16_hbm_traffic.json).
The record carries the hash of gemm_f16.hip and the launches per step so that bench.py can tell a stale record.
usage: python scratch/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<command>" <steps run> """
import csv, hashlib, json, os, sys

def total(path, counter):
    s, n = 0.0, 0
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and ("gemm_f16_kernel" in r["Kernel_Name"] or "gemm_f16_wgrad_group_kernel" in r["Kernel_Name"]):
                s += float(r["Counter_Value"]); n += 1
    return s, n

fs, fn = total(sys.argv[1], "FETCH_SIZE")
ws, wn = total(sys.argv[2], "WRITE_SIZE")
assert fn == wn and fn > 0, (fn, wn)
fetch = fs * 1024 / fn            # counters are in KiB
write = ws * 1024 / wn
steps = int(sys.argv[5])
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hmmc_amd", "csrc", "gemm_f16.hip")
rec = {"gemm_f16_hip_sha256_16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "steps_profiled": steps,
       "launches_per_step": fn // steps, "counter_unit": "KiB (x1024)", "launches": fn, "fetch_size_per_launch_bytes": fetch, "fetch_corrected_x2_bytes": 2 * fetch,
       "write_size_per_launch_bytes": write, "hbm_traffic_per_launch_bytes": 2 * fetch + write, "command": sys.argv[4],
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); separate passes; all "
               "gemm_f16_kernel variants (256x256 and 128x128 tiles) and the grouped weight-gradient launches of every step of the run"}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(rec)
