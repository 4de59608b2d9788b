"""Per-K-tile and per-item cost of the 256x256 GEMM tile: exact rounds (M=65536 -> 256 m-tiles), K sweep.
usage: python scratch/gemm_sweep.py [reps]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import os
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
for lay in ("kk", "km"):
    for N in (768, 3072):
        rounds = (M // 256) * (N // 256) // 256
        pts = []
        for K in (256, 512, 768, 1536, 3072):
            a = torch.randn(M, K, device="cuda", generator=g).half()
            b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half() if lay == "kk" else (torch.randn(K, N, device="cuda", generator=g) * 0.05).half()
            f = (lambda: ops.gemm_f16(a, b, M, N, K)) if lay == "kk" else (lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False))
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            pts.append((K // 64, us / rounds))
            del a, b
        (k0, t0), (k1, t1) = pts[2], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print(f"{lay} N={N:5d} rounds={rounds:3d} per-item us by nkt: " + " ".join(f"{k}:{t:.1f}" for k, t in pts) +
              f" | slope {slope:.3f} us/K-tile, intercept {t0 - slope * k0:.2f} us/item", flush=True)
