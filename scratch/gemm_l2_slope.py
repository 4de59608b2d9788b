"""K-loop slope of the 256x256 kernel when BOTH operands stay in L2 (one round of tiles, small operands) against the streaming
shapes of gemm_sweep.py: is the K-loop of the big forward GEMMs held back by where its operands come from?
usage: python scratch/gemm_l2_slope.py"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for lay in ("kk", "km"):
    for (M, N) in ((4096, 4096), (2048, 8192), (16384, 1024), (65536, 3072)):
        pts = []
        for K in (768, 1536, 3072, 6144):
            a = torch.randn(M, K, device="cuda", generator=g).half()
            b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half() if lay == "kk" else (torch.randn(K, N, device="cuda", generator=g) * 0.05).half()
            out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            f = (lambda: ops.gemm_f16(a, b, M, N, K, out=out)) if lay == "kk" else (lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False, out=out))
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            tiles = (M // 256) * (N // 256)
            rounds = (tiles + 255) // 256
            pts.append((K // 64, us / rounds, 2.0 * M * N * K / us / 1e6))
            del a, b
        (k0, t0, _), (k1, t1, _) = pts[1], pts[-1]
        slope = (t1 - t0) / (k1 - k0)
        print(f"{lay} M={M:6d} N={N:5d} tiles={(M//256)*(N//256):5d}: " + " ".join(f"{k}:{t:.1f}us/{tf:.0f}TF" for k, t, tf in pts) + f" | slope {slope:.3f} us/K-tile", flush=True)
