"""Rounds of the persistent 256x256 grid: a shape of exactly 7 rounds vs 7.03 rounds, with and without the K split of the short
last round (HMMC_NO_GEMM_TAIL).  usage: python scratch/tail_probe.py"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
def run(M, N, K, lay):
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
    print(f"{lay} M={M} N={N} K={K}: tiles {tiles} = {tiles/256:.2f} rounds  {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
for lay in ("kk", "km"):
    for K in (3072, 768):
        run(256 * 597, 768, K, lay)      # 1791 tiles
        run(256 * 600, 768, K, lay)      # 1800 tiles
        run(256 * 682, 768, K, lay)      # 2046 tiles = 7.99 rounds
