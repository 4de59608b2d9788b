"""GEMM microbenchmark at the tower shapes (random data). usage: python scratch/gemm_bench.py [reps]"""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = 153600
shapes = [("kk qkv", "kk", T, 2304, 768), ("kk out", "kk", T, 768, 768), ("kk fc", "kk", T, 3072, 768), ("kk proj", "kk", T, 768, 3072),
          ("km dfc", "km", T, 768, 3072), ("km dproj", "km", T, 3072, 768), ("km dqkv", "km", T, 768, 2304),
          ("mm wqkv", "mm", 2304, 768, T), ("mm wfc", "mm", 3072, 768, T), ("mm wproj", "mm", 768, 3072, T)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, lay, M, N, K in shapes:
    if lay == "kk":
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K)
    elif lay == "km":
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(K, N, device="cuda", generator=g) * 0.05).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False)
    else:
        a = torch.randn(K, M, device="cuda", generator=g).half(); b = torch.randn(K, N, device="cuda", generator=g).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=False, b_kmajor=False)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:6d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
    del a, b
