import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
tot = 0
for T in (3072, 384):
    for name, M, N, K, lay in (("qkv", T, 1536, 512, "kk"), ("out", T, 512, 512, "kk"), ("fc", T, 2048, 512, "kk"), ("proj", T, 512, 2048, "kk"),
                               ("dfc", T, 512, 2048, "km"), ("wfc", 2048, 512, T, "mm"), ("wout", 512, 512, T, "mm")):
        if lay == "kk":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(N, K, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K))
        elif lay == "km":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (N, 1))
        else:
            a = torch.randn(K, M, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (1, M), (N, 1))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tot += us
        print(f"T={T:5d} {name:5s} {M}x{N}x{K}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
print("total", tot)
