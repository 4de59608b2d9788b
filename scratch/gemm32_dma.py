"""fp32 GEMM kernels by shape: time and max error against a float64 product.  HMMC_F32_PICK (a -DHMMC_SCRATCH build named by
HMMC_LIB) forces one kernel: 0 dispatcher, 2 64x64 register-staged, 7 / 8 LDS-DMA 64x64 / 128x64.  usage: python scratch/gemm32_dma.py [tokens]"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
g = torch.Generator(device="cuda").manual_seed(0)
def tm(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = []
for nm, N, K in (("qkv", 1536, 512), ("out", 512, 512), ("fc", 2048, 512), ("proj", 512, 2048)):
    shapes += [(nm, T, N, K, "kk"), ("d" + nm, T, K, N, "km"), ("w" + nm, N, K, T, "mm")]
shapes += [("mlp1", 2 * T // 3, 4096, 512, "kk"), ("mlp2", 2 * T // 3, 512, 4096, "kk"), ("moco", 2 * T // 3, 1024 * 13, 512, "km"),
           ("mlm", 1536, 49408, 512, "kk"), ("dmlm", 1536, 512, 49408, "km"), ("wmlm", 49408, 512, 1536, "mm")]
tot = 0.0
for name, M, N, K, lay in shapes:
    if lay == "kk":
        a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(N, K, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K)); ref = lambda: a.double() @ b.double().t()
    elif lay == "km":
        a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (N, 1)); ref = lambda: a.double() @ b.double()
    else:
        a = torch.randn(K, M, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (1, M), (N, 1)); ref = lambda: a.double().t() @ b.double()
    err = (f().double() - ref()).abs().max().item()
    us = tm(f); tot += us
    print(f"{name:6s} {lay} {M:6d}x{N:6d}x{K:6d}: {us:8.1f} us {2.0*M*N*K/us/1e6:6.1f} TF  err {err:.2e}", flush=True)
    del a, b
print(f"total {tot:.0f} us")
