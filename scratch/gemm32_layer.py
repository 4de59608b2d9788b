"""The twelve fp32 GEMMs of one temporal-transformer layer (D = 512, T = 3072 / 384 tokens) beside torch.matmul (rocBLAS / hipBLASLt
fp32: reference point only, the product never calls it).  usage: python scratch/gemm32_layer.py"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
def tm(f, reps=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for T in (3072, 384):
    tot = tot_ref = 0.0
    shapes = []
    for nm, N, K in (("qkv", 1536, 512), ("out", 512, 512), ("fc", 2048, 512), ("proj", 512, 2048)):
        shapes += [(nm, T, N, K, "kk"), ("d" + nm, T, K, N, "km"), ("w" + nm, N, K, T, "mm")]
    for name, M, N, K, lay in shapes:
        if lay == "kk":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(N, K, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K)); r = lambda: a @ b.t()
        elif lay == "km":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (N, 1)); r = lambda: a @ b
        else:
            a = torch.randn(K, M, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
            f = lambda: ops.gemm_f32(a, b, M, N, K, (1, M), (N, 1)); r = lambda: a.t() @ b
        us, ur = tm(f), tm(r)
        tot += us; tot_ref += ur
        print(f"T={T:5d} {name:6s} {M:5d}x{N:5d}x{K:5d}: {us:7.1f} us {2.0*M*N*K/us/1e6:6.1f} TF | torch {ur:7.1f} us {2.0*M*N*K/ur/1e6:6.1f} TF", flush=True)
    print(f"T={T} layer total {tot:.0f} us | torch {tot_ref:.0f} us")
