"""What does the vendor library reach on the same shapes?  torch.matmul (hipBLASLt / rocBLAS) on the twelve GEMMs of a ViT-B/32
layer, plain (no epilogue), against hmmc_gemm_f16 plain.  Reference point only: the product never calls it."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import ops
T = 153600
g = torch.Generator(device="cuda").manual_seed(0)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
_a = torch.randn(T, 768, device="cuda", generator=g).half(); _w = torch.randn(2304, 768, device="cuda", generator=g).half()
for _ in range(60): torch.matmul(_a, _w.t())
torch.cuda.synchronize(); del _a, _w
shapes = [("kk qkv", "kk", T, 2304, 768), ("kk out", "kk", T, 768, 768), ("kk fc", "kk", T, 3072, 768), ("kk proj", "kk", T, 768, 3072),
          ("km dfc", "km", T, 768, 3072), ("km dproj", "km", T, 3072, 768), ("km dqkv", "km", T, 768, 2304), ("km dout", "km", T, 768, 768),
          ("mm wqkv", "mm", 2304, 768, T), ("mm wfc", "mm", 3072, 768, T), ("mm wproj", "mm", 768, 3072, T), ("mm wout", "mm", 768, 768, T)]
tot = [0.0, 0.0]
for name, lay, M, N, K in shapes:
    if lay == "kk":
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        f_lib = lambda: torch.matmul(a, b.t(), out=out)
        f_own = lambda: ops.gemm_f16(a, b, M, N, K, out=out)
    elif lay == "km":
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(K, N, device="cuda", generator=g) * 0.05).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        f_lib = lambda: torch.matmul(a, b, out=out)
        f_own = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False, out=out)
    else:
        a = torch.randn(K, M, device="cuda", generator=g).half(); b = torch.randn(K, N, device="cuda", generator=g).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        f_lib = lambda: torch.matmul(a.t(), b, out=out)
        f_own = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=False, b_kmajor=False, out=out)
    tl, to = t(f_lib), t(f_own)
    tot[0] += tl; tot[1] += to
    fl = 2.0 * M * N * K
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:6d}   library {tl:7.1f} us {fl/tl/1e6:7.1f} TF   this repo {to:7.1f} us {fl/to/1e6:7.1f} TF", flush=True)
    del a, b, out
print(f"layer: library {tot[0]:.0f} us, this repo {tot[1]:.0f} us")
