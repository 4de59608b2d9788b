"""exactness of the 256x256 GEMM of the library named by HMMC_LIB on integer data, three layouts (A/B experiments)"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(1)
ok = True
for lay, M, N, K in (("kk", 20000, 2304, 768), ("km", 16384, 768, 2304), ("mm", 768, 3072, 40000), ("kk", 70000, 768, 3072), ("mm", 2304, 768, 153600)):
    A = torch.randint(-3, 4, (M, K), device="cuda", generator=g).half(); B = torch.randint(-3, 4, (N, K), device="cuda", generator=g).half()
    ref = (A.float() @ B.float().t())
    if lay == "kk": c = ops.gemm_f16(A, B, M, N, K)
    elif lay == "km": c = ops.gemm_f16(A, B.t().contiguous(), M, N, K, a_kmajor=True, b_kmajor=False)
    else: c = ops.gemm_f16(A.t().contiguous(), B.t().contiguous(), M, N, K, a_kmajor=False, b_kmajor=False)
    good = bool(torch.equal(c.float(), ref.half().float()))
    ok &= good
    print(lay, M, N, K, "exact" if good else f"MISMATCH max {float((c.float()-ref).abs().max())}")
sys.exit(0 if ok else 1)
