"""What each epilogue variant costs at the tower's forward shapes (plain vs +bias vs +bias+residual vs QuickGELU).
usage: python scratch/epi_cost.py [reps]"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T = 153600
g = torch.Generator(device="cuda").manual_seed(0)
def run(name, N, K, **kw):
    a = torch.randn(T, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    out = torch.empty(T, N, device="cuda", dtype=torch.float16)
    f = lambda: ops.gemm_f16(a, b, T, N, K, out=out, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    items = (T // 256) * (N // 256)
    print(f"{name:28s} N={N:5d} K={K:5d} {us:8.1f} us  {2.0*T*N*K/us/1e6:7.1f} TF  {us/((items+255)//256):6.2f} us/round", flush=True)
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    bias = torch.randn(N, device="cuda", generator=g).half()
    resid = torch.randn(T, N, device="cuda", generator=g).half()
    run("plain", N, K)
    run("+bias", N, K, bias=bias)
    run("+bias+resid", N, K, bias=bias, resid=resid)
    if N == 3072:
        run("+bias+qgelu (1 out)", N, K, bias=bias, epilogue=ops.EPI_QGELU)
        run("+bias+qgelu+dgelu (2 out)", N, K, bias=bias, epilogue=ops.EPI_QGELU | ops.EPI_SAVE_DGELU, want_aux=True)
    del resid
