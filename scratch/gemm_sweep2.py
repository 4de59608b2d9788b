"""Per-item cost of the 256x256 GEMM tile at exact rounds (kk, M=65536, N=3072; K = 256, 768, 3072). usage: python scratch/gemm_sweep2.py [reps]"""
import sys, torch, os
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
out = []
for N in (3072, 768):
    rounds = (M // 256) * (N // 256) // 256
    for K in (256, 768, 3072):
        a = torch.randn(M, K, device="cuda", generator=g).half()
        b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        out.append(f"N{N}/nkt{K//64}:{us / rounds:.1f}")
        del a, b
print(os.environ.get('HMMC_LIB', 'default'), " ".join(out), flush=True)
