"""K-loop slope under ablations (diagnostic build): HMMC_STAG_NPH bits: 1 = no LDS-DMA in the loop, 2 = no MFMA."""
import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
_lib.LIB_PATH = '/root/repo/scratch/_dbg/libhmmc_stamps.so'
from hmmc_amd import ops
M, N = 65536, 3072
g = torch.Generator(device="cuda").manual_seed(0)
res = []
for K in (768, 3072):
    a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    f = lambda: ops.gemm_f16(a, b, M, N, K)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 10 * 1e3 / 12)
print(f"per-item us: K=768 {res[0]:.1f}  K=3072 {res[1]:.1f}  slope {(res[1]-res[0])/36:.3f} us/K-tile")
