"""Two contrasting GEMMs for PMC passes: kk fc (many tiles, short K, L2 reuse) and mm wqkv (256 items, K = all tokens)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
T = 153600
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(T, 768, device="cuda", generator=g).half(); w = (torch.randn(3072, 768, device="cuda", generator=g) * 0.05).half()
dy = torch.randn(T, 2304, device="cuda", generator=g).half()
for _ in range(4):
    ops.gemm_f16(a, w, T, 3072, 768)
    ops.gemm_f16(dy, a, 2304, 768, T, a_kmajor=False, b_kmajor=False)
torch.cuda.synchronize()
