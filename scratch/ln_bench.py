"""LayerNorm forward / backward at the ViT-B/32 tower shape [153600, 768] fp16 (HBM-bound). usage: HMMC_LIB=... python scratch/ln_bench.py"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
T, D = int(sys.argv[1]) if len(sys.argv) > 1 else 153600, 768
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, D, device="cuda", generator=g).half(); dy = torch.randn(T, D, device="cuda", generator=g).half()
res = torch.randn(T, D, device="cuda", generator=g).half()
gm, bt = torch.randn(D, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-5)
us = t(lambda: ops.layernorm_fwd(x, gm, bt, 1e-5))
print(f"ln_fwd  {us:7.1f} us  {2 * T * D * 2 / us / 1e6:.2f} TB/s")
us = t(lambda: ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=res, want_colsum=True))
print(f"ln_bwd  {us:7.1f} us  {4 * T * D * 2 / us / 1e6:.2f} TB/s (dy, x, dres in; dx out; + reduce)")
