import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
nseq, L, H = 3072, 50, 12
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(nseq * L, 3 * H * 64, device="cuda", generator=g).half()
dout = torch.randn(nseq * L, H * 64, device="cuda", generator=g).half()
out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, False)
for name, f in (("fwd", lambda: ops.attention_f16_fwd(qkv, nseq, L, H, False)), ("bwd", lambda: ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, False))):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"attention {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
