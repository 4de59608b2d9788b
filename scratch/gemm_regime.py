"""Is the slow K-tile of the weight gradients the operand layout or the streaming regime?  kk / km / mm at 256 items, long K, odd strides."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
from hmmc_amd._lib import call, ptr, query
g = torch.Generator(device="cuda").manual_seed(0)
def run(lay, M, N, K, pad):
    ak, bk = lay[0] == "k", lay[1] == "k"
    a = torch.randn((M, K + pad) if ak else (K, M + pad), device="cuda", generator=g).half()
    b = torch.randn((N, K + pad) if bk else (K, N + pad), device="cuda", generator=g).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    wsb = query("hmmc_gemm_f16_workspace", M, N, K)
    ws = ops.workspace(wsb, a.device, "gemm")
    f = lambda: call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, a.shape[1], b.shape[1], N, int(ak), int(bk), None, None, None, None, 0, ptr(ws), wsb)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for lay in ("kk", "km", "mm"):
    for pad in (0, 72):
        t1, t2 = run(lay, 2048, 2048, 30720, pad), run(lay, 2048, 2048, 61440, pad)
        print(f"{lay} pad {pad:3d}: {(t2 - t1) / 120:.3f} us per K-tile")
