"""Per-K-tile time of the 256x256 tile by operand layout (256 work items, long K)."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
def run(lay, M, N, K):
    if lay == "mm":
        a = torch.randn(K, M, device="cuda", generator=g).half(); b = torch.randn(K, N, device="cuda", generator=g).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=False, b_kmajor=False)
    elif lay == "kk":
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = torch.randn(N, K, device="cuda", generator=g).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K)
    else:
        a = torch.randn(M, K, device="cuda", generator=g).half(); b = torch.randn(K, N, device="cuda", generator=g).half()
        f = lambda: ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for lay in ("kk", "km", "mm"):
    # 64 output tiles x split-K 4 = 256 items; K-tiles per item = K / 64 / 4
    t1, t2 = run(lay, 2048, 2048, 32768), run(lay, 2048, 2048, 65536)
    print(f"{lay}: K=32768 {t1:.0f} us, K=65536 {t2:.0f} us -> {(t2 - t1) / 128:.3f} us per K-tile ({2*256*256*64*256/((t2-t1)/128)/1e6:.0f} TFLOP/s steady)")

def run_mm(M, N, K, pad=0):
    a = torch.randn(K, M + pad, device="cuda", generator=g).half(); b = torch.randn(K, N + pad, device="cuda", generator=g).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    from hmmc_amd._lib import call, ptr, query
    wsb = query("hmmc_gemm_f16_workspace", M, N, K)
    ws = ops.workspace(wsb, a.device, "gemm")
    f = lambda: call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, M + pad, N + pad, N, 0, 0, None, None, None, None, 0, ptr(ws), wsb)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for M, N in ((2304, 768), (3072, 768), (768, 3072)):
    for pad in (0, 64):
        t1, t2 = run_mm(M, N, 76800, pad), run_mm(M, N, 153600, pad)
        tiles = (M // 256) * (N // 256); sk = 256 // tiles
        print(f"mm wgrad {M}x{N} pad {pad}: T=76800 {t1:.0f} us, T=153600 {t2:.0f} us -> {(t2 - t1) / (1200 / sk):.3f} us per K-tile (splitk {sk}, {tiles * sk} items)")
