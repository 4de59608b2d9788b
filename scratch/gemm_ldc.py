"""Row strides of the activation operands vs GEMM speed (plain epilogue): the twelve GEMM shapes of a ViT-B/32 layer with the
token-major operands' leading dimensions as they are (dense) and padded by 64 bytes.  usage: python scratch/gemm_ldc.py [pad halves]"""
import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
from hmmc_amd.ops import call, ptr, query, workspace
T = 153600
PAD = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g = torch.Generator(device="cuda").manual_seed(0)
def run(name, lay, M, N, K, pa, pb, pc):
    # token-major operands: kk/km: A [M, K] and C [M, N]; mm: A [K, M] and B [K, N]
    if lay == "kk": sa, sb, ak, bk = (M, K + pa), (N, K), 1, 1
    elif lay == "km": sa, sb, ak, bk = (M, K + pa), (K, N), 1, 0
    else: sa, sb, ak, bk = (K, M + pa), (K, N + pb), 0, 0
    a = torch.randn(sa, device="cuda", generator=g).half(); b = (torch.randn(sb, device="cuda", generator=g) * 0.05).half()
    out = torch.empty(M, N + pc, device="cuda", dtype=torch.float16)
    wsb = query("hmmc_gemm_f16_workspace", M, N, K); ws = workspace(wsb, a.device, "gemm") if wsb else None
    f = lambda: call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(out), M, N, K, sa[1], sb[1], N + pc, ak, bk, None, None, None, None, 0, ptr(ws), wsb)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    return ms * 1e3
shapes = [("kk qkv", "kk", T, 2304, 768), ("kk out", "kk", T, 768, 768), ("kk fc", "kk", T, 3072, 768), ("kk proj", "kk", T, 768, 3072),
          ("km dfc", "km", T, 768, 3072), ("km dproj", "km", T, 3072, 768), ("km dqkv", "km", T, 768, 2304), ("km dout", "km", T, 768, 768),
          ("mm wqkv", "mm", 2304, 768, T), ("mm wfc", "mm", 3072, 768, T), ("mm wproj", "mm", 768, 3072, T), ("mm wout", "mm", 768, 768, T)]
# clocks ramp up over the first tens of milliseconds of load: warm the board before the first measured case
_a = torch.randn(T, 768, device="cuda", generator=g).half(); _w = torch.randn(768, 768, device="cuda", generator=g).half()
for _ in range(300): ops.gemm_f16(_a, _w, T, 768, 768)
torch.cuda.synchronize(); del _a, _w
print(f"pad = {PAD} halves; us per launch: dense | A padded | C (mm: B) padded | both")
tot = [0.0] * 4
for name, lay, M, N, K in shapes:
    r = []
    for pa, pc in ((0, 0), (PAD, 0), (0, PAD), (PAD, PAD), (0, 0)):
        r.append(run(name, lay, M, N, K, pa, pc if lay == "mm" else 0, 0 if lay == "mm" else pc))
    for i in range(4): tot[i] += r[i]
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:6d}  " + "  ".join(f"{x:7.1f}" for x in r[:4]) + f"  (dense again {r[4]:7.1f})" + f"   {2.0*M*N*K/r[0]/1e6:7.1f} -> {2.0*M*N*K/min(r)/1e6:7.1f} TFLOP/s", flush=True)
print("layer      " + " " * 28 + "  ".join(f"{x:7.1f}" for x in tot))
