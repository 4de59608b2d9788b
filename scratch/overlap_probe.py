"""Could the HBM-bound kernels run beside the GEMMs?  One layer's forward GEMMs (frame-tower shapes, half a batch) on stream A
with R compute units kept out of their grids, LayerNorm + attention of the other half-batch on stream B, against the same work
serialised on one stream with full grids.  A probe for a micro-batch-interleaved tower, not product code.
usage: python scratch/overlap_probe.py [reserved CUs]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import ops, _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, D, L, H = 76800, 768, 50, 12          # half of the B=256, F=12 batch
nseq = T // L
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, D, device="cuda", generator=g).half()
w_qkv = (torch.randn(3 * D, D, device="cuda", generator=g) * 0.05).half(); w_o = (torch.randn(D, D, device="cuda", generator=g) * 0.05).half()
w_fc = (torch.randn(4 * D, D, device="cuda", generator=g) * 0.05).half(); w_pr = (torch.randn(D, 4 * D, device="cuda", generator=g) * 0.05).half()
b3, b1, b4 = torch.zeros(3 * D, device="cuda").half(), torch.zeros(D, device="cuda").half(), torch.zeros(4 * D, device="cuda").half()
gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
qkv = torch.empty(T, 3 * D, device="cuda", dtype=torch.float16); att = torch.empty(T, D, device="cuda", dtype=torch.float16)
h = torch.empty(T, 4 * D, device="cuda", dtype=torch.float16); y = torch.empty(T, D, device="cuda", dtype=torch.float16)
qkv_b = (torch.randn(T, 3 * D, device="cuda", generator=g) * 0.5).half()
def gemms():
    ops.gemm_f16(x, w_qkv, T, 3 * D, D, bias=b3, out=qkv)
    ops.gemm_f16(att, w_o, T, D, D, bias=b1, resid=x, out=y)
    ops.gemm_f16(x, w_fc, T, 4 * D, D, bias=b4, epilogue=ops.EPI_QGELU, out=h)
    ops.gemm_f16(h, w_pr, T, D, 4 * D, bias=b1, resid=x, out=y)
def hbm():
    ops.layernorm_fwd(x, gam, bet, 1e-5)
    ops.attention_f16_fwd(qkv_b, nseq, L, H, False)
    ops.layernorm_fwd(x, gam, bet, 1e-5)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tg, th = timeit(gemms), timeit(hbm)
side = torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side): hbm()
    gemms()
    cur.wait_stream(side)
t_full = timeit(both)
_lib.load().hmmc_gemm_reserve_cus(R)
tg_r = timeit(gemms)
t_res = timeit(both)
print(f"GEMMs alone {tg:.0f} us, LayerNorm + attention alone {th:.0f} us, serialised {tg + th:.0f} us | two streams, full GEMM grids {t_full:.0f} us | "
      f"GEMMs with {R} CUs reserved alone {tg_r:.0f} us, two streams {t_res:.0f} us")
