"""in_proj / c_fc with the LayerNorm folded in, beside the unfolded epilogues. usage: python scratch/fold_shapes.py [T ...]"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
Ts = [int(a) for a in sys.argv[1:]] or [153600]
reps = 20
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
_a = torch.randn(Ts[0], 768, device="cuda", generator=g).half(); _w = torch.randn(2304, 768, device="cuda", generator=g).half(); _o = torch.empty(Ts[0], 2304, device="cuda", dtype=torch.float16)
for _ in range(60): ops.gemm_f16(_a, _w, Ts[0], 2304, 768, out=_o)
torch.cuda.synchronize(); del _a, _w, _o
for T in Ts:
    for name, N, K, gelu in (("in_proj", 2304, 768, False), ("c_fc", 3072, 768, True)):
        a = torch.randn(T, K, device="cuda", generator=g).half(); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(N, device="cuda", generator=g).half()
        gm = torch.ones(K, device="cuda"); bt = torch.zeros(K, device="cuda")
        (wf, cd), = ops.ln_fold_prep([(w, gm, bt, bias)])
        st = ops.rowstat(a)
        out = torch.empty(T, N, device="cuda", dtype=torch.float16)
        e = ops.EPI_QGELU if gelu else 0
        t0 = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, epilogue=e, out=out))
        t1 = timeit(lambda: ops.gemm_f16_fold(a, wf, epilogue=e, rowstat=st, colterms=cd, out=out))
        t0b = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, epilogue=e, out=out))
        t1b = timeit(lambda: ops.gemm_f16_fold(a, wf, epilogue=e, rowstat=st, colterms=cd, out=out))
        print(f"T={T} {name}: unfolded {t0:.1f} / {t0b:.1f} us, folded {t1:.1f} / {t1b:.1f} us", flush=True)
    for name, N, K in (("out_proj", 768, 768), ("c_proj", 768, 3072)):
        a = torch.randn(T, K, device="cuda", generator=g).half(); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(N, device="cuda", generator=g).half(); r = torch.randn(T, N, device="cuda", generator=g).half()
        out = torch.empty(T, N, device="cuda", dtype=torch.float16)
        t0 = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, resid=r, out=out))
        t1 = timeit(lambda: ops.gemm_f16_fold(a, w, bias=bias, resid=r, want_stat=True, out=out))
        print(f"T={T} {name}: plain {t0:.1f} us, +rowstat {t1:.1f} us", flush=True)
