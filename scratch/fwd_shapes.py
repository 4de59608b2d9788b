"""Forward (k-major x k-major) GEMM shapes of a CLIP block, each with: no epilogue / the eval epilogue / the training epilogue.
usage: python scratch/fwd_shapes.py [T ...]   (default 153600 19200)"""
import os, sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
Ts = [int(a) for a in sys.argv[1:]] or [153600, 19200]
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
    tot = {"plain": 0.0, "eval": 0.0, "train": 0.0}
    for name, N, K, epi in (("in_proj", 2304, 768, "b"), ("out_proj", 768, 768, "br"), ("c_fc", 3072, 768, "bg"), ("c_proj", 768, 3072, "br")):
        a = torch.randn(T, K, device="cuda", generator=g).half(); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
        bias = torch.randn(N, device="cuda", generator=g).half()
        resid = torch.randn(T, N, device="cuda", generator=g).half() if "r" in epi else None
        out = torch.empty(T, N, device="cuda", dtype=torch.float16)
        t_plain = timeit(lambda: ops.gemm_f16(a, w, T, N, K, out=out))
        if "g" in epi:
            t_eval = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, epilogue=ops.EPI_QGELU, out=out))
            t_train = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, epilogue=ops.EPI_QGELU | ops.EPI_SAVE_DGELU, want_aux=True, out=out))
        else:
            t_eval = timeit(lambda: ops.gemm_f16(a, w, T, N, K, bias=bias, resid=resid, out=out))
            t_train = t_eval
        fl = 2.0 * T * N * K
        tot["plain"] += t_plain; tot["eval"] += t_eval; tot["train"] += t_train
        print(f"T={T:6d} {name:8s} N={N:4d} K={K:4d}  plain {t_plain:7.1f} us {fl/t_plain/1e6:6.0f} TF | eval {t_eval:7.1f} us {fl/t_eval/1e6:6.0f} TF | train {t_train:7.1f} us {fl/t_train/1e6:6.0f} TF", flush=True)
        del a, w, bias, resid, out
    fl = 2.0 * T * 768 * 768 * 12
    print(f"T={T:6d} layer fwd GEMMs: plain {tot['plain']:.0f} us ({fl/tot['plain']/1e6:.0f} TF)  eval {tot['eval']:.0f} us ({fl/tot['eval']/1e6:.0f} TF)  train {tot['train']:.0f} us ({fl/tot['train']/1e6:.0f} TF)", flush=True)
    x = torch.randn(T, 768, device="cuda", generator=g).half(); gm = torch.ones(768, device="cuda"); bt = torch.zeros(768, device="cuda")
    print(f"T={T:6d} ln_fwd {timeit(lambda: ops.layernorm_fwd(x, gm, bt, 1e-5)):.1f} us", flush=True)
    qkv = torch.randn(T, 2304, device="cuda", generator=g).half()
    print(f"T={T:6d} attn_fwd L=50 {timeit(lambda: ops.attention_f16_fwd(qkv, T // 50, 50, 12, False)):.1f} us", flush=True)
    del x, qkv
