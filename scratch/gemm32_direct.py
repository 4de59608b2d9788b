"""fp32 GEMM timings with the ctypes entry point called directly (prebuilt arguments, ~2 us per call of host time), so
kernels of 10-30 us are not hidden behind Python overhead.  HMMC_LIB selects the library (A/B runs)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = {}
for T in [int(t) for t in os.environ.get("TS", "3072,384,96").split(",")]:
    for name, M, N, K, lay in (("qkv", T, 1536, 512, "kk"), ("out", T, 512, 512, "kk"), ("fc", T, 2048, 512, "kk"), ("proj", T, 512, 2048, "kk"),
                               ("dqkv", T, 512, 1536, "km"), ("dout", T, 512, 512, "km"), ("dproj", T, 2048, 512, "km"), ("dfc", T, 512, 2048, "km"),
                               ("wqkv", 1536, 512, T, "mm"), ("wfc", 2048, 512, T, "mm"), ("wproj", 512, 2048, T, "mm"), ("wout", 512, 512, T, "mm")):
        if lay == "kk":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(N, K, device="cuda", generator=g); s = (K, 1, 1, K)
        elif lay == "km":
            a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g); s = (K, 1, N, 1)
        else:
            a = torch.randn(K, M, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g); s = (1, M, N, 1)
        c = torch.empty(M, N, device="cuda")
        args = (ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(c.data_ptr()), M, N, K, s[0], s[1], s[2], s[3], N,
                ctypes.c_float(1.0), None, None, None, None, 0, st)
        f = lib.hmmc_gemm_f32
        for _ in range(3): f(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f(*args)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        tot[T] = tot.get(T, 0) + us
        print(f"T={T:5d} {name:5s} {M}x{N}x{K}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF")
print("totals", {k: round(v, 1) for k, v in tot.items()})
