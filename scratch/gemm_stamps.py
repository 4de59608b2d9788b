"""In-kernel timeline of the ping-pong GEMM (diagnostic build scratch/_dbg/libhmmc_stamps.so, -DHMMC_GEMM_STAMPS)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
_lib.LIB_PATH = '/root/repo/scratch/_dbg/libhmmc_stamps.so'
from hmmc_amd import ops
M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
import os
for N, K in ((3072, 768),):
    a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    for _ in range(3): ops.gemm_f16(a, b, M, N, K)
    torch.cuda.synchronize()
    out = np.zeros((8, 2, 64), dtype=np.uint64)
    lib = _lib.load()
    lib.hmmc_gemm_debug_stamps.argtypes = [ctypes.c_void_p]
    assert lib.hmmc_gemm_debug_stamps(out.ctypes.data) == 0
    print(f"N={N} K={K} nkt={K//64}")
    print('grid', os.environ.get('HMMC_GEMM_GRID'))
    for blk in (0, 5):
        for w in (0, 1):
            t = out[blk, w].astype(np.int64)
            t0 = t[0]
            rel = (t[:14] - t0) / 100.0
            print(f"  blk{blk} grp{w}: " + " ".join(f"{x:7.2f}" for x in rel))
            ph = t[32:48]
            print(f"      phase cycles (after each barrier, 2 K-tiles): " + " ".join(f"{int(x):5d}" for x in (ph[1:] - ph[:-1])))
