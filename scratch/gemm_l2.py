"""Experiment: same GEMM with lda=0 (all A rows alias one row -> always L2-resident) vs real A."""
import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
from hmmc_amd._lib import call, ptr
T = 153600
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(T, 768, 768), (T, 2304, 768), (T, 768, 3072)]:
    a = torch.randn(M, K, device="cuda", generator=g).half(); b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.half)
    for lda, tag in ((K, "real A (HBM)"), (0, "lda=0 (L2)  ")):
        f = lambda: call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, lda, K, N, 1, 1, None, None, None, None, 0, None, 0)
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={M} N={N} K={K} {tag}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TF", flush=True)
