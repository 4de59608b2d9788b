"""Every fp32 GEMM kernel on every fp32 shape of the path (HMMC_F32_PICK forces one): which is fastest where.
usage: HMMC_F32_PICK=k python scratch/gemm32_pick.py   (k = 0 auto, 1 small, 2 tiled, 3 wave-split-K 64x64, 4 wave-split-K 32x64)"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
shapes = []
for T in (3072, 1536, 384):
    shapes += [(f"T{T} qkv", T, 1536, 512, "kk"), (f"T{T} out", T, 512, 512, "kk"), (f"T{T} fc", T, 2048, 512, "kk"), (f"T{T} proj", T, 512, 2048, "kk"),
               (f"T{T} dqkv", T, 512, 1536, "kn"), (f"T{T} dout", T, 512, 512, "kn"), (f"T{T} dfc", T, 512, 2048, "kn"), (f"T{T} dproj", T, 2048, 512, "kn"),
               (f"T{T} wqkv", 1536, 512, T, "mm"), (f"T{T} wfc", 2048, 512, T, "mm"), (f"T{T} wproj", 512, 2048, T, "mm"), (f"T{T} wout", 512, 512, T, "mm")]
shapes += [("moco S", 2816, 12288, 512, "kn"), ("moco dq", 2816, 512, 12288, "kk"), ("moco dS^T q", 12288, 512, 2816, "mm"),
           ("mlm logits", 1100, 49408, 512, "kk"), ("mlm dt", 1100, 512, 49408, "kn"), ("mlm dW", 49408, 512, 1100, "mm"),
           ("mlp fc1", 1536, 4096, 512, "kk"), ("mlp fc2", 1536, 512, 4096, "kk"), ("mlp dW1", 4096, 512, 1536, "mm"), ("mlp dx", 1536, 512, 4096, "kn"),
           ("sim 256", 256, 3072, 512, "kk"), ("sim 3072", 3072, 256, 512, "kk")]
for name, M, N, K, lay in shapes:
    if lay == "kk":
        a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(N, K, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K))
    elif lay == "kn":
        a = torch.randn(M, K, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (K, 1), (N, 1))
    else:
        a = torch.randn(K, M, device="cuda", generator=g); b = torch.randn(K, N, device="cuda", generator=g)
        f = lambda: ops.gemm_f32(a, b, M, N, K, (1, M), (N, 1))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"{name:14s} {lay} {M}x{N}x{K} {us:8.1f}", flush=True)
