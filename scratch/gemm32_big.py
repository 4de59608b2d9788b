"""fp32 GEMM at the pre-training heads' shapes (MoCo logits against the frame queue, MLM vocabulary head, projector MLPs)."""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
tot = 0
shapes = [("moco S    q.queue", 2816, 12288, 512, "kn"), ("moco dq   dS.queue^T", 2816, 512, 12288, "kk"),
          ("mlm logits", 1100, 49408, 512, "kk"), ("mlm dt     dl.W", 1100, 512, 49408, "kn"), ("mlm dW     dl^T.t", 49408, 512, 1100, "mm"),
          ("mlp fc1", 1536, 4096, 512, "kk"), ("mlp fc2", 1536, 512, 4096, "kk"), ("mlp dW1", 4096, 512, 1536, "mm"),
          ("temporal qkv", 3072, 1536, 512, "kk"), ("temporal fc", 3072, 2048, 512, "kk")]
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
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tot += us
    print(f"{name:22s} {M}x{N}x{K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:6.1f} TF", flush=True)
print("total", tot)
