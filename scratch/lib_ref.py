"""Reference points for the HBM-bound kernels: PyTorch's own attention (scaled_dot_product_attention) and LayerNorm on the path's
shapes beside this repo's kernels (forward and backward).  The product never calls them."""
import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import ops
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for nseq, L, H, causal in ((3072, 50, 12, False), (384, 197, 12, False), (256, 32, 8, True)):
    D = H * 64
    qkv = (torch.randn(nseq * L, 3 * D, device="cuda") * 0.5).half()
    dout = torch.randn(nseq * L, D, device="cuda").half()
    out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, causal)
    tf = t(lambda: ops.attention_f16_fwd(qkv, nseq, L, H, causal))
    tb = t(lambda: ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, want_dbias=True))
    q, k, v = [x.contiguous().requires_grad_(True) for x in qkv.view(nseq, L, 3, H, 64).permute(2, 0, 3, 1, 4)]   # [nseq, H, L, 64], as SDPA wants
    do = dout.view(nseq, L, H, 64).permute(0, 2, 1, 3).contiguous()
    res = {}
    for name, be in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION), ("math", torch.nn.attention.SDPBackend.MATH)):
        try:
            with torch.nn.attention.sdpa_kernel(be):
                o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
                lf = t(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=causal))
                def fb():
                    o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
                    o.backward(do)
                lfb = t(fb)
            res[name] = (lf, lfb - lf)
        except Exception as e:
            res[name] = str(e)[:60]
    print(f"attention nseq={nseq} L={L} H={H} causal={causal}: this repo fwd {tf:.1f} us, bwd {tb:.1f} us (reads the packed [tokens, 3D] layout in place) | torch SDPA (pre-permuted contiguous q, k, v): " +
          "; ".join(f"{n}: fwd {r[0]:.1f} bwd {r[1]:.1f}" if isinstance(r, tuple) else f"{n}: {r}" for n, r in res.items()), flush=True)
for rows, D in ((153600, 768), (8192, 512)):
    x = torch.randn(rows, D, device="cuda").half().requires_grad_(True); g = torch.randn(D, device="cuda"); b = torch.randn(D, device="cuda")
    dy = torch.randn(rows, D, device="cuda").half(); dres = torch.randn(rows, D, device="cuda").half()
    y, mean, rstd = ops.layernorm_fwd(x.detach(), g, b, 1e-5)
    tf = t(lambda: ops.layernorm_fwd(x.detach(), g, b, 1e-5))
    tb = t(lambda: ops.layernorm_bwd(dy, x.detach(), g, mean, rstd, dres=dres, want_colsum=True))
    gh, bh = g.half().requires_grad_(True), b.half().requires_grad_(True)
    lf = t(lambda: F.layer_norm(x, (D,), gh, bh, 1e-5))
    def fb():
        F.layer_norm(x, (D,), gh, bh, 1e-5).backward(dy)
    lfb = t(fb)
    print(f"LayerNorm [{rows}, {D}] fp16: this repo fwd {tf:.1f} us, bwd {tb:.1f} us (+ residual-gradient add, + dx column sums) | torch fwd {lf:.1f} us, bwd {lfb - lf:.1f} us", flush=True)
