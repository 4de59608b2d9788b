"""Tensor-level wrappers over the C-ABI (no autograd here; see functional.py).

Every function allocates its outputs with torch (caching allocator), checks dtypes /
contiguity on the host so a kernel can never be launched on a shape it does not expect,
and enqueues on torch's current stream.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call, ptr, query

EPI_BIAS, EPI_RESID, EPI_QGELU, EPI_DGELU = 1, 2, 4, 8

_ws_cache = {}


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag); kernels never allocate."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=True, bias=None, resid=None, aux_in=None, epilogue=0,
             want_aux=False, out=None):
    """C[M,N] = epi(Aop[M,K] @ Bop[N,K]^T).  a: [M,K] if a_kmajor else [K,M]; b: [N,K] if b_kmajor else [K,N]."""
    _chk(a, torch.float16, "a")
    _chk(b, torch.float16, "b")
    assert a.dim() == 2 and b.dim() == 2
    assert tuple(a.shape) == ((M, K) if a_kmajor else (K, M)), (a.shape, M, K, a_kmajor)
    assert tuple(b.shape) == ((N, K) if b_kmajor else (K, N)), (b.shape, N, K, b_kmajor)
    c = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=a.device)
    _chk(c, torch.float16, "out")
    aux_out = torch.empty_like(c) if want_aux else None
    for t, n in ((bias, "bias"), (resid, "resid"), (aux_in, "aux_in")):
        if t is not None:
            _chk(t, torch.float16, n)
    if bias is not None:
        assert bias.numel() == N
        epilogue |= EPI_BIAS
    if resid is not None:
        assert tuple(resid.shape) == (M, N)
    if aux_in is not None:
        assert tuple(aux_in.shape) == (M, N)
    wsb = 0 if epilogue else query("hmmc_gemm_f16_workspace", M, N, K)
    ws = workspace(wsb, a.device, "gemm") if wsb else None
    call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, a.shape[1], b.shape[1], N, int(a_kmajor), int(b_kmajor),
         ptr(bias), ptr(resid), ptr(aux_out), ptr(aux_in), epilogue, ptr(ws), wsb)
    return (c, aux_out) if want_aux else c


def layernorm_fwd(x, gamma, beta, eps, rows=None, row_index=None, in_stride=None):
    """x: [..., D] fp16 or fp32.  Optional row gather: output row r reads x_flat[row_index[r]]."""
    D = x.shape[-1]
    dt = 0 if x.dtype == torch.float16 else 1
    _chk(x, torch.float16 if dt == 0 else torch.float32, "x")
    _chk(gamma, torch.float32, "gamma")
    _chk(beta, torch.float32, "beta")
    if row_index is not None:
        _chk(row_index, torch.int32, "row_index")
        rows = row_index.numel()
    elif rows is None:
        rows = x.numel() // D
    y = torch.empty((rows, D), dtype=x.dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("hmmc_layernorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), ptr(row_index), rows, D,
         in_stride or D, float(eps), dt)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, row_index=None, dx=None):
    """Returns (dx, dgamma, dbeta).  With row_index, dx must be a pre-zeroed tensor shaped like x."""
    D = x.shape[-1]
    dt = 0 if x.dtype == torch.float16 else 1
    rows = mean.numel()
    _chk(dy, x.dtype, "dy")
    if dx is None:
        assert row_index is None
        dx = torch.empty_like(x)
    if dres is not None:
        _chk(dres, x.dtype, "dres")
        assert dres.shape == x.shape
    dgamma = torch.empty(D, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(D, dtype=torch.float32, device=x.device)
    wsb = query("hmmc_layernorm_bwd_workspace", rows, D)
    ws = workspace(wsb, x.device, "ln")
    call("hmmc_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dgamma),
         ptr(dbeta), ptr(row_index), rows, D, D, dt, ptr(ws), wsb)
    return dx, dgamma, dbeta


def colsum(x2d, out_dtype=None, round_f16=False):
    """out[n] = sum_m x[m][n] (fp32 accumulation)."""
    M, N = x2d.shape
    in_dt = 0 if x2d.dtype == torch.float16 else 1
    out_dtype = out_dtype or x2d.dtype
    _chk(x2d, x2d.dtype, "x")
    out = torch.empty(N, dtype=out_dtype, device=x2d.device)
    wsb = query("hmmc_colsum_workspace", M, N)
    ws = workspace(wsb, x2d.device, "colsum")
    call("hmmc_colsum", ptr(x2d), ptr(out), M, N, N, in_dt, 0 if out_dtype == torch.float16 else 1, int(round_f16),
         ptr(ws), wsb)
    return out


def patchify(video4d, patch):
    """fp32 [n,3,H,W] -> fp16 [n*(g*g+1), 3*p*p] with a zero row in every frame's class-token slot."""
    _chk(video4d, torch.float32, "video")
    n, c, H, W = video4d.shape
    assert c == 3
    g = H // patch
    out = torch.empty((n * (g * g + 1), 3 * patch * patch), dtype=torch.float16, device=video4d.device)
    call("hmmc_patchify", ptr(video4d), ptr(out), n, H, W, patch)
    return out


def vit_embed_(x, cls, pos, L):
    _chk(x, torch.float16, "x")
    _chk(cls, torch.float32, "cls")
    _chk(pos, torch.float32, "pos")
    call("hmmc_vit_embed", ptr(x), ptr(cls), ptr(pos), x.shape[0], L, x.shape[1])
    return x


def text_embed(ids, table, pos):
    _chk(ids, torch.int64, "ids")
    _chk(table, torch.float32, "table")
    _chk(pos, torch.float32, "pos")
    b, L = ids.shape
    D = table.shape[1]
    x = torch.empty((b * L, D), dtype=torch.float16, device=ids.device)
    call("hmmc_text_embed", ptr(ids), ptr(table), ptr(pos), ptr(x), b * L, L, D)
    return x


def text_embed_bwd(ids, dx, vocab):
    _chk(dx, torch.float16, "dx")
    D = dx.shape[-1]
    dtable = torch.zeros((vocab, D), dtype=torch.float32, device=dx.device)
    call("hmmc_text_embed_bwd", ptr(ids), ptr(dx), ptr(dtable), ids.numel(), D)
    return dtable


def attention_f16_fwd(qkv, nseq, L, H, causal):
    _chk(qkv, torch.float16, "qkv")
    D = H * 64
    assert tuple(qkv.shape) == (nseq * L, 3 * D)
    out = torch.empty((nseq * L, D), dtype=torch.float16, device=qkv.device)
    lse = torch.empty((nseq, H, L), dtype=torch.float32, device=qkv.device)
    call("hmmc_attention_f16_fwd", ptr(qkv), ptr(out), ptr(lse), nseq, L, H, int(causal))
    return out, lse


def attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal):
    _chk(dout, torch.float16, "dout")
    dqkv = torch.empty_like(qkv)
    call("hmmc_attention_f16_bwd", ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), nseq, L, H, int(causal))
    return dqkv
