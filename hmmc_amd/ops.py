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
EPI_COLSUM = 32
EPI_SAVE_DGELU, EPI_MULAUX = 64, 128
EPI_LNFOLD, EPI_ROWSTAT = 256, 512

_ws_cache = {}

_reserved = None


def reserve_cus_for_collectives():
    """Called by the towers when they run under data parallelism: keep `HMMC_RCCL_CUS` (default 16 of 256) compute units
    out of the persistent GEMM grids so RCCL's all-reduce workgroups run beside them (include/hmmc_hip.h,
    hmmc_gemm_reserve_cus).  Idempotent."""
    global _reserved
    if _reserved is None:
        import os
        _reserved = int(os.environ.get("HMMC_RCCL_CUS", "16"))
        rc = _lib.load().hmmc_gemm_reserve_cus(_reserved)
        if rc:
            raise RuntimeError(f"hmmc_gemm_reserve_cus({_reserved}) failed: {_lib.ERRORS.get(rc, rc)}")
    return _reserved


def gemm_profile_start():
    """bench.py: time every hmmc_gemm_f16 / hmmc_gemm_f32 launch (also those issued by the native tower runtime) with HIP
    events recorded on the launch stream."""
    _lib.load().hmmc_gemm_profile_start()


def gemm_profile_stop():
    """-> {layout: {"flops", "bytes", "seconds", "launches"}} (synchronises)."""
    import ctypes
    flops, nbytes, secs, n = (ctypes.c_double * 4)(), (ctypes.c_double * 4)(), (ctypes.c_double * 4)(), (ctypes.c_long * 4)()
    rc = _lib.load().hmmc_gemm_profile_stop(flops, nbytes, secs, n)
    if rc:
        raise RuntimeError(f"hmmc_gemm_profile_stop failed: {rc}")
    names = ("fwd_kk", "dgrad_km", "wgrad_mm", "f32")             # three operand layouts of hmmc_gemm_f16; hmmc_gemm_f32
    return {names[i]: {"flops": flops[i], "bytes": nbytes[i], "seconds": secs[i], "launches": n[i]} for i in range(4) if n[i]}


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, stream, tag); kernels never allocate.  Keyed by stream because the two
    encoder towers may run on different streams at the same time."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_aux_streams = {}


def aux_stream(device, tag):
    """A helper stream tied to (device, current stream, tag): e.g. the weight-gradient stream of a tower backward."""
    cur = torch.cuda.current_stream(device)
    key = (str(device), cur.cuda_stream, tag)
    st = _aux_streams.get(key)
    if st is None:
        st = _aux_streams[key] = torch.cuda.Stream(device=device)
    return st


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=True, bias=None, resid=None, aux_in=None, epilogue=0,
             want_aux=False, out=None, want_colsum=False):
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
        epilogue |= EPI_RESID
    if aux_in is not None:
        assert tuple(aux_in.shape) == (M, N) and epilogue & (EPI_DGELU | EPI_MULAUX)
    if want_colsum:                               # fp32 partial column sums of C, one row per 128 / 64 output rows
        rows = query("hmmc_gemm_f16_colsum_rows", M, N, K)
        part = torch.empty((rows, N), dtype=torch.float32, device=a.device)
        call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, a.shape[1], b.shape[1], N, int(a_kmajor), int(b_kmajor),
             ptr(bias), ptr(resid), ptr(aux_out), ptr(aux_in), epilogue | EPI_COLSUM, ptr(part), part.numel() * 4)
        return (c, aux_out, part) if want_aux else (c, part)
    wsb = 0 if epilogue else query("hmmc_gemm_f16_workspace", M, N, K)
    ws = workspace(wsb, a.device, "gemm") if wsb else None
    call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, a.shape[1], b.shape[1], N, int(a_kmajor), int(b_kmajor),
         ptr(bias), ptr(resid), ptr(aux_out), ptr(aux_in), epilogue, ptr(ws), wsb)
    return (c, aux_out) if want_aux else c


def ln_fold_prep(items):
    """items: [(W [N,K] fp16, gamma [K] fp32, beta [K] fp32, bias [N] fp16 or None)] -> [(gamma o W fp16, cd [2,N] fp32)]:
    the operands of a GEMM with the LayerNorm in front of it folded in (include/hmmc_hip.h, hmmc_ln_fold_prep)."""
    import ctypes
    n = len(items)
    K = items[0][0].shape[1]
    outs = []
    for W, gm, bt, bias in items:
        _chk(W, torch.float16, "W"); _chk(gm, torch.float32, "gamma"); _chk(bt, torch.float32, "beta")
        assert W.shape[1] == K and gm.numel() == K and bt.numel() == K
        if bias is not None:
            _chk(bias, torch.float16, "bias")
        outs.append((torch.empty_like(W), torch.empty((2, W.shape[0]), dtype=torch.float32, device=W.device)))
    P = ctypes.c_void_p * n
    pa = lambda ts: P(*[None if t is None else t.data_ptr() for t in ts])
    call("hmmc_ln_fold_prep", pa([i[0] for i in items]), pa([i[1] for i in items]), pa([i[2] for i in items]), pa([i[3] for i in items]),
         pa([o[0] for o in outs]), pa([o[1] for o in outs]), (ctypes.c_int * n)(*[i[0].shape[0] for i in items]), K, n)
    return outs


def rowstat(x):
    """[rows, 2] fp32 = (rstd_r, -rstd_r mean_r) of fp16 rows (eps 1e-5: the CLIP LayerNorm)."""
    _chk(x, torch.float16, "x")
    rows, D = x.shape
    out = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    call("hmmc_rowstat", ptr(x), ptr(out), rows, D, D, 1e-5)
    return out


def rowstat_finalize(part, D):
    """part [D / 64, rows, 2] (sum, sum of squares) from an EPI_ROWSTAT launch -> [rows, 2] as rowstat()."""
    _chk(part, torch.float32, "part")
    nparts, rows, _ = part.shape
    out = torch.empty((rows, 2), dtype=torch.float32, device=part.device)
    call("hmmc_rowstat_finalize", ptr(part), ptr(out), nparts, rows, D, 1e-5)
    return out


def gemm_f16_fold(a, b, bias=None, resid=None, epilogue=0, rowstat=None, colterms=None, want_stat=False, out=None):
    """C = epi(a[M,K] b[N,K]^T) with a LayerNorm folded in (EPI_LNFOLD: rowstat [M,2], colterms [2,N]) and / or the row
    statistics of C emitted per 64-column block (want_stat -> second result [N/64, M, 2])."""
    _chk(a, torch.float16, "a"); _chk(b, torch.float16, "b")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    c = out if out is not None else torch.empty((M, N), dtype=torch.float16, device=a.device)
    if bias is not None:
        _chk(bias, torch.float16, "bias"); epilogue |= EPI_BIAS
    if resid is not None:
        _chk(resid, torch.float16, "resid"); assert tuple(resid.shape) == (M, N); epilogue |= EPI_RESID
    if rowstat is not None:
        _chk(rowstat, torch.float32, "rowstat"); _chk(colterms, torch.float32, "colterms")
        assert tuple(rowstat.shape) == (M, 2) and tuple(colterms.shape) == (2, N)
        epilogue |= EPI_LNFOLD
    part = None
    if want_stat:
        part = torch.empty((N // 64, M, 2), dtype=torch.float32, device=a.device)
        epilogue |= EPI_ROWSTAT
    call("hmmc_gemm_f16_fold", ptr(a), ptr(b), ptr(c), M, N, K, K, K, N, 1, ptr(bias), ptr(resid), None, None, epilogue, ptr(rowstat),
         ptr(colterms), ptr(part), None, 0)
    return (c, part) if want_stat else c


def gemm_f16_rowscaled_dgrad(dy, w, aux, rowstat):
    """rstd_r x [(dy[M,N'] w[N',K']) o aux[M,K']] with the column sums of the unscaled product (EPI_MULAUX | COLSUM | ROWSCALE): the
    data gradient in front of a folded LayerNorm.  -> (scaled gradient [M,K'], partial column sums fp32 [rows, K'])."""
    _chk(dy, torch.float16, "dy"); _chk(w, torch.float16, "w"); _chk(aux, torch.float16, "aux"); _chk(rowstat, torch.float32, "rowstat")
    M, Np = dy.shape
    Kp = w.shape[1]
    assert w.shape[0] == Np and tuple(aux.shape) == (M, Kp) and tuple(rowstat.shape) == (M, 2)
    c = torch.empty((M, Kp), dtype=torch.float16, device=dy.device)
    rows = query("hmmc_gemm_f16_colsum_rows", M, Kp, Np)
    part = torch.empty((rows, Kp), dtype=torch.float32, device=dy.device)
    call("hmmc_gemm_f16_fold", ptr(dy), ptr(w), ptr(c), M, Kp, Np, Np, Kp, Kp, 0, None, None, None, ptr(aux),
         EPI_MULAUX | EPI_COLSUM | 1024, ptr(rowstat), None, None, ptr(part), part.numel() * 4)
    return c, part


def layernorm_bwd_fold(dut, x, stat, dres=None, want_colsum=False, reduce=True):
    """dx = du~ - mean(du~) - u mean(du~ o u) (+ dres), u = stat[:, 0] x + stat[:, 1] (include/hmmc_hip.h) [, column sums of dx]"""
    _chk(dut, torch.float16, "dut"); _chk(x, torch.float16, "x"); _chk(stat, torch.float32, "stat")
    rows, D = x.shape
    dx = torch.empty_like(x)
    part = torch.empty((query("hmmc_layernorm_bwd_fold_rows", rows), D), dtype=torch.float32, device=x.device) if want_colsum else None
    call("hmmc_layernorm_bwd_fold", ptr(dut), ptr(x), ptr(stat), ptr(dres), ptr(dx), ptr(part), int(want_colsum), rows, D, D)
    return (dx, part.sum(0) if reduce else part) if want_colsum else dx


def fold_grad_finish(items):
    """items: [(S fp32 [N,K], W fp16 [N,K], gamma, beta fp32 [K], db fp16 [N])] -> [(dW fp16 [N,K], dgamma, dbeta fp32 [K])]"""
    import ctypes
    n = len(items)
    K = items[0][0].shape[1]
    outs, vms = [], []
    for S, W, gm, bt, db in items:
        _chk(S, torch.float32, "S"); _chk(W, torch.float16, "W"); _chk(gm, torch.float32, "gamma"); _chk(bt, torch.float32, "beta")
        _chk(db, torch.float16, "db")
        outs.append((torch.empty_like(W), torch.empty(K, dtype=torch.float32, device=W.device), torch.empty(K, dtype=torch.float32, device=W.device)))
        vms.append(torch.empty(query("hmmc_fold_grad_scratch_floats", W.shape[0], K), dtype=torch.float32, device=W.device))
    P = ctypes.c_void_p * n
    pa = lambda ts: P(*[t.data_ptr() for t in ts])
    call("hmmc_fold_grad_finish", pa([i[0] for i in items]), pa([i[1] for i in items]), pa([i[2] for i in items]), pa([i[3] for i in items]),
         pa([i[4] for i in items]), pa([o[0] for o in outs]), pa([o[1] for o in outs]), pa([o[2] for o in outs]), pa(vms),
         (ctypes.c_int * n)(*[i[0].shape[0] for i in items]), K, n)
    return outs


def wgrad_group(dys, xs):
    """[dY_j^T X_j for j] (fp16, up to four problems over the same token count) in one grouped launch; None when the shapes
    should take one gemm_f16 call each (hmmc_gemm_f16_wgrad_group_workspace == 0)."""
    import ctypes
    n = len(dys)
    T = dys[0].shape[0]
    for dy, x in zip(dys, xs):
        _chk(dy, torch.float16, "dy")
        _chk(x, torch.float16, "x")
        assert dy.shape[0] == T and x.shape[0] == T
    Np = (ctypes.c_int * n)(*[dy.shape[1] for dy in dys])
    Kp = (ctypes.c_int * n)(*[x.shape[1] for x in xs])
    wsb = query("hmmc_gemm_f16_wgrad_group_workspace", Np, Kp, n, T)
    if wsb == 0:
        return None
    ws = workspace(wsb, dys[0].device, "gemm_group")
    outs = [torch.empty((dy.shape[1], x.shape[1]), dtype=torch.float16, device=dy.device) for dy, x in zip(dys, xs)]
    P = ctypes.c_void_p * n
    call("hmmc_gemm_f16_wgrad_group", P(*[t.data_ptr() for t in dys]), P(*[t.data_ptr() for t in xs]), P(*[t.data_ptr() for t in outs]),
         None, Np, Kp, n, T, ptr(ws), wsb)
    return outs


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


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, row_index=None, dx=None, want_colsum=False):
    """Returns (dx, dgamma, dbeta[, colsum(dx)]).  With row_index, dx must be a pre-zeroed tensor shaped like x."""
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
    dxs = torch.empty(D, dtype=x.dtype, device=x.device) if want_colsum else None
    call("hmmc_layernorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dgamma),
         ptr(dbeta), ptr(dxs), ptr(row_index), rows, D, D, dt, ptr(ws), wsb)
    return (dx, dgamma, dbeta, dxs) if want_colsum else (dx, dgamma, dbeta)


def colsum(x2d, out_dtype=None, round_f16=False, out=None):
    """out[n] = sum_m x[m][n] (fp32 accumulation).  out: a contiguous [N] destination (a slice of a larger buffer)."""
    M, N = x2d.shape
    in_dt = 0 if x2d.dtype == torch.float16 else 1
    out_dtype = out_dtype or (out.dtype if out is not None else x2d.dtype)
    _chk(x2d, x2d.dtype, "x")
    if out is None:
        out = torch.empty(N, dtype=out_dtype, device=x2d.device)
    else:
        _chk(out, out_dtype, "out")
        assert out.numel() == N
    wsb = query("hmmc_colsum_workspace", M, N)
    ws = workspace(wsb, x2d.device, "colsum")
    call("hmmc_colsum", ptr(x2d), ptr(out), M, N, N, in_dt, 0 if out_dtype == torch.float16 else 1, int(round_f16),
         ptr(ws), wsb)
    return out


def _dt(dtype):
    """dtype code of the C-ABI: 0 fp16 (the towers as written), 1 fp32 (the reference after model.float())."""
    if dtype == torch.float16:
        return 0
    if dtype == torch.float32:
        return 1
    raise TypeError(f"the CLIP towers run in fp16 (as written) or fp32 (model.float()), not {dtype}")


def patchify(video4d, patch, dtype=torch.float16):
    """fp32 [n,3,H,W] -> `dtype` [n*(g*g+1), 3*p*p] with a zero row in every frame's class-token slot."""
    _chk(video4d, torch.float32, "video")
    n, c, H, W = video4d.shape
    assert c == 3
    g = H // patch
    out = torch.empty((n * (g * g + 1), 3 * patch * patch), dtype=dtype, device=video4d.device)
    call("hmmc_patchify", ptr(video4d), ptr(out), n, H, W, patch, _dt(dtype))
    return out


CLIP_PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)      # dataloaders/rawvideo_util.py / dataloader_*: Normalize(...)
CLIP_PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def patchify_u8(video4d_u8, patch, mean=CLIP_PIXEL_MEAN, std=CLIP_PIXEL_STD, frame_index=None, dtype=torch.float16):
    """uint8 [n,3,H,W] -> fp16 patches as patchify(), with x/255 and the per-channel normalisation fused in.
    frame_index (int32 [m] on the device): patches of the m frames video4d_u8[frame_index[i]] instead (frame sampling)."""
    import ctypes
    _chk(video4d_u8, torch.uint8, "video")
    n, c, H, W = video4d_u8.shape
    assert c == 3
    if frame_index is not None:
        _chk(frame_index, torch.int32, "frame_index")
        n = frame_index.numel()
    g = H // patch
    out = torch.empty((n * (g * g + 1), 3 * patch * patch), dtype=dtype, device=video4d_u8.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    call("hmmc_patchify_u8", ptr(video4d_u8), ptr(frame_index), ptr(out), n, H, W, patch, m3, s3, _dt(dtype))
    return out


def vit_embed_(x, cls, pos, L):
    _chk(x, x.dtype, "x")
    _chk(cls, torch.float32, "cls")
    _chk(pos, torch.float32, "pos")
    call("hmmc_vit_embed", ptr(x), ptr(cls), ptr(pos), x.shape[0], L, x.shape[1], _dt(x.dtype))
    return x


def vit_embed_ln_(x0, cls, pos, gamma, beta, L, want_stat=False, write_x0=True):
    """vit_embed_ + layernorm_fwd(eps 1e-5) in one pass (fp16): x0 is rewritten with the embedded rows when write_x0.
    -> (y, mean, rstd, stat or None); stat [rows, 2] = (rstd, -rstd mean) of the rows of y."""
    _chk(x0, torch.float16, "x0")
    rows, D = x0.shape
    y = torch.empty_like(x0)
    mean = torch.empty(rows, dtype=torch.float32, device=x0.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x0.device)
    stat = torch.empty((rows, 2), dtype=torch.float32, device=x0.device) if want_stat else None
    call("hmmc_vit_embed_ln", ptr(x0), ptr(cls), ptr(pos), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), ptr(stat), rows, L, D,
         1e-5, int(write_x0))
    return y, mean, rstd, stat


def text_embed(ids, table, pos, dtype=torch.float16):
    _chk(ids, torch.int64, "ids")
    _chk(table, torch.float32, "table")
    _chk(pos, torch.float32, "pos")
    b, L = ids.shape
    D = table.shape[1]
    x = torch.empty((b * L, D), dtype=dtype, device=ids.device)
    call("hmmc_text_embed", ptr(ids), ptr(table), ptr(pos), ptr(x), b * L, L, D, table.shape[0], ptr(device_error_flag(ids.device)),
         _dt(dtype))
    return x


def eot_index(ids, base=0, stride=None):
    """int32 [b]: base + i * stride + argmax(ids[i]) - row of caption i's EOT token (largest id, first occurrence)."""
    _chk(ids, torch.int64, "ids")
    b, L = ids.shape
    out = torch.empty(b, dtype=torch.int32, device=ids.device)
    call("hmmc_eot_index", ptr(ids), ptr(out), b, L, int(base), int(L if stride is None else stride))
    return out


def text_embed_bwd(ids, dx, vocab):
    _chk(dx, dx.dtype, "dx")
    D = dx.shape[-1]
    dtable = torch.zeros((vocab, D), dtype=torch.float32, device=dx.device)
    call("hmmc_text_embed_bwd", ptr(ids), ptr(dx), ptr(dtable), ids.numel(), D, vocab, _dt(dx.dtype))
    return dtable


_ERR_FLAGS = {}


def _device_key(device):
    idx = torch.device(device).index
    return idx if idx is not None else torch.cuda.current_device()      # a bare "cuda" means the current device, on every rank


def device_error_flag(device):
    """int32 [1] on `device` that kernels set instead of faulting (token ids outside the embedding table)."""
    key = _device_key(device)
    if key not in _ERR_FLAGS:
        _ERR_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _ERR_FLAGS[key]


def raise_on_device_errors(device=None):
    """Synchronising check of the device error flags (what nn.Embedding's IndexError is in the reference)."""
    keys = list(_ERR_FLAGS) if device is None else [_device_key(device)]
    for k in keys:
        f = _ERR_FLAGS.get(k)
        if f is not None and int(f.item()) != 0:
            f.zero_()
            raise IndexError("a device-side check failed since the last call: a token id outside the embedding table (the row was "
                             "embedded as zeros), or more MLM-labelled positions than the head's row buffer holds")


def attention_f16_fwd(qkv, nseq, L, H, causal):
    _chk(qkv, torch.float16, "qkv")
    D = H * 64
    assert tuple(qkv.shape) == (nseq * L, 3 * D)
    out = torch.empty((nseq * L, D), dtype=torch.float16, device=qkv.device)
    lse = torch.empty((nseq, H, L), dtype=torch.float32, device=qkv.device)
    call("hmmc_attention_f16_fwd", ptr(qkv), ptr(out), ptr(lse), nseq, L, H, int(causal))
    return out, lse


def attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, want_dbias=False, rowstat=None):
    """-> dqkv [, per-sequence column sums of dqkv: fp32 [nseq, 3*64*H] ]; rowstat [tokens, 2]: row r of dqkv leaves multiplied by
    rowstat[r, 0] (the bias partials stay unscaled)"""
    _chk(dout, torch.float16, "dout")
    dqkv = torch.empty_like(qkv)
    part = torch.empty((nseq, qkv.shape[1]), dtype=torch.float32, device=qkv.device) if want_dbias else None
    if rowstat is not None:
        _chk(rowstat, torch.float32, "rowstat")
        call("hmmc_attention_f16_bwd_scaled", ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), ptr(part), ptr(rowstat), nseq, L, H,
             int(causal))
    else:
        call("hmmc_attention_f16_bwd", ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), ptr(part), nseq, L, H, int(causal))
    return (dqkv, part) if want_dbias else dqkv


def attention_f16_fwd_lead(qkv, nseq, L, H, causal, out=None, lse=None):
    """Query 0 of every sequence only: row n*L of `out` and lse[n, h, 0] (the other rows / entries are not written; fresh buffers
    are filled with NaN so that a consumer of an unwritten row is noticed).  Reads Q at token 0 only."""
    _chk(qkv, torch.float16, "qkv")
    D = H * 64
    assert tuple(qkv.shape) == (nseq * L, 3 * D)
    if out is None:
        out = torch.full((nseq * L, D), float("nan"), dtype=torch.float16, device=qkv.device)
    if lse is None:
        lse = torch.full((nseq, H, L), float("nan"), dtype=torch.float32, device=qkv.device)
    call("hmmc_attention_f16_fwd_lead", ptr(qkv), ptr(out), ptr(lse), nseq, L, H, int(causal))
    return out, lse


def attention_f16_bwd_lead(qkv, lse, dout, nseq, L, H, causal, want_dbias=False, rowstat=None, dqkv=None, out=None):
    """Backward of attention_f16_fwd_lead: dout (and out, needed above 64 tokens) is read at row n*L only; dK, dV of every token
    and dQ of token 0 are written."""
    _chk(dout, torch.float16, "dout")
    if dqkv is None:
        dqkv = torch.full_like(qkv, float("nan"))
    part = torch.empty((nseq, qkv.shape[1]), dtype=torch.float32, device=qkv.device) if want_dbias else None
    if rowstat is not None:
        _chk(rowstat, torch.float32, "rowstat")
    call("hmmc_attention_f16_bwd_lead", ptr(qkv), ptr(out), ptr(lse), ptr(dout), ptr(dqkv), ptr(part), ptr(rowstat), nseq, L, H, int(causal))
    return (dqkv, part) if want_dbias else dqkv


# ----------------------------------------------------------------------------- fp32 side

EPI_RELU = 16


def gemm_f32(a, b, M, N, K, sa, sb, alpha=1.0, bias=None, resid=None, aux_in=None, epilogue=0, want_aux=False, out=None):
    """C[M,N] = epi(alpha * sum_k A[m*sa[0] + k*sa[1]] * B[k*sb[0] + n*sb[1]]); a, b are fp32 buffers (any shape)."""
    _chk(a, torch.float32, "a")
    _chk(b, torch.float32, "b")
    assert (M - 1) * sa[0] + (K - 1) * sa[1] < a.numel(), ("A out of range", M, K, sa, a.shape)
    assert (K - 1) * sb[0] + (N - 1) * sb[1] < b.numel(), ("B out of range", K, N, sb, b.shape)
    c = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=a.device)
    _chk(c, torch.float32, "out")
    assert c.numel() == M * N
    aux_out = torch.empty_like(c) if want_aux else None
    for t, n in ((bias, "bias"), (resid, "resid"), (aux_in, "aux_in")):
        if t is not None:
            _chk(t, torch.float32, n)
    if bias is not None:
        assert bias.numel() == N
        epilogue |= EPI_BIAS
    if resid is not None:
        assert resid.numel() == M * N
        epilogue |= EPI_RESID
    if aux_in is not None:
        assert aux_in.numel() == M * N
    call("hmmc_gemm_f32", ptr(a), ptr(b), ptr(c), M, N, K, sa[0], sa[1], sb[0], sb[1], N, float(alpha), ptr(bias),
         ptr(resid), ptr(aux_out), ptr(aux_in), epilogue)
    return (c, aux_out) if want_aux else c


def linear_f32(x, w, bias=None, resid=None, epilogue=0, want_aux=False, aux_in=None):
    """y = x @ w.T (+bias): x [M,K], w [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    return gemm_f32(x, w, M, N, K, (K, 1), (1, K), bias=bias, resid=resid, epilogue=epilogue, want_aux=want_aux,
                    aux_in=aux_in)


def dgrad_f32(dy, w, aux_in=None, epilogue=0):
    """dx = dy @ w: dy [M,N'], w [N',K'] -> [M,K']."""
    M, Np = dy.shape
    Kp = w.shape[1]
    return gemm_f32(dy, w, M, Kp, Np, (Np, 1), (Kp, 1), aux_in=aux_in, epilogue=epilogue)


def wgrad_f32(dy, x):
    """dW = dy.T @ x: dy [T,N'], x [T,K'] -> [N',K']."""
    T, Np = dy.shape
    Kp = x.shape[1]
    return gemm_f32(dy, x, Np, Kp, T, (1, Np), (Kp, 1))


def l2norm_fwd(x, eps=0.0, out=None):
    _chk(x, torch.float32, "x")
    rows, D = x.shape
    y = out if out is not None else torch.empty_like(x)
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    call("hmmc_l2norm_fwd", ptr(x), ptr(y), ptr(norm), rows, D, float(eps))
    return y, norm


def l2norm_bwd(dy, y, norm):
    _chk(dy, torch.float32, "dy")
    rows, D = y.shape
    dx = torch.empty_like(y)
    call("hmmc_l2norm_bwd", ptr(dy), ptr(y), ptr(norm), ptr(dx), rows, D)
    return dx


def infonce_fwd(S, B, F, w_video, w_frame):
    _chk(S, torch.float32, "S")
    assert tuple(S.shape) == (B, B * (1 + F))
    lse_row = torch.empty((B, 1 + F), dtype=torch.float32, device=S.device)
    lse_col = torch.empty(B * (1 + F), dtype=torch.float32, device=S.device)
    loss = torch.empty((), dtype=torch.float32, device=S.device)
    call("hmmc_infonce_fwd", ptr(S), ptr(lse_row), ptr(lse_col), ptr(loss), B, F, float(w_video), float(w_frame))
    return loss, lse_row, lse_col


def infonce_bwd(S, lse_row, lse_col, gout, B, F, w_video, w_frame):
    gout = gout.contiguous().float()
    dS = torch.empty_like(S)
    call("hmmc_infonce_bwd", ptr(S), ptr(lse_row), ptr(lse_col), ptr(gout), ptr(dS), B, F, float(w_video), float(w_frame))
    return dS


def topk_mean(S_frame, bq, bv, F, k, base=None, lds=None, ldb=None):
    """out[i][b] = base[i][b] + mean(topk_f S_frame[i][b*F+f])."""
    _chk(S_frame, torch.float32, "S_frame")
    out = torch.empty((bq, bv), dtype=torch.float32, device=S_frame.device)
    call("hmmc_topk_mean", ptr(S_frame), ptr(base), ptr(out), bq, bv, F, k, lds or bv * F, ldb or bv)
    return out


def eval_slots(F):
    """rows per video of the packed candidate matrix of the fused eval scorer (0: too many frames for it)."""
    return query("hmmc_eval_slots", int(F))


def eval_pack(visual, frames):
    """[nv, E], [nv, F, E] -> unit rows [nv * slots, E] (slot 0 video embedding, 1..F frames, rest zero)."""
    _chk(visual, torch.float32, "visual")
    _chk(frames, torch.float32, "frames")
    nv, F, E = frames.shape
    P = eval_slots(F)
    packed = torch.empty((nv * P, E), dtype=torch.float32, device=visual.device)
    call("hmmc_eval_pack", ptr(visual.contiguous()), ptr(frames.contiguous()), ptr(packed), nv, F, E)
    return packed


def eval_score(queries_unit, packed, nv, F, k, scale, want=("video", "frame")):
    """fused scale * Q [V; U]^T -> (video logits, mean top-k frame logits, their sum), each [nq, nv] or None."""
    _chk(queries_unit, torch.float32, "queries_unit")
    nq, E = queries_unit.shape
    outs = [torch.empty((nq, nv), dtype=torch.float32, device=packed.device) if w in want else None
            for w in ("video", "frame", "score")]
    call("hmmc_eval_score", ptr(queries_unit), ptr(packed), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), nq, nv, F, E, int(k), float(scale))
    return outs


def segment_max(sim, offsets):
    """out[g][v] = max over rows offsets[g] <= s < offsets[g+1] of sim[s][v]; offsets int32 [G + 1] on the device."""
    _chk(sim, torch.float32, "sim")
    _chk(offsets, torch.int32, "offsets")
    G, V = offsets.numel() - 1, sim.shape[1]
    out = torch.empty((G, V), dtype=torch.float32, device=sim.device)
    call("hmmc_segment_max", ptr(sim), ptr(offsets), ptr(out), G, V, sim.stride(0))
    return out


def retrieval_rank(sim, target=None, transposed=False):
    """rank[q] = #{j : S(q, j) > S(q, target[q])} (int32); S(q, j) = sim[q, j], or sim[j, q] with transposed."""
    _chk(sim, torch.float32, "sim")
    assert sim.dim() == 2
    R, C = sim.shape
    Q, V = (C, R) if transposed else (R, C)
    if target is not None:
        _chk(target, torch.int32, "target")
        assert target.numel() == Q
    rank = torch.empty(Q, dtype=torch.int32, device=sim.device)
    call("hmmc_retrieval_rank", ptr(sim), ptr(target), ptr(rank), Q, V, C, int(transposed))
    return rank


def temporal_pool_fwd(h, u, b, F, D):
    out = torch.empty((b, D), dtype=torch.float32, device=h.device)
    norms = torch.empty((b, F), dtype=torch.float32, device=h.device)
    call("hmmc_temporal_pool_fwd", ptr(h), ptr(u), ptr(out), ptr(norms), b, F, D)
    return out, norms


def temporal_pool_bwd(h, u, norms, dout, b, F, D):
    dout = dout.contiguous()
    dvf = torch.empty((b * F, D), dtype=torch.float32, device=h.device)
    call("hmmc_temporal_pool_bwd", ptr(h), ptr(u), ptr(norms), ptr(dout), ptr(dvf), b, F, D)
    return dvf


def add_rowbias(x, table, period):
    rows, D = x.shape
    out = torch.empty_like(x)
    call("hmmc_add_rowbias", ptr(x), ptr(table), ptr(out), rows, period, D)
    return out


def attention_f32_fwd(qkv, b, F, H, causal=False):
    _chk(qkv, torch.float32, "qkv")
    D = H * 64
    out = torch.empty((b * F, D), dtype=torch.float32, device=qkv.device)
    probs = torch.empty((b, H, F, F), dtype=torch.float32, device=qkv.device)
    call("hmmc_temporal_attention_fwd", ptr(qkv), ptr(out), ptr(probs), b, F, H, int(causal))
    return out, probs


def attention_f32_bwd(qkv, probs, dout, b, F, H):
    dqkv = torch.empty_like(qkv)
    call("hmmc_temporal_attention_bwd", ptr(qkv), ptr(probs), ptr(dout), ptr(dqkv), b, F, H)
    return dqkv


def enqueue(keys, queue, col0):
    _chk(keys, torch.float32, "keys")
    _chk(queue, torch.float32, "queue")
    R, E = keys.shape
    assert queue.shape[0] == E
    call("hmmc_enqueue", ptr(keys), ptr(queue), R, E, queue.shape[1], int(col0))


# ----------------------------------------------------------------------------- pre-training (MoCo) heads

def bn_stats(h):
    """-> sums [2, N]: column sum and sum of squares of this rank's rows."""
    M, N = h.shape
    sums = torch.empty((2, N), dtype=torch.float32, device=h.device)
    wsb = query("hmmc_bn_workspace", M, N)
    ws = workspace(wsb, h.device, "bn")
    call("hmmc_bn_stats", ptr(h), ptr(sums), M, N, ptr(ws), wsb)
    return sums


def bn_finalize(sums, n, eps, momentum=0.0, running_mean=None, running_var=None, num_batches_tracked=None):
    """sums [2, N] (hmmc_bn_stats, summed over ranks) -> (mean, biased var, rstd); n: the global row count, a python number or a
    one-element device tensor.  With running_mean / running_var the train-mode running statistics are updated in place."""
    N = sums.shape[-1] if sums.dim() == 2 else sums.numel() // 2
    _chk(sums, torch.float32, "sums")
    out = torch.empty((3, N), dtype=torch.float32, device=sums.device)
    n_dev, n_host = (n, 0.0) if isinstance(n, torch.Tensor) else (None, float(n))
    if n_dev is not None:
        _chk(n_dev, torch.float32, "n")
    for t in (running_mean, running_var):
        if t is not None:
            _chk(t, torch.float32, "running statistics")
    if num_batches_tracked is not None:
        _chk(num_batches_tracked, torch.int64, "num_batches_tracked")
    call("hmmc_bn_finalize", ptr(sums), ptr(n_dev), n_host, float(eps), float(momentum), ptr(out[0]), ptr(out[1]), ptr(out[2]),
         ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), N)
    return out[0], out[1], out[2]


def bn_apply_relu(h, mean, rstd, gamma, beta):
    M, N = h.shape
    y = torch.empty_like(h)
    call("hmmc_bn_apply_relu", ptr(h), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(y), M, N)
    return y


def bn_bwd_reduce(dy, y, h, mean, rstd):
    M, N = h.shape
    sums = torch.empty((2, N), dtype=torch.float32, device=h.device)
    wsb = query("hmmc_bn_workspace", M, N)
    ws = workspace(wsb, h.device, "bn")
    call("hmmc_bn_bwd_reduce", ptr(dy), ptr(y), ptr(h), ptr(mean), ptr(rstd), ptr(sums), M, N, ptr(ws), wsb)
    return sums


def bn_bwd_apply(dy, y, h, mean, rstd, gamma, sums, n_global):
    M, N = h.shape
    dh = torch.empty_like(h)
    call("hmmc_bn_bwd_apply", ptr(dy), ptr(y), ptr(h), ptr(mean), ptr(rstd), ptr(gamma), ptr(sums), ptr(dh), M, N,
         1.0 / float(n_global))
    return dh


def rowdot(a, b):
    rows, D = a.shape
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    call("hmmc_rowdot", ptr(a), ptr(b), ptr(out), rows, D)
    return out


def moco_loss_fwd(S, lpos, temperature, w):
    R, Kq = S.shape
    lse = torch.empty(R, dtype=torch.float32, device=S.device)
    rowloss = torch.empty(R, dtype=torch.float32, device=S.device)
    loss = torch.empty((), dtype=torch.float32, device=S.device)
    call("hmmc_moco_loss_fwd", ptr(S), ptr(lpos), ptr(lse), ptr(rowloss), ptr(loss), R, Kq, float(temperature), float(w))
    return loss, lse


def moco_loss_bwd_(S, lpos, lse, gout, temperature, w):
    """Overwrites S with dS; returns dlpos."""
    R, Kq = S.shape
    gout = gout.contiguous().float()
    dlpos = torch.empty(R, dtype=torch.float32, device=S.device)
    call("hmmc_moco_loss_bwd", ptr(S), ptr(lpos), ptr(lse), ptr(gout), ptr(dlpos), R, Kq, float(temperature), float(w))
    return dlpos


def row_axpy_(y, s, x):
    rows, D = y.shape
    call("hmmc_row_axpy", ptr(y), ptr(s), ptr(x), rows, D)
    return y


def gelu_erf_fwd(x):
    y = torch.empty_like(x)
    call("hmmc_gelu_erf_fwd", ptr(x), ptr(y), x.numel())
    return y


def gelu_erf_bwd(x, dy):
    dx = torch.empty_like(x)
    call("hmmc_gelu_erf_bwd", ptr(x), ptr(dy), ptr(dx), x.numel())
    return dx


def ce_fwd(logits, labels):
    """-> (loss_sum, lse, count) with ignore_index < 0."""
    _chk(logits, torch.float32, "logits")
    _chk(labels, torch.int64, "labels")
    R, V = logits.shape
    lse = torch.empty(R, dtype=torch.float32, device=logits.device)
    rowloss = torch.empty(R, dtype=torch.float32, device=logits.device)
    count = torch.empty(1, dtype=torch.float32, device=logits.device)
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    call("hmmc_ce_fwd", ptr(logits), ptr(labels), ptr(lse), ptr(rowloss), ptr(count), ptr(loss), R, V)
    return loss, lse, count


def ce_bwd_(logits, labels, lse, gout, count):
    R, V = logits.shape
    gout = gout.contiguous().float()
    call("hmmc_ce_bwd", ptr(logits), ptr(labels), ptr(lse), ptr(gout), ptr(count), R, V)
    return logits
