"""GPU unit tests of the individual HIP kernels through the C-ABI, each against a plain PyTorch
fp32 reference of the same op (bit-exact on integer-valued data where the op is exact)."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

from hmmc_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def ints(*shape, lo=-2, hi=3, dtype=torch.float16, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return torch.randint(lo, hi, shape, generator=g).to(dtype).to(DEV)


def relerr(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


# ----------------------------------------------------------------------------- GEMM

@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (800, 384, 128), (130, 136, 192), (2048, 768, 768)])
def test_gemm_kk_exact_integers(M, N, K):
    a, b = ints(M, K), ints(N, K, seed=1)
    c = ops.gemm_f16(a, b, M, N, K)
    ref = a.float() @ b.float().t()
    assert torch.equal(c.float(), ref), f"max diff {(c.float()-ref).abs().max()}"


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (800, 384, 200), (1000, 512, 136), (4096, 768, 2304)])
def test_gemm_km_dgrad_exact_integers(M, N, K):
    # dx[M,N] = dy[M,K] @ W[K,N]  (A k-major, B m-major: Bop[n][k] = W[k][n])
    dy, w = ints(M, K), ints(K, N, seed=1)
    if K % 64:
        pytest.skip("k-major operand needs K % 64 == 0")
    c = ops.gemm_f16(dy, w, M, N, K, a_kmajor=True, b_kmajor=False)
    ref = dy.float() @ w.float()
    assert torch.equal(c.float(), ref), f"max diff {(c.float()-ref).abs().max()}"


@pytest.mark.parametrize("T,N,K", [(64, 128, 128), (800, 384, 128), (1000, 136, 264), (12800, 768, 384)])
def test_gemm_mm_wgrad_exact_integers(T, N, K):
    # dW[N,K] = dy[T,N]^T @ x[T,K]   (both m-major, reduction over T incl. ragged tail and split-K)
    dy, x = ints(T, N, lo=-1, hi=2), ints(T, K, lo=-1, hi=2, seed=1)
    c = ops.gemm_f16(dy, x, N, K, T, a_kmajor=False, b_kmajor=False)
    ref = dy.float().t() @ x.float()
    assert float(ref.abs().max()) <= 2048
    assert torch.equal(c.float(), ref), f"max diff {(c.float()-ref).abs().max()}"


@pytest.mark.parametrize("layout,M,N,K", [("kk", 16384, 768, 768), ("kk", 20000, 2304, 768), ("km", 16384, 768, 2304),
                                          ("mm", 768, 768, 12800), ("mm", 3072, 768, 25000)])
def test_gemm_big_tile_exact_integers(layout, M, N, K):
    """Shapes that select the 256x256 / 8-wave configuration (and split-K for the weight-gradient layout)."""
    if layout == "kk":
        a, b = ints(M, K, lo=-1, hi=2), ints(N, K, lo=-1, hi=2, seed=1)
        c = ops.gemm_f16(a, b, M, N, K)
        ref = a.float() @ b.float().t()
    elif layout == "km":
        a, b = ints(M, K, lo=-1, hi=2), ints(K, N, lo=-1, hi=2, seed=1)
        c = ops.gemm_f16(a, b, M, N, K, a_kmajor=True, b_kmajor=False)
        ref = a.float() @ b.float()
    else:
        a, b = ints(K, M, lo=-1, hi=2), ints(K, N, lo=-1, hi=2, seed=1)
        c = ops.gemm_f16(a, b, M, N, K, a_kmajor=False, b_kmajor=False)
        ref = a.float().t() @ b.float()
    assert float(ref.abs().max()) <= 2048
    assert torch.equal(c.float(), ref), f"max diff {(c.float()-ref).abs().max()}"


@pytest.mark.parametrize("T,D", [(2048, 256), (4480, 512), (19200, 768), (40000, 768), (2056, 256), (5000, 512), (12344, 768), (2120, 768),
                                 (512, 512), (1024, 512), (1000, 512), (520, 256), (1440, 512)])
def test_gemm_grouped_weight_gradients_exact_integers(T, D):
    """The four weight gradients of a layer in one grouped launch (hmmc_gemm_f16_wgrad_group): bit-exact on integer data
    against fp32 torch for every problem, and the same results as one hmmc_gemm_f16 call per gradient.  Token counts: multiples
    of the 64-token K tile (2048 ... 40000), counts that are not (2056, 5000, 12344: the K tail of the grouped kernel; T % 8 == 0
    is the operand alignment), and 2120 tokens = 34 K-tiles, whose last split holds fewer K-tiles than the others.  Round 5: the
    launch takes 512 tokens and more (the text tower at 32 captions per GPU has 1 024; 1 440 = 32 titles of 45): 512 / 1024 / 1000 /
    520 / 1440 tokens - 8 to 23 K-tiles, one or two K-tiles per split at the short end."""
    g = torch.Generator().manual_seed(T + D)
    dims = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]                      # c_proj, c_fc, out_proj, in_proj
    dys = [torch.randint(-2, 3, (T, n), generator=g).half().to(DEV) for n, _ in dims]
    xs = [torch.randint(-2, 3, (T, k), generator=g).half().to(DEV) for _, k in dims]
    outs = ops.wgrad_group(dys, xs)
    assert outs is not None
    for dy, x, o in zip(dys, xs, outs):
        ref = dy.float().t() @ x.float()
        assert torch.equal(o.float(), ref.half().float()), float((o.float() - ref).abs().max())
        single = ops.gemm_f16(dy, x, dy.shape[1], x.shape[1], T, a_kmajor=False, b_kmajor=False)
        assert torch.equal(o, single)
    assert ops.wgrad_group([d[:504] for d in dys], [x[:504] for x in xs]) is None             # too few tokens: per-gradient calls
    small = ops.wgrad_group([torch.zeros(4096, 128, dtype=torch.float16, device=DEV)], [torch.zeros(4096, 384, dtype=torch.float16, device=DEV)])
    assert small is None                                                                       # not 256-tile shaped


@pytest.mark.parametrize("layout", ["kk", "km", "mm"])
def test_gemm_operands_beyond_2gib(layout):
    """An operand of 2 GiB or more (SURVEY config 5 on one GPU: 605 184 tokens x 3072 fp16 = 3.7 GB) exceeds what one launch's
    32-bit buffer offsets reach: the entry point cuts it into pieces (rows of a k-major A, token ranges of the weight-gradient
    operands whose split-K slabs meet in one reduce) and the result is still exact."""
    T, W = 360448, 3072                                  # 2.2 GB per [T, W] fp16 operand
    g = torch.Generator(device=DEV).manual_seed(7)
    big = torch.randint(-1, 2, (T, W), device=DEV, dtype=torch.int8, generator=g).half()
    if layout == "kk":                                   # forward: y[T, 256] = x[T, W] w[256, W]^T + bias, + residual
        w, bias = ints(256, W, lo=-1, hi=2, seed=1), ints(256, lo=-2, hi=3, seed=2)
        res = torch.randint(-2, 3, (T, 256), device=DEV, dtype=torch.int8, generator=g).half()
        c = ops.gemm_f16(big, w, T, 256, W, bias=bias, resid=res)
        for r0 in (0, T // 2 - 1000, T - 4096):
            ref = big[r0:r0 + 4096].float() @ w.float().t() + bias.float() + res[r0:r0 + 4096].float()
            assert torch.equal(c[r0:r0 + 4096].float(), ref), r0
    elif layout == "km":                                 # data gradient: dx[T, 256] = dy[T, W] w[W, 256]
        w = ints(W, 256, lo=-1, hi=2, seed=1)
        c = ops.gemm_f16(big, w, T, 256, W, a_kmajor=True, b_kmajor=False)
        for r0 in (0, T // 2 - 1000, T - 4096):
            assert torch.equal(c[r0:r0 + 4096].float(), big[r0:r0 + 4096].float() @ w.float()), r0
    else:                                                # weight gradient: dW[W, 256] = dy[T, W]^T x[T, 256], tokens = K
        x = torch.randint(0, 2, (T, 256), device=DEV, dtype=torch.int8, generator=g).half()
        big01 = (big != 0).half() * (torch.arange(T, device=DEV) % 7 == 0).half()[:, None]     # sums stay below 2048 * 32
        c = ops.gemm_f16(big01, x, W, 256, T, a_kmajor=False, b_kmajor=False)
        ref = torch.zeros(W, 256, device=DEV)
        for r0 in range(0, T, 65536):
            ref += big01[r0:r0 + 65536].float().t() @ x[r0:r0 + 65536].float()
        assert float(ref.max()) < 60000
        assert relerr(c, ref) < 1e-3 and torch.equal(c.float(), ref.half().float())


@pytest.mark.parametrize("nseq,L,N,K", [(96, 50, 768, 768), (37, 10, 128, 512)])
def test_gemm_strided_rows(nseq, L, N, K):
    """Row strides larger than the row length on A, C and the residual (the class-token rows of a [tokens, D] buffer,
    hmmc_tower_fwd's lead_only path): only the addressed rows are read and written."""
    from hmmc_amd._lib import call, ptr
    full_a = ints(nseq * L, K, lo=-1, hi=2)
    w, bias = ints(N, K, lo=-1, hi=2, seed=1), ints(N, lo=-2, hi=3, seed=2)
    full_r = ints(nseq * L, N, lo=-2, hi=3, seed=3)
    out = torch.full((nseq * L, N), 7.0, dtype=torch.float16, device=DEV)
    call("hmmc_gemm_f16", ptr(full_a), ptr(w), ptr(out), nseq, N, K, L * K, K, L * N, 1, 1, ptr(bias), ptr(full_r), None, None,
         ops.EPI_BIAS | ops.EPI_RESID, None, 0)
    a_rows, r_rows = full_a.view(nseq, L, K)[:, 0], full_r.view(nseq, L, N)[:, 0]
    ref = a_rows.float() @ w.float().t() + bias.float() + r_rows.float()
    o3 = out.view(nseq, L, N)
    assert torch.equal(o3[:, 0].float(), ref)
    assert bool((o3[:, 1:] == 7.0).all()), "rows between the addressed ones were written"
    # weight-gradient layout with strided operands: dW[N, K] = sum over the addressed rows of dy^T x
    dw = torch.empty(N, K, dtype=torch.float16, device=DEV)
    call("hmmc_gemm_f16", ptr(full_r), ptr(full_a), ptr(dw), N, K, nseq, L * N, L * K, K, 0, 0, None, None, None, None, 0, None, 0)
    assert torch.equal(dw.float(), r_rows.float().t() @ a_rows.float())


def test_gemm_with_reserved_cus():
    """hmmc_gemm_reserve_cus only shrinks the persistent grid: results are unchanged, bad counts are refused."""
    from hmmc_amd import _lib
    lib = _lib.load()
    assert lib.hmmc_gemm_reserve_cus(-1) == -1 and lib.hmmc_gemm_reserve_cus(129) == -1
    M, N, K = 20000, 2304, 768
    a, b = ints(M, K, lo=-1, hi=2), ints(N, K, lo=-1, hi=2, seed=1)
    ref = ops.gemm_f16(a, b, M, N, K)
    try:
        for cus in (16, 100):
            assert lib.hmmc_gemm_reserve_cus(cus) == 0
            assert torch.equal(ops.gemm_f16(a, b, M, N, K), ref)
            dy, x = ints(12800, 768, lo=-1, hi=2), ints(12800, 384, lo=-1, hi=2, seed=1)
            c = ops.gemm_f16(dy, x, 768, 384, 12800, a_kmajor=False, b_kmajor=False)
            assert torch.equal(c.float(), dy.float().t() @ x.float())
    finally:
        assert lib.hmmc_gemm_reserve_cus(0) == 0
    assert torch.equal(ref.float(), a.float() @ b.float().t())


def test_gemm_random_and_epilogues():
    M, N, K = 1000, 384, 256
    a, w = rnd(M, K, scale=0.5), rnd(N, K, scale=0.1, seed=1)
    bias, res = rnd(N, scale=0.1, seed=2), rnd(M, N, seed=3)
    acc = a.float() @ w.float().t()
    c = ops.gemm_f16(a, w, M, N, K, bias=bias)
    assert relerr(c, acc + bias.float()) < 2e-3
    c = ops.gemm_f16(a, w, M, N, K, bias=bias, resid=res, epilogue=ops.EPI_RESID)
    ref = (res.float() + (acc + bias.float()).half().float()).half()
    assert relerr(c, ref) < 1e-3 and float((c.float() - ref.float()).abs().max()) < 2e-2
    g, h = ops.gemm_f16(a, w, M, N, K, bias=bias, epilogue=ops.EPI_QGELU, want_aux=True)
    href = (acc + bias.float()).half()
    assert relerr(h, href) < 1e-3
    gref = h.float() * torch.sigmoid(1.702 * h.float())
    assert relerr(g, gref) < 2e-3
    dh = ops.gemm_f16(a, w, M, N, K, aux_in=h, epilogue=ops.EPI_DGELU)
    s = torch.sigmoid(1.702 * h.float())
    dref = acc * (s * (1 + 1.702 * h.float() * (1 - s)))
    assert relerr(dh, dref) < 2e-3
    # the towers' pairing: the forward saves QuickGELU'(h), the backward multiplies by it
    g2, gf = ops.gemm_f16(a, w, M, N, K, bias=bias, epilogue=ops.EPI_QGELU | ops.EPI_SAVE_DGELU, want_aux=True)
    assert torch.equal(g2, g)
    assert relerr(gf, s * (1 + 1.702 * h.float() * (1 - s))) < 1e-3
    dh2 = ops.gemm_f16(a, w, M, N, K, aux_in=gf, epilogue=ops.EPI_MULAUX)
    assert relerr(dh2, dref) < 2e-3


@pytest.mark.parametrize("b_kmajor", [True, False])
def test_gemm_partial_last_round(b_kmajor):
    """261 tiles of 256x256 on a 256-CU persistent grid (a last round of 5 tiles), ragged M, fused epilogues."""
    M, N, K = 87 * 256 - 100, 768, 128
    a = rnd(M, K, scale=0.5)
    w = rnd(N, K, scale=0.1, seed=1) if b_kmajor else rnd(K, N, scale=0.1, seed=1)
    bias, res = rnd(N, scale=0.1, seed=2), rnd(M, N, seed=3)
    acc = a.float() @ (w.float().t() if b_kmajor else w.float())
    c = ops.gemm_f16(a, w, M, N, K, b_kmajor=b_kmajor, bias=bias, resid=res, epilogue=ops.EPI_RESID)
    ref = (res.float() + (acc + bias.float()).half().float()).half()
    assert relerr(c, ref) < 1e-3 and float((c.float() - ref.float()).abs().max()) < 2e-2
    assert relerr(c[-600:], ref[-600:]) < 1e-3
    g, h = ops.gemm_f16(a, w, M, N, K, b_kmajor=b_kmajor, bias=bias, epilogue=ops.EPI_QGELU, want_aux=True)
    assert relerr(h, (acc + bias.float()).half()) < 1e-3 and relerr(h[-600:], (acc + bias.float()).half()[-600:]) < 1e-3
    assert relerr(g, h.float() * torch.sigmoid(1.702 * h.float())) < 2e-3


@pytest.mark.parametrize("M,N,K,b_kmajor", [(1000, 384, 256, True), (4096, 768, 256, False), (2048 + 77, 512, 128, False)])
def test_gemm_fused_column_sums(M, N, K, b_kmajor):
    """EPI_COLSUM: partial column sums of the fp16 values written (bias gradient without re-reading C)."""
    a = rnd(M, K, scale=0.5)
    w = rnd(N, K, scale=0.1, seed=1) if b_kmajor else rnd(K, N, scale=0.1, seed=1)
    h = rnd(M, N, seed=2)
    c, part = ops.gemm_f16(a, w, M, N, K, b_kmajor=b_kmajor, aux_in=h, epilogue=ops.EPI_DGELU, want_colsum=True)
    c2 = ops.gemm_f16(a, w, M, N, K, b_kmajor=b_kmajor, aux_in=h, epilogue=ops.EPI_DGELU)
    assert torch.equal(c, c2)
    ref = c.float().sum(0)
    got = part.sum(0)
    assert float((got - ref).abs().max()) < 1e-3 * float(ref.abs().max()) + 1e-3
    assert torch.equal(ops.colsum(part, out_dtype=torch.float16), ops.colsum(c)) or \
        float((ops.colsum(part, out_dtype=torch.float16).float() - ops.colsum(c).float()).abs().max()) < 2e-2


# ----------------------------------------------------------------------------- LayerNorm / rows

@pytest.mark.parametrize("dtype,D,eps", [(torch.float16, 768, 1e-5), (torch.float16, 128, 1e-5), (torch.float32, 512, 1e-12)])
def test_layernorm_fwd_bwd(dtype, D, eps):
    rows = 1003
    x = rnd(rows, D, dtype=dtype)
    gamma = rnd(D, dtype=torch.float32, seed=1) * 0.1 + 1
    beta = rnd(D, dtype=torch.float32, seed=2) * 0.1
    dy, dres = rnd(rows, D, dtype=dtype, seed=3), rnd(rows, D, dtype=dtype, seed=4)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, eps)
    xr = x.float().requires_grad_()
    gr, br = gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, br, eps)
    tol = 2e-3 if dtype == torch.float16 else 1e-5
    assert relerr(y, yr) < tol
    yr.backward(dy.float())
    dx, dg, db = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres)
    assert relerr(dx, xr.grad + dres.float()) < tol
    assert relerr(dg, gr.grad) < 1e-4 and relerr(db, br.grad) < 1e-4
    dx2, dg2, db2, cs = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, want_colsum=True)   # fused colsum(dx)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    assert cs.dtype == dtype and relerr(cs, dx.float().sum(0)) < (2e-3 if dtype == torch.float16 else 1e-5)


def test_layernorm_row_gather():
    n, L, D = 37, 50, 768
    x = rnd(n * L, D)
    gamma, beta = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    idx = (torch.arange(n, device=DEV, dtype=torch.int32) * L).contiguous()
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5, row_index=idx)
    ref = torch.nn.functional.layer_norm(x.view(n, L, D)[:, 0].float(), (D,))
    assert relerr(y, ref) < 2e-3
    dy = rnd(n, D, seed=5)
    dx = torch.zeros_like(x)
    dx, dg, db = ops.layernorm_bwd(dy, x, gamma, mean, rstd, row_index=idx, dx=dx)
    xr = x.view(n, L, D)[:, 0].float().requires_grad_()
    torch.nn.functional.layer_norm(xr, (D,)).backward(dy.float())
    assert relerr(dx.view(n, L, D)[:, 0], xr.grad) < 2e-3
    assert float(dx.view(n, L, D)[:, 1:].abs().max()) == 0.0


def test_colsum_patchify_embed():
    x = rnd(1234, 768)
    s = ops.colsum(x, out_dtype=torch.float32)
    assert relerr(s, x.float().sum(0)) < 1e-5
    xf = rnd(777, 512, dtype=torch.float32)
    assert relerr(ops.colsum(xf), xf.sum(0)) < 1e-5
    # patchify == unfold of the stride-p conv, (c, ky, kx) column order, zero class row
    for p in (32, 16):
        vid = rnd(3, 3, 224, 224, dtype=torch.float32)
        out = ops.patchify(vid, p)
        g = 224 // p
        ref = torch.nn.functional.unfold(vid, kernel_size=p, stride=p).transpose(1, 2)     # [n, g*g, 3*p*p]
        out = out.view(3, g * g + 1, -1)
        assert torch.equal(out[:, 1:], ref.half())
        assert float(out[:, 0].abs().max()) == 0.0
    # vit_embed
    L, D = 50, 768
    t = rnd(4 * L, D)
    t.view(4, L, D)[:, 0] = 0
    cls, pos = rnd(D, dtype=torch.float32, seed=1), rnd(L, D, dtype=torch.float32, seed=2)
    ref = t.clone().view(4, L, D)
    ref[:, 0] = cls.half()
    ref = (ref + pos.half()).view(4 * L, D)
    got = ops.vit_embed_(t.clone(), cls, pos, L)
    assert torch.equal(got, ref)
    # text embed fwd / bwd
    ids = torch.randint(0, 1000, (6, 32), device=DEV)
    table, tpos = rnd(1000, 512, dtype=torch.float32, scale=0.02), rnd(77, 512, dtype=torch.float32, scale=0.01, seed=3)
    xe = ops.text_embed(ids, table, tpos)
    ref = (table[ids].half() + tpos[:32].half()).view(-1, 512)
    assert torch.equal(xe, ref)
    dx = rnd(6 * 32, 512, seed=9)
    dt = ops.text_embed_bwd(ids, dx, 1000)
    ref = torch.zeros(1000, 512, device=DEV).index_add_(0, ids.view(-1), dx.float())
    assert relerr(dt, ref) < 1e-5
    # heavy hitters (the padding id fills most of a real batch), more occurrences than one list batch, bit-stable
    ids2 = torch.randint(0, 1000, (600, 32), device=DEV)
    ids2[:, 12:] = 0
    ids2[:, 0] = 999
    dx2 = rnd(600 * 32, 512, seed=11)
    dt2 = ops.text_embed_bwd(ids2, dx2, 1000)
    ref2 = torch.zeros(1000, 512, device=DEV, dtype=torch.float64).index_add_(0, ids2.view(-1), dx2.double())
    assert relerr(dt2, ref2.float()) < 1e-5
    assert torch.equal(dt2, ops.text_embed_bwd(ids2, dx2, 1000))
    # more rows than a 16-bit offset spans (a list batch ends where `t - start` would pass 65 535): 65 536 and 66 000 rows
    # truncated offsets in round 2's kernel; heavy hitters in the first and in the last rows
    for nrows in (65536, 66000, 140000):
        ids3 = torch.randint(0, 300, (nrows,), device=DEV)
        ids3[::7] = 0
        ids3[-5:] = 299
        ids3[:3] = 299
        dx3 = rnd(nrows, 128, seed=13)
        dt3 = ops.text_embed_bwd(ids3.view(-1, 1), dx3, 300)
        ref3 = torch.zeros(300, 128, device=DEV, dtype=torch.float64).index_add_(0, ids3, dx3.double())
        assert relerr(dt3, ref3.float()) < 1e-5, nrows
    # ids outside the table: no fault, zero row + position embedding, flag raised on the host check, no gradient
    bad = ids.clone()
    bad[0, 3], bad[2, 5] = 1000, -7
    xb = ops.text_embed(bad, table, tpos)
    good = bad.clamp(0, 999)
    refb = table[good].half()
    refb[0, 3], refb[2, 5] = 0, 0
    assert torch.equal(xb, (refb + tpos[:32].half()).view(-1, 512))
    with pytest.raises(IndexError):
        ops.raise_on_device_errors()
    ops.raise_on_device_errors()                     # the flag was cleared
    dtb = ops.text_embed_bwd(bad, dx, 1000)
    keep = ((bad >= 0) & (bad < 1000)).view(-1)
    refd = torch.zeros(1000, 512, device=DEV).index_add_(0, good.view(-1)[keep], dx.float()[keep])
    assert relerr(dtb, refd) < 1e-5


# ----------------------------------------------------------------------------- attention

def attn_ref(qkv, nseq, L, H, causal):
    D = H * 64
    q, k, v = qkv.float().view(nseq, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=qkv.device).triu_(1)
    p = torch.softmax(s, -1)
    o = (p @ v).permute(0, 2, 1, 3).reshape(nseq * L, D)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("nseq,L,H,causal", [(5, 50, 2, False), (3, 32, 8, True), (2, 45, 2, True), (7, 25, 2, True),
                                             (4, 64, 12, False), (3, 17, 2, False), (3, 197, 2, False), (2, 77, 8, True),
                                             (2, 130, 2, True), (1, 256, 1, False),
                                             (5, 197, 12, False)])      # ViT-B/16's real shape: 197 tokens x 12 heads
def test_attention_fwd_bwd(nseq, L, H, causal):
    D = H * 64
    qkv = rnd(nseq * L, 3 * D, scale=1.0)
    out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, causal)
    qr = qkv.float().requires_grad_()
    oref, lref = attn_ref(qr, nseq, L, H, causal)
    assert relerr(out, oref) < 3e-3, relerr(out, oref)
    assert float((lse - lref).abs().max()) < 2e-3
    dout = rnd(nseq * L, D, seed=11)
    oref.backward(dout.float())
    dqkv = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal)
    g = qr.grad.view(nseq * L, 3, D)
    d = dqkv.view(nseq * L, 3, D)
    for i, nm in enumerate("qkv"):
        e = relerr(d[:, i], g[:, i])
        assert e < 8e-3, f"d{nm} rel err {e}"
    # fused in_proj bias-gradient partials: per-sequence column sums of dqkv (short and long kernels)
    dqkv2, part = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, want_dbias=True)
    assert torch.equal(dqkv, dqkv2)
    ref = dqkv.float().view(nseq, L, 3 * D).sum(1)
    assert float((part - ref).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-4


@pytest.mark.parametrize("nseq,L,H,causal", [(5, 50, 2, False), (3, 32, 8, False), (7, 25, 2, True), (4, 64, 12, False), (3, 17, 2, False),
                                             (96, 50, 12, False), (5, 197, 12, False), (3, 77, 8, True), (2, 130, 2, False), (1, 256, 1, False),
                                             (300, 197, 12, False),    # more heads than the persistent grid has workgroups
                                             (3072, 50, 12, False)])   # BASELINE config 2's frame tower: 256 videos x 12 frames
def test_attention_query0_only_equals_the_all_query_kernels(nseq, L, H, causal):
    """hmmc_attention_f16_fwd_lead / _bwd_lead (the last block of a tower read at its class token, modules/module_cross.py:228-230):
    row n*L of the output and its log-sum-exp must be BIT-identical to the all-query kernel's, the backward (output gradient at
    query 0 only, the other rows of the buffer poisoned) must give the all-query kernel's dK, dV, dQ[token 0] and bias partials
    bit for bit, and nothing but the announced rows may be written or read: the Q columns of the other tokens are NaN on the
    way in and untouched on the way out."""
    D = H * 64
    qkv = rnd(nseq * L, 3 * D, scale=1.0)
    out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, causal)
    lead = torch.arange(nseq, device=DEV) * L
    qkv_p = qkv.clone().view(nseq, L, 3 * D)
    qkv_p[:, 1:, :D] = float("nan")                                          # only token 0's Q may be read
    qkv_p = qkv_p.view(nseq * L, 3 * D)
    out1, lse1 = ops.attention_f16_fwd_lead(qkv_p, nseq, L, H, causal)
    assert torch.equal(out1[lead], out[lead]), "class-token rows of the output"
    assert torch.equal(lse1[:, :, 0], lse[:, :, 0]), "their log-sum-exp"
    rest = torch.ones(nseq * L, dtype=torch.bool, device=DEV); rest[lead] = False
    assert bool(torch.isnan(out1[rest]).all()) and bool(torch.isnan(lse1[:, :, 1:]).all()), "rows that must not be written"
    # backward: the reference is the all-query kernel on an output gradient that is zero off the class rows
    dout = torch.zeros(nseq * L, D, dtype=torch.float16, device=DEV)
    dout[lead] = rnd(nseq, D, seed=11)
    dref, pref = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, want_dbias=True)
    dout_p = torch.full_like(dout, float("nan")); dout_p[lead] = dout[lead]
    lse_p = torch.full_like(lse, float("nan")); lse_p[:, :, 0] = lse[:, :, 0]
    out_p = torch.full_like(out, float("nan")); out_p[lead] = out[lead]      # the long-sequence kernel reads O at the class rows
    d1, p1 = ops.attention_f16_bwd_lead(qkv_p, lse_p, dout_p, nseq, L, H, causal, want_dbias=True, out=out_p)
    assert torch.equal(d1[:, D:], dref[:, D:]), "dK, dV of every token"
    assert torch.equal(d1[lead, :D], dref[lead, :D]), "dQ of the class tokens"
    assert bool(torch.isnan(d1[rest, :D]).all()), "dQ of the other tokens must stay untouched"
    assert float(dref[rest, :D].abs().max()) == 0.0, "(and is exactly zero in the all-query computation)"
    assert torch.equal(p1, pref), "in_proj bias partials"
    # scaled variant (folded ln_1): rows leave multiplied by their factor
    stat = torch.rand(nseq * L, 2, device=DEV) + 0.5
    dsc = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, rowstat=stat)
    d2 = ops.attention_f16_bwd_lead(qkv_p, lse_p, dout_p, nseq, L, H, causal, rowstat=stat, out=out_p)
    assert torch.equal(d2[:, D:], dsc[:, D:]) and torch.equal(d2[lead, :D], dsc[lead, :D]), "scaled rows"


def test_multi_colreduce_many_tasks_ragged_and_unaligned():
    """hmmc_multi_colreduce (round 5: 128 columns per block, 16-byte loads where the rows allow): several tasks in one launch -
    widths that are / are not multiples of 4 and of 128, a partial matrix at an address that is not 16-byte aligned, one to three
    output segments in fp16 / fp32, a missing (NULL) segment, row counts from 1 to 700 - against float64 column sums."""
    import ctypes
    from hmmc_amd import _lib

    class Task(ctypes.Structure):
        _fields_ = [("partial", ctypes.c_void_p), ("R", ctypes.c_int), ("N", ctypes.c_int), ("seg", ctypes.c_int),
                    ("out", ctypes.c_void_p * 3), ("dtype", ctypes.c_int * 3)]
    g = torch.Generator(device=DEV).manual_seed(17)
    specs = [(600, 2304, 768, (1, 1, 0)), (150, 3072, 3072, (0,)), (384, 2304, 2304, (0,)), (1, 130, 130, (1,)), (7, 1026, 342, (1, None, 0)),
             (700, 512, 256, (0, 1)), (33, 96, 96, (1,)), (257, 100, 100, (0,))]
    keep, tasks, checks = [], (Task * len(specs))(), []
    for i, (R, N, seg, outs) in enumerate(specs):
        buf = torch.randn(R * N + 1, device=DEV, generator=g)
        part = buf[1:] if i % 3 == 2 else buf[:-1]                 # every third partial matrix starts 4 bytes off a 16-byte boundary
        keep.append(buf)
        tasks[i].partial, tasks[i].R, tasks[i].N, tasks[i].seg = part.data_ptr(), R, N, seg
        ref = part.view(R, N).double().sum(0)
        for sgm, dt in enumerate(outs):
            if dt is None:
                tasks[i].out[sgm], tasks[i].dtype[sgm] = None, 1
                continue
            o = torch.full((seg,), float("nan"), dtype=torch.float16 if dt == 0 else torch.float32, device=DEV)
            keep.append(o)
            tasks[i].out[sgm], tasks[i].dtype[sgm] = o.data_ptr(), dt
            checks.append((o, ref[sgm * seg:(sgm + 1) * seg], dt, (R, N, seg, sgm)))
    _lib.call("hmmc_multi_colreduce", ctypes.cast(tasks, ctypes.c_void_p), len(specs))
    torch.cuda.synchronize()
    for o, ref, dt, what in checks:
        tol = 2e-3 if dt == 0 else 1e-5
        err = float((o.double() - ref).abs().max() / (ref.abs().max() + 1e-9))
        assert err < tol, (what, err)


def test_retrieval_rank_ties_and_targets():
    """metrics.py:20-28: the rank is the first position of the target's score in the descending sort."""
    torch.manual_seed(3)
    S = torch.randint(-4, 5, (37, 53), device=DEV).float()            # many exact ties
    tgt = torch.randint(0, 53, (37,), device=DEV, dtype=torch.int32)
    got = ops.retrieval_rank(S, target=tgt).cpu()
    ref = (S > S.gather(1, tgt.long()[:, None])).sum(1).cpu()
    assert torch.equal(got.long(), ref)
    sq = S[:, :37].contiguous()
    sx = np.sort(-sq.cpu().numpy(), axis=1)
    d = np.diag(-sq.cpu().numpy())[:, None]
    first = np.array([np.where(sx[i] == d[i])[0][0] for i in range(37)])
    assert np.array_equal(ops.retrieval_rank(sq).cpu().numpy(), first)
    assert np.array_equal(ops.retrieval_rank(sq, transposed=True).cpu().numpy(), ops.retrieval_rank(sq.t().contiguous()).cpu().numpy())


def test_patchify_u8_matches_loader_normalisation():
    """uint8 frames + fused ToTensor / Normalize (dataloader_msrvtt_retrieval.py:242-247) == patchify of the fp32 frames."""
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (5, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean = torch.tensor(ops.CLIP_PIXEL_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ops.CLIP_PIXEL_STD).view(1, 3, 1, 1)
    f32 = (u8.float().div(255.0) - mean) / std
    for patch in (32, 16):
        a = ops.patchify_u8(u8.to(DEV), patch)
        b = ops.patchify(f32.to(DEV).contiguous(), patch)
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())


def test_error_statuses_instead_of_faults():
    """Shapes the kernels cannot take come back as HMMC_ERR_* (RuntimeError in the host layer), never as a launch."""
    from hmmc_amd._lib import call, ptr, query
    a = rnd(64, 72)                                    # K = 72: not a multiple of the 64-deep K-tile of a k-major operand
    w = rnd(64, 72, seed=1)
    with pytest.raises(RuntimeError, match="unsupported"):
        ops.gemm_f16(a, w, 64, 64, 72)
    with pytest.raises(RuntimeError, match="invalid argument"):
        call("hmmc_gemm_f16", ptr(a), None, ptr(a), 64, 64, 64, 72, 72, 64, 1, 1, None, None, None, None, 0, None, 0)
    qkv = rnd(300 * 2, 3 * 64)
    with pytest.raises(RuntimeError, match="unsupported"):          # sequences longer than 256 tokens do not occur on the path
        ops.attention_f16_fwd(qkv, 2, 300, 1, False)
    x = rnd(128, 768)
    g = torch.ones(768, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, g, torch.zeros(768, device=DEV), 1e-5)
    dx = torch.empty_like(x)
    dg, db = torch.empty(768, device=DEV), torch.empty(768, device=DEV)
    small = torch.empty(16, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError, match="workspace"):
        call("hmmc_layernorm_bwd", ptr(x), ptr(x), ptr(g), ptr(mean), ptr(rstd), None, ptr(dx), ptr(dg), ptr(db), None, None,
             128, 768, 768, 0, ptr(small), 16)
    with pytest.raises(TypeError):
        ops.gemm_f16(a.cpu(), w, 64, 64, 72)                        # host tensors are rejected: there is no CPU path
    # lead_only is an fp16-tower mode
    from hmmc_amd import functional as Fn
    xf = torch.randn(4 * 6, 64, device=DEV)
    prm = [torch.randn(s, device=DEV) for s in ((64,), (64,), (192, 64), (192,), (64, 64), (64,), (64,), (64,), (256, 64), (256,),
                                                 (64, 256), (64,))]
    with pytest.raises(RuntimeError, match="unsupported"):
        Fn._tower_forward(xf, prm, 4, 6, 1, False, 1e-12, True, False, lead_only=True)
