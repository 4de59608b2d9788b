"""LayerNorm folded into the GEMM behind it (hmmc_gemm_f16_fold / hmmc_tower_fwd_fused: the fp16 towers' forward when no
activations are kept; reference modules/module_clip.py:217-223,252-256).  The folded form rounds gamma o W to fp16 where the
reference rounds LN(x): its contract is the fp32 value of the same expression and the reference's own fp16 envelope
(tests/test_gpu_model.py::test_envelope_at_true_vit_b32_dims and test_retrieval_ranks_b32_vs_reference run on this path),
not torch's rounding sequence."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from hmmc_amd import ops  # noqa: E402
from hmmc_amd import functional as Fn  # noqa: E402

DEV = "cuda"


def relerr(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (2400, 2304, 768), (1000, 512, 512), (300, 192, 128)])
def test_fold_prep_and_rowstat(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M + N)
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    gm = 1.0 + 0.2 * torch.randn(K, device=DEV, generator=g)
    bt = 0.1 * torch.randn(K, device=DEV, generator=g)
    b = (0.1 * torch.randn(N, device=DEV, generator=g)).half()
    (Wf, cd), (Wf2, cd2) = ops.ln_fold_prep([(W, gm, bt, b), (W, gm, bt, None)])
    ref_wf = (gm[None, :] * W.float()).half()
    assert torch.equal(Wf, ref_wf) and torch.equal(Wf2, ref_wf)
    torch.testing.assert_close(cd[0], ref_wf.float().sum(1), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(cd[1], (W.float() * bt[None, :]).sum(1) + b.float(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cd2[1], (W.float() * bt[None, :]).sum(1), rtol=1e-5, atol=1e-5)
    x = (torch.randn(M, K, device=DEV, generator=g) * 1.5 + 0.3).half()
    st = ops.rowstat(x)
    mean = x.float().mean(1)
    rstd = torch.rsqrt(x.float().var(1, unbiased=False) + 1e-5)
    torch.testing.assert_close(st[:, 0], rstd, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st[:, 1], -rstd * mean, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K,gelu", [(4096, 2304, 768, False), (4096, 3072, 768, True), (2400, 768, 768, False),
                                        (1000, 1536, 512, False), (300, 384, 128, True), (130, 192, 64, False)])
def test_folded_gemm_equals_layernorm_then_gemm(M, N, K, gelu):
    """rstd_r (x W'^T) - rstd_r mean_r c + d against the same expression in fp64 from the same fp16 operands (one fp16 rounding
    of the result: 1e-3), and against LayerNorm -> fp16 -> linear as the reference evaluates it (two more roundings)."""
    g = torch.Generator(device=DEV).manual_seed(N + K)
    x = (torch.randn(M, K, device=DEV, generator=g) * 1.3 + 0.4).half()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.04).half()
    gm = 1.0 + 0.2 * torch.randn(K, device=DEV, generator=g)
    bt = 0.1 * torch.randn(K, device=DEV, generator=g)
    b = (0.1 * torch.randn(N, device=DEV, generator=g)).half()
    (Wf, cd), = ops.ln_fold_prep([(W, gm, bt, b)])
    st = ops.rowstat(x)
    y = ops.gemm_f16_fold(x, Wf, rowstat=st, colterms=cd, epilogue=ops.EPI_QGELU if gelu else 0)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + 1e-5)
    exact = rstd * (xd @ Wf.double().t()) - rstd * mean * Wf.double().sum(1)[None, :] + (W.double() * bt.double()[None, :]).sum(1)[None, :] + b.double()[None, :]
    ln = (((xd - mean) * rstd) * gm.double() + bt.double()).half()
    ref = (ln.double() @ W.double().t() + b.double()).half().double()
    if gelu:
        exact = exact * torch.sigmoid(1.702 * exact)
        ref = ref * torch.sigmoid(1.702 * ref)
    err = (y.double() - exact).abs()
    lim = 2.5e-3 * exact.abs() + 2e-3        # fp16 rounding of h, of sigmoid and of the product (QuickGELU) or of the sum
    assert (err <= lim).all(), f"max err {float(err.max()):.3e}, worst ratio {float((err / lim).max()):.2f}"
    assert relerr(y, ref) < 2e-3, relerr(y, ref)


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (4096, 768, 3072), (2400, 512, 2048), (1000, 128, 512), (130, 192, 64)])
def test_rowstat_from_the_gemm_epilogue(M, N, K):
    """HMMC_EPI_ROWSTAT: the statistics of the fp16 rows the GEMM wrote, bit-for-bit the values a pass over them would see."""
    g = torch.Generator(device=DEV).manual_seed(M + K)
    a = torch.randn(M, K, device=DEV, generator=g).half()
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
    b = (0.1 * torch.randn(N, device=DEV, generator=g)).half()
    r = (torch.randn(M, N, device=DEV, generator=g) * 1.5 + 0.5).half()
    y, part = ops.gemm_f16_fold(a, W, bias=b, resid=r, want_stat=True)
    y0 = ops.gemm_f16(a, W, M, N, K, bias=b, resid=r)
    assert torch.equal(y, y0)
    yb = y.float().view(M, N // 64, 64)
    torch.testing.assert_close(part[:, :, 0].t(), yb.sum(2), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(part[:, :, 1].t(), (yb * yb).sum(2), rtol=1e-5, atol=1e-4)
    st = ops.rowstat_finalize(part, N)
    st0 = ops.rowstat(y)
    torch.testing.assert_close(st, st0, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("width,heads,L,nseq,layers,causal,lead", [(128, 2, 10, 37, 3, False, False), (768, 12, 50, 96, 3, False, True),
                                                                    (768, 12, 50, 64, 2, False, False), (512, 8, 32, 64, 3, True, False)])
def test_fused_tower_forward_against_the_unfolded_tower(width, heads, L, nseq, layers, causal, lead):
    from hmmc_amd import module_clip
    torch.manual_seed(5)
    tw = module_clip.Transformer(width, layers, heads, attn_mask="causal" if causal else None)
    for prm in tw.parameters():
        torch.nn.init.normal_(prm, std=0.04 if prm.dim() > 1 else 0.1)
    for blk in tw.resblocks:
        blk.ln_1.weight.data.add_(1.0)
        blk.ln_2.weight.data.add_(1.0)
    module_clip.convert_weights(tw)
    tw = tw.to(DEV)
    x0 = (torch.randn(nseq * L, width) * 0.7 + 0.1).half().to(DEV)
    outs = {}
    for fold in (False, True):
        tw.fold_ln = fold
        with torch.no_grad():
            y = tw(x0, nseq, L, lead_only=lead)
        outs[fold] = (y.view(nseq, L, width)[:, 0, :] if lead else y).float()
    # fp32 evaluation of the same blocks from the same fp16 weights
    with torch.no_grad():
        h = x0.float().view(nseq, L, width)
        mask = torch.full((L, L), float("-inf"), device=DEV).triu_(1) if causal else None
        for blk in tw.resblocks:
            p = [q.float() for q in Fn.block_params(blk)]
            a = torch.nn.functional.layer_norm(h, (width,), p[0], p[1], 1e-5)
            qkv = a @ p[2].t() + p[3]
            q, k, v = [t.view(nseq, L, heads, 64).transpose(1, 2) for t in qkv.chunk(3, -1)]
            s = (q @ k.transpose(-1, -2)) / 8.0
            if mask is not None:
                s = s + mask
            o = (s.softmax(-1) @ v).transpose(1, 2).reshape(nseq, L, width)
            h = h + o @ p[4].t() + p[5]
            a = torch.nn.functional.layer_norm(h, (width,), p[6], p[7], 1e-5)
            f = a @ p[8].t() + p[9]
            h = h + (f * torch.sigmoid(1.702 * f)) @ p[10].t() + p[11]
        ref = h[:, 0, :] if lead else h.reshape(nseq * L, width)
    e_exact, e_fold = relerr(outs[False], ref), relerr(outs[True], ref)
    print(f"rel-L2 against fp32: unfolded kernels {e_exact:.3e}, folded {e_fold:.3e}; folded vs unfolded {relerr(outs[True], outs[False]):.3e}")
    assert e_fold <= 1.5 * e_exact + 1e-4, (e_fold, e_exact)
    assert torch.isfinite(outs[True]).all()
    assert not torch.equal(outs[True], outs[False]), "the folded kernels did not run"


def test_both_towers_folded_against_both_regimes_of_the_reference(monkeypatch):
    """HMMC_FOLD_LN=all at true ViT-B/32 dims (tests/golden/enc_b32x8_*.npz).  The folded kernels round gamma o W where the
    reference rounds LN(x), so their fp16 errors are independent of the as-written run's: what is asserted is that they are as
    close to the reference's fp32 regime as the reference's own fp16 regime is (rel-L2 <= 1.25 x, max <= 1.5 x), and within
    1.5 x the regime gap of the as-written run in rel-L2.  (The element-wise maximum against the as-written run is 1.85 x the
    regime gap for text_feat: why the default policy folds the frame tower only.)"""
    import numpy as np
    from conftest import golden
    from test_gpu_model import build
    from hmmc_amd import synth
    monkeypatch.setattr(Fn, "_FOLD_LN", "all")
    ga, gf = golden("enc_b32x8_aswritten"), golden("enc_b32x8_fp32")
    B, Fr, L, k = int(ga["B"]), int(ga["F"]), int(ga["L"]), int(ga["k"])
    model, sd = build(synth.VIT_B32, max_frames=Fr, top_frames=k)
    ids, mask, vid, vf, idx = [t.to(DEV) for t in synth.finetune_batch(B, Fr, L, synth.VIT_B32.image_res, tag="enc_b32x8")]
    with torch.no_grad():
        q = model.text_encoder(ids, mask)
        v, u = model.visual_encoder(vid, vf)
        sv, fk = model.eval_scores(q, v, u, top_frames=k)
    for key, mine in {"text_feat": q, "video_emb": v, "frame_output": u, "S_video": sv, "S_frame_topk": fk}.items():
        mine = mine.cpu().numpy()
        nrm = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        own_max, own_l2 = float(np.abs(ga[key] - gf[key]).max()), nrm(ga[key], gf[key])
        print(f"{key}: vs fp32 regime max {np.abs(mine - gf[key]).max():.3e} rel-L2 {nrm(mine, gf[key]):.3e}; vs as-written max "
              f"{np.abs(mine - ga[key]).max():.3e} rel-L2 {nrm(mine, ga[key]):.3e}; the reference's regimes: {own_max:.3e} / {own_l2:.3e}")
        assert nrm(mine, gf[key]) <= 1.25 * own_l2 and float(np.abs(mine - gf[key]).max()) <= 1.5 * own_max, key
        assert nrm(mine, ga[key]) <= 1.5 * own_l2, key


@pytest.mark.parametrize("nframes,L,D", [(37, 50, 768), (5, 197, 768), (9, 10, 128),
                                         (700, 50, 768), (200, 197, 768)])    # enough rows for the one-position-per-wave grid (round 5)
def test_vit_embed_ln_is_bit_identical_to_its_two_stages(nframes, L, D):
    """hmmc_vit_embed_ln = hmmc_vit_embed + hmmc_layernorm_fwd (modules/module_clip.py:311-313) in one pass: same bits, plus
    the row pairs of its output."""
    g = torch.Generator(device=DEV).manual_seed(D + L)
    x0 = torch.randn(nframes * L, D, device=DEV, generator=g).half()
    x0.view(nframes, L, D)[:, 0, :] = 0                       # class rows of the patch GEMM's output are zero
    cls = torch.randn(D, device=DEV, generator=g) * 0.04
    pos = torch.randn(L, D, device=DEV, generator=g) * 0.04
    gm = 1.0 + 0.2 * torch.randn(D, device=DEV, generator=g)
    bt = 0.1 * torch.randn(D, device=DEV, generator=g)
    a = x0.clone()
    ops.vit_embed_(a, cls, pos, L)
    ya, ma, ra = ops.layernorm_fwd(a, gm, bt, 1e-5)
    b = x0.clone()
    yb, mb, rb, st = ops.vit_embed_ln_(b, cls, pos, gm, bt, L, want_stat=True, write_x0=True)
    assert torch.equal(a, b) and torch.equal(ya, yb) and torch.equal(ma, mb) and torch.equal(ra, rb)
    torch.testing.assert_close(st, ops.rowstat(yb), rtol=1e-5, atol=1e-6)
    c = x0.clone()
    yc, _, _, none = ops.vit_embed_ln_(c, cls, pos, gm, bt, L, want_stat=False, write_x0=False)
    assert none is None and torch.equal(c, x0) and torch.equal(yc, ya)


# ----------------------------------------------------------------------------- the fold through the backward pass (training)

def test_rowscaled_dgrad_and_its_bias_partials():
    """HMMC_EPI_MULAUX | COLSUM | ROWSCALE: rstd_r x [(dy W) o aux] stored, column sums of the UNSCALED fp16-rounded product."""
    g = torch.Generator(device=DEV).manual_seed(11)
    M, Np, Kp = 4096, 768, 3072
    dy = (torch.randn(M, Np, device=DEV, generator=g) * 0.1).half()
    w = (torch.randn(Np, Kp, device=DEV, generator=g) * 0.05).half()
    aux = torch.rand(M, Kp, device=DEV, generator=g).half()
    x = (torch.randn(M, 64, device=DEV, generator=g) * 2.0 + 0.5).half()
    st = ops.rowstat(x)
    out, part = ops.gemm_f16_rowscaled_dgrad(dy, w, aux, st)
    base = (dy.double() @ w.double()) * aux.double()
    ref = base * st[:, :1].double()
    err = (out.double() - ref).abs()
    assert (err <= 1.5e-3 * ref.abs() + 1e-4).all(), float((err / (1.5e-3 * ref.abs() + 1e-4)).max())
    torch.testing.assert_close(part.sum(0).double(), base.half().double().sum(0), rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("rows,D", [(4096, 768), (1000, 512), (300, 256)])
def test_layernorm_backward_folded(rows, D):
    """dx of hmmc_layernorm_bwd_fold against autograd through (x - mean) rstd with the upstream gradient pre-scaled by rstd."""
    g = torch.Generator(device=DEV).manual_seed(rows)
    x = (torch.randn(rows, D, device=DEV, generator=g) * 1.7 + 0.3).half()
    du = (torch.randn(rows, D, device=DEV, generator=g) * 0.05).half()          # gradient w.r.t. u = (x - mean) rstd
    dres = (torch.randn(rows, D, device=DEV, generator=g) * 0.05).half()
    st = ops.rowstat(x)
    dut = (du.float() * st[:, :1]).half()
    dx, csum = ops.layernorm_bwd_fold(dut, x, st, dres=dres, want_colsum=True)
    xd = x.double().requires_grad_()
    u = (xd - xd.mean(1, keepdim=True)) * torch.rsqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)
    (u * (dut.double() / st[:, :1].double())).sum().backward()
    ref = xd.grad + dres.double()
    assert relerr(dx, ref) < 1e-3, relerr(dx, ref)
    torch.testing.assert_close(csum.double(), dx.double().sum(0), rtol=1e-4, atol=1e-3)
    dx2 = ops.layernorm_bwd_fold(dut, x, st)
    assert relerr(dx2, xd.grad) < 1e-3


@pytest.mark.parametrize("nseq,L,H", [(96, 50, 12), (24, 197, 12), (40, 77, 8)])
def test_scaled_attention_backward(nseq, L, H):
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(nseq * L, 3 * 64 * H, device=DEV, generator=g).half()
    out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, False)
    dout = (torch.randn(nseq * L, 64 * H, device=DEV, generator=g) * 0.1).half()
    st = torch.rand(nseq * L, 2, device=DEV, generator=g) + 0.5
    d0, p0 = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, False, want_dbias=True)
    d1, p1 = ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, False, want_dbias=True, rowstat=st)
    assert torch.equal(p0, p1)                                                    # bias partials: the unscaled gradient's
    ref = d0.float() * st[:, :1]
    err = (d1.float() - ref).abs()
    assert (err <= 2e-3 * ref.abs() + 1e-6).all()                                 # one more fp16 rounding of the same values


def test_fold_grad_finish_against_the_definitions():
    g = torch.Generator(device=DEV).manual_seed(9)
    T, N, K = 3000, 2304, 768
    x = (torch.randn(T, K, device=DEV, generator=g) * 1.5 + 0.4).half()
    dyt = (torch.randn(T, N, device=DEV, generator=g) * 0.05).half()              # rstd_r x dy
    W = (torch.randn(N, K, device=DEV, generator=g) * 0.04).half()
    gm = 1.0 + 0.2 * torch.randn(K, device=DEV, generator=g)
    bt = 0.1 * torch.randn(K, device=DEV, generator=g)
    db = (torch.randn(N, device=DEV, generator=g) * 0.5).half()
    S = (dyt.double().t() @ x.double()).float()
    (dW, dg, dbeta), = ops.fold_grad_finish([(S, W, gm, bt, db)])
    xc = x.double() - x.double().mean(1, keepdim=True)
    G = dyt.double().t() @ xc
    ref_dW = gm.double()[None, :] * G + bt.double()[None, :] * db.double()[:, None]
    assert relerr(dW, ref_dW) < 1e-3, relerr(dW, ref_dW)
    torch.testing.assert_close(dg.double(), (W.double() * G).sum(0), rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(dbeta.double(), (W.double() * db.double()[:, None]).sum(0), rtol=1e-4, atol=1e-4)


def _tower_and_reference(width, heads, L, nseq, layers, causal, outlier=None, gscale=0.1):
    """outlier = (every n-th row, factor): those rows of the input are multiplied by `factor` (large-variance tokens: rstd << 1);
    gscale: size of the upstream gradient."""
    from hmmc_amd import module_clip
    torch.manual_seed(7)
    tw = module_clip.Transformer(width, layers, heads, attn_mask="causal" if causal else None)
    for prm in tw.parameters():
        torch.nn.init.normal_(prm, std=0.04 if prm.dim() > 1 else 0.1)
    for blk in tw.resblocks:
        blk.ln_1.weight.data.add_(1.0)
        blk.ln_2.weight.data.add_(1.0)
    module_clip.convert_weights(tw)
    tw = tw.to(DEV)
    x0 = torch.randn(nseq * L, width) * 0.7 + 0.1
    if outlier is not None:
        x0[::outlier[0]] *= outlier[1]
    x0 = x0.half().to(DEV)
    wsel = torch.randn(nseq * L, width).to(DEV) * gscale

    def fp32_run():
        P = [[q.detach().double().requires_grad_() for q in Fn.block_params(blk)] for blk in tw.resblocks]
        xd = x0.double().requires_grad_()
        h = xd.view(nseq, L, width)
        mask = torch.full((L, L), float("-inf"), device=DEV, dtype=torch.float64).triu_(1) if causal else None
        for p in P:
            a = torch.nn.functional.layer_norm(h, (width,), p[0], p[1], 1e-5)
            qkv = a @ p[2].t() + p[3]
            q, k, v = [t.view(nseq, L, heads, 64).transpose(1, 2) for t in qkv.chunk(3, -1)]
            s_ = (q @ k.transpose(-1, -2)) / 8.0
            if mask is not None:
                s_ = s_ + mask
            o = (s_.softmax(-1) @ v).transpose(1, 2).reshape(nseq, L, width)
            h = h + o @ p[4].t() + p[5]
            a = torch.nn.functional.layer_norm(h, (width,), p[6], p[7], 1e-5)
            f = a @ p[8].t() + p[9]
            h = h + (f * torch.sigmoid(1.702 * f)) @ p[10].t() + p[11]
        (h.reshape(nseq * L, width) * wsel.double()).sum().backward()
        return h.reshape(nseq * L, width).detach(), xd.grad, [[q.grad for q in p] for p in P]
    return tw, x0, wsel, fp32_run


@pytest.mark.parametrize("width,heads,L,nseq,layers,causal", [(256, 4, 10, 210, 3, False), (768, 12, 50, 64, 3, False), (512, 8, 32, 96, 2, True),
                                                              (768, 12, 197, 16, 2, False)])
def test_folded_training_tower_against_fp32_autograd(width, heads, L, nseq, layers, causal, monkeypatch):
    """hmmc_tower_fwd_fused(keep_acts) + hmmc_tower_bwd_fold: output, input gradient and EVERY parameter gradient against fp64
    autograd through the same blocks - as close as the unfolded kernels are (x 1.5), tensor by tensor."""
    tw, x0, wsel, fp32_run = _tower_and_reference(width, heads, L, nseq, layers, causal)
    yr, dxr, gr = fp32_run()
    monkeypatch.setattr(Fn, "_FOLD_LN_TRAIN", "vit")            # the tower's own flag decides (fold_ln below)
    res = {}
    for fold in (False, True):
        tw.fold_ln = fold
        for prm in tw.parameters():
            prm.grad = None
        x = x0.clone().requires_grad_()
        y = tw(x, nseq, L)
        (y.float() * wsel).sum().backward()
        res[fold] = (y.detach(), x.grad, [[q.grad for q in Fn.block_params(blk)] for blk in tw.resblocks])
    assert not torch.equal(res[True][0], res[False][0]), "the folded kernels did not run"
    names = ["ln_1.w", "ln_1.b", "in_proj.w", "in_proj.b", "out_proj.w", "out_proj.b", "ln_2.w", "ln_2.b", "c_fc.w", "c_fc.b", "c_proj.w", "c_proj.b"]
    e0, e1 = relerr(res[False][0], yr), relerr(res[True][0], yr)
    assert e1 <= 1.5 * e0 + 1e-4, ("y", e1, e0)
    e0, e1 = relerr(res[False][1], dxr), relerr(res[True][1], dxr)
    print(f"dx: unfolded {e0:.3e} folded {e1:.3e}")
    assert e1 <= 1.5 * e0 + 2e-3, ("dx", e1, e0)
    worst = []
    for li in range(layers):
        for j, nm in enumerate(names):
            a0, a1 = relerr(res[False][2][li][j], gr[li][j]), relerr(res[True][2][li][j], gr[li][j])
            worst.append((a1 / max(a0, 1e-4), f"layer {li} {nm}", a0, a1))
            assert a1 <= 2.0 * a0 + 3e-3, (li, nm, a1, a0)
    print("worst ratios:", sorted(worst, reverse=True)[:4])


@pytest.mark.parametrize("factor,gscale", [(30.0, 1e-3), (100.0, 1e-4)])
def test_folded_backward_with_large_variance_rows_and_small_gradients(factor, gscale, monkeypatch):
    """Advisor, round 4: in the folded backward the data gradient in front of a LayerNorm is stored in fp16 ALREADY multiplied by
    the row's rstd, and the c_fc bias gradient's column sums are rebuilt from those rounded values x 1 / rstd.  Rows with a large
    variance (CLIP's outlier tokens: here every 16th row x 30 / x 100, rstd ~ 1/20 .. 1/70) push small gradients towards the fp16
    subnormals.  The gradients that pass through that hand-over - c_fc bias and weight, ln_2 / ln_1 gamma and beta, in_proj - must
    stay as close to fp64 autograd as the unfolded kernels' are (x 2 + 3e-3), whose own fp16 gradients are the reference's regime.
    Measured (round 5): at x 30 / 1e-3 folded = unfolded on every tensor (7e-4 .. 1.1e-3 rel-L2 either way); at x 100 / 1e-4 the
    tensors behind the scaled hand-over of the FOLDED layer (ln_1 / ln_2, in_proj / c_fc weight and bias) are 1.9 - 2.6 x further
    from fp64 than unfolded (3.2e-3 against 1.2e-3): rstd ~ 1/70 times gradients that are fp16 subnormals already costs one to
    two more bits there.  Inside the bound, recorded in DESIGN.md; `HMMC_FOLD_LN_TRAIN=0` is the exact path."""
    width, heads, L, nseq, layers = 768, 12, 50, 64, 2
    tw, x0, wsel, fp32_run = _tower_and_reference(width, heads, L, nseq, layers, False, outlier=(16, factor), gscale=gscale)
    yr, dxr, gr = fp32_run()
    monkeypatch.setattr(Fn, "_FOLD_LN_TRAIN", "vit")
    res = {}
    for fold in (False, True):
        tw.fold_ln = fold
        for prm in tw.parameters():
            prm.grad = None
        x = x0.clone().requires_grad_()
        y = tw(x, nseq, L)
        (y.float() * wsel).sum().backward()
        res[fold] = (y.detach(), x.grad, [[q.grad for q in Fn.block_params(blk)] for blk in tw.resblocks])
    assert not torch.equal(res[True][0], res[False][0]), "the folded kernels did not run"
    names = ["ln_1.w", "ln_1.b", "in_proj.w", "in_proj.b", "out_proj.w", "out_proj.b", "ln_2.w", "ln_2.b", "c_fc.w", "c_fc.b", "c_proj.w", "c_proj.b"]
    report, bad = [], []
    e0, e1 = relerr(res[False][1], dxr), relerr(res[True][1], dxr)
    report.append(("dx", e0, e1))
    if e1 > 2.0 * e0 + 2e-3:
        bad.append(("dx", e0, e1))
    for li in range(layers):
        for j, nm in enumerate(names):
            a0, a1 = relerr(res[False][2][li][j], gr[li][j]), relerr(res[True][2][li][j], gr[li][j])
            report.append((f"layer {li} {nm}", a0, a1))
            if a1 > 2.0 * a0 + 3e-3:
                bad.append((f"layer {li} {nm}", a0, a1))
    print(f"outlier rows x {factor}, upstream gradient {gscale}: (tensor, unfolded rel-L2 vs fp64, folded)")
    for r in report:
        print("   %-22s %.3e %.3e" % r)
    assert not bad, bad


def test_unsupported_fused_training_forward_falls_back_to_the_unfolded_kernels(monkeypatch):
    """Advisor, round 4: when hmmc_tower_fwd_fused(keep_acts = 1) answers HMMC_ERR_UNSUPPORTED (here: the grouped weight-gradient
    launch switched off through hmmc_set_option AFTER Python decided to fold) training must fall back to the unfolded kernels -
    same function, the reference's rounding points - instead of raising."""
    from hmmc_amd import _lib
    width, heads, L, nseq, layers = 256, 4, 10, 210, 2
    tw, x0, wsel, _ = _tower_and_reference(width, heads, L, nseq, layers, False)
    monkeypatch.setattr(Fn, "_FOLD_LN_TRAIN", "vit")
    _lib.set_option("no_wgrad_group", True)                                       # the library cannot run the folded training forward ...
    try:
        tw.fold_ln = False
        assert not Fn.fold_train_enabled(True, nseq * L, width, L)                # (and Python, asking the library, knows)
        x = x0.clone().requires_grad_()
        y0 = tw(x, nseq, L)
        (y0.float() * wsel).sum().backward()
        ref = (y0.detach().clone(), x.grad.clone(), [q.grad.clone() for blk in tw.resblocks for q in Fn.block_params(blk)])
        for prm in tw.parameters():
            prm.grad = None
        tw.fold_ln = True
        monkeypatch.setattr(Fn, "fold_train_enabled", lambda *a, **k: True)       # ... while Python says "fold"
        x = x0.clone().requires_grad_()
        y1 = tw(x, nseq, L)
        (y1.float() * wsel).sum().backward()
    finally:
        _lib.set_option("no_wgrad_group", False)
    got = (y1.detach(), x.grad, [q.grad for blk in tw.resblocks for q in Fn.block_params(blk)])
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    assert all(torch.equal(a, b) for a, b in zip(ref[2], got[2]))


def test_folded_training_tower_is_deterministic_and_chunks_agree(monkeypatch):
    """(1) Output, input gradient and every parameter gradient of the folded training tower bit-identical from run to run and with
    the weight-gradient stream on or off (the fp32 weight-gradient sums, the finish kernels and the deferred reductions add in a
    fixed order; no atomics).  (2) The tower cut into autograd nodes the way data parallelism cuts it (module_clip.Transformer.
    ddp_layers_per_node; here 2 + 2 layers by hand) gives the same results up to the rounding of the second chunk's input
    statistics, which then come from a pass over x instead of the producing GEMM's epilogue."""
    monkeypatch.setattr(Fn, "_FOLD_LN_TRAIN", "vit")
    width, heads, L, nseq, layers = 768, 12, 50, 48, 4
    tw, x0, wsel, _ = _tower_and_reference(width, heads, L, nseq, layers, False)
    tw.fold_ln = True

    def run(chunks=None):
        for prm in tw.parameters():
            prm.grad = None
        x = x0.clone().requires_grad_()
        if chunks is None:
            y = tw(x, nseq, L)
        else:
            y, blocks, i = x, list(tw.resblocks), 0
            for n in chunks:
                params = []
                for blk in blocks[i:i + n]:
                    params += Fn.block_params(blk)
                i += n
                y = Fn.clip_transformer(y, nseq, L, heads, False, False, *params, fold_train="last_exact" if i == layers else "all")
        (y.float() * wsel).sum().backward()
        torch.cuda.synchronize()
        return y.detach().clone(), x.grad.clone(), [q.grad.clone() for blk in tw.resblocks for q in Fn.block_params(blk)]
    saved = Fn._WGRAD_STREAM
    try:
        a, b = run(), run()
        Fn._WGRAD_STREAM = False
        c = run()
    finally:
        Fn._WGRAD_STREAM = saved
    for u, v, w in zip([a[0], a[1]] + a[2], [b[0], b[1]] + b[2], [c[0], c[1]] + c[2]):
        assert torch.equal(u, v), "run-to-run difference"
        assert torch.equal(u, w), "weight-gradient stream on / off difference"
    d = run(chunks=(2, 2))
    assert relerr(d[0], a[0]) < 2e-3 and relerr(d[1], a[1]) < 5e-3
    for u, v in zip(d[2], a[2]):
        assert relerr(u, v) < 5e-3, relerr(u, v)
