"""Random-shape sweep of the C-ABI kernels against torch references (a one-off robustness run, not a test):
fp16 GEMM in the three operand layouts with every tower epilogue on integer data (bit-exact), attention at random lengths,
LayerNorm forward / backward with row gathers, the multi-task column reduce.  usage: python scratch/fuzz_kernels.py [cases] [seed]"""
import sys, os, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import ops
DEV = "cuda"
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
def ints(*shape, lo=-2, hi=3):
    return torch.randint(lo, hi, shape, device=DEV).half()
def report(ok, what):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", what, flush=True)

# ---- fp16 GEMM
for case in range(ncases):
    lay = rng.choice(["kk", "km", "mm"])
    M = rng.choice([8, 64, 130, 256, 300, 1024, 2048, 4104]) if lay != "mm" else rng.choice([64, 128, 512, 768])
    N = rng.choice([64, 128, 200, 512, 768, 1536]) if lay != "mm" else rng.choice([64, 128, 512, 768])
    K = rng.choice([64, 128, 192, 512, 768]) if lay != "mm" else rng.choice([64, 200, 1000, 4096, 20000])
    M -= M % 8; N -= N % 8
    if lay == "mm": K -= K % 8
    epi = rng.choice(["", "b", "br", "bg", "bgs", "m"]) if lay != "mm" else ""
    try:
        if lay == "kk":
            a, b = ints(M, K), ints(N, K)
            ref = a.float() @ b.float().t()
        elif lay == "km":
            a, b = ints(M, K), ints(K, N)
            ref = a.float() @ b.float()
        else:
            a, b = ints(K, M, lo=-1, hi=2), ints(K, N, lo=-1, hi=2)
            ref = a.float().t() @ b.float()
        kw = dict(a_kmajor=lay != "mm", b_kmajor=lay == "kk")
        bias = ints(N) if "b" in epi else None
        resid = ints(M, N) if "r" in epi else None
        if bias is not None: ref = (ref + bias.float()).half().float()
        if "g" in epi:
            kw.update(epilogue=ops.EPI_QGELU | (ops.EPI_SAVE_DGELU if "s" in epi else 0), want_aux=True)
            h = ref.half()
            refo = (h * torch.sigmoid(1.702 * h)).float()
            out, aux = ops.gemm_f16(a, b, M, N, K, bias=bias, **kw)
            ok = torch.equal(out.float(), refo)
            if "s" not in epi: ok = ok and torch.equal(aux.float(), ref)
            else:
                s = torch.sigmoid(1.702 * ref)
                ok = ok and float((aux.float() - (s + 1.702 * ref * s * (1 - s))).abs().max()) < 4e-3
            report(ok, (lay, M, N, K, epi))
            continue
        if "m" in epi:
            aux_in = ints(M, N)
            out, part = ops.gemm_f16(a, b, M, N, K, aux_in=aux_in, epilogue=ops.EPI_MULAUX, want_colsum=True, **kw)
            refo = (ref * aux_in.float()).half().float()
            ok = torch.equal(out.float(), refo) and float((part.sum(0) - refo.sum(0)).abs().max()) < 1e-2 * (1 + float(refo.sum(0).abs().max()))
            report(ok, (lay, M, N, K, epi))
            continue
        if resid is not None: ref = (resid.float() + ref).half().float()
        out = ops.gemm_f16(a, b, M, N, K, bias=bias, resid=resid, **kw)
        report(torch.equal(out.float(), ref.half().float()), (lay, M, N, K, epi))
    except Exception as e:                                     # an error status for an unsupported shape is fine; a wrong answer is not
        print("status", (lay, M, N, K, epi), str(e)[:80], flush=True)
torch.cuda.synchronize()
print("gemm done, mismatches", bad, flush=True)

# ---- LayerNorm-fold epilogues (round 4): LNFOLD [+ QuickGELU], ROWSTAT, ROWSCALE against fp64 on random (ragged) shapes
for case in range(ncases):
    M = rng.choice([8, 64, 130, 256, 300, 1000, 2048, 4104, 5000]); M -= M % 8
    K = rng.choice([64, 128, 192, 512, 768])
    kind = rng.choice(["fold", "foldg", "stat", "scale"])
    N = rng.choice([64, 128, 192, 512, 768, 1536, 2304]) if kind != "fold" and kind != "foldg" else rng.choice([64, 128, 200, 512, 768, 1536])
    N -= N % 8
    try:
        g = torch.Generator(device=DEV).manual_seed(case)
        x = (torch.randn(M, K, device=DEV, generator=g) * 1.2 + 0.3).half()
        W = (torch.randn(N, K, device=DEV, generator=g) * 0.05).half()
        b = (0.1 * torch.randn(N, device=DEV, generator=g)).half()
        if kind in ("fold", "foldg"):
            gm = 1.0 + 0.2 * torch.randn(K, device=DEV, generator=g); bt = 0.1 * torch.randn(K, device=DEV, generator=g)
            (Wf, cd), = ops.ln_fold_prep([(W, gm, bt, b)])
            st = ops.rowstat(x)
            y = ops.gemm_f16_fold(x, Wf, rowstat=st, colterms=cd, epilogue=ops.EPI_QGELU if kind == "foldg" else 0)
            xd = x.double(); mean = xd.mean(1, keepdim=True); rstd = torch.rsqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)
            ex = rstd * (xd @ Wf.double().t()) - rstd * mean * Wf.double().sum(1)[None] + (W.double() * bt.double()[None]).sum(1)[None] + b.double()[None]
            if kind == "foldg": ex = ex * torch.sigmoid(1.702 * ex)
            err = (y.double() - ex).abs()
            report(bool((err <= 2.5e-3 * ex.abs() + 2e-3).all()), (kind, M, N, K, float(err.max())))
        elif kind == "stat":
            r = (torch.randn(M, N, device=DEV, generator=g) + 0.5).half()
            y, part = ops.gemm_f16_fold(x, W, bias=b, resid=r, want_stat=True)
            y0 = ops.gemm_f16(x, W, M, N, K, bias=b, resid=r)
            yb = y.float().view(M, N // 64, 64)
            ok = torch.equal(y, y0) and torch.allclose(part[:, :, 0].t(), yb.sum(2), rtol=1e-5, atol=1e-3) and \
                torch.allclose(part[:, :, 1].t(), (yb * yb).sum(2), rtol=1e-5, atol=1e-3)
            report(ok, (kind, M, N, K))
        else:
            dy = (torch.randn(M, K, device=DEV, generator=g) * 0.1).half()             # dy [M, Np = K], w [Np, Kp = N]
            w2 = (torch.randn(K, N, device=DEV, generator=g) * 0.05).half()
            aux = torch.rand(M, N, device=DEV, generator=g).half()
            st = ops.rowstat(x)
            out, part = ops.gemm_f16_rowscaled_dgrad(dy, w2, aux, st)
            base = (dy.double() @ w2.double()) * aux.double()
            ref = base * st[:, :1].double()
            err = (out.double() - ref).abs()
            ok = bool((err <= 1.5e-3 * ref.abs() + 1e-4).all()) and torch.allclose(part.sum(0).double(), base.sum(0), rtol=3e-3, atol=3e-2)
            report(ok, (kind, M, N, K))
    except Exception as e:
        print("status", (kind, M, N, K), str(e)[:80], flush=True)
torch.cuda.synchronize()
print("fold epilogues done, mismatches", bad, flush=True)

# ---- attention
def attn_ref(qkv, nseq, L, H, causal):
    D = H * 64
    q, k, v = qkv.float().view(nseq, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal: s = s + torch.full((L, L), float("-inf"), device=qkv.device).triu_(1)
    p = torch.softmax(s, -1)
    return (p @ v).permute(0, 2, 1, 3).reshape(nseq * L, D)
for case in range(ncases // 3):
    L = rng.randint(1, 256); H = rng.choice([1, 2, 8, 12]); nseq = rng.randint(1, 9); causal = rng.random() < 0.5
    D = H * 64
    qkv = (torch.randn(nseq * L, 3 * D, device=DEV) * 0.7).half().requires_grad_(True)
    out, lse = ops.attention_f16_fwd(qkv.detach(), nseq, L, H, causal)
    ref = attn_ref(qkv, nseq, L, H, causal)
    e = float((out.float() - ref).norm() / (ref.norm() + 1e-9))
    dout = (torch.randn(nseq * L, D, device=DEV)).half()
    ref.backward(dout.float())
    dqkv, part = ops.attention_f16_bwd(qkv.detach(), out, lse, dout, nseq, L, H, causal, want_dbias=True)
    e2 = float((dqkv.float() - qkv.grad.float()).norm() / (qkv.grad.float().norm() + 1e-9))
    e3 = float((part.sum(0) - dqkv.float().sum(0)).abs().max()) / (1e-3 + float(dqkv.float().sum(0).abs().max()))
    report(e < 4e-3 and e2 < 1e-2 and e3 < 2e-3, ("attn", nseq, L, H, causal, e, e2, e3))
print("attention done, mismatches", bad, flush=True)

# ---- LayerNorm
for case in range(ncases // 3):
    D = rng.choice([64, 128, 512, 768, 1024]); rows = rng.randint(1, 3000)
    x = torch.randn(rows, D, device=DEV).half(); g = torch.randn(D, device=DEV); b = torch.randn(D, device=DEV)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    xr = x.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), g, b, 1e-5)
    dy = torch.randn(rows, D, device=DEV).half()
    ref.backward(dy.float())
    res = ops.layernorm_bwd(dy, x, g, mean, rstd, want_colsum=True)
    dx, dg, db = res[0], res[1], res[2]
    e1 = float((y.float() - ref).abs().max()); e2 = float((dx.float() - xr.grad).norm() / (xr.grad.norm() + 1e-9))
    rg = (dy.float() * ((x.float() - x.float().mean(1, keepdim=True)) * rstd[:, None])).sum(0)
    e3 = float((dg.float() - rg).abs().max()) / (1 + float(rg.abs().max()))
    report(e1 < 2e-2 and e2 < 3e-3 and e3 < 2e-3, ("ln", rows, D, e1, e2, e3))
torch.cuda.synchronize()
ops.raise_on_device_errors()
print("done, mismatches", bad)

# ---- round 3: fp32 attention of the fp32 regime (1..256 tokens, both kernels) and grouped weight gradients
for case in range(max(ncases // 5, 20)):
    nseq, L, H, causal = rng.randint(1, 5), rng.randint(1, 256), rng.choice([1, 2, 8, 12]), rng.random() < 0.4
    D = 64 * H
    qkv = torch.randn(nseq * L, 3 * D, device=DEV)
    dout = torch.randn(nseq * L, D, device=DEV)
    out, probs = ops.attention_f32_fwd(qkv, nseq, L, H, causal)
    qr = qkv.clone().requires_grad_()
    q, k, v = qr.view(nseq, L, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / 8.0
    if causal:
        s = s + torch.full((L, L), float("-inf"), device=DEV).triu_(1)
    oref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(nseq * L, D)
    oref.backward(dout)
    dqkv = ops.attention_f32_bwd(qkv, probs, dout, nseq, L, H)
    e1 = float((out - oref).abs().max()); e2 = float((dqkv - qr.grad).abs().max())
    report(e1 < 2e-5 and e2 < 2e-4, ("attention_f32", nseq, L, H, causal, e1, e2))
print("fp32 attention done, mismatches", bad, flush=True)
for case in range(max(ncases // 10, 10)):
    T = rng.choice([512, 520, 1000, 1024, 1440, 2048, 2056, 5000, 12345 - 12345 % 8, 30000])     # round 5: the launch takes >= 512 tokens
    D = rng.choice([256, 512, 768])
    n = rng.randint(1, 4)
    dims = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)][:n]
    dys = [ints(T, a) for a, _ in dims]
    xs = [ints(T, b) for _, b in dims]
    outs = ops.wgrad_group(dys, xs)
    report(outs is not None and all(torch.equal(o.float(), (dy.float().t() @ x.float()).half().float()) for dy, x, o in zip(dys, xs, outs)),
           ("wgrad_group", T, D, n))
print("grouped weight gradients done, mismatches", bad, flush=True)
print("TOTAL mismatches", bad)
