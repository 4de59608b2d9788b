"""Pin the oracle (oracle/hmmc_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from hmmc_amd import synth
from oracle import hmmc_oracle as O
from conftest import golden


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    assert np.allclose(a, b, atol=atol, rtol=rtol), f"{what}: max abs err {err:.3e} (atol {atol}, rtol {rtol})"


@pytest.mark.parametrize("tag", ["head_ft_small", "head_ft_c2"])
def test_finetune_head(tag):
    g = golden(tag)
    B, Fr = int(g["B"]), int(g["F"])
    q = synth.normal(f"{tag}.q", (B, 512)).requires_grad_()
    v = synth.normal(f"{tag}.v", (B, 512)).requires_grad_()
    u = synth.normal(f"{tag}.u", (B, Fr, 512)).requires_grad_()
    loss = O.finetune_head(q, v, u)
    loss.backward()
    close(loss.detach(), g["loss"], 1e-5, what="loss")
    s = O.loose_similarity(q, v).detach()
    if "S_video" in g:
        close(s, g["S_video"], 1e-3, what="S_video")       # north_star: fp32 logits within 1e-3
        close(q.grad, g["dQ"], 1e-6, what="dQ")
        close(v.grad, g["dV"], 1e-6, what="dV")
        close(u.grad, g["dU"], 1e-6, what="dU")
    else:
        close(s[:8], g["S_video_rows"], 1e-3, what="S_video_rows")
        close(q.grad[:8], g["dQ_rows"], 1e-6, what="dQ_rows")
        close(u.grad[:4], g["dU_rows"], 1e-6, what="dU_rows")
        close(q.grad.norm(), g["dQ_norm"], 1e-6, what="dQ_norm")


def test_eval_scorer_and_metrics():
    g = golden("head_eval")
    q = synth.normal("head_eval.q", (48, 512))
    v = synth.normal("head_eval.v", (48, 512))
    u = synth.normal("head_eval.u", (48, 12, 512))
    q = q + 0.7 * v
    for k in (1, 2, 3, 12):
        sv, sf = O.eval_scores(q, v, u, k)
        close(sv, g["S_video"], 1e-3, what="S_video")
        close(sf, g[f"topk{k}"], 1e-3, what=f"topk{k}")
        m = O.compute_metrics((sv + sf).numpy())
        close([m["R1"], m["R5"], m["R10"], m["MR"], m["MeanR"]], g[f"metrics{k}"], 1e-9, what=f"metrics{k}")
        # retrieval ranks identical
        ref_rank = np.argsort(-(g["S_video"] + g[f"topk{k}"]), axis=1)
        assert np.array_equal(np.argsort(-(sv + sf).numpy(), axis=1), ref_rank)
    m = O.compute_metrics(O.loose_similarity(q, v).numpy().T)
    close([m["R1"], m["R5"], m["R10"], m["MR"], m["MeanR"]], g["metrics_video_v2t"], 1e-9, what="v2t")


ENC = [("enc_tiny", "fp32", synth.TINY, True), ("enc_tiny", "aswritten", synth.TINY, True),
       ("enc_tiny_notemp", "fp32", synth.TINY, False),
       ("enc_tiny16", "fp32", synth.TINY16, True), ("enc_tiny16", "aswritten", synth.TINY16, True),
       ("enc_b32", "fp32", synth.VIT_B32, True), ("enc_b32", "aswritten", synth.VIT_B32, True),
       ("enc_b16", "fp32", synth.VIT_B16, True)]          # true ViT-B/16 dims (197 tokens x 12 heads)


@pytest.mark.parametrize("name,mode,dims,use_temp", ENC)
def test_encoders_and_loss(name, mode, dims, use_temp):
    g = golden(f"{name}_{mode}")
    assert json.loads(str(g["dims"])) == dims.to_dict()
    B, Fr, L = int(g["B"]), int(g["F"]), int(g["L"])
    sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in synth.finetune_state(dims, use_temp=use_temp).items()}
    ids, mask, vid, vf, idx = synth.finetune_batch(B, Fr, L, dims.image_res, tag=name)
    loss, (q, v, u) = O.finetune_loss(ids, vid, sd, mode=mode, use_temp=use_temp)
    # fp32: same math, different op fusion (the reference goes through nn.MultiheadAttention/SDPA);
    # as-written: fp16 rounding points differ between SDPA and the explicit restatement.
    tol = 2e-4 if mode == "fp32" else 3e-2
    close(q.detach(), g["text_feat"], tol, tol, "text_feat")
    close(u.detach(), g["frame_output"], tol, tol, "frame_output")
    close(v.detach(), g["video_emb"], tol, tol, "video_emb")
    close(loss.detach(), g["loss"], 1e-4 if mode == "fp32" else 2e-2, what="loss")
    if mode == "fp32":
        loss.backward()
        names = [str(n) for n in g["grad_norm_names"]]
        ref = dict(zip(names, g["grad_norm_values"]))
        for n in names:
            gn = float(sd[n].grad.norm())
            assert abs(gn - ref[n]) <= 2e-3 * max(ref[n], 1e-6) + 1e-7, f"grad norm {n}: {gn} vs {ref[n]}"
        for key in g.files:
            if key.startswith("g:") and "[" not in key:
                ref_slice = g[key]
                full = sd[key[2:]].grad
                idx_ = tuple(slice(0, s) for s in ref_slice.shape)
                if key.endswith("conv1.weight"):
                    got = full[0:2, 0, 0:4, 0:4]
                else:
                    got = full[idx_]
                close(got, ref_slice, 1e-5, 2e-3, key)
        tg = sd["text_encoder.token_embedding.weight"].grad
        close(tg[synth.SOT, :8], g["g:text_encoder.token_embedding.weight[SOT]"], 1e-6, 2e-3, "tok SOT")
        close(tg[synth.EOT, :8], g["g:text_encoder.token_embedding.weight[EOT]"], 1e-6, 2e-3, "tok EOT")


def test_encoders_b32x8_scores():
    """The 8 x 8 retrieval problem at true ViT-B/32 dims (fp32 regime): features, x100 logits and the mean top-k frame
    logits of the oracle against the reference's, and identical argsort of every row."""
    g = golden("enc_b32x8_fp32")
    dims = synth.VIT_B32
    B, Fr, L, k = int(g["B"]), int(g["F"]), int(g["L"]), int(g["k"])
    sd = synth.finetune_state(dims)
    ids, mask, vid, vf, idx = synth.finetune_batch(B, Fr, L, dims.image_res, tag="enc_b32x8")
    with torch.no_grad():
        loss, (q, v, u) = O.finetune_loss(ids, vid, sd, mode="fp32")
        sv, fk = O.eval_scores(q, v, u, k)
    close(q, g["text_feat"], 2e-4, 2e-4, "text_feat")
    close(u, g["frame_output"], 2e-4, 2e-4, "frame_output")
    close(v, g["video_emb"], 2e-4, 2e-4, "video_emb")
    close(sv, g["S_video"], 1e-3, what="S_video (1e-3 logits)")
    close(fk, g["S_frame_topk"], 1e-3, what="top-k frame logits")
    close(loss, g["loss"], 1e-4, what="loss")
    assert np.array_equal(np.argsort(-sv.numpy(), 1), np.argsort(-g["S_video"], 1))
    assert np.array_equal(np.argsort(-(sv + fk).numpy(), 1), np.argsort(-(g["S_video"] + g["S_frame_topk"]), 1))


BERTADAM_SPECS = [("a32", (37,), torch.float32, 0.2, 1e-4, 3.0), ("b32", (8, 9), torch.float32, 0.0, 3e-5, 0.01),
                  ("c16", (64,), torch.float16, 0.2, 1e-4, 2.0), ("d16", (4, 32), torch.float16, 0.0, 1e-7, 0.05)]


def test_bertadam_bit_exact():
    """BertAdam.step op by op in each tensor's dtype.  fp16 tensors are sized as multiples of the CPU vector
    width: torch's CPU kernels round differently in their scalar tail loop (Half arithmetic) than in the
    vectorised body (float arithmetic), an artefact of the host library, not of the reference's algorithm."""
    g = golden("bertadam")
    for name, shape, dt, wd, lr, gs in BERTADAM_SPECS:
        p = synth.normal(f"bertadam.{name}.p", shape, 0.5).to(dt)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in range(5):
            gr = synth.normal(f"bertadam.{name}.g{step}", shape, gs).to(dt)
            p, m, v, gc = O.bert_adam_step(p, gr, m, v, step, lr, 20, 0.1, wd)
            for nm, mine in (("g", gc), ("m", m), ("v", v), ("p", p)):
                ref = t(g[f"{name}.{nm}{step}"])
                assert torch.equal(mine.float(), ref), f"{name}.{nm}{step} differs from the reference"
    close(g["lr1"], [1e-4 * O.warmup_cosine(2 / 20, 0.1) * s for s in (1, 0.3, 1, 1e-3)], 1e-12, what="lr")  # get_lr() reads the already incremented step


def _groups(names, lr, text_lr, coef_lr, wd):
    """main_task_retrieval.py:171-205: (lr, weight_decay) per parameter name."""
    out = {}
    for n in names:
        nod = any(nd in n for nd in ("bias", "LayerNorm.bias", "LayerNorm.weight"))
        if "visual_encoder.visual." in n:
            l = lr * coef_lr
        elif "text_encoder." in n:
            l = text_lr
        else:
            l = lr
        out[n] = (l, 0.0 if nod else wd)
    return out


def test_train_steps_fp32():
    """4 steps of forward / backward / clip_grad_norm_(1.0) / BertAdam (main_task_retrieval.py:272-302)."""
    g = golden("train_ft_fp32")
    sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in synth.finetune_state(synth.TINY).items()}
    names = [k for k, v in sd.items() if v.requires_grad]
    hp = _groups(names, 2e-3, 1e-3, 0.5, 0.2)
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    for step in range(4):
        ids, mask, vid, vf, idx = synth.finetune_batch(4, 4, 32, tag=f"train_ft.s{step}")
        for k in names:
            sd[k].grad = None
        loss, _ = O.finetune_loss(ids, vid, sd, mode="fp32")
        loss.backward()
        tn = torch.nn.utils.clip_grad_norm_([sd[k] for k in names], 1.0)
        with torch.no_grad():
            for k in names:
                m, v = state[k]
                p, m, v, _ = O.bert_adam_step(sd[k].data, sd[k].grad, m, v, step, hp[k][0], 10, 0.1, hp[k][1])
                sd[k].data.copy_(p)
                state[k] = (m, v)
        close(loss.detach(), g[f"loss{step}"], 2e-3, what=f"loss{step}")
        close(tn, g[f"gnorm{step}"], 0, 5e-3, f"gnorm{step}")
        if step < 2:
            for key in g.files:
                if key.startswith(f"p{step}:"):
                    close(sd[key.split(":", 1)[1]].detach().reshape(-1)[:16], g[key], 2e-4, 1e-3, key)


def _moco_oracle_run(mode, nsteps=5):
    """Reference main_pretrain.py loop on the oracle: forward (EMA, queues, MLM), backward, clip, BertAdam."""
    g = golden(f"moco_{mode}")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    raw = synth.pretrain_state(synth.TINY, K, Fr)
    sd = {}
    for k, v in raw.items():
        trainable = v.is_floating_point() and not any(s in k for s in ("_k.", "queue_", "running_", "num_batches"))
        sd[k] = v.clone().requires_grad_(trainable)
    sd["cls.decoder.bias"] = sd["cls.bias"]
    queues = {k: sd[k] for k in sd if k.startswith("queue_") and k != "queue_ptr"}
    names = [k for k, v in sd.items() if v.requires_grad and k != "cls.decoder.bias"]
    hp = _groups(names, 2e-3, 1e-3, 0.5, 0.05)
    state = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    ptr = 0
    out = []
    for step in range(nsteps):
        batch = synth.pretrain_batch(B, Fr, tag=f"moco.s{step}")
        draws = [t(g[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
        draws = [d.bool() if i < 3 else d for i, d in enumerate(draws)]
        for k in names:
            sd[k].grad = None
        loss, parts, ptr = O.pretrain_loss(batch, sd, queues, ptr, K, mode=mode, mlm_draws=draws)
        loss.backward()
        used = [sd[k] for k in names if sd[k].grad is not None]
        tn = torch.nn.utils.clip_grad_norm_(used, 1.0)
        with torch.no_grad():
            for k in names:
                if sd[k].grad is None:
                    continue
                m, v = state[k]
                p, m, v, _ = O.bert_adam_step(sd[k].data, sd[k].grad, m, v, step, hp[k][0], 10, 0.1, hp[k][1])
                sd[k].data.copy_(p)
                state[k] = (m, v)
        out.append((loss.detach(), parts, tn, ptr))
        yield step, g, sd, queues, loss.detach(), parts, tn, ptr


def test_pretrain_steps_fp32():
    for step, g, sd, queues, loss, parts, tn, ptr in _moco_oracle_run("fp32"):
        fam, vtm, ftm, mlm = [float(x) for x in parts]
        close(fam, g[f"fam{step}"], 2e-3, what=f"fam{step}")
        close(ftm, g[f"ftm{step}"], 2e-3, what=f"ftm{step}")
        close(mlm, g[f"mlm{step}"], 2e-3, what=f"mlm{step}")
        close(loss, g[f"loss{step}"], 2e-3, what=f"loss{step}")
        assert ptr == int(g[f"ptr{step}"][0])
        if step == 0:
            close(tn, g["gnorm0"], 0, 5e-3, "gnorm0")
            for key in g.files:
                if key.startswith("s0:"):
                    close(sd[key[3:]].detach().reshape(-1)[:16], g[key], 1e-5, 1e-4, key)
                if key.startswith("q0:"):
                    close(queues[key[3:]][:32], g[key], 1e-5, what=key)


def test_module_members_fp32():
    """MLP.forward (train + eval), BertLMPredictionHead.forward and loose_similarity with gradients: the oracle's restatement
    against what the reference's own modules returned (tests/golden/modules_fp32.npz, make_golden.py:fx_modules)."""
    g = golden("modules_fp32")
    sd = {k: v.clone() for k, v in synth.pretrain_state(synth.TINY, 16, 4).items()}
    p = "v_projector."
    x = synth.normal("modules.mlp.x", (24, 512)).requires_grad_()
    w = synth.normal("modules.mlp.w", (24, 512))
    for k in list(sd):
        if k.startswith(p) and sd[k].is_floating_point() and "running" not in k:
            sd[k] = sd[k].float().requires_grad_()
    y, mean, var = O.mlp_forward(x, sd, p)
    (y * w).sum().backward()
    close(y.detach(), g["mlp_y"], 2e-5, 1e-5, "mlp y")
    close(x.grad, g["mlp_dx"], 2e-6, 1e-4, "mlp dx")
    close(sd[p + "linear_hidden.2.weight"].grad[:64], g["mlp_dgamma"], 2e-5, 1e-4, "dgamma")
    close(sd[p + "linear_out.weight"].grad[:8, :16], g["mlp_dw2"], 2e-5, 1e-4, "dw2")
    rm = 0.9 * sd[p + "linear_hidden.2.running_mean"] + 0.1 * mean.detach()
    rv = 0.9 * sd[p + "linear_hidden.2.running_var"] + 0.1 * var.detach()
    close(rm[:64], g["mlp_running_mean1"], 1e-6, 1e-5, "running mean")
    close(rv[:64], g["mlp_running_var1"], 1e-6, 1e-5, "running var")
    x3 = synth.normal("modules.mlp.x3", (3, 8, 512)).reshape(-1, 512)
    with torch.no_grad():
        y3, mean3, var3 = O.mlp_forward(x3, sd, p)
        close(y3.view(3, 8, 512), g["mlp_y3"], 2e-5, 1e-5, "mlp y3")
        sde = dict(sd)
        sde[p + "linear_hidden.2.running_mean"] = 0.9 * rm + 0.1 * mean3
        sde[p + "linear_hidden.2.running_var"] = 0.9 * rv + 0.1 * var3
        close(sde[p + "linear_hidden.2.running_var"][:64], g["mlp_running_var2"], 1e-6, 1e-5, "running var 2")
        ye, _, _ = O.mlp_forward(synth.normal("modules.mlp.xe", (10, 512)), sde, p, training=False)
        close(ye, g["mlp_y_eval"], 2e-5, 1e-5, "mlp eval")
    h = synth.normal("modules.lm.h", (3, 7, 512)).requires_grad_()
    wl = synth.normal("modules.lm.w", (3, 7, 64))
    sdl = {k: (v.float().requires_grad_() if k.startswith("cls.") else v) for k, v in sd.items()}
    logits = O.mlm_head(h, sdl)
    (logits[..., :64] * wl).sum().backward()
    close(logits.detach()[..., :128], g["lm_logits_head"], 2e-5, 1e-5, "lm logits")
    close(logits.detach().sum(-1), g["lm_logits_rowsum"], 5e-3, 1e-5, "lm logits row sums")
    close(h.grad, g["lm_dh"], 2e-6, 1e-4, "lm dh")
    close(sdl["cls.decoder.weight"].grad[:8, :16], g["lm_ddec"], 2e-6, 1e-4, "lm ddec")
    close(sdl["cls.bias"].grad[:128], g["lm_dbias"], 2e-6, 1e-4, "lm dbias")
    q = synth.normal("modules.sim.q", (6, 512)).requires_grad_()
    v = synth.normal("modules.sim.v", (5, 512)).requires_grad_()
    u = synth.normal("modules.sim.u", (5, 3, 512)).requires_grad_()
    ws, wu = synth.normal("modules.sim.ws", (6, 5)), synth.normal("modules.sim.wu", (6, 5, 3))
    s2, s3 = O.loose_similarity(q, v), O.loose_similarity(q, u)
    ((s2 * ws).sum() + (s3 * wu).sum()).backward()
    close(s2.detach(), g["sim2"], 1e-3, what="sim2")
    close(s3.detach(), g["sim3"], 1e-3, what="sim3")
    close(q.grad, g["sim_dq"], 2e-5, 1e-4, "sim dq")
    close(u.grad, g["sim_du"], 2e-5, 1e-4, "sim du")


def test_pretrain_forward_at_true_vit_b32_dims_fp32():
    """BASELINE config 4's model at true ViT-B/32 dimensions (K = 1 024, title 45 / tag 25, B = 4, F = 2): every loss part of the
    first step and the enqueued keys against the reference (tests/golden/moco_b32_fp32.npz; forward only - the CPU suite's time)."""
    g = golden("moco_b32_fp32")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    sd = {k: v.clone() for k, v in synth.pretrain_state(synth.VIT_B32, K, Fr).items()}
    sd["cls.decoder.bias"] = sd["cls.bias"]
    queues = {k: sd[k] for k in sd if k.startswith("queue_") and k != "queue_ptr"}
    batch = synth.pretrain_batch(B, Fr, tag="moco_b32.s0")
    draws = [t(g[f"mlm_{n}0"]) for n in ("masked", "replaced", "randsel", "words")]
    draws = [d.bool() if i < 3 else d for i, d in enumerate(draws)]
    with torch.no_grad():
        loss, parts, ptr = O.pretrain_loss(batch, sd, queues, 0, K, mode="fp32", mlm_draws=draws)
    fam, vtm, ftm, mlm = [float(x) for x in parts]
    close(fam, g["fam0"], 2e-3, what="fam0")
    close(ftm, g["ftm0"], 2e-3, what="ftm0")
    close(mlm, g["mlm0"], 2e-3, what="mlm0")
    close(loss, g["loss0"], 2e-3, what="loss0")
    assert ptr == int(g["ptr0"][0]) == B
    for key in g.files:
        if key.startswith("q0:"):
            close(queues[key[3:]][:32, :64], g[key], 1e-5, what=key)
