"""GPU parity tests of the pre-training path (BirdPreTrainedModel: MoCo queues, EMA, FAM/VTM/FTM/MLM) against the
reference's golden vectors (tests/golden/moco_*.npz, 5 steps with queue wrap-around) and the CPU oracle."""
from argparse import Namespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden  # noqa: E402
from hmmc_amd import ops, synth  # noqa: E402
from hmmc_amd import functional as Fn  # noqa: E402
from oracle import hmmc_oracle as O  # noqa: E402
from test_gpu_model import close, prep_optimizer, task_config  # noqa: E402

DEV = "cuda"


def test_mlp_fn_vs_torch_batchnorm():
    torch.manual_seed(0)
    x = torch.randn(48, 512)
    lin1, bn, lin2 = torch.nn.Linear(512, 4096), torch.nn.BatchNorm1d(4096), torch.nn.Linear(4096, 512)
    with torch.no_grad():
        bn.weight.normal_(1, 0.1)
        bn.bias.normal_(0, 0.1)
    ref_in = x.clone().requires_grad_()
    ref = lin2(torch.relu(bn(lin1(ref_in))))
    w = torch.randn(48, 512)
    (ref * w).sum().backward()
    P = [p.detach().clone().to(DEV).requires_grad_() for p in (lin1.weight, lin1.bias, bn.weight, bn.bias, lin2.weight, lin2.bias)]
    xg = x.to(DEV).requires_grad_()
    rm, rv, nbt = torch.zeros(4096, device=DEV), torch.ones(4096, device=DEV), torch.zeros((), dtype=torch.long, device=DEV)
    out, mean, var = Fn.MlpFn.apply(xg, *P, bn.eps, (rm, rv, nbt, bn.momentum))
    (out * w.to(DEV)).sum().backward()
    close(out, ref, 2e-4, 1e-4, "mlp out")
    close(xg.grad, ref_in.grad, 2e-4, 2e-3, "dx")
    for mine, refp, nm in zip(P, (lin1.weight, lin1.bias, bn.weight, bn.bias, lin2.weight, lin2.bias),
                              ("w1", "b1", "gamma", "beta", "w2", "b2")):
        close(mine.grad, refp.grad, 3e-4, 2e-3, nm)
    close(mean * 0.1, bn.running_mean, 1e-5, 1e-4, "running mean")
    # the train-mode running statistics, updated by the same launch (hmmc_bn_finalize), against nn.BatchNorm1d's own
    close(rm, bn.running_mean, 1e-6, 1e-5, "running_mean")
    close(rv, bn.running_var, 1e-6, 1e-5, "running_var")
    assert int(nbt) == int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("R,Kq", [(4, 16), (24, 64), (16, 64), (40, 192), (352, 12288)])
def test_moco_loss_fn_vs_oracle(R, Kq):
    E = 512
    q, k = synth.normal("moco.q", (R, E)), synth.normal("moco.k", (R, E))
    queue = torch.nn.functional.normalize(synth.normal("moco.queue", (E, Kq)), dim=0)
    qo = q.clone().requires_grad_()
    ref = O.contrastive_loss(qo, k, queue, 0.07)
    ref.backward()
    qg = q.to(DEV).requires_grad_()
    loss = Fn.MocoLossFn.apply(qg, k.to(DEV), queue.to(DEV), 0.07, 1.0 / R)
    loss.backward()
    close(loss, ref, 1e-5, 1e-5, "loss")
    close(qg.grad, qo.grad, 1e-6, 1e-3, "dq")


def test_mlm_head_fn_vs_oracle():
    sd = synth.pretrain_state(synth.TINY, 16, 4)
    hidden = synth.normal("mlm.hidden", (3, 20, 512), 0.5)
    labels = torch.full((3, 20), -100, dtype=torch.long)
    labels[0, 3], labels[1, 7], labels[2, 19], labels[2, 1] = 17, 40000, 49407, 5
    ho = hidden.clone().requires_grad_()
    sdo = {k: sd[k].clone().requires_grad_() for k in sd if k.startswith("cls.")}
    scores = O.mlm_head(ho, sdo)
    ref = torch.nn.functional.cross_entropy(scores.view(-1, scores.shape[-1]), labels.view(-1), ignore_index=-100)
    ref.backward()
    hg = hidden.to(DEV).requires_grad_()
    P = {k: sd[k].clone().to(DEV).requires_grad_() for k in sdo}
    loss = Fn.MlmHeadFn.apply(hg, labels.to(DEV), P["cls.transform.dense.weight"], P["cls.transform.dense.bias"],
                              P["cls.transform.LayerNorm.weight"], P["cls.transform.LayerNorm.bias"],
                              P["cls.decoder.weight"], P["cls.bias"])
    loss.backward()
    close(loss, ref, 1e-5, 1e-5, "mlm loss")
    close(hg.grad, ho.grad, 1e-6, 2e-3, "dhidden")
    close(P["cls.decoder.weight"].grad, sdo["cls.decoder.weight"].grad, 1e-6, 2e-3, "ddecoder")
    close(P["cls.bias"].grad, sdo["cls.bias"].grad, 1e-6, 2e-3, "dbias")
    close(P["cls.transform.dense.weight"].grad, sdo["cls.transform.dense.weight"].grad, 1e-6, 2e-3, "ddense")
    # no labelled position at all: zero loss and zero gradients (the head is evaluated on labelled rows only)
    h0 = hidden.to(DEV).requires_grad_()
    for q in P.values():
        q.grad = None
    loss0 = Fn.MlmHeadFn.apply(h0, torch.full((3, 20), -100, dtype=torch.long, device=DEV), P["cls.transform.dense.weight"],
                               P["cls.transform.dense.bias"], P["cls.transform.LayerNorm.weight"],
                               P["cls.transform.LayerNorm.bias"], P["cls.decoder.weight"], P["cls.bias"])
    loss0.backward()
    assert float(loss0) == 0.0 and float(h0.grad.abs().max()) == 0.0
    assert all(q.grad is None or float(q.grad.abs().max()) == 0.0 for q in P.values())
    assert P["cls.decoder.weight"].grad is not None and P["cls.transform.dense.weight"].grad is not None


def test_mlm_head_row_buffer_capacity_and_overflow_flag():
    """Above 4 096 positions the head runs on a fixed-capacity buffer of compacted rows (no host read of the labelled
    count): same loss and gradients as the all-rows evaluation, and a count beyond the capacity raises the device flag."""
    sd = synth.pretrain_state(synth.TINY, 16, 4)
    P = {k: v.to(DEV) for k, v in sd.items() if k.startswith("cls.")}
    args = (P["cls.transform.dense.weight"], P["cls.transform.dense.bias"], P["cls.transform.LayerNorm.weight"],
            P["cls.transform.LayerNorm.bias"], P["cls.decoder.weight"], P["cls.bias"])
    n = 6000
    hidden = synth.normal("mlm.cap.hidden", (n, 512), 0.5).to(DEV)
    gen = torch.Generator().manual_seed(3)
    labels = torch.where(torch.rand(n, generator=gen) < 0.15, torch.randint(0, 49408, (n,), generator=gen), torch.full((n,), -100))
    labels = labels.to(DEV)
    assert Fn._mlm_capacity(n, 0.15) < n // 4
    outs = []
    for prob in (0.15, 1.0):                       # 1.0: capacity = every row
        h = hidden.clone().requires_grad_()
        w = [a.clone().requires_grad_() for a in args]
        loss = Fn.MlmHeadFn.apply(h, labels, *w, prob)
        loss.backward()
        outs.append((loss.detach(), h.grad, w[4].grad, w[0].grad))
    ops.raise_on_device_errors()                   # nothing flagged
    close(outs[0][0], outs[1][0], 1e-6, 1e-6, "loss")
    close(outs[0][1], outs[1][1], 1e-7, 1e-4, "dhidden")
    close(outs[0][2], outs[1][2], 1e-7, 1e-4, "ddecoder")
    close(outs[0][3], outs[1][3], 1e-7, 1e-4, "ddense")
    Fn.MlmHeadFn.apply(hidden, torch.full((n,), 7, dtype=torch.long, device=DEV), *args, 0.15)
    with pytest.raises(IndexError):
        ops.raise_on_device_errors()


def test_pretrain_steps_vs_reference_golden():
    """5 steps of main_pretrain.py's loop; K=16, B=4 wraps the queue pointer at step 4."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_
    g = golden("moco_aswritten")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    sd = synth.pretrain_state(synth.TINY, K, Fr)
    cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt", lr=2e-3, text_lr=1e-3, coef_lr=0.5,
                      weight_decay=0.05)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=sd, task_config=cfg).to(DEV).train()
    opt = prep_optimizer(model, cfg, 10)
    for step in range(5):
        vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(B, Fr, tag=f"moco.s{step}")]
        model._mlm_draws = [torch.from_numpy(g[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
        loss = model(vid, vf, tg, gm, ti, tm, step + 1)
        loss.backward()
        tn = clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        fam, vtm, ftm, mlm = [float(x) for x in model.last_losses]
        tol = 5e-2 if step < 2 else 0.4          # weights move (chaotically in fp16, see test_gpu_model) from step 2 on
        close(fam, g[f"fam{step}"], tol, what=f"fam{step}")
        close(ftm, g[f"ftm{step}"], tol, what=f"ftm{step}")
        close(mlm, g[f"mlm{step}"], tol, what=f"mlm{step}")
        close(loss, g[f"loss{step}"], tol, what=f"loss{step}")
        assert int(model.queue_ptr) == int(g[f"ptr{step}"][0]), "queue pointer"
        S = model.state_dict()
        if step == 0:
            close(tn, g["gnorm0"], 0, 0.15, "gnorm0")
            # EMA of fp16 / fp32 key-encoder weights after the first update: bit patterns
            for k in ("visual_encoder_k.visual.conv1.weight", "text_encoder_k.text_projection",
                      "visual_encoder_k.temporal_transformer.resblocks.0.mlp.c_fc.weight",
                      "v_projector_k.linear_out.weight", "text_encoder_k.ln_final.weight"):
                mine = S[k].reshape(-1)[:16].float().cpu()
                ref = torch.from_numpy(g[f"s0:{k}"])
                assert torch.equal(mine, ref), f"EMA {k}: {mine} vs {ref}"
            for k in ("v_projector.linear_hidden.2.running_mean", "v_projector.linear_hidden.2.running_var",
                      "v_projector_k.linear_hidden.2.running_mean", "v_predictor.linear_hidden.2.running_var"):
                close(S[k].reshape(-1)[:16], g[f"s0:{k}"], 2e-3, 2e-2, k)
            for qn in ("queue_v_cross_ng", "queue_title_cross_ng", "queue_tag_cross_ng", "queue_frame_proj_ng",
                       "queue_frame_cross_ng"):
                close(S[qn][:32], g[f"q0:{qn}"], 2e-3, what=qn)
        if step == 4:
            # after the wrap-around every column has been overwritten once more: untouched columns must be intact
            for qn in ("queue_v_cross_ng", "queue_frame_cross_ng"):
                q = S[qn][:32].float().cpu().numpy()
                assert np.isfinite(q).all()
                close(np.linalg.norm(S[qn].float().cpu().numpy(), axis=0), 1.0, 1e-4, what=f"{qn} column norms")
    # the model keeps a host copy of queue_ptr (no device read per step); a checkpoint load must invalidate it
    S = {k: v.clone() for k, v in model.state_dict().items()}
    S["queue_ptr"] = torch.tensor([2 * B], dtype=torch.long)
    model.load_state_dict(S)
    vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(B, Fr, tag="moco.s0")]
    model._mlm_draws = None
    before = model.queue_v_cross_ng.clone()
    model(vid, vf, tg, gm, ti, tm, 6)
    assert int(model.queue_ptr) == (3 * B) % K
    changed = (model.queue_v_cross_ng != before).any(dim=0).nonzero().view(-1).tolist()
    assert changed == list(range(2 * B, 3 * B)), changed


def test_pretrain_gradients_vs_oracle():
    """Every parameter gradient of step 0 against the fp32 oracle: direction (cosine) and norm."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    g = golden("moco_fp32")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    raw = synth.pretrain_state(synth.TINY, K, Fr)
    cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt")
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=raw, task_config=cfg).to(DEV).train()
    batch = synth.pretrain_batch(B, Fr, tag="moco.s0")
    draws = [torch.from_numpy(g[f"mlm_{n}0"]) for n in ("masked", "replaced", "randsel", "words")]
    model._mlm_draws = draws
    loss = model(*[t.to(DEV) for t in batch], 1)
    loss.backward()
    sd = {}
    for k, v in raw.items():
        trainable = v.is_floating_point() and not any(s in k for s in ("_k.", "queue_", "running_", "num_batches"))
        sd[k] = v.clone().requires_grad_(trainable)
    sd["cls.decoder.bias"] = sd["cls.bias"]
    queues = {k: sd[k] for k in sd if k.startswith("queue_") and k != "queue_ptr"}
    d2 = [d.bool() if i < 3 else d for i, d in enumerate(draws)]
    ref, parts, _ = O.pretrain_loss(batch, sd, queues, 0, K, mode="fp32", mlm_draws=d2)
    ref.backward()
    close(loss, ref, 5e-2, what="loss")
    bad = []
    for n, p in model.named_parameters():
        if not p.requires_grad or "t_projector" in n:
            continue
        key = "cls.bias" if n == "cls.decoder.bias" else n
        assert p.grad is not None, n
        a, b = p.grad.float().cpu().flatten(), sd[key].grad.flatten()
        if float(b.norm()) < 1e-6:        # biases in front of BatchNorm: the true gradient is zero, both sides hold noise
            assert float(a.norm()) < 1e-4, n
            continue
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20))
        ratio = float(a.norm() / (b.norm() + 1e-20))
        if cos < 0.97 or abs(ratio - 1) > 0.1:
            bad.append((n, round(cos, 4), round(ratio, 3), float(b.norm())))
    assert not bad, f"{len(bad)} gradients disagree with the oracle: {bad[:12]}"


@pytest.mark.parametrize("kind", ["finetune", "pretrain"])
def test_resume_from_checkpoint_is_bit_identical(kind):
    """model + optimizer state saved after 3 steps (torch.save of the state_dicts, loaded with
    weights_only=True into freshly built objects), 3 more steps: every tensor - weights, momentum encoders, queues,
    queue_ptr, BatchNorm statistics, Adam moments - equals the uninterrupted run bit for bit."""
    import io
    from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_

    def make():
        if kind == "finetune":
            cfg = task_config(max_frames=4)
            m = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY), task_config=cfg)
            batch = [t.to(DEV) for t in synth.finetune_batch(8, 4, 32, synth.TINY.image_res, tag="resume")]
        else:
            cfg = task_config(max_frames=4, dataset="chvtt", contrast_num_negative=16, lr=2e-3, text_lr=1e-3, coef_lr=0.5)
            m = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(synth.TINY, 16, 4), task_config=cfg)
            batch = [t.to(DEV) for t in synth.pretrain_batch(4, 4, res=synth.TINY.image_res, tag="resume")]
        m = m.to(DEV).train()
        return m, prep_optimizer(m, cfg, 50), batch

    def steps(m, opt, batch, lo, hi):
        params = [p for p in m.parameters() if p.requires_grad]
        for i in range(lo, hi):
            torch.manual_seed(100 + i)                  # the MLM mask draws
            m(*batch, i + 1).backward()
            clip_grad_norm_(params, 1.0)
            opt.step()
            opt.zero_grad()

    m, opt, batch = make()
    steps(m, opt, batch, 0, 3)
    buf = io.BytesIO()
    torch.save({"model": m.state_dict(), "opt": opt.state_dict()}, buf)
    steps(m, opt, batch, 3, 6)
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    m2, opt2, batch2 = make()
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    m2.load_state_dict(ck["model"])
    opt2.load_state_dict(ck["opt"])
    steps(m2, opt2, batch2, 3, 6)
    bad = [k for k, v in m2.state_dict().items() if not torch.equal(v, ref[k])]
    assert not bad, f"{len(bad)} tensors differ after the resume: {bad[:8]}"


def test_module_members_vs_reference_golden():
    """MLP.forward (train mode twice, then eval mode), BertLMPredictionHead.forward and the differentiable loose_similarity of
    the pre-training model as callables of their own (reference modules/modeling.py:788-807,207-229, modules/module_cross.py:
    308-322) against what the reference's modules returned on the same seeded inputs (tests/golden/modules_fp32.npz)."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    g = golden("modules_fp32")
    cfg = task_config(contrast_num_negative=16, max_frames=4, dataset="chvtt")
    sd = synth.pretrain_state(synth.TINY, 16, 4)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=sd, task_config=cfg).to(DEV).train()
    mlp = model.v_projector
    x = synth.normal("modules.mlp.x", (24, 512)).to(DEV).requires_grad_()
    w = synth.normal("modules.mlp.w", (24, 512)).to(DEV)
    y = mlp(x)
    (y * w).sum().backward()
    close(y, g["mlp_y"], 5e-5, 1e-4, "mlp y")
    close(x.grad, g["mlp_dx"], 5e-6, 1e-3, "mlp dx")
    close(mlp.linear_hidden[1].weight.grad[:8, :16], g["mlp_dw1"], 5e-5, 1e-3, "dw1")
    close(mlp.linear_hidden[2].weight.grad[:64], g["mlp_dgamma"], 5e-5, 1e-3, "dgamma")
    close(mlp.linear_hidden[2].bias.grad[:64], g["mlp_dbeta"], 5e-5, 1e-3, "dbeta")
    close(mlp.linear_out.weight.grad[:8, :16], g["mlp_dw2"], 5e-5, 1e-3, "dw2")
    close(mlp.linear_out.bias.grad[:64], g["mlp_db2"], 5e-5, 1e-3, "db2")
    close(mlp.linear_hidden[2].running_mean[:64], g["mlp_running_mean1"], 1e-6, 1e-4, "running mean")
    close(mlp.linear_hidden[2].running_var[:64], g["mlp_running_var1"], 1e-6, 1e-4, "running var")
    with torch.no_grad():
        y3 = mlp(synth.normal("modules.mlp.x3", (3, 8, 512)).to(DEV))            # leading dims are kept
    close(y3, g["mlp_y3"], 5e-5, 1e-4, "mlp y3")
    close(mlp.linear_hidden[2].running_var[:64], g["mlp_running_var2"], 1e-6, 1e-4, "running var 2")
    assert int(mlp.linear_hidden[2].num_batches_tracked) == 2
    mlp.eval()
    with torch.no_grad():
        ye = mlp(synth.normal("modules.mlp.xe", (10, 512)).to(DEV))
    close(ye, g["mlp_y_eval"], 5e-5, 1e-4, "mlp eval")
    with pytest.raises(RuntimeError):
        mlp(x)                                                                   # eval mode is forward-only
    mlp.train()
    # MLM head
    h = synth.normal("modules.lm.h", (3, 7, 512)).to(DEV).requires_grad_()
    wl = synth.normal("modules.lm.w", (3, 7, 64)).to(DEV)
    logits = model.cls(h)
    assert logits.shape == (3, 7, 49408)
    (logits[..., :64] * wl).sum().backward()
    close(logits[..., :128], g["lm_logits_head"], 5e-5, 1e-4, "lm logits")
    close(logits.sum(-1), g["lm_logits_rowsum"], 2e-2, 1e-4, "lm logits row sums")
    close(h.grad, g["lm_dh"], 5e-6, 1e-3, "lm dh")
    close(model.cls.transform.dense.weight.grad[:8, :16], g["lm_ddense"], 5e-6, 1e-3, "lm ddense")
    close(model.cls.transform.LayerNorm.weight.grad[:64], g["lm_dln"], 5e-6, 1e-3, "lm dln")
    close(model.cls.decoder.weight.grad[:8, :16], g["lm_ddec"], 5e-6, 1e-3, "lm ddec")
    close(model.cls.bias.grad[:128], g["lm_dbias"], 5e-6, 1e-3, "lm dbias")
    # loose_similarity: x100 logits within 1e-3 (north_star), gradients to both operands
    q = synth.normal("modules.sim.q", (6, 512)).to(DEV).requires_grad_()
    v = synth.normal("modules.sim.v", (5, 512)).to(DEV).requires_grad_()
    u = synth.normal("modules.sim.u", (5, 3, 512)).to(DEV).requires_grad_()
    ws, wu = synth.normal("modules.sim.ws", (6, 5)).to(DEV), synth.normal("modules.sim.wu", (6, 5, 3)).to(DEV)
    s2, s3 = model.loose_similarity(q, v), model.loose_similarity(q, u)
    ((s2 * ws).sum() + (s3 * wu).sum()).backward()
    close(s2, g["sim2"], 1e-3, what="sim2")
    close(s3, g["sim3"], 1e-3, what="sim3")
    close(q.grad, g["sim_dq"], 5e-5, 1e-3, "sim dq")
    close(v.grad, g["sim_dv"], 5e-5, 1e-3, "sim dv")
    close(u.grad, g["sim_du"], 5e-5, 1e-3, "sim du")


def test_pretrain_as_written_at_true_vit_b32_dims_vs_reference_golden():
    """The same two steps at true ViT-B/32 dimensions in the as-written (fp16 tower) regime, against the reference's as-written
    run inside the reference's own envelope: every loss part of step 0 within 1.5 x |as-written - fp32| (or 2e-2), the queue
    columns the first enqueue wrote within 2e-3, step 1 finite and within 0.25 of the reference's."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_
    ga, gf = golden("moco_b32_aswritten"), golden("moco_b32_fp32")
    K, B, Fr = int(ga["K"]), int(ga["B"]), int(ga["F"])
    cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt", lr=2e-3, text_lr=1e-3, coef_lr=0.5, weight_decay=0.05)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(synth.VIT_B32, K, Fr), task_config=cfg)
    model = model.to(DEV).train()
    opt = prep_optimizer(model, cfg, 10)
    for step in range(2):
        vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(B, Fr, tag=f"moco_b32.s{step}")]
        model._mlm_draws = [torch.from_numpy(ga[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
        loss = model(vid, vf, tg, gm, ti, tm, step + 1)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        parts = dict(zip(("fam", "vtm", "ftm", "mlm"), [float(x.detach()) for x in model.last_losses]))
        for nm in ("fam", "ftm", "mlm"):
            ref, env = float(ga[f"{nm}{step}"]), 1.5 * abs(float(ga[f"{nm}{step}"]) - float(gf[f"{nm}{step}"]))
            print(f"step {step} {nm}: {parts[nm]:.5f} (as-written reference {ref:.5f}, its fp32 regime {float(gf[f'{nm}{step}']):.5f})")
            assert abs(parts[nm] - ref) <= (max(env, 2e-2) if step == 0 else 0.25), (step, nm, parts[nm], ref)
        assert int(model.queue_ptr) == int(ga[f"ptr{step}"][0])
        if step == 0:
            S = model.state_dict()
            for key in ga.files:
                if key.startswith("q0:"):
                    close(S[key[3:]][:32, :64], ga[key], 2e-3, what=key)
    ops.raise_on_device_errors()
