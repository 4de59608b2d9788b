"""The BASELINE configs at their FULL per-GPU sizes (round-4 review, missing #2): until round 5 the largest model-level parity
problems were B = 8 (enc_b32x8) and kernel tests stopped at 16 384 - 40 000 rows; the 153 600-token paths (600 M-tiles on the
persistent 256 x 256 grid, grouped weight gradients at split 7, the fold at 153 600 rows, ~50 GiB of slabs) were reached by
bench.py alone, whose only check is a finite loss.  One test per config, true dimensions:

  config 2  ViT-B/32 fine-tune, B = 256, F = 12 (153 600 frame tokens)            reference modules/modeling.py:682-722
  config 5  ViT-B/16 fine-tune, one rank's share b = 16, F = 24 (75 648 tokens)   same, 197-token attention path
  config 4  ViT-B/32 pre-train, B = 128, F = 12, K = 1 024, title 45 / tag 25     reference modules/modeling.py:334-436

The reference cannot run these sizes here in any useful time, so the statements are the size-independent ones:
  (i)   the towers are per-sample: the first videos / captions of the big batch ARE the reference's golden problem
        (enc_b32x8 / enc_b16 / moco_b32), and their features must sit inside the same envelope against the reference's goldens
        as they do at B = 8 - batch size may change tile paths, not values.  The features are taken from the TRAINING forward
        (hooks), i.e. from the folded training path the default step runs at these sizes (advisor, round 4: its parity was
        pinned by no reference-held fixture);
  (ii)  the loss equals the oracle's head evaluated on the CPU from the GPU's own features (1e-3);
  (iii) a second run is bit-identical (loss and every gradient);
  (iv)  the same batch run as 2 x B/2 through the towers and joined before the head - the arithmetic of two data-parallel
        ranks behind `_AllGatherCat` - gives the same loss and the same global gradient norm within fp16 rounding.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden  # noqa: E402
from hmmc_amd import ops, synth  # noqa: E402
from hmmc_amd import functional as Fn  # noqa: E402
from oracle import hmmc_oracle as O  # noqa: E402
from test_gpu_model import DEV, ENVELOPE, build, task_config  # noqa: E402


def _big_finetune_batch(B, Fr, L, res, small_tag, bs, fs, seed):
    """[B, Fr] batch whose first `bs` captions and the first `fs` frames of the first `bs` videos are the golden problem's."""
    ids_s, mask_s, vid_s, _, _ = synth.finetune_batch(bs, fs, L, res, tag=small_tag)
    ids, mask = synth.token_ids(f"full.{small_tag}.ids", B, L)
    ids[:bs], mask[:bs] = ids_s, mask_s
    g = torch.Generator(device=DEV).manual_seed(seed)
    vid = torch.randn((B, Fr, 3, res, res), generator=g, device=DEV)
    vid[:bs, :fs] = vid_s.to(DEV)
    return ids.to(DEV), mask.to(DEV), vid, torch.full((B,), Fr, dtype=torch.long, device=DEV), torch.arange(B, device=DEV)


def _grads(model):
    return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}


def _total_norm(grads):
    return float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))


@pytest.mark.parametrize("name,dims,B,Fr,gname", [("config2", synth.VIT_B32, 256, 12, "enc_b32x8"), ("config5", synth.VIT_B16, 16, 24, "enc_b16")])
def test_finetune_step_at_full_size(name, dims, B, Fr, gname):
    ga, gf = golden(f"{gname}_aswritten"), golden(f"{gname}_fp32")
    bs, fs, L = int(ga["B"]), int(ga["F"]), int(ga["L"])
    model, sd = build(dims, max_frames=Fr, top_frames=3)
    batch = _big_finetune_batch(B, Fr, L, dims.image_res, gname, bs, fs, seed=99)
    T = B * Fr * model.visual_encoder.visual.tokens
    assert Fn.fold_train_enabled(True, T, dims.vision_width, model.visual_encoder.visual.tokens), "the default training fold must be in force here"
    cap = {}
    hooks = [model.text_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("q", o.detach())),
             model.visual_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("vu", (o[0].detach(), o[1].detach())))]

    def run():
        model.zero_grad(set_to_none=True)
        loss = model(*batch, 1)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), _grads(model)

    loss1, g1 = run()
    q, (v, u) = cap["q"].clone(), [t.clone() for t in cap["vu"]]
    for h in hooks:
        h.remove()
    assert q.shape == (B, 512) and u.shape == (B, Fr, 512) and torch.isfinite(loss1)
    # (i) the golden rows inside the envelope they are held to at the golden's own batch size
    fails = []
    for key, mine in (("text_feat", q[:bs]), ("frame_output", u[:bs, :fs])):
        mine = mine.float().cpu().numpy()
        own_max, own_l2 = float(np.abs(ga[key] - gf[key]).max()), float(np.linalg.norm(ga[key] - gf[key]) / np.linalg.norm(gf[key]))
        my_max, my_l2 = float(np.abs(mine - ga[key]).max()), float(np.linalg.norm(mine - ga[key]) / np.linalg.norm(ga[key]))
        print(f"{name} {key}: max |HIP - as-written| {my_max:.3e} (reference regimes {own_max:.3e}), rel-L2 {my_l2:.3e} ({own_l2:.3e})")
        if my_max > ENVELOPE * own_max:
            fails.append(f"{key}: max abs {my_max:.3e} > {ENVELOPE} x {own_max:.3e}")
        if my_l2 > ENVELOPE * own_l2:
            fails.append(f"{key}: rel-L2 {my_l2:.3e} > {ENVELOPE} x {own_l2:.3e}")
    assert not fails, fails
    # (ii) the head: the oracle on the CPU from the GPU's own features
    ref = O.finetune_head(q.float().cpu(), v.float().cpu(), u.float().cpu(), model.weight_VTM_finetune, model.weight_FTM_finetune)
    print(f"{name} loss {float(loss1):.6f}, oracle head on the same features {float(ref):.6f}")
    assert abs(float(loss1) - float(ref)) <= 1e-3 * max(1.0, abs(float(ref)))
    # (iii) bit-identical from run to run
    loss2, g2 = run()
    assert torch.equal(loss1, loss2), (float(loss1), float(loss2))
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, f"{len(bad)} gradients differ between two runs: {bad[:5]}"
    n1 = _total_norm(g1)
    del g2
    # (iv) two halves through the towers, joined before the head (two ranks' arithmetic behind _AllGatherCat)
    ids, mask, vid, vf, idx = batch
    model.zero_grad(set_to_none=True)
    parts = []
    for sl in (slice(0, B // 2), slice(B // 2, B)):
        qh = model.text_encoder(ids[sl], mask[sl])
        vh, uh = model.visual_encoder(vid[sl], vf[sl])
        parts.append((qh, vh, uh))
    qc, vc, uc = [torch.cat([p[i] for p in parts], 0) for i in range(3)]
    scale = min(float(np.exp(float(model.text_encoder.logit_scale))), 100.0)
    loss_h = Fn.FinetuneHeadFn.apply(qc, vc, uc, model.weight_VTM_finetune, model.weight_FTM_finetune, scale)
    loss_h.backward()
    torch.cuda.synchronize()
    gh = _grads(model)
    nh = _total_norm(gh)
    fdiff = max(float((qc.detach() - q).abs().max()), float((uc.detach() - u).abs().max()), float((vc.detach() - v).abs().max()))
    worst = max(float((gh[n].double() - g1[n].double()).norm() / (g1[n].double().norm() + 1e-30)) for n in g1)
    print(f"{name} 2 x B/2: loss {float(loss_h):.6f} vs {float(loss1):.6f}; global grad norm {nh:.6e} vs {n1:.6e}; max feature diff {fdiff:.2e}; "
          f"worst per-tensor gradient rel-L2 {worst:.2e}")
    assert abs(float(loss_h) - float(loss1)) <= 1e-4 * max(1.0, abs(float(loss1)))
    assert abs(nh / n1 - 1) <= HALVES_NORM_TOL, (nh, n1)
    ops.raise_on_device_errors()


# | norm(2 x B/2) / norm(B) - 1 | of the global gradient norm: the features of the two runs are bit-identical or differ by one
# fp32 ulp (the towers are per-sample), only the order of the fp16-rounded partial sums of the weight gradients differs.
# Measured (round 5): 3e-6 at config 2 (164.1492 vs 164.1487), 3e-7 at config 5; per tensor the worst rel-L2 is 4e-2 (a bias whose
# gradient is a difference of large fp16 sums), which is why the statement is made on the global norm.
HALVES_NORM_TOL = 1e-4


def test_pretrain_step_at_full_size():
    """Config 4: B = 128, F = 12, K = 1 024, title 45 / tag 25 at true ViT-B/32 dims.  The first 4 videos' first 2 frames, titles
    and tags are `moco_b32`'s (the reference's fixture at B = 4, F = 2): their per-sample KEY features - what step 0 enqueues
    for them: title, tag and frame keys through the momentum towers - must equal the columns the reference enqueued (2e-3, the
    tolerance of the B = 4 test); each loss part must equal the oracle's restatement (modules/modeling.py:286-332) evaluated
    on the CPU from the GPU's own features and the old queues (1e-3); queue_ptr advances by B; a second run from the same
    state is bit-identical."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    ga = golden("moco_b32_aswritten")
    K, bs, fs = int(ga["K"]), int(ga["B"]), int(ga["F"])
    B, Fr = 128, 12
    cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt")
    state = synth.pretrain_state(synth.VIT_B32, K, Fr)
    vid_s, _, tg_s, gm_s, ti_s, tm_s = synth.pretrain_batch(bs, fs, tag="moco_b32.s0")
    vid, vf, tg, gm, ti, tm = synth.pretrain_batch(B, 1, tag="full.moco")             # ids for everyone; frames below
    tg[:bs], gm[:bs], ti[:bs], tm[:bs] = tg_s, gm_s, ti_s, tm_s
    g = torch.Generator(device=DEV).manual_seed(7)
    vid = torch.randn((B, Fr, 3, 224, 224), generator=g, device=DEV)
    vid[:bs, :fs] = vid_s.to(DEV)
    vf = torch.full((B,), Fr, dtype=torch.long)
    gdraw = torch.Generator().manual_seed(3)
    draws = [torch.bernoulli(torch.full(ti.shape, 0.15), generator=gdraw).long(), torch.bernoulli(torch.full(ti.shape, 0.8), generator=gdraw).long(),
             torch.bernoulli(torch.full(ti.shape, 0.5), generator=gdraw).long(), torch.randint(49408, ti.shape, generator=gdraw)]
    batch = [t.to(DEV) for t in (vid, vf, tg, gm, ti, tm)]

    def run():
        model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict={k: x.clone() for k, x in state.items()}, task_config=cfg)
        model = model.to(DEV).train()
        model._mlm_draws = [d.clone() for d in draws]
        cap = {}
        hk = [model.visual_encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("on", (o[0].detach(), o[1].detach()))),
              model.visual_encoder_k.register_forward_hook(lambda m, i, o: cap.__setitem__("key", (o[0].detach(), o[1].detach()))),
              model.v_predictor.register_forward_hook(lambda m, i, o: cap.__setitem__("pred", o.detach())),
              model.v_projector_k.register_forward_hook(lambda m, i, o: cap.__setitem__("proj_k", o.detach()))]
        for enc, nm in ((model.text_encoder, "text"), (model.text_encoder_k, "text_k")):
            orig = enc.encode_many

            def wrapped(lists, kinds, _o=orig, _n=nm):
                out = _o(lists, kinds)
                cap[_n] = [t.detach() for t in out]
                return out
            enc.encode_many = wrapped
        old_q = {k: getattr(model, k).detach().clone() for k in ("queue_frame_proj_ng", "queue_frame_cross_ng", "queue_title_cross_ng", "queue_v_cross_ng")}
        loss = model(*batch, 1)
        loss.backward()
        torch.cuda.synchronize()
        for h in hk:
            h.remove()
        return model, loss.detach().clone(), [float(x.detach()) for x in model.last_losses], cap, old_q, _grads(model)

    model, loss1, parts, cap, old_q, g1 = run()
    assert int(model.queue_ptr) == B % K
    S = model.state_dict()
    # (i) the golden samples' keys, as enqueued at columns 0.. (queue_ptr was 0): first 32 dims, the reference's columns
    close_cols = []
    for key, cols in (("queue_title_cross_ng", list(range(bs))), ("queue_tag_cross_ng", list(range(bs))),
                      ("queue_frame_cross_ng", [b * Fr + f for b in range(bs) for f in range(fs)])):
        mine = S[key][:32, cols].float().cpu().numpy()
        refc = ga["q0:" + key][:, :len(cols)]
        err = float(np.abs(mine - refc).max())
        close_cols.append((key, err))
        assert err <= 2e-3, (key, err)
    print("config4 enqueued keys of the golden samples vs the reference's columns:", close_cols)
    # (ii) every loss part from the GPU's own features, on the CPU
    v_fea, frame_fea = [t.float().cpu() for t in cap["on"]]
    v_k, frame_k = [t.float().cpu() for t in cap["key"]]
    title_fea = cap["text"][0].float().cpu()
    title_k = cap["text_k"][1].float().cpu()
    frame_pred = cap["pred"].float().cpu().view(B, Fr, -1)
    frame_proj_k = cap["proj_k"].float().cpu().view(B, Fr, -1)
    oq = {k: x.float().cpu() for k, x in old_q.items()}
    fam = O.frame_self_loss(frame_pred, frame_proj_k, oq["queue_frame_proj_ng"])
    vtm = O.contrastive_loss(v_fea, title_k, oq["queue_title_cross_ng"]) + O.contrastive_loss(title_fea, v_k, oq["queue_v_cross_ng"])
    ftm = O.frame_cross_loss(frame_fea, frame_k, oq["queue_frame_cross_ng"], title_fea, title_k, oq["queue_title_cross_ng"])
    for nm, mine, ref in (("FAM", parts[0], float(fam)), ("VTM", parts[1], float(vtm)), ("FTM", parts[2], float(ftm))):
        print(f"config4 {nm}: {mine:.6f}, oracle on the same features {ref:.6f}")
        assert abs(mine - ref) <= 1e-3 * max(1.0, abs(ref)), (nm, mine, ref)
    assert np.isfinite(parts[3]) and 5.0 < parts[3] < 13.0            # MLM over 49 408 classes at random init: ~ln(49408) = 10.8
    # (iii) bit-identical second run from the same state
    del model
    model2, loss2, parts2, _, _, g2 = run()
    assert torch.equal(loss1, loss2) and parts == parts2, (float(loss1), float(loss2), parts, parts2)
    bad = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    assert not bad, f"{len(bad)} gradients differ between two runs: {bad[:5]}"
    ops.raise_on_device_errors()
