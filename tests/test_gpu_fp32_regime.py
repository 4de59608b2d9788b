"""The reference's fp32-upcast regime on the HIP path: `model.float(); model.text_encoder.dtype = torch.float32`
(how the reference itself is run fully upcast: CLIP(...).float() at modules/module_clip.py:566-577 followed by .float()
on the built model; the `fp32` half of tests/golden/*.npz was generated exactly so, tests/golden/make_golden.py).

In this regime every stage is exact fp32 (exact-f32 MFMA GEMMs, fp32 LayerNorm / attention / embeddings), so the
north-star tolerances become HARD assertions through the whole model, not only on the heads:
features 2e-4, x100 logits 1e-3, identical argsort, losses of consecutive optimizer steps 1e-3 ... 2e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden  # noqa: E402
from hmmc_amd import synth  # noqa: E402
from test_gpu_model import DEV, SAMPLED, close, prep_optimizer, task_config  # noqa: E402


def to_fp32(model):
    """the reference's recipe (make_golden.py:build_reference_model, mode 'fp32')"""
    model.float()
    model.text_encoder.dtype = torch.float32
    if hasattr(model, "text_encoder_k"):
        model.text_encoder_k.dtype = torch.float32
    return model


def build(dims, use_temp=True, cls=None, sd=None, **tc):
    from hmmc_amd.modeling import BirdModel
    sd = sd if sd is not None else synth.finetune_state(dims, use_temp=use_temp)
    model = (cls or BirdModel).from_pretrained("cross-base", state_dict=sd, task_config=task_config(use_temp=use_temp, **tc))
    return to_fp32(model).to(DEV).train()


def test_model_float_alone_fails_like_the_reference():
    """TextEncoder.dtype is a stored attribute, so model.float() without it feeds fp16 activations to fp32
    weights and the reference raises; so does this path (no silent cast)."""
    from hmmc_amd.modeling import BirdModel
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY), task_config=task_config())
    model.float().to(DEV).train()
    ids, mask, vid, vf, idx = [t.to(DEV) for t in synth.finetune_batch(2, 2, 32, tag="mixed")]
    with pytest.raises(RuntimeError, match="same dtype"):
        model.text_encoder(ids, mask)
    v, u = model.visual_encoder(vid, vf)                      # the visual side follows its weights, as the reference's does
    assert v.dtype == torch.float32 and bool(torch.isfinite(v).all())


ENC = [("enc_tiny", synth.TINY, True), ("enc_tiny_notemp", synth.TINY, False), ("enc_tiny16", synth.TINY16, True),
       ("enc_b32", synth.VIT_B32, True), ("enc_b16", synth.VIT_B16, True)]


@pytest.mark.parametrize("name,dims,use_temp", ENC)
def test_fp32_model_vs_reference_golden(name, dims, use_temp):
    """Features, loss, every per-parameter gradient norm and the stored gradient slices against the reference's fp32 run."""
    g = golden(f"{name}_fp32")
    B, Fr, L = int(g["B"]), int(g["F"]), int(g["L"])
    model = build(dims, use_temp)
    ids, mask, vid, vf, idx = [t.to(DEV) for t in synth.finetune_batch(B, Fr, L, dims.image_res, tag=name)]
    q = model.text_encoder(ids, mask)
    v, u = model.visual_encoder(vid, vf)
    loss = model(ids, mask, vid, vf, idx, 1)
    loss.backward()
    assert q.dtype == v.dtype == u.dtype == torch.float32
    close(q, g["text_feat"], 2e-4, 2e-4, "text_feat")
    close(u, g["frame_output"], 2e-4, 2e-4, "frame_output")
    close(v, g["video_emb"], 2e-4, 2e-4, "video_emb")
    close(loss, g["loss"], 1e-4, what="loss")
    names = [str(n) for n in g["grad_norm_names"]]
    ref = dict(zip(names, g["grad_norm_values"]))
    P = dict(model.named_parameters())
    for n in names:
        assert P[n].grad is not None and P[n].grad.dtype == torch.float32, n
        gn = float(P[n].grad.norm())
        assert abs(gn - ref[n]) <= 2e-3 * max(ref[n], 1e-6) + 1e-7, f"grad norm {n}: {gn} vs {ref[n]}"
    for key in g.files:
        if key.startswith("g:") and "[" not in key:
            full = P[key[2:]].grad
            got = full[0:2, 0, 0:4, 0:4] if key.endswith("conv1.weight") else full[tuple(slice(0, s) for s in g[key].shape)]
            close(got, g[key], 1e-5, 2e-3, key)
    tg = P["text_encoder.token_embedding.weight"].grad
    close(tg[synth.SOT, :8], g["g:text_encoder.token_embedding.weight[SOT]"], 1e-6, 2e-3, "tok SOT")
    close(tg[synth.EOT, :8], g["g:text_encoder.token_embedding.weight[EOT]"], 1e-6, 2e-3, "tok EOT")


@pytest.mark.parametrize("name,dims", [("enc_b32x8", synth.VIT_B32), ("enc_rank", synth.TINY)])
def test_fp32_logits_1e3_and_identical_ranks(name, dims):
    """north_star's acceptance through the TOWERS: x100 logits within 1e-3 of the reference and identical retrieval ranks
    (argsort of every row of the video-text matrix and of the eval score S_video + mean top-k frame logits,
    main_task_retrieval.py:332-336) - the 8 x 8 problem at true ViT-B/32 dims and the 32 x 32 problem on the small model."""
    from hmmc_amd import metrics as M
    g = golden(f"{name}_fp32")
    B, Fr, L, k = int(g["B"]), int(g["F"]), int(g["L"]), int(g["k"])
    model = build(dims, max_frames=Fr, top_frames=k)
    ids, mask, vid, vf, idx = [t.to(DEV) for t in synth.finetune_batch(B, Fr, L, dims.image_res, tag=name)]
    with torch.no_grad():
        q = model.text_encoder(ids, mask)
        v, u = model.visual_encoder(vid, vf)
        sv, fk = model.eval_scores(q, v, u, top_frames=k)
        loss = model(ids, mask, vid, vf, idx, 1)
    close(q, g["text_feat"], 2e-4, 2e-4, "text_feat")
    close(u, g["frame_output"], 2e-4, 2e-4, "frame_output")
    close(v, g["video_emb"], 2e-4, 2e-4, "video_emb")
    close(sv, g["S_video"], 1e-3, what="S_video (x100 logits, 1e-3)")
    close(fk, g["S_frame_topk"], 1e-3, what="mean top-k frame logits (1e-3)")
    close(loss, g["loss"], 1e-4, what="loss")
    sv, fk = sv.cpu().numpy(), fk.cpu().numpy()
    assert np.array_equal(np.argsort(-sv, 1), np.argsort(-g["S_video"], 1)), "video-text ranks differ"
    assert np.array_equal(np.argsort(-(sv + fk), 1), np.argsort(-(g["S_video"] + g["S_frame_topk"]), 1)), "score ranks differ"
    if "metrics_score" in g.files:
        mt = M.metrics_from_ranks(M.ranks(torch.from_numpy(np.ascontiguousarray(sv + fk)).to(DEV)))
        close([mt["R1"], mt["R5"], mt["R10"], mt["MR"], mt["MeanR"]], g["metrics_score"], 1e-9, what="R@K / median / mean rank")


def test_fp32_train_steps_vs_reference_golden():
    """4 full steps (forward, backward, global clip, BertAdam) of main_task_retrieval.py:272-302 in fp32: losses and the global
    gradient norm of EVERY step, and the sampled weights after the updates - hard assertions (the fp16 regime's test can
    only compare directions, test_gpu_model.test_train_steps_vs_reference_golden)."""
    from hmmc_amd.optimization import clip_grad_norm_
    g = golden("train_ft_fp32")
    model = build(synth.TINY, lr=2e-3, text_lr=1e-3, coef_lr=0.5)
    cfg = model.task_config
    opt = prep_optimizer(model, cfg, 10)
    P = dict(model.named_parameters())
    for step in range(4):
        ids, mask, vid, vf, idx = [t.to(DEV) for t in synth.finetune_batch(4, 4, 32, tag=f"train_ft.s{step}")]
        loss = model(ids, mask, vid, vf, idx, step + 1)
        loss.backward()
        tn = clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        close(loss, g[f"loss{step}"], 2e-3, what=f"loss{step}")
        close(tn, g[f"gnorm{step}"], 0, 5e-3, f"gnorm{step}")
        if step < 2:
            for k in SAMPLED:
                close(P[k].data.reshape(-1)[:16], g[f"p{step}:{k}"], 2e-4, 1e-3, f"p{step}:{k}")


def test_fp32_pretrain_steps_vs_reference_golden():
    """5 steps of main_pretrain.py's loop (EMA, five queues with wrap-around, FAM / VTM / FTM / MLM, clip, BertAdam) in fp32:
    every loss part of every step, queue pointer, EMA'd key weights, BN running statistics and queue contents."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_
    g = golden("moco_fp32")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    model = build(synth.TINY, cls=BirdPreTrainedModel, sd=synth.pretrain_state(synth.TINY, K, Fr), contrast_num_negative=K,
                  max_frames=Fr, dataset="chvtt", lr=2e-3, text_lr=1e-3, coef_lr=0.5, weight_decay=0.05)
    opt = prep_optimizer(model, model.task_config, 10)
    for step in range(5):
        vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(B, Fr, tag=f"moco.s{step}")]
        model._mlm_draws = [torch.from_numpy(g[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
        loss = model(vid, vf, tg, gm, ti, tm, step + 1)
        loss.backward()
        tn = clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        fam, vtm, ftm, mlm = [float(x) for x in model.last_losses]
        close(fam, g[f"fam{step}"], 2e-3, what=f"fam{step}")
        close(ftm, g[f"ftm{step}"], 2e-3, what=f"ftm{step}")
        close(mlm, g[f"mlm{step}"], 2e-3, what=f"mlm{step}")
        close(loss, g[f"loss{step}"], 2e-3, what=f"loss{step}")
        assert int(model.queue_ptr) == int(g[f"ptr{step}"][0]), "queue pointer"
        if step == 0:
            close(tn, g["gnorm0"], 0, 5e-3, "gnorm0")
            S = model.state_dict()
            for key in g.files:
                if key.startswith("s0:"):
                    close(S[key[3:]].reshape(-1)[:16], g[key], 1e-5, 1e-4, key)
                if key.startswith("q0:"):
                    close(S[key[3:]][:32], g[key], 1e-5, what=key)


def test_fp32_pretrain_at_true_vit_b32_dims_vs_reference_golden():
    """BASELINE config 4's model at TRUE ViT-B/32 dimensions - four CLIP towers of width 768 / 512, K = 1 024 negatives, title
    45 / tag 25 tokens - B = 4, F = 2, two optimizer steps in fp32 against the reference (tests/golden/moco_b32_fp32.npz): every
    loss part 2e-3, the global gradient norm, EMA'd key weights, BN running statistics, and the queue columns written 1e-5."""
    from hmmc_amd.modeling import BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_
    g = golden("moco_b32_fp32")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    model = build(synth.VIT_B32, cls=BirdPreTrainedModel, sd=synth.pretrain_state(synth.VIT_B32, K, Fr), contrast_num_negative=K,
                  max_frames=Fr, dataset="chvtt", lr=2e-3, text_lr=1e-3, coef_lr=0.5, weight_decay=0.05)
    opt = prep_optimizer(model, model.task_config, 10)
    for step in range(2):
        vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(B, Fr, tag=f"moco_b32.s{step}")]
        model._mlm_draws = [torch.from_numpy(g[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
        loss = model(vid, vf, tg, gm, ti, tm, step + 1)
        loss.backward()
        tn = clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        fam, vtm, ftm, mlm = [float(x.detach()) for x in model.last_losses]
        close(fam, g[f"fam{step}"], 2e-3, what=f"fam{step}")
        close(ftm, g[f"ftm{step}"], 2e-3, what=f"ftm{step}")
        close(mlm, g[f"mlm{step}"], 2e-3, what=f"mlm{step}")
        close(loss, g[f"loss{step}"], 2e-3, what=f"loss{step}")
        close(tn, g[f"gnorm{step}"], 0, 5e-3, f"gnorm{step}")
        assert int(model.queue_ptr) == int(g[f"ptr{step}"][0]) == B * (step + 1), "queue pointer"
        S = model.state_dict()
        for key in g.files:
            if key.startswith(f"s{step}:"):
                close(S[key.split(":", 1)[1]].reshape(-1)[:16], g[key], 2e-5 if step == 0 else 2e-4, 1e-3, key)
            if key.startswith(f"q{step}:"):
                close(S[key.split(":", 1)[1]][:32, :64], g[key], 1e-5 if step == 0 else 2e-4, what=key)
