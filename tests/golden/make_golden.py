#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself on CPU.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  The reference is imported unmodified; the harness below
only (a) provides empty stand-ins for third-party *Python packages that are
not installed and are not on the english hot path* (boto3/botocore: S3 cache,
ftfy: tokenizer text cleanup) and maps `diffdist.functional.all_gather`
(un-vendored, requirements.txt:3) onto torch.distributed.nn.functional.all_gather,
(b) replaces the two loaders that need the network (CLIP.get_config download,
AutoConfig hub fetch) by synthetic CLIP-shaped weights / a local config.json,
(c) patches torch for CPU: Tensor.cuda -> identity (modeling.py:311),
SyncBatchNorm conversion -> no-op (identical numerics at world size 1).

Outputs are plain arrays (.npz) — inputs/weights are regenerated from seeds by
hmmc_amd/synth.py on both sides and never stored.

    python tests/golden/make_golden.py [--only NAME]
"""
import argparse
import importlib.machinery
import json
import os
import sys
import tempfile
import types
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from hmmc_amd import synth  # noqa: E402

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import torch.distributed as dist
    import torch.distributed.nn.functional as dnf
    _stub("boto3")
    b = _stub("botocore")
    b.exceptions = _stub("botocore.exceptions", ClientError=Exception)
    _stub("ftfy", fix_text=lambda s: s)
    f = _stub("diffdist.functional", all_gather=lambda out_list, x: list(dnf.all_gather(x)))
    _stub("diffdist", functional=f)
    sys.path.insert(0, REF)
    if not dist.is_initialized():
        store = tempfile.mktemp(prefix="hmmc_golden_store_")
        dist.init_process_group("gloo", init_method=f"file://{store}", rank=0, world_size=1)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.SyncBatchNorm.convert_sync_batchnorm = classmethod(lambda cls, module, process_group=None: module)
    import modules.module_clip as mclip
    import modules.modeling as mmodel
    import modules.optimization as mopt
    import metrics as mmetrics
    return mclip, mmodel, mopt, mmetrics


def task_config(**kw):
    d = dict(local_rank=0, rank=0, use_temp=True, language="english", top_frames=2, max_frames=4, n_display=100000,
             logdir=None, use_frame_fea=True, dataset="msrvtt", contrast_momentum=0.99, contrast_temperature=0.07,
             contrast_num_negative=16, pretrained_text=None, lr=1e-4, text_lr=3e-5, coef_lr=1e-3, weight_decay=0.2,
             warmup_proportion=0.1)
    d.update(kw)
    return Namespace(**d)


def build_reference_model(mclip, cls, dims, sd, mode, **tc):
    """Construct the reference model with synthetic weights and load the full synthetic state."""
    clip_sd = synth.clip_state_from(sd, dims)
    mclip.CLIP.get_config = staticmethod(lambda pretrained_clip_name="ViT-B/32": {k: v.clone() for k, v in clip_sd.items()})
    cfg = task_config(**tc)
    if cfg.pretrained_text is None:
        d = tempfile.mkdtemp(prefix="hmmc_textcfg_")
        with open(os.path.join(d, "config.json"), "w") as fh:
            json.dump({"model_type": "bert", "hidden_act": "gelu"}, fh)
        cfg.pretrained_text = d
    model = cls.from_pretrained("cross-base", cache_dir=tempfile.mkdtemp(), state_dict=None, task_config=cfg)
    missing, unexpected = model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("t_projector" in k for k in missing) or not missing, missing
    if mode == "fp32":
        model.float()
        model.text_encoder.dtype = torch.float32
        if hasattr(model, "text_encoder_k"):
            model.text_encoder_k.dtype = torch.float32
    model.train()
    return model, cfg


def grad_norms(model):
    return {n: float(p.grad.float().norm()) for n, p in model.named_parameters() if p.grad is not None}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().float().numpy() if v.is_floating_point() else v.detach().numpy()
        out[k] = v
    out["_versions"] = np.array(f"torch {torch.__version__} numpy {np.__version__}")
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


# ----------------------------------------------------------------------------- fixtures

def fx_head(mclip, mmodel, mopt, mmetrics):
    dims = synth.TINY
    sd = synth.finetune_state(dims)
    model, _ = build_reference_model(mclip, mmodel.BirdModel, dims, sd, "fp32")
    for tag, B, Fr in (("head_ft_small", 48, 5), ("head_ft_c2", 256, 12)):
        q = synth.normal(f"{tag}.q", (B, 512)).requires_grad_()
        v = synth.normal(f"{tag}.v", (B, 512)).requires_grad_()
        u = synth.normal(f"{tag}.u", (B, Fr, 512)).requires_grad_()
        fl = model.frame_loss(q, u)
        s = model.loose_similarity(q, v)
        sl = model.loss_fct(s) + model.loss_fct(s.T)
        loss = model.weight_FTM_finetune * fl + model.weight_VTM_finetune * sl
        loss.backward()
        s0 = model.loose_similarity(q, u[:, 0, :])
        if B <= 64:
            save(tag, B=B, F=Fr, loss=loss, frame_loss=fl, sim_loss=sl, S_video=s, S_frame0=s0,
                 dQ=q.grad, dV=v.grad, dU=u.grad)
        else:
            save(tag, B=B, F=Fr, loss=loss, frame_loss=fl, sim_loss=sl, S_video_rows=s[:8], S_frame0_rows=s0[:8],
                 dQ_rows=q.grad[:8], dV_rows=v.grad[:8], dU_rows=u.grad[:4], dQ_norm=q.grad.norm(), dV_norm=v.grad.norm(),
                 dU_norm=u.grad.norm())
    # eval scorer + metrics (main_task_retrieval.py:332-336, metrics.py:12-39)
    q = synth.normal("head_eval.q", (48, 512))
    v = synth.normal("head_eval.v", (48, 512))
    u = synth.normal("head_eval.u", (48, 12, 512))
    # make it a retrieval problem with structure: text i is close to video i
    q = q + 0.7 * v
    out = {}
    with torch.no_grad():
        sv = model.loose_similarity(q, v)
        sf = model.loose_similarity(q, u)
        out["S_video"], out["S_frame"] = sv, sf
        for k in (1, 2, 3, 12):
            fk = torch.topk(sf, k=k, dim=2)[0].mean(dim=2)
            out[f"topk{k}"] = fk
            mt = mmetrics.compute_metrics((sv + fk).numpy())
            out[f"metrics{k}"] = np.array([mt["R1"], mt["R5"], mt["R10"], mt["MR"], mt["MeanR"]])
        mt = mmetrics.compute_metrics(sv.numpy())
        out["metrics_video"] = np.array([mt["R1"], mt["R5"], mt["R10"], mt["MR"], mt["MeanR"]])
        mt = mmetrics.compute_metrics(sv.numpy().T)
        out["metrics_video_v2t"] = np.array([mt["R1"], mt["R5"], mt["R10"], mt["MR"], mt["MeanR"]])
    save("head_eval", **out)


def _enc_fixture(mclip, mmodel, name, dims, B, Fr, L, modes, use_temp=True):
    sd = synth.finetune_state(dims, use_temp=use_temp)
    ids, mask, vid, vf, idx = synth.finetune_batch(B, Fr, L, dims.image_res, tag=name)
    for mode in modes:
        model, _ = build_reference_model(mclip, mmodel.BirdModel, dims, sd, mode, use_temp=use_temp, max_frames=Fr)
        q = model.text_encoder(ids, mask)
        v, u = model.visual_encoder(vid, vf)
        loss = model(ids, mask, vid, vf, idx, 1)
        loss.backward()
        gn = grad_norms(model)
        names = sorted(gn)
        P = dict(model.named_parameters())
        sl = {}
        for key, slc in (("text_encoder.text_projection", (slice(0, 4), slice(0, 8))),
                         ("visual_encoder.visual.conv1.weight", (slice(0, 2), 0, slice(0, 4), slice(0, 4))),
                         ("visual_encoder.visual.class_embedding", (slice(0, 8),)),
                         ("visual_encoder.visual.positional_embedding", (slice(0, 3), slice(0, 8))),
                         ("visual_encoder.visual.ln_pre.weight", (slice(0, 8),)),
                         ("visual_encoder.visual.transformer.resblocks.0.attn.in_proj_weight", (slice(0, 4), slice(0, 8))),
                         ("visual_encoder.visual.transformer.resblocks.0.mlp.c_fc.bias", (slice(0, 8),)),
                         ("text_encoder.transformer.resblocks.1.attn.out_proj.weight", (slice(0, 4), slice(0, 8))),
                         ("text_encoder.positional_embedding", (slice(0, 3), slice(0, 8))),
                         ("visual_encoder.visual.proj", (slice(0, 4), slice(0, 8)))):
            if key in P and P[key].grad is not None:
                sl["g:" + key] = P[key].grad[slc]
        if use_temp:
            k = "visual_encoder.temporal_transformer.resblocks.0.mlp.c_fc.weight"
            sl["g:" + k] = P[k].grad[:4, :8]
            k = "visual_encoder.frame_position_embeddings.weight"
            sl["g:" + k] = P[k].grad[:3, :8]
        tg = P["text_encoder.token_embedding.weight"].grad
        sl["g:text_encoder.token_embedding.weight[SOT]"] = tg[synth.SOT, :8]
        sl["g:text_encoder.token_embedding.weight[EOT]"] = tg[synth.EOT, :8]
        save(f"{name}_{mode}", dims=json.dumps(dims.to_dict()), B=B, F=Fr, L=L, text_feat=q, video_emb=v, frame_output=u,
             loss=loss, grad_norm_names=np.array(names), grad_norm_values=np.array([gn[n] for n in names]), **sl)


def fx_enc_tiny(mclip, mmodel, mopt, mmetrics):
    _enc_fixture(mclip, mmodel, "enc_tiny", synth.TINY, 4, 4, 32, ("fp32", "aswritten"))
    _enc_fixture(mclip, mmodel, "enc_tiny_notemp", synth.TINY, 3, 2, 20, ("fp32",), use_temp=False)


def fx_enc_rank(mclip, mmodel, mopt, mmetrics):
    """A batch large enough for retrieval ranks to differ: 32 captions x 32 videos of 4 frames through the reference's
    encoders, its similarity (modeling.py:207-229) and its eval score S_video + mean top-k frame logits
    (main_task_retrieval.py:332-336), with the rank metrics of metrics.py:12-39."""
    dims, B, Fr, L, k = synth.TINY, 32, 4, 32, 2
    sd = synth.finetune_state(dims)
    ids, mask, vid, vf, idx = synth.finetune_batch(B, Fr, L, dims.image_res, tag="enc_rank")
    for mode in ("fp32", "aswritten"):
        model, _ = build_reference_model(mclip, mmodel.BirdModel, dims, sd, mode, max_frames=Fr)
        with torch.no_grad():
            q = model.text_encoder(ids, mask)
            v, u = model.visual_encoder(vid, vf)
            sv = model.loose_similarity(q, v)
            sf = model.loose_similarity(q, u)
            fk = torch.topk(sf, k=k, dim=2)[0].mean(dim=2)
        loss = model(ids, mask, vid, vf, idx, 1)
        mt = mmetrics.compute_metrics((sv + fk).numpy())
        mv = mmetrics.compute_metrics(sv.numpy())
        save(f"enc_rank_{mode}", dims=json.dumps(dims.to_dict()), B=B, F=Fr, L=L, k=k, text_feat=q, video_emb=v, frame_output=u,
             S_video=sv, S_frame_topk=fk, loss=loss,
             metrics_score=np.array([mt["R1"], mt["R5"], mt["R10"], mt["MR"], mt["MeanR"]]),
             metrics_video=np.array([mv["R1"], mv["R5"], mv["R10"], mv["MR"], mv["MeanR"]]))


def fx_multisent(mclip, mmodel, mopt, mmetrics):
    """Multi-sentence retrieval metrics (MSVD / VATEX style: several captions per video) through the reference's own
    logging_rank (metrics.py:89-144): uneven caption counts; case t repeats case a with exact ties planted in one row."""
    import logging
    out = {}
    for tag, counts, seed in (("a", [3, 1, 9, 2, 5, 1, 4, 7, 2, 6, 1, 8], 7), ("b", [10] * 30, 11), ("c", [1, 2, 3, 4, 20, 1, 1, 6], 13),
                              ("t", [3, 1, 9, 2, 5, 1, 4, 7, 2, 6, 1, 8], 7)):
        n_video, n_sent = len(counts), sum(counts)
        rng = np.random.Generator(np.random.Philox(key=seed))
        sim = rng.normal(size=(n_sent, n_video)).astype(np.float32) * 3.0
        vid = np.repeat(np.arange(n_video), counts)
        sim[np.arange(n_sent), vid] += 2.0                       # the right video tends to win
        if tag == "t":
            sim[5, :] = np.round(sim[5, :])                      # exact ties in one row: the reference's rank is then whatever
                                                                 # position torch.argsort (unstable) gives the ground truth
        cut = list(np.cumsum(counts) - 1)                        # eval_epoch's cut_off_points_ (index of a video's last sentence)
        captured = {}
        orig = mmetrics.compute_metrics

        def spy(x, _orig=orig, _c=captured):
            _c["vt"] = _orig(x)
            return _c["vt"]
        mmetrics.compute_metrics = spy
        try:
            tv = mmetrics.logging_rank(sim.copy(), True, cut, logging.getLogger("golden"))
        finally:
            mmetrics.compute_metrics = orig
        vt = captured["vt"]
        out[f"{tag}.sim"], out[f"{tag}.cut"] = sim, np.asarray(cut)
        out[f"{tag}.tv"] = np.array([tv["R1"], tv["R5"], tv["R10"], tv["MedianR"], tv["MeanR"], tv["Std_Rank"]])
        out[f"{tag}.vt"] = np.array([vt["R1"], vt["R5"], vt["R10"], vt["MR"], vt["MeanR"]])
    save("multisent", **out)


def fx_frame_sampling(mclip, mmodel, mopt, mmetrics):
    """Frame indices the reference's loader draws (dataloaders/dataloader_msrvtt_retrieval.py:296-312, `uniform`,
    `random`, `uniform_random` out of g_lmdb_frames stored frames), recorded by driving MSRVTT_TrainDataLoader._get_rawvideo itself with a fake
    LMDB transaction.  lmdb / cv2 / torchvision are not installed and only decode pixels: empty stand-ins."""
    import random
    _stub("lmdb")
    _stub("cv2", imdecode=lambda buf, flag: np.zeros((2, 2, 3), np.uint8), cvtColor=lambda x, code: x, IMREAD_COLOR=1,
          COLOR_BGR2RGB=4)
    tv = _stub("torchvision")
    tv.datasets = _stub("torchvision.datasets", VisionDataset=object)
    tv.transforms = _stub("torchvision.transforms")
    _stub("dataloaders.rawvideo_util", RawVideoExtractor=object)
    _stub("dataloaders.randaugment", RandomAugment=object)
    import importlib
    if not hasattr(np, "long"):
        np.long = np.int64                     # the reference targets numpy 1.x (dataloader_msrvtt_retrieval.py:297)
    mod = importlib.import_module("dataloaders.dataloader_msrvtt_retrieval")
    out = {}
    for stored in (30, 48):
        mod.g_lmdb_frames = stored
        for policy in ("uniform", "random", "uniform_random"):
            for frames in (7, 12, 24):
                for seed in (0, 1):
                    keys = []

                    class Txn:
                        def get(self, key, _keys=keys):
                            _keys.append(int(key.decode().rsplit("_", 1)[1]))
                            return b"\x00" * 8
                    fake = Namespace(max_frames=frames, frame_sample=policy, _txn=Txn(), resolution=1,
                                     transform=lambda img: np.zeros((3, 1, 1), np.float32))
                    random.seed(seed)
                    mod.MSRVTT_TrainDataLoader._get_rawvideo(fake, ["video0"], frames)
                    out[f"{policy}.{stored}.{frames}.{seed}"] = np.asarray(keys, dtype=np.int32)
    save("frame_sampling", **out)


def fx_enc_tiny16(mclip, mmodel, mopt, mmetrics):
    _enc_fixture(mclip, mmodel, "enc_tiny16", synth.TINY16, 2, 3, 32, ("fp32", "aswritten"))


def fx_enc_b32(mclip, mmodel, mopt, mmetrics):
    _enc_fixture(mclip, mmodel, "enc_b32", synth.VIT_B32, 2, 2, 32, ("fp32", "aswritten"))


def fx_enc_b16(mclip, mmodel, mopt, mmetrics):
    """True ViT-B/16 dimensions (197 tokens per frame, 12 heads: SURVEY config 5's attention path).  Two videos of two
    frames: with one video the 1 x 1 InfoNCE is identically zero and no gradient would be pinned."""
    _enc_fixture(mclip, mmodel, "enc_b16", synth.VIT_B16, 2, 2, 32, ("fp32", "aswritten"))


def fx_enc_b32x8(mclip, mmodel, mopt, mmetrics):
    """True ViT-B/32 dimensions on a batch whose 8 x 8 score matrices can reorder: features, the similarity of
    modeling.py:207-229 and the eval score S_video + mean top-k frame logits (main_task_retrieval.py:332-336) in BOTH
    regimes of the reference, so the GPU test can assert the reference's own fp16-vs-fp32 envelope at real width."""
    dims, B, Fr, L, k = synth.VIT_B32, 8, 4, 32, 2
    sd = synth.finetune_state(dims)
    ids, mask, vid, vf, idx = synth.finetune_batch(B, Fr, L, dims.image_res, tag="enc_b32x8")
    for mode in ("fp32", "aswritten"):
        model, _ = build_reference_model(mclip, mmodel.BirdModel, dims, sd, mode, max_frames=Fr)
        with torch.no_grad():
            q = model.text_encoder(ids, mask)
            v, u = model.visual_encoder(vid, vf)
            sv = model.loose_similarity(q, v)
            sf = model.loose_similarity(q, u)
            fk = torch.topk(sf, k=k, dim=2)[0].mean(dim=2)
            loss = model(ids, mask, vid, vf, idx, 1)
        save(f"enc_b32x8_{mode}", dims=json.dumps(dims.to_dict()), B=B, F=Fr, L=L, k=k, text_feat=q, video_emb=v, frame_output=u,
             S_video=sv, S_frame_topk=fk, loss=loss)


def fx_bertadam(mclip, mmodel, mopt, mmetrics):
    out = {}
    specs = [("a32", (37,), torch.float32, 0.2, 1e-4, 3.0), ("b32", (8, 9), torch.float32, 0.0, 3e-5, 0.01),
             ("c16", (64,), torch.float16, 0.2, 1e-4, 2.0), ("d16", (4, 32), torch.float16, 0.0, 1e-7, 0.05)]
    params, groups = [], []
    for name, shape, dt, wd, lr, gscale in specs:
        p = torch.nn.Parameter(synth.normal(f"bertadam.{name}.p", shape, 0.5).to(dt))
        params.append(p)
        groups.append({"params": [p], "weight_decay": wd, "lr": lr})
    opt = mopt.BertAdam(groups, lr=1e-4, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=20,
                        weight_decay=0.2, max_grad_norm=1.0)
    for step in range(5):
        for (name, shape, dt, wd, lr, gscale), p in zip(specs, params):
            p.grad = synth.normal(f"bertadam.{name}.g{step}", shape, gscale).to(dt)
        opt.step()
        out[f"lr{step}"] = np.array(opt.get_lr())
        for (name, *_), p in zip(specs, params):
            st = opt.state[p]
            out[f"{name}.p{step}"], out[f"{name}.m{step}"], out[f"{name}.v{step}"] = p.data.clone(), st["next_m"].clone(), st["next_v"].clone()
            out[f"{name}.g{step}"] = p.grad.clone()
    save("bertadam", **out)


def _prep_optimizer(mopt, model, cfg, t_total):
    """main_task_retrieval.py:171-205 grouping, verbatim semantics."""
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    dec = [(n, p) for n, p in named if not any(nd in n for nd in no_decay)]
    nod = [(n, p) for n, p in named if any(nd in n for nd in no_decay)]
    wd = cfg.weight_decay
    lrc = cfg.lr * cfg.coef_lr
    groups = [
        {"params": [p for n, p in dec if "visual_encoder.visual." in n], "weight_decay": wd, "lr": lrc},
        {"params": [p for n, p in dec if "text_encoder." in n], "weight_decay": wd, "lr": cfg.text_lr},
        {"params": [p for n, p in dec if "visual_encoder.visual." not in n and "text_encoder." not in n], "weight_decay": wd},
        {"params": [p for n, p in nod if "visual_encoder.visual." in n], "weight_decay": 0.0, "lr": lrc},
        {"params": [p for n, p in nod if "text_encoder." in n], "weight_decay": 0.0, "lr": cfg.text_lr},
        {"params": [p for n, p in nod if "visual_encoder.visual." not in n and "text_encoder." not in n], "weight_decay": 0.0},
    ]
    return mopt.BertAdam(groups, lr=cfg.lr, warmup=cfg.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98,
                         e=1e-6, t_total=t_total, weight_decay=wd, max_grad_norm=1.0)


SAMPLED = ["text_encoder.text_projection", "visual_encoder.visual.conv1.weight", "visual_encoder.visual.class_embedding",
           "visual_encoder.visual.transformer.resblocks.1.mlp.c_proj.weight",
           "visual_encoder.visual.transformer.resblocks.0.ln_1.weight",
           "visual_encoder.temporal_transformer.resblocks.2.attn.in_proj_weight",
           "text_encoder.transformer.resblocks.0.attn.in_proj_bias", "text_encoder.ln_final.bias"]


def fx_train_ft(mclip, mmodel, mopt, mmetrics):
    dims = synth.TINY
    sd = synth.finetune_state(dims)
    for mode in ("fp32", "aswritten"):
        # larger lr than the defaults so 3 steps move the sampled weights measurably
        model, cfg = build_reference_model(mclip, mmodel.BirdModel, dims, sd, mode, lr=2e-3, text_lr=1e-3, coef_lr=0.5)
        opt = _prep_optimizer(mopt, model, cfg, t_total=10)
        out = {}
        P = dict(model.named_parameters())
        for step in range(4):
            ids, mask, vid, vf, idx = synth.finetune_batch(4, 4, 32, tag=f"train_ft.s{step}")
            loss = model(ids, mask, vid, vf, idx, step + 1)
            loss.backward()
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            out[f"loss{step}"], out[f"gnorm{step}"] = loss.detach().clone(), tn.detach().clone()
            for k in SAMPLED:
                out[f"p{step}:{k}"] = P[k].data.reshape(-1)[:16].clone()
        save(f"train_ft_{mode}", **out)


class _Recorder:
    """Record torch.bernoulli / torch.randint draws made inside get_mlm_loss."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._b, self._r = torch.bernoulli, torch.randint

        def bern(*a, **k):
            o = self._b(*a, **k)
            self.draws.append(o.clone())
            return o

        def rint(*a, **k):
            o = self._r(*a, **k)
            self.draws.append(o.clone())
            return o

        torch.bernoulli, torch.randint = bern, rint
        return self

    def __exit__(self, *exc):
        torch.bernoulli, torch.randint = self._b, self._r


def fx_moco(mclip, mmodel, mopt, mmetrics):
    _moco_fixture(mclip, mmodel, mopt, "moco", synth.TINY, K=16, B=4, Fr=4, steps=5)


def fx_moco_b32(mclip, mmodel, mopt, mmetrics):
    """BASELINE config 4's model at TRUE ViT-B/32 dimensions (four CLIP towers, K = 1 024 negatives, title 45 / tag 25), B = 4,
    F = 2, two optimizer steps, both regimes (round 4: the pre-training step had only been pinned at width 128)."""
    _moco_fixture(mclip, mmodel, mopt, "moco_b32", synth.VIT_B32, K=1024, B=4, Fr=2, steps=2, queue_steps=(0, 1))


def _moco_fixture(mclip, mmodel, mopt, name, dims, K, B, Fr, steps, queue_steps=(0, 3, 4)):
    sd = synth.pretrain_state(dims, K, Fr)
    for mode in ("fp32", "aswritten"):
        model, cfg = build_reference_model(mclip, mmodel.BirdPreTrainedModel, dims, sd, mode, contrast_num_negative=K,
                                           max_frames=Fr, dataset="chvtt", lr=2e-3, text_lr=1e-3, coef_lr=0.5,
                                           weight_decay=0.05)
        opt = _prep_optimizer(mopt, model, cfg, t_total=10)
        parts = {}
        for nm in ("frame_self_loss", "frame_cross_loss", "calculate_mlm_loss"):
            orig = getattr(model, nm)

            def wrap(*a, _o=orig, _n=nm, **k):
                r = _o(*a, **k)
                parts[_n] = r.detach().clone()
                return r
            setattr(model, nm, wrap)
        out = {"K": K, "B": B, "F": Fr}
        torch.manual_seed(1234)
        for step in range(steps):
            batch = synth.pretrain_batch(B, Fr, res=dims.image_res, tag=f"{name}.s{step}")
            vid, vf, tg, gm, ti, tm = batch
            with _Recorder() as rec:
                loss = model(vid, vf, tg, gm, ti, tm, step + 1)
            assert len(rec.draws) == 4, len(rec.draws)
            out[f"mlm_masked{step}"], out[f"mlm_replaced{step}"], out[f"mlm_randsel{step}"], out[f"mlm_words{step}"] = \
                [d.numpy().astype(np.int64) for d in rec.draws]
            loss.backward()
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            out[f"loss{step}"], out[f"gnorm{step}"] = loss.detach().clone(), tn.detach().clone()
            out[f"fam{step}"], out[f"ftm{step}"], out[f"mlm{step}"] = parts["frame_self_loss"], parts["frame_cross_loss"], \
                parts["calculate_mlm_loss"]
            out[f"ptr{step}"] = model.queue_ptr.clone()
            S = model.state_dict()
            for k in ("visual_encoder_k.visual.conv1.weight", "text_encoder_k.text_projection",
                      "visual_encoder_k.temporal_transformer.resblocks.0.mlp.c_fc.weight",
                      "v_projector_k.linear_out.weight", "text_encoder_k.ln_final.weight",
                      "v_projector.linear_hidden.2.running_mean", "v_projector.linear_hidden.2.running_var",
                      "v_projector_k.linear_hidden.2.running_mean", "v_predictor.linear_hidden.2.running_var"):
                out[f"s{step}:{k}"] = S[k].reshape(-1)[:16].clone()
            if step in queue_steps:
                for qn in ("queue_v_cross_ng", "queue_title_cross_ng", "queue_tag_cross_ng", "queue_frame_proj_ng",
                           "queue_frame_cross_ng"):
                    out[f"q{step}:{qn}"] = S[qn][:32, :64].clone()   # first 32 of 512 feature rows, the columns written so far
        save(f"{name}_{mode}", **out)


def fx_modules(mclip, mmodel, mopt, mmetrics):
    """The callable members of the pre-training model on their own (round 4): MLP.forward in train and eval mode
    (modules/modeling.py:788-807), BertLMPredictionHead.forward (modules/module_cross.py:308-322) and loose_similarity with
    gradients (modules/modeling.py:207-229), fp32 regime, seeded inputs."""
    dims = synth.TINY
    K, Fr = 16, 4
    sd = synth.pretrain_state(dims, K, Fr)
    model, cfg = build_reference_model(mclip, mmodel.BirdPreTrainedModel, dims, sd, "fp32", contrast_num_negative=K, max_frames=Fr,
                                       dataset="chvtt")
    out = {}
    # MLP: two train-mode calls (batch statistics, running statistics updated), then eval mode on other rows
    mlp = model.v_projector
    x = synth.normal("modules.mlp.x", (24, 512)).requires_grad_()
    w = synth.normal("modules.mlp.w", (24, 512))
    y = mlp(x)
    (y * w).sum().backward()
    out["mlp_y"], out["mlp_dx"] = y, x.grad
    out["mlp_dw1"] = mlp.linear_hidden[1].weight.grad[:8, :16]
    out["mlp_dgamma"], out["mlp_dbeta"] = mlp.linear_hidden[2].weight.grad[:64], mlp.linear_hidden[2].bias.grad[:64]
    out["mlp_dw2"], out["mlp_db2"] = mlp.linear_out.weight.grad[:8, :16], mlp.linear_out.bias.grad[:64]
    out["mlp_running_mean1"], out["mlp_running_var1"] = mlp.linear_hidden[2].running_mean[:64].clone(), mlp.linear_hidden[2].running_var[:64].clone()
    x3 = synth.normal("modules.mlp.x3", (3, 8, 512))
    with torch.no_grad():
        out["mlp_y3"] = mlp(x3.reshape(-1, 512)).view(3, 8, 512)
    out["mlp_running_mean2"], out["mlp_running_var2"] = mlp.linear_hidden[2].running_mean[:64].clone(), mlp.linear_hidden[2].running_var[:64].clone()
    mlp.eval()
    xe = synth.normal("modules.mlp.xe", (10, 512))
    with torch.no_grad():
        out["mlp_y_eval"] = mlp(xe)
    mlp.train()
    # MLM head logits
    h = synth.normal("modules.lm.h", (3, 7, 512)).requires_grad_()
    wl = synth.normal("modules.lm.w", (3, 7, 64))
    logits = model.cls(h)
    (logits[..., :64] * wl).sum().backward()
    out["lm_logits_head"], out["lm_logits_rowsum"] = logits[..., :128], logits.sum(-1)
    out["lm_dh"] = h.grad
    out["lm_ddense"] = model.cls.transform.dense.weight.grad[:8, :16]
    out["lm_dln"] = model.cls.transform.LayerNorm.weight.grad[:64]
    out["lm_ddec"] = model.cls.decoder.weight.grad[:8, :16]
    out["lm_dbias"] = model.cls.bias.grad[:128]
    # loose_similarity, differentiable, 2-D and 3-D candidates
    q = synth.normal("modules.sim.q", (6, 512)).requires_grad_()
    v = synth.normal("modules.sim.v", (5, 512)).requires_grad_()
    u = synth.normal("modules.sim.u", (5, 3, 512)).requires_grad_()
    ws, wu = synth.normal("modules.sim.ws", (6, 5)), synth.normal("modules.sim.wu", (6, 5, 3))
    s2, s3 = model.loose_similarity(q, v), model.loose_similarity(q, u)
    ((s2 * ws).sum() + (s3 * wu).sum()).backward()
    out.update(sim2=s2, sim3=s3, sim_dq=q.grad, sim_dv=v.grad, sim_du=u.grad)
    save("modules_fp32", **out)


def fx_manifest(mclip, mmodel, mopt, mmetrics):
    """state_dict layout (key -> shape, dtype) of the reference models as written (fp16 CLIP weights): what a
    pytorch_model.bin.N checkpoint of the reference holds (main_task_retrieval.py:215-222)."""
    out = {}
    for tag, cls, dims, sd, tc in (
            ("finetune_tiny", mmodel.BirdModel, synth.TINY, synth.finetune_state(synth.TINY), {}),
            ("finetune_b32", mmodel.BirdModel, synth.VIT_B32, synth.finetune_state(synth.VIT_B32), {"max_frames": 12}),
            ("pretrain_tiny", mmodel.BirdPreTrainedModel, synth.TINY, synth.pretrain_state(synth.TINY, 16, 4),
             {"dataset": "chvtt", "contrast_num_negative": 16})):
        model, _ = build_reference_model(mclip, cls, dims, sd, "aswritten", **tc)
        out[tag] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
        out[tag + ":trainable"] = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    path = os.path.join(HERE, "statedict_manifest.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path)/1024:.1f} KiB)")


FIXTURES = {"manifest": fx_manifest, "head": fx_head, "enc_tiny": fx_enc_tiny, "enc_rank": fx_enc_rank, "multisent": fx_multisent, "frame_sampling": fx_frame_sampling, "enc_tiny16": fx_enc_tiny16, "enc_b32": fx_enc_b32,
            "enc_b16": fx_enc_b16, "enc_b32x8": fx_enc_b32x8,
            "bertadam": fx_bertadam, "train_ft": fx_train_ft, "moco": fx_moco, "moco_b32": fx_moco_b32, "modules": fx_modules}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.set_num_threads(8)
    mods = import_reference()
    for name, fn in FIXTURES.items():
        if args.only and name != args.only:
            continue
        print(f"== {name}")
        fn(*mods)


if __name__ == "__main__":
    main()
