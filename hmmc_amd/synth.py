"""Deterministic synthetic weights and batches for the HMMC hot path.

There is no network for CLIP checkpoints or MSR-VTT, so every test, golden
fixture and benchmark draws its weights and inputs from here.  Each tensor is
drawn from its own numpy Philox stream keyed by (seed, tensor name), so the
golden-vector generator (which feeds the reference) and the tests (which feed
this repo's HIP path and the oracle) regenerate identical values from a name
and a shape alone; nothing large is ever committed.

Key names and dtypes follow the reference's state_dict
(reference: modules/module_clip.py:271-325,328-416, modules/module_cross.py:152-169,241-256,
modules/modeling.py:88-155,788-807; layout table in SURVEY.md section 8b).
Weight scales follow CLIP.initialize_parameters (modules/module_clip.py:389-416).
CLIP-tower tensors that the reference stores in fp16 (convert_weights,
modules/module_clip.py:506-527) are rounded to fp16-representable values so the
fp32-upcast and as-written regimes share identical parameters.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, asdict

import numpy as np
import torch

SOT, EOT, MASK_ID, VOCAB = 49406, 49407, 49394, 49408  # modules/tokenization_clip.py:75-87


@dataclass(frozen=True)
class Dims:
    """Architecture of one HMMC model (inferred by the reference from tensor shapes,
    modules/module_clip.py:531-553)."""
    vision_width: int = 768
    vision_layers: int = 12
    patch: int = 32
    image_res: int = 224
    embed_dim: int = 512
    context_length: int = 77
    vocab: int = VOCAB
    text_width: int = 512
    text_layers: int = 12
    temporal_layers: int = 4      # cross_config.json:5
    temporal_heads: int = 8       # cross_config.json:4
    max_position_embeddings: int = 48  # cross_config.json:2

    @property
    def grid(self):
        return self.image_res // self.patch

    @property
    def vision_tokens(self):
        return self.grid * self.grid + 1

    def to_dict(self):
        return asdict(self)


VIT_B32 = Dims()
VIT_B16 = Dims(patch=16)
TINY = Dims(vision_width=128, vision_layers=2, patch=32, image_res=224, text_width=128, text_layers=2)
# tiny model with a ragged (non multiple-of-16) token count larger than one 64-row tile
TINY16 = Dims(vision_width=128, vision_layers=2, patch=16, image_res=224, text_width=128, text_layers=2)

NAMED = {"ViT-B/32": VIT_B32, "ViT-B/16": VIT_B16, "tiny": TINY, "tiny16": TINY16}


def _gen(seed: int, name: str) -> np.random.Generator:
    key = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
    return np.random.Generator(np.random.Philox(key=key))


def normal(name, shape, std=1.0, mean=0.0, seed=42, fp16_round=False) -> torch.Tensor:
    a = _gen(seed, name).standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean)
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if fp16_round:
        t = t.half().float()
    return t


def _block(sd, prefix, width, layers, half, seed, attn_std, proj_std, fc_std):
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}."
        sd[p + "attn.in_proj_weight"] = normal(p + "attn.in_proj_weight", (3 * width, width), attn_std, seed=seed, fp16_round=half)
        sd[p + "attn.in_proj_bias"] = normal(p + "attn.in_proj_bias", (3 * width,), 0.02, seed=seed, fp16_round=half)
        sd[p + "attn.out_proj.weight"] = normal(p + "attn.out_proj.weight", (width, width), proj_std, seed=seed, fp16_round=half)
        sd[p + "attn.out_proj.bias"] = normal(p + "attn.out_proj.bias", (width,), 0.02, seed=seed, fp16_round=half)
        sd[p + "ln_1.weight"] = normal(p + "ln_1.weight", (width,), 0.1, 1.0, seed=seed)
        sd[p + "ln_1.bias"] = normal(p + "ln_1.bias", (width,), 0.05, seed=seed)
        sd[p + "mlp.c_fc.weight"] = normal(p + "mlp.c_fc.weight", (4 * width, width), fc_std, seed=seed, fp16_round=half)
        sd[p + "mlp.c_fc.bias"] = normal(p + "mlp.c_fc.bias", (4 * width,), 0.02, seed=seed, fp16_round=half)
        sd[p + "mlp.c_proj.weight"] = normal(p + "mlp.c_proj.weight", (width, 4 * width), proj_std, seed=seed, fp16_round=half)
        sd[p + "mlp.c_proj.bias"] = normal(p + "mlp.c_proj.bias", (width,), 0.02, seed=seed, fp16_round=half)
        sd[p + "ln_2.weight"] = normal(p + "ln_2.weight", (width,), 0.1, 1.0, seed=seed)
        sd[p + "ln_2.bias"] = normal(p + "ln_2.bias", (width,), 0.05, seed=seed)


def visual_encoder_state(dims: Dims, prefix="visual_encoder.", seed=42, use_temp=True):
    """fp32 tensors (fp16-representable where the reference stores fp16)."""
    sd = {}
    w, p = dims.vision_width, dims.patch
    scale = w ** -0.5
    v = prefix + "visual."
    sd[v + "class_embedding"] = normal(v + "class_embedding", (w,), scale, seed=seed)
    sd[v + "positional_embedding"] = normal(v + "positional_embedding", (dims.vision_tokens, w), scale, seed=seed)
    sd[v + "proj"] = normal(v + "proj", (w, dims.embed_dim), scale, seed=seed, fp16_round=True)
    sd[v + "conv1.weight"] = normal(v + "conv1.weight", (w, 3, p, p), (3 * p * p) ** -0.5, seed=seed, fp16_round=True)
    for ln in ("ln_pre", "ln_post"):
        sd[v + ln + ".weight"] = normal(v + ln + ".weight", (w,), 0.1, 1.0, seed=seed)
        sd[v + ln + ".bias"] = normal(v + ln + ".bias", (w,), 0.05, seed=seed)
    proj_std = (w ** -0.5) * ((2 * dims.vision_layers) ** -0.5)
    _block(sd, v + "transformer", w, dims.vision_layers, True, seed, w ** -0.5, proj_std, (2 * w) ** -0.5)
    if use_temp:
        e = dims.embed_dim
        _block(sd, prefix + "temporal_transformer", e, dims.temporal_layers, False, seed,
               e ** -0.5, (e ** -0.5) * ((2 * dims.temporal_layers) ** -0.5), (2 * e) ** -0.5)
        sd[prefix + "frame_position_embeddings.weight"] = normal(
            prefix + "frame_position_embeddings.weight", (dims.max_position_embeddings, e), 0.02, seed=seed)
    return sd


def text_encoder_state(dims: Dims, prefix="text_encoder.", seed=42):
    sd = {}
    w = dims.text_width
    sd[prefix + "token_embedding.weight"] = normal(prefix + "token_embedding.weight", (dims.vocab, w), 0.02, seed=seed)
    sd[prefix + "positional_embedding"] = normal(prefix + "positional_embedding", (dims.context_length, w), 0.01, seed=seed)
    proj_std = (w ** -0.5) * ((2 * dims.text_layers) ** -0.5)
    _block(sd, prefix + "transformer", w, dims.text_layers, True, seed, w ** -0.5, proj_std, (2 * w) ** -0.5)
    sd[prefix + "ln_final.weight"] = normal(prefix + "ln_final.weight", (w,), 0.1, 1.0, seed=seed)
    sd[prefix + "ln_final.bias"] = normal(prefix + "ln_final.bias", (w,), 0.05, seed=seed)
    sd[prefix + "text_projection"] = normal(prefix + "text_projection", (w, dims.embed_dim), w ** -0.5, seed=seed, fp16_round=True)
    return sd


def mlp_state(prefix, seed=42, in_dim=512, inner=4096, out_dim=512):
    """MLP projector / predictor (modules/modeling.py:788-807)."""
    sd = {}
    sd[prefix + "linear_hidden.1.weight"] = normal(prefix + "linear_hidden.1.weight", (inner, in_dim), in_dim ** -0.5, seed=seed)
    sd[prefix + "linear_hidden.1.bias"] = normal(prefix + "linear_hidden.1.bias", (inner,), 0.02, seed=seed)
    sd[prefix + "linear_hidden.2.weight"] = normal(prefix + "linear_hidden.2.weight", (inner,), 0.1, 1.0, seed=seed)
    sd[prefix + "linear_hidden.2.bias"] = normal(prefix + "linear_hidden.2.bias", (inner,), 0.05, seed=seed)
    sd[prefix + "linear_hidden.2.running_mean"] = torch.zeros(inner)
    sd[prefix + "linear_hidden.2.running_var"] = torch.ones(inner)
    sd[prefix + "linear_hidden.2.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    sd[prefix + "linear_out.weight"] = normal(prefix + "linear_out.weight", (out_dim, inner), inner ** -0.5, seed=seed)
    sd[prefix + "linear_out.bias"] = normal(prefix + "linear_out.bias", (out_dim,), 0.02, seed=seed)
    return sd


def finetune_state(dims: Dims, seed=42, use_temp=True):
    """BirdModel state_dict (350 tensors at ViT-B/32)."""
    sd = {}
    sd.update(text_encoder_state(dims, seed=seed))
    sd.update(visual_encoder_state(dims, seed=seed, use_temp=use_temp))
    return sd


def pretrain_state(dims: Dims, K: int, max_frames: int, seed=42):
    """BirdPreTrainedModel state_dict; the *_k copies equal the online weights
    (copy_params, modules/modeling.py:231-236)."""
    sd = {}
    e = dims.embed_dim
    t = text_encoder_state(dims, seed=seed)
    v = visual_encoder_state(dims, seed=seed)
    sd.update(t)
    sd.update({k.replace("text_encoder.", "text_encoder_k.", 1): x.clone() for k, x in t.items()})
    for name in ("t_projector", "v_projector", "v_predictor"):
        m = mlp_state(name + ".", seed=seed)
        sd.update(m)
        if name != "v_predictor":
            sd.update({k.replace(name + ".", name + "_k.", 1): x.clone() for k, x in m.items()})
    # MLM head (modules/module_cross.py:308-357)
    sd["cls.bias"] = normal("cls.bias", (dims.vocab,), 0.02, seed=seed)
    sd["cls.transform.dense.weight"] = normal("cls.transform.dense.weight", (e, e), e ** -0.5, seed=seed)
    sd["cls.transform.dense.bias"] = normal("cls.transform.dense.bias", (e,), 0.02, seed=seed)
    sd["cls.transform.LayerNorm.weight"] = normal("cls.transform.LayerNorm.weight", (e,), 0.1, 1.0, seed=seed)
    sd["cls.transform.LayerNorm.bias"] = normal("cls.transform.LayerNorm.bias", (e,), 0.05, seed=seed)
    sd["cls.decoder.weight"] = normal("cls.decoder.weight", (dims.vocab, e), 0.02, seed=seed)
    sd["cls.decoder.bias"] = sd["cls.bias"]
    sd.update(v)
    sd.update({k.replace("visual_encoder.", "visual_encoder_k.", 1): x.clone() for k, x in v.items()})
    for qn, width in (("queue_v_cross_ng", K), ("queue_frame_proj_ng", K * max_frames),
                      ("queue_frame_cross_ng", K * max_frames), ("queue_title_cross_ng", K),
                      ("queue_tag_cross_ng", K)):
        q = normal(qn, (e, width), 1.0, seed=seed)
        sd[qn] = q / q.norm(dim=0, keepdim=True).clamp_min(1e-12)   # modeling.py:138-149
    sd["queue_ptr"] = torch.zeros(1, dtype=torch.long)
    return sd


def clip_state_from(sd, dims: Dims):
    """CLIP-checkpoint-shaped state dict (what CLIP.get_config returns,
    modules/module_clip.py:418-439) carrying the tower weights of `sd`."""
    out = {}
    for k, x in sd.items():
        if k.startswith("visual_encoder.visual."):
            out["visual." + k[len("visual_encoder.visual."):]] = x.clone()
        elif k.startswith("text_encoder."):
            out[k[len("text_encoder."):]] = x.clone()
    out["logit_scale"] = torch.tensor(float(np.log(100.0)))
    return out


# ----------------------------------------------------------------------------- batches

def video(name, b, f, res=224, seed=42):
    """CLIP-normalised frames are ~N(0,1) (dataloader_msrvtt_retrieval.py:242-247)."""
    return normal(name, (b, f, 3, res, res), 1.0, seed=seed)


def token_ids(name, b, length, seed=42, lo=4, hi=None):
    """SOT + n random BPE ids + EOT + zero pad (dataloader_msrvtt_retrieval.py:263-288)."""
    g = _gen(seed, name)
    hi = min(hi or (length - 4), length - 2)
    ids = np.zeros((b, length), dtype=np.int64)
    for i in range(b):
        n = int(g.integers(lo, hi + 1))
        ids[i, 0] = SOT
        ids[i, 1:1 + n] = g.integers(1, MASK_ID, size=n)
        ids[i, 1 + n] = EOT
    t = torch.from_numpy(ids)
    return t, (t != 0).long()


def finetune_batch(b, f, length=32, res=224, seed=42, tag="ft"):
    """(query_ids, query_mask, video, video_frame, idx) as dataloader_msrvtt_retrieval.py:346."""
    ids, mask = token_ids(f"{tag}.query_ids", b, length, seed=seed)
    vid = video(f"{tag}.video", b, f, res, seed=seed)
    return ids, mask, vid, torch.full((b,), f, dtype=torch.long), torch.arange(b)


def pretrain_batch(b, f, title_len=45, tag_len=25, res=224, seed=42, tag="pt"):
    """(video, video_frame, tag_ids, tag_mask, title_ids, title_mask) as dataloader_bird.py:250."""
    title, tmask = token_ids(f"{tag}.title_ids", b, title_len, seed=seed)
    tg, gmask = token_ids(f"{tag}.tag_ids", b, tag_len, seed=seed)
    vid = video(f"{tag}.video", b, f, res, seed=seed)
    return vid, torch.full((b,), f, dtype=torch.long), tg, gmask, title, tmask
