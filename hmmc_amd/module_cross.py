"""Encoders with the reference's names and forward signatures (reference: modules/module_cross.py:47-108
CrossConfig, :110-149 temporal blocks, :152-237 VisualEncoder, :240-305 TextEncoder, :308-357 MLM head)."""
from __future__ import annotations

import copy
import json
import logging
import math
import os
from collections import OrderedDict

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .module_clip import CLIP, build_model
from .until_module import LayerNorm

logger = logging.getLogger(__name__)

CONFIG_NAME = "cross_config.json"


class CrossConfig(object):
    """Holds cross-base/cross_config.json (reference modules/module_cross.py:47-108, until_config.py:41-99)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as fh:
            return cls(**json.load(fh))

    @classmethod
    def get_config(cls, pretrained_model_name, cache_dir=None, type_vocab_size=2, state_dict=None, task_config=None):
        here = os.path.dirname(os.path.abspath(__file__))
        cand = os.path.join(here, pretrained_model_name)
        path = cand if os.path.exists(cand) else pretrained_model_name
        cfg_file = os.path.join(path, CONFIG_NAME) if os.path.isdir(path) else path
        if not os.path.exists(cfg_file):
            if task_config is None or getattr(task_config, "local_rank", 0) == 0:
                logger.error("Model name '%s' was not found (looked for %s)", pretrained_model_name, cfg_file)
            return None
        config = cls.from_json_file(cfg_file)
        config.type_vocab_size = type_vocab_size
        return config, state_dict

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def __repr__(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True)


class QuickGELU(nn.Module):
    def forward(self, x):
        raise RuntimeError("QuickGELU is fused into hmmc_gemm_f32's epilogue; call the enclosing Transformer")


class ResidualAttentionBlock(nn.Module):
    """fp32 temporal block container, TF-style LN eps 1e-12 (reference modules/module_cross.py:114-139)."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.n_head = n_head


class Transformer(nn.Module):
    """Temporal transformer container (reference modules/module_cross.py:141-149); executed inside
    functional.TemporalFn together with the position add, the residual and the pooling."""

    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])

    def flat_params(self):
        out = []
        for blk in self.resblocks:
            out += Fn.block_params(blk)
        return out


def _clip_for(task_config, cross_config):
    name = getattr(task_config, "pretrained_clip_name", None) or cross_config.pretrained_clip_name
    sd = getattr(cross_config, "_clip_state_dict", None)
    if sd is None:
        sd = CLIP.get_config(pretrained_clip_name=name)
    return sd, build_model(sd, local_rank=getattr(task_config, "local_rank", 0))


class VisualEncoder(nn.Module):
    """video [b,F,3,H,W] -> (video_emb [b,512], frame_output [b,F,512]) fp32
    (reference modules/module_cross.py:152-237)."""

    def __init__(self, task_config, cross_config):
        super().__init__()
        _, clip = _clip_for(task_config, cross_config)
        self.use_temp = task_config.use_temp
        self.is_vit = True
        self.visual = copy.deepcopy(clip.visual)
        if self.use_temp:
            self.temporal_transformer = Transformer(width=cross_config.temporal_hidden_size,
                                                    layers=cross_config.temporal_hidden_layers,
                                                    heads=cross_config.temporal_attention_heads)
            self.frame_position_embeddings = nn.Embedding(cross_config.max_position_embeddings,
                                                          cross_config.temporal_hidden_size)

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def forward(self, video, video_frames=None, frame_index=None):
        """frame_index (int32 [bs, frames] from hmmc_amd.sampling.batch_frame_index): `video` then holds every STORED frame,
        uint8 [bs, stored, 3, H, W], and the sampled ones are read in place by the patch kernel."""
        bs, frames, channel, h, w = video.shape
        video = video.reshape(bs * frames, channel, h, w)
        if frame_index is not None:
            frames = frame_index.shape[1]
            frame_index = frame_index.reshape(-1).contiguous()
        frame_output = self.encode_image(video, video_frame=frames, frame_index=frame_index).view(bs, frames, -1)
        if self.use_temp:
            if frames > self.frame_position_embeddings.weight.shape[0]:
                raise ValueError("more frames than max_position_embeddings")
            visual_output = Fn.temporal(frame_output, self.temporal_transformer.heads,
                                                self.frame_position_embeddings.weight,
                                                *self.temporal_transformer.flat_params())
        else:
            visual_output = Fn.temporal(frame_output, 0, None)
        return visual_output, frame_output

    def encode_image(self, image, return_hidden=False, video_frame=-1, frame_index=None):
        """ln_post(hidden) @ proj, CLS row, .float() (reference modules/module_cross.py:222-237).  Only the CLS
        rows are normalised and projected unless return_hidden asks for all tokens."""
        n = image.shape[0] if frame_index is None else frame_index.numel()
        L = self.visual.tokens
        # without return_hidden only the class-token row of the last block is consumed: its per-token half runs on that row alone
        tokens = self.visual.hidden_tokens(image, lead_only=not return_hidden, frame_index=frame_index)
        v = self.visual
        if return_hidden:
            hidden = Fn.LnProjFn.apply(tokens, None, v.ln_post.weight, v.ln_post.bias, v.proj).view(n, L, -1)
            return hidden[:, 0, :], hidden
        idx = torch.arange(n, device=tokens.device, dtype=torch.int32) * L
        # the fp16 tower ran lead_only: its backward reads the gradient of `tokens` at the class rows alone
        return Fn.LnProjFn.apply(tokens, idx, v.ln_post.weight, v.ln_post.bias, v.proj, tokens.dtype == torch.float16)


class TextEncoder(nn.Module):
    """CLIP text transformer (english branch; reference modules/module_cross.py:240-305)."""

    def __init__(self, task_config, cross_config):
        super().__init__()
        self.language = task_config.language
        if self.language != "english":
            raise NotImplementedError("only the CLIP (english) text transformer is on the HMMC hot path")
        clip_state_dict, clip = _clip_for(task_config, cross_config)
        self.logit_scale = copy.deepcopy(clip_state_dict["logit_scale"]).float()    # plain tensor attribute, ln(100)
        self.token_embedding = copy.deepcopy(clip.token_embedding)
        self.positional_embedding = copy.deepcopy(clip.positional_embedding)
        self.transformer = copy.deepcopy(clip.transformer)
        self.ln_final = copy.deepcopy(clip.ln_final)
        self.text_projection = copy.deepcopy(clip.text_projection)
        self.dtype = clip.visual.conv1.weight.dtype

    def forward(self, input_ids, attention_mask=None, return_hidden=False):
        bs_pair = input_ids.size(0)
        text_output, hidden = self.encode_text(input_ids, return_hidden=True, _want=("hidden" if return_hidden else "feat"))
        if return_hidden:
            return hidden.view(bs_pair, -1, hidden.size(-1))
        return text_output.view(bs_pair, text_output.size(-1))

    def encode_text(self, text, return_hidden=False, _want="both"):
        b, L = text.shape
        if L > self.positional_embedding.shape[0]:
            raise ValueError("sequence longer than the positional embedding table")
        # activations in self.dtype, a STORED attribute as in the reference (module_cross.py:256,288): model.float() alone leaves
        # it at fp16 and the text path then fails on mixed dtypes; the fp32 regime sets text_encoder.dtype = torch.float32 too
        x = Fn.TextEmbedFn.apply(text, self.token_embedding.weight, self.positional_embedding, self.dtype)
        x = self.transformer(x, b, L)
        feat = hidden = None
        if _want in ("both", "hidden"):
            hidden = Fn.LnProjFn.apply(x, None, self.ln_final.weight, self.ln_final.bias, self.text_projection).view(b, L, -1)
        if _want in ("both", "feat"):
            idx = ops.eot_index(text.contiguous())                                                  # EOT = largest id
            feat = Fn.LnProjFn.apply(x, idx, self.ln_final.weight, self.ln_final.bias, self.text_projection)
        if return_hidden:
            return feat, hidden
        return feat


def _encode_many(self, texts, wants):
    """Several id tensors [b_i, L_i] through ONE pass of the text tower (padded with id 0 to the longest; the tower is
    causal and its rows independent, so every sequence's feature / hidden states are those of its own pass): one set of
    ~600 small launches instead of one per input, and ONE gradient per parameter where separate passes make autograd
    add two (the pre-training step runs the online title pass and the MLM pass of the masked titles on the same
    weights, modules/modeling.py:347-352,160-169, and two momentum passes, :369-373).
    wants[i]: "feat" -> [b_i, 512] (EOT row), "hidden" -> [b_i, L_i, 512]."""
    Lmax = max(t.shape[1] for t in texts)
    if Lmax > self.positional_embedding.shape[0]:
        raise ValueError("sequence longer than the positional embedding table")
    ids = torch.cat([t if t.shape[1] == Lmax else torch.nn.functional.pad(t, (0, Lmax - t.shape[1])) for t in texts], dim=0)
    x = Fn.TextEmbedFn.apply(ids, self.token_embedding.weight, self.positional_embedding, self.dtype)
    x = self.transformer(x, ids.shape[0], Lmax)
    outs, row0 = [], 0
    for t, want in zip(texts, wants):
        b, L = t.shape
        if want == "feat":
            idx = ops.eot_index(t.contiguous(), base=row0 * Lmax, stride=Lmax)   # EOT = largest id
            outs.append(Fn.LnProjFn.apply(x, idx, self.ln_final.weight, self.ln_final.bias, self.text_projection))
        elif L == Lmax:
            rows = x[row0 * Lmax:(row0 + b) * Lmax]
            outs.append(Fn.LnProjFn.apply(rows, None, self.ln_final.weight, self.ln_final.bias, self.text_projection).view(b, L, -1))
        else:
            seq = torch.arange(row0, row0 + b, device=t.device)
            idx = (seq[:, None] * Lmax + torch.arange(L, device=t.device)[None, :]).reshape(-1).to(torch.int32)
            outs.append(Fn.LnProjFn.apply(x, idx, self.ln_final.weight, self.ln_final.bias, self.text_projection).view(b, L, -1))
        row0 += b
    return outs


TextEncoder.encode_many = _encode_many


class BertLayerNorm(LayerNorm):
    pass


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, hidden_size, hidden_act="gelu"):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.hidden_act = hidden_act
        self.LayerNorm = BertLayerNorm(hidden_size, eps=1e-12)


class BertLMPredictionHead(nn.Module):
    """MLM head container (reference modules/module_cross.py:308-357): dense, erf-GELU, LN(1e-12), decoder to vocab."""

    def __init__(self, hidden_size, vocab_size, hidden_act="gelu"):
        super().__init__()
        self.transform = BertPredictionHeadTransform(hidden_size, hidden_act)
        self.decoder = nn.Linear(hidden_size, vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab_size))
        self.decoder.bias = self.bias

    def forward(self, hidden_states):
        """[..., hidden] -> [..., vocab] fp32 logits, differentiable (reference modules/module_cross.py:319-322).  The
        pre-training loss does not call this: calculate_mlm_loss evaluates head + cross-entropy on the labelled rows only."""
        t = self.transform
        return Fn.LmLogitsFn.apply(hidden_states.float(), t.dense.weight, t.dense.bias, t.LayerNorm.weight, t.LayerNorm.bias,
                                   self.decoder.weight, self.bias)
