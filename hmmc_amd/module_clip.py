"""CLIP backbone with the reference's module names and state_dict keys, running on the HIP kernels
(reference: modules/module_clip.py:217-325 LayerNorm/QuickGELU/ResidualAttentionBlock/Transformer/
VisualTransformer, :328-503 CLIP, :506-527 convert_weights, :530-579 build_model).

The torch.nn modules below are parameter containers only: they give the reference's key names
(`...resblocks.3.attn.in_proj_weight`, `...mlp.c_fc.bias`, ...) and dtypes (fp16 for Conv/Linear/
MultiheadAttention/proj, fp32 for LayerNorm and embeddings); every forward goes through
hmmc_amd.functional (one autograd node per tower) and therefore through libhmmc_hip.so.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import torch
from torch import nn

from . import functional as Fn
from . import synth


class LayerNorm(nn.LayerNorm):
    """fp32 LayerNorm over fp16 activations, eps 1e-5 (reference modules/module_clip.py:217-223)."""

    def forward(self, x):
        from . import ops
        shape = x.shape
        y, _, _ = ops.layernorm_fwd(x.contiguous().view(-1, shape[-1]), self.weight, self.bias, self.eps)
        return y.view(shape)


class QuickGELU(nn.Module):
    """x * sigmoid(1.702 x); fused into the c_fc GEMM epilogue (reference modules/module_clip.py:226-228)."""

    def forward(self, x):
        raise RuntimeError("QuickGELU is fused into hmmc_gemm_f16's epilogue; call the enclosing Transformer")


class ResidualAttentionBlock(nn.Module):
    """Parameter container of one block (reference modules/module_clip.py:231-257)."""

    def __init__(self, d_model, n_head, attn_mask=None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask
        self.n_head = n_head


class Transformer(nn.Module):
    """N residual attention blocks executed as ONE autograd node on token-major fp16 activations
    (reference modules/module_clip.py:260-268).  `causal` replaces the additive -inf mask of
    CLIP.build_attention_mask (:441-447)."""

    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.causal = attn_mask is not None
        self.ddp_layers_per_node = 3           # world_size > 1 only: granularity of gradient hand-over to DDP
        self.fold_ln = False                   # no-grad passes with ln_1 / ln_2 folded into the GEMMs (functional.fold_enabled)
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def flat_params(self):
        out = []
        for blk in self.resblocks:
            out += Fn.block_params(blk)
        return out

    def forward(self, x, nseq, L, lead_only=False, x_stat=None):
        """x: [nseq*L, width] fp16 -> same shape.  lead_only: the caller reads only token 0 of every sequence of the result
        (the class token), so the last block's per-token half runs on those rows alone; the other rows are undefined."""
        if x.dtype not in (torch.float16, torch.float32):
            raise TypeError(f"the CLIP towers run in fp16 (as written) or fp32 (after model.float()), not {x.dtype}")
        # One native call per direction on a single GPU.  Under data parallelism the tower is cut into runs of
        # `ddp_layers_per_node` layers, one autograd node each, so that the gradients of the upper layers reach DDP's
        # bucket hooks (and the xGMI all-reduce starts) while the lower layers are still in their backward pass.
        per = self.layers
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            per = max(1, min(self.layers, int(self.ddp_layers_per_node)))
            from . import ops
            ops.reserve_cus_for_collectives()
        blocks = list(self.resblocks)
        train_fold = x.dtype == torch.float16 and Fn.fold_train_enabled(self.fold_ln, x.shape[0], self.width, L)
        for i in range(0, self.layers, per):
            params = []
            for blk in blocks[i:i + per]:
                params += Fn.block_params(blk)
            final = i + per >= self.layers
            # training fold: every layer but the tower's last (which, lead-only, works on 1 / L of the rows anyway)
            ft = ("last_exact" if final else "all") if train_fold and not (final and len(params) == Fn.PER_LAYER) else False
            x = Fn.clip_transformer(x, nseq, L, self.heads, self.causal, bool(lead_only and final), *params,
                                    x_stat=x_stat if i == 0 else None, fold=Fn.fold_enabled(self.fold_ln), fold_train=ft)
        return x


class VisualTransformer(nn.Module):
    """ViT frame encoder (reference modules/module_clip.py:271-325, '2d' patch branch)."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, linear_patch="2d"):
        super().__init__()
        assert linear_patch == "2d", "only the 2d patch branch is on the HMMC hot path"
        self.input_resolution, self.output_dim, self.patch_size = input_resolution, output_dim, patch_size
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.transformer.fold_ln = True
        self.ln_post = LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        self.linear_patch = linear_patch

    @property
    def tokens(self):
        return self.positional_embedding.shape[0]

    def forward(self, x, video_frame=-1):
        """x: [N,3,H,W] -> hidden [N, L, width] fp16 (NLD, as the reference returns)."""
        n = x.shape[0]
        h = self.hidden_tokens(x)
        return h.view(n, self.tokens, -1)

    def hidden_tokens(self, x, lead_only=False, frame_index=None):
        n = x.shape[0] if frame_index is None else frame_index.numel()
        # the kernel casts fp32 pixels to fp16 while patchifying (image.type(fp16)); raw uint8 frames are normalised there too
        x = x.contiguous() if x.dtype == torch.uint8 else x.float().contiguous()
        t, stat = Fn.vit_embed(x, self.conv1.weight, self.class_embedding, self.positional_embedding,
                               self.ln_pre.weight, self.ln_pre.bias, frame_index)
        return self.transformer(t, n, self.tokens, lead_only, x_stat=stat)


def convert_weights(model: nn.Module):
    """Cast Conv/Linear/MultiheadAttention/proj/text_projection parameters to fp16
    (reference modules/module_clip.py:506-527)."""

    def _convert(l):
        if isinstance(l, (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear)):
            l.weight.data = l.weight.data.half()
            if l.bias is not None:
                l.bias.data = l.bias.data.half()
        if isinstance(l, nn.MultiheadAttention):
            for attr in ["in_proj_weight", "q_proj_weight", "k_proj_weight", "v_proj_weight", "in_proj_bias", "bias_k", "bias_v"]:
                t = getattr(l, attr, None)
                if t is not None:
                    t.data = t.data.half()
        for name in ["text_projection", "proj"]:
            if hasattr(l, name):
                attr = getattr(l, name)
                if attr is not None:
                    attr.data = attr.data.half()

    model.apply(_convert)


class CLIP(nn.Module):
    """Container with the reference CLIP's members (reference modules/module_clip.py:328-387); the
    encoders copy `visual`, `token_embedding`, `positional_embedding`, `transformer`, `ln_final`,
    `text_projection` out of it (modules/module_cross.py:159-161,250-256)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers, linear_patch="2d"):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise NotImplementedError("CLIP ResNet towers are outside the HMMC hot path (cross_config selects ViT)")
        self.context_length = context_length
        self.vit = True
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim, linear_patch)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads, attn_mask="causal")
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))

    @staticmethod
    def get_config(pretrained_clip_name="ViT-B/32"):
        """Reference: load modules/ViT-B-32.pt or download (modules/module_clip.py:418-439).  There is no
        network here: a checkpoint FILE path (or ViT-B-32.pt next to this module) is loaded; a known
        architecture name ("ViT-B/32", "ViT-B/16", "tiny", ...) without a file yields CLIP-shaped
        random weights drawn as CLIP.initialize_parameters does (hmmc_amd.synth)."""
        local = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ViT-B-32.pt")
        path = None
        if pretrained_clip_name == "ViT-B/32" and os.path.exists(local):
            path = local
        elif os.path.isfile(pretrained_clip_name):
            path = pretrained_clip_name
        if path is not None:
            # A plain state_dict file loads with the weights-only unpickler (nothing in the file is executed).  OpenAI's
            # ViT-B-32.pt is a TorchScript archive (the reference opens it with torch.jit.load, modules/module_clip.py:425-
            # 439): deserialising it runs the archive's code, so it is taken only for a zip archive that weights_only
            # refused AND with HMMC_ALLOW_TORCHSCRIPT_CHECKPOINT=1 set; only its state_dict is kept.
            try:
                return torch.load(path, map_location="cpu", weights_only=True)
            except Exception as err:
                import zipfile
                if not (zipfile.is_zipfile(path) and os.environ.get("HMMC_ALLOW_TORCHSCRIPT_CHECKPOINT") == "1"):
                    raise RuntimeError(f"{path}: not loadable as a weights-only state_dict ({type(err).__name__}: {err}); a "
                                       "TorchScript archive needs HMMC_ALLOW_TORCHSCRIPT_CHECKPOINT=1") from err
                return torch.jit.load(path, map_location="cpu").eval().state_dict()
        if pretrained_clip_name in synth.NAMED:
            dims = synth.NAMED[pretrained_clip_name]
            return synth.clip_state_from(synth.finetune_state(dims, use_temp=False), dims)
        raise RuntimeError(f"Model {pretrained_clip_name} not found; available models = {sorted(synth.NAMED)}")


def dims_from_state_dict(state_dict, visual_prefix="visual.", text_prefix=""):
    """Infer every dimension from tensor shapes, as build_model does (reference modules/module_clip.py:531-553)."""
    vw = state_dict[visual_prefix + "conv1.weight"].shape[0]
    vl = len([k for k in state_dict if k.startswith(visual_prefix) and k.endswith(".attn.in_proj_weight")])
    ps = state_dict[visual_prefix + "conv1.weight"].shape[-1]
    grid = round((state_dict[visual_prefix + "positional_embedding"].shape[0] - 1) ** 0.5)
    tp = text_prefix
    return dict(embed_dim=state_dict[tp + "text_projection"].shape[1], image_resolution=ps * grid, vision_layers=vl,
                vision_width=vw, vision_patch_size=ps, context_length=state_dict[tp + "positional_embedding"].shape[0],
                vocab_size=state_dict[tp + "token_embedding.weight"].shape[0],
                transformer_width=state_dict[tp + "ln_final.weight"].shape[0],
                transformer_heads=state_dict[tp + "ln_final.weight"].shape[0] // 64,
                transformer_layers=len({k.split(".")[2 + tp.count(".")] for k in state_dict
                                        if k.startswith(tp + "transformer.resblocks")}))


def build_model(state_dict: dict, local_rank=0):
    """state_dict -> CLIP with fp16 tower weights (reference modules/module_clip.py:530-579)."""
    if "visual.proj" not in state_dict:
        raise NotImplementedError("CLIP ResNet checkpoints are outside the HMMC hot path")
    state_dict = dict(state_dict)
    model = CLIP(**dims_from_state_dict(state_dict)).float()
    for key in ["input_resolution", "context_length", "vocab_size"]:
        state_dict.pop(key, None)
    convert_weights(model)
    model.load_state_dict(state_dict)
    return model
