"""HMMC models with the reference's class names, constructor and forward() signatures
(reference: modules/modeling.py:25-36 dist_collect, :39-67 from_pretrained, :88-436 BirdPreTrainedModel,
:648-722 BirdModel, :788-807 MLP) running on the MI355X kernels of libhmmc_hip.so.
"""
from __future__ import annotations

import logging
import math

import torch
import torch.distributed as dist
from torch import nn

from . import functional as Fn
from . import ops
from .module_clip import dims_from_state_dict
from .module_cross import BertLMPredictionHead, CrossConfig, TextEncoder, VisualEncoder
from .until_module import CrossEn, PreTrainedModel

logger = logging.getLogger(__name__)


# ----------------------------------------------------------------------------- collectives

class _AllGatherCat(torch.autograd.Function):
    """Differentiable all-gather along dim 0 (the reference uses diffdist.functional.all_gather,
    modules/modeling.py:25-36): forward = concat in rank order (one RCCL all-gather), backward = sum over
    ranks of each rank's slice (one reduce-scatter)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        world = dist.get_world_size()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        rank = dist.get_rank()
        gx = torch.empty((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        try:
            dist.reduce_scatter_tensor(gx, g, op=dist.ReduceOp.SUM)
        except (RuntimeError, NotImplementedError):      # gloo: no reduce_scatter
            g = g.clone()
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            gx = g[rank * ctx.rows:(rank + 1) * ctx.rows].contiguous()
        return gx


def dist_collect(x):
    """collect a tensor from all ranks: [b, ...] -> [b * world, ...] (rank order), differentiable."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x.contiguous()
    return _AllGatherCat.apply(x)


# ----------------------------------------------------------------------------- base classes

class CLIP4ClipPreTrainedModel(PreTrainedModel, nn.Module):
    def __init__(self, cross_config, *inputs, **kwargs):
        super().__init__(cross_config)
        self.cross_config = cross_config

    @classmethod
    def from_pretrained(cls, cross_model_name, state_dict=None, cache_dir=None, type_vocab_size=2, *inputs, **kwargs):
        """Same contract as the reference (modules/modeling.py:48-67).  When a state_dict is given the CLIP
        dimensions are inferred from it (as build_model does from a checkpoint), so a full HMMC checkpoint
        (`pytorch_model.bin.N`) constructs and fills the model without any download."""
        task_config = kwargs.get("task_config")
        if task_config is not None:
            if not hasattr(task_config, "local_rank"):
                task_config.__dict__["local_rank"] = 0
            elif task_config.local_rank == -1:
                task_config.local_rank = 0
        cross_config, _ = CrossConfig.get_config(cross_model_name, cache_dir, type_vocab_size, state_dict=None,
                                                 task_config=task_config)
        if state_dict is not None and "visual_encoder.visual.conv1.weight" in state_dict:
            from . import synth
            clip_sd = {}
            for k, v in state_dict.items():
                if k.startswith("visual_encoder.visual."):
                    clip_sd["visual." + k[len("visual_encoder.visual."):]] = v
                elif k.startswith("text_encoder."):
                    clip_sd[k[len("text_encoder."):]] = v
            clip_sd["logit_scale"] = torch.tensor(math.log(100.0))
            cross_config._clip_state_dict = clip_sd
        model = cls(cross_config, *inputs, **kwargs)
        if hasattr(cross_config, "_clip_state_dict"):
            del cross_config._clip_state_dict
        if state_dict is not None:
            model = cls.init_preweight(model, dict(state_dict), task_config=task_config)
        return model


class BirdPreTrainedModel(CLIP4ClipPreTrainedModel):
    """Pre-training model (MoCo queues, FAM/VTM/FTM/MLM); filled in by hmmc_amd.pretrain (see __init__)."""

    def loose_similarity(self, sequence_output, visual_output):
        """100 * n(q) n(v)^T; visual may be [bv,512] or [bv,F,512] -> [bq,bv,F]
        (reference modules/modeling.py:207-229).  Inference-only entry point (eval scorer); training
        goes through the fused InfoNCE head."""
        if torch.is_grad_enabled() and (sequence_output.requires_grad or visual_output.requires_grad):
            raise RuntimeError("loose_similarity is the eval entry point; training uses the fused InfoNCE head")
        q = sequence_output.contiguous().float().view(-1, sequence_output.shape[-1])
        v = visual_output.contiguous().float()
        E = q.shape[-1]
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        qn, _ = ops.l2norm_fwd(q)
        three_d = v.dim() == 3
        vn, _ = ops.l2norm_fwd(v.view(-1, E))
        S = ops.gemm_f32(qn, vn, qn.shape[0], vn.shape[0], E, (E, 1), (1, E), alpha=scale)
        return S.view(qn.shape[0], v.shape[0], v.shape[1]) if three_d else S

    def eval_scores(self, query_output, visual_output, frame_output, top_frames=None):
        """video-text logits + mean of the top-k frame-text logits (main_task_retrieval.py:332-336,512-513)."""
        k = top_frames or self.top_frames
        sv = self.loose_similarity(query_output, visual_output)
        sf = self.loose_similarity(query_output, frame_output)
        bq, bv, F = sf.shape
        return sv, ops.topk_mean(sf.view(bq, bv * F), bq, bv, F, k)


class BirdModel(BirdPreTrainedModel):
    """Fine-tuning model (reference modules/modeling.py:648-722)."""

    def __init__(self, cross_config, task_config):
        super(BirdPreTrainedModel, self).__init__(cross_config)
        self.task_config = task_config
        self.rank = task_config.local_rank
        self.weight_VTM_finetune = cross_config.weight_VTM_finetune
        self.weight_FTM_finetune = cross_config.weight_FTM_finetune
        self.top_frames = task_config.top_frames
        self.text_encoder = TextEncoder(self.task_config, cross_config)
        self.visual_encoder = VisualEncoder(self.task_config, cross_config)
        self.loss_fct = CrossEn()

    def forward(self, query_ids, query_mask, video_data, video_frame, idx, global_step):
        query_ids = query_ids.view(-1, query_ids.shape[-1])
        video = torch.as_tensor(video_data)
        if not self.training:
            return None
        query_output = self.text_encoder(query_ids, query_mask)
        visual_output, frame_output = self.visual_encoder(video, video_frame)
        b, F, E = frame_output.shape
        # one packed all-gather [b, (F+2)*E] instead of the reference's three (modules/modeling.py:698-700)
        packed = dist_collect(torch.cat([visual_output, query_output, frame_output.reshape(b, F * E)], dim=1))
        visual_output, query_output = packed[:, :E], packed[:, E:2 * E]
        frame_output = packed[:, 2 * E:].reshape(-1, F, E)
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        use_frames = bool(self.task_config.use_frame_fea)
        loss = Fn.FinetuneHeadFn.apply(query_output, visual_output, frame_output if use_frames else None,
                                       self.weight_VTM_finetune, self.weight_FTM_finetune, scale)
        if self.task_config.local_rank == 0 and getattr(self.task_config, "logdir", None):
            self.task_config.writer.add_scalar("loss", float(loss), global_step=global_step)
        return loss


class MLP(nn.Module):
    """Projector / predictor parameter container: Identity, Linear(512,4096), BatchNorm1d(4096), ReLU, Linear(4096,512)
    (reference modules/modeling.py:788-807)."""

    def __init__(self, in_dim=512, inner_dim=4096, out_dim=512, num_layers=2):
        super().__init__()
        hidden = [nn.Identity()]
        for i in range(num_layers - 1):
            hidden.append(nn.Linear(in_dim if i == 0 else inner_dim, inner_dim))
            hidden.append(nn.BatchNorm1d(inner_dim))
            hidden.append(nn.ReLU(inplace=True))
        self.linear_hidden = nn.Sequential(*hidden)
        self.linear_out = nn.Linear(in_dim if num_layers == 1 else inner_dim, out_dim) if num_layers >= 1 else nn.Identity()
