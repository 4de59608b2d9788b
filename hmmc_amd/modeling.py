"""HMMC models with the reference's class names, constructor and forward() signatures
(reference: modules/modeling.py:25-36 dist_collect, :39-67 from_pretrained, :88-436 BirdPreTrainedModel,
:648-722 BirdModel, :788-807 MLP) running on the MI355X kernels of libhmmc_hip.so.
"""
from __future__ import annotations

import logging
import math
import os

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

_OVERLAP_TOWERS = os.environ.get("HMMC_OVERLAP_TOWERS", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _AllGatherCat(torch.autograd.Function):
    """Differentiable all-gather along dim 0 (the reference uses diffdist.functional.all_gather,
    modules/modeling.py:25-36): forward = concat in rank order (one RCCL all-gather), backward = sum over
    ranks of each rank's slice (one reduce-scatter).  The collective is chosen ONCE from the backend's name, never by
    catching an exception: a failed collective must propagate, not be followed by a different one on this rank alone."""

    @staticmethod
    def _flat():
        return dist.get_backend() == "nccl"              # RCCL has the flat forms; gloo (CPU tests) does not

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        world = dist.get_world_size()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        if _AllGatherCat._flat():
            dist.all_gather_into_tensor(out, x)
        else:
            dist.all_gather(list(out.chunk(world, dim=0)), x)
        ctx.rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if _AllGatherCat._flat():
            gx = torch.empty((ctx.rows,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(gx, g, op=dist.ReduceOp.SUM)
            return gx
        rank = dist.get_rank()
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g[rank * ctx.rows:(rank + 1) * ctx.rows].contiguous()


def dist_collect(x):
    """collect a tensor from all ranks: [b, ...] -> [b * world, ...] (rank order), differentiable."""
    if not Fn.collectives_active():
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
    """Pre-training model: online + momentum encoders, five negative queues, FAM / VTM / FTM / MLM losses
    (reference modules/modeling.py:88-436).  forward() keeps the reference's order: online encoders, EMA update,
    key encoders, losses against the OLD queues, enqueue, MLM."""

    PAD_ID, CLS_ID, MASK_ID, VOCAB = 49407, 49406, 49394, 49408     # ClipTokenizer ids (tokenization_clip.py:75-87)

    def __init__(self, cross_config, task_config):
        super(BirdPreTrainedModel, self).__init__(cross_config)
        self.task_config = task_config
        self.rank = task_config.local_rank
        self.mlm_probability = cross_config.mlm_probability
        self.top_frames = task_config.top_frames
        self.weight_FAM, self.weight_VTM = cross_config.weight_FAM, cross_config.weight_VTM
        self.weight_FTM, self.weight_MLM = cross_config.weight_FTM, cross_config.weight_MLM
        self.contrast_momentum = task_config.contrast_momentum
        self.contrast_temperature = task_config.contrast_temperature
        self.contrast_num_negative = task_config.contrast_num_negative
        E = cross_config.temporal_hidden_size
        self.text_encoder = TextEncoder(self.task_config, cross_config)
        self.text_encoder_k = TextEncoder(self.task_config, cross_config)
        self.t_projector = MLP(num_layers=cross_config.proj_num_layers)
        self.t_projector_k = MLP(num_layers=cross_config.proj_num_layers)
        self.cls = BertLMPredictionHead(E, self.VOCAB, "gelu")
        self.visual_encoder = VisualEncoder(self.task_config, cross_config)
        self.visual_encoder_k = VisualEncoder(self.task_config, cross_config)
        self.v_projector = MLP(num_layers=cross_config.proj_num_layers)
        self.v_projector_k = MLP(num_layers=cross_config.proj_num_layers)
        self.v_predictor = MLP(num_layers=cross_config.pred_num_layers)
        self.model_pairs = [[self.visual_encoder, self.visual_encoder_k], [self.text_encoder, self.text_encoder_k],
                            [self.v_projector, self.v_projector_k], [self.t_projector, self.t_projector_k]]
        self.copy_params()
        K, Fm = self.contrast_num_negative, self.task_config.max_frames
        for name, width in (("queue_v_cross_ng", K), ("queue_frame_proj_ng", K * Fm), ("queue_frame_cross_ng", K * Fm),
                            ("queue_title_cross_ng", K), ("queue_tag_cross_ng", K)):
            q = torch.randn(E, width)
            self.register_buffer(name, torch.nn.functional.normalize(q, dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        # host copy of queue_ptr: the reference reads the device buffer with int() every step (modeling.py:270), which stalls
        # the launch stream; the buffer stays the source of truth and is re-read once after construction / load_state_dict
        self._queue_ptr_host = None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: setattr(module, "_queue_ptr_host", None))
        self.loss_fct = CrossEn()
        self._ema_table = None
        self._mlm_draws = None          # tests inject the reference's recorded random draws here

    # ---- momentum encoders
    @torch.no_grad()
    def copy_params(self):
        for online, key in self.model_pairs:
            for param, param_k in zip(online.parameters(), key.parameters()):
                param_k.data.copy_(param.data)
                param_k.requires_grad = False

    @torch.no_grad()
    def _momentum_update(self):
        """p_k = p_k * m + p * (1 - m) for the 4 model pairs, one multi-tensor launch
        (reference modules/modeling.py:238-242, 362 tensors in three kernels each)."""
        from .optimization import _TensorTable, _dtype_flag
        from ._lib import call, ptr
        rows = []
        for online, key in self.model_pairs:
            for param, param_k in zip(online.parameters(), key.parameters()):
                rows.append((param_k.data_ptr(), param.data_ptr(), 0, 0, param.numel(), _dtype_flag(param)))
        dev = next(self.parameters()).device
        if self._ema_table is None or self._ema_table.device != dev:
            self._ema_table = _TensorTable(dev)
        tbl = self._ema_table.build(rows)
        m = float(self.contrast_momentum)
        call("hmmc_mt_ema", ptr(tbl.tab), ptr(tbl.chunk), tbl.nchunks, m, 1.0 - m)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, v_fea_k, tag_fea_k, title_fea_k, frame_fea_k, frame_proj_k):
        """gather keys from every rank, normalise, overwrite queue columns [ptr, ptr+B) (frame queues: [ptr*F, (ptr+B)*F))
        (reference modules/modeling.py:244-284)."""
        b, F, E = frame_fea_k.shape
        if Fn.collectives_active():
            packed = torch.cat([v_fea_k, tag_fea_k, title_fea_k, frame_fea_k.reshape(b, F * E), frame_proj_k.reshape(b, F * E)], dim=1)
            packed = dist_collect(packed)
            B = packed.shape[0]
            v, tag, title = packed[:, :E], packed[:, E:2 * E], packed[:, 2 * E:3 * E]
            fr = packed[:, 3 * E:3 * E + F * E].reshape(B * F, E)
            fp = packed[:, 3 * E + F * E:].reshape(B * F, E)
        else:                                            # one rank: nothing to gather, nothing to pack and slice apart again
            B = b
            v, tag, title = v_fea_k, tag_fea_k, title_fea_k
            fr, fp = frame_fea_k.reshape(B * F, E), frame_proj_k.reshape(B * F, E)
        ptr_ = self._queue_ptr_host if self._queue_ptr_host is not None else int(self.queue_ptr)
        ops.enqueue(v.contiguous(), self.queue_v_cross_ng, ptr_)
        ops.enqueue(tag.contiguous(), self.queue_tag_cross_ng, ptr_)
        ops.enqueue(title.contiguous(), self.queue_title_cross_ng, ptr_)
        ops.enqueue(fp.contiguous(), self.queue_frame_proj_ng, ptr_ * F)
        ops.enqueue(fr.contiguous(), self.queue_frame_cross_ng, ptr_ * F)
        self._queue_ptr_host = (ptr_ + B) % self.contrast_num_negative
        self.queue_ptr.fill_(self._queue_ptr_host)

    # ---- MLP with train-mode BatchNorm (batch statistics shared over ranks): MLP.forward below
    def _mlp(self, mlp, x):
        return mlp(x)

    # ---- losses
    def contrastive_loss(self, q, k, queue):
        """reference modules/modeling.py:286-313 (mean over rows)."""
        q, k = q.reshape(-1, q.shape[-1]), k.reshape(-1, k.shape[-1])
        return Fn.MocoLossFn.apply(q, k, queue, self.contrast_temperature, 1.0 / q.shape[0])

    def frame_self_loss(self, frame_fea, frame_fea_k, queue_frame_ng):
        """FAM (reference modules/modeling.py:315-323): all 2(F-1) contrastive_loss calls in one batched launch set."""
        b, F, E = frame_fea.shape
        q = torch.cat([frame_fea[:, :-1].reshape(-1, E), frame_fea[:, 1:].reshape(-1, E)])
        k = torch.cat([frame_fea_k[:, 1:].reshape(-1, E), frame_fea_k[:, :-1].reshape(-1, E)])
        return Fn.MocoLossFn.apply(q, k, queue_frame_ng, self.contrast_temperature, 1.0 / (b * (F - 1)))

    def frame_cross_loss(self, frame_fea, frame_fea_k, queue_frame_ng, text_fea, text_fea_k, queue_text_ng):
        """FTM (reference modules/modeling.py:325-332): 2F calls batched into two (one per queue)."""
        b, F, E = frame_fea.shape
        w = 1.0 / (b * F)
        t2f = Fn.MocoLossFn.apply(text_fea.unsqueeze(1).expand(b, F, E).reshape(-1, E), frame_fea_k.reshape(-1, E),
                                  queue_frame_ng, self.contrast_temperature, w)
        f2t = Fn.MocoLossFn.apply(frame_fea.reshape(-1, E), text_fea_k.unsqueeze(1).expand(b, F, E).reshape(-1, E),
                                  queue_text_ng, self.contrast_temperature, w)
        return t2f + f2t

    def mask(self, input_ids, vocab_size, device, targets=None, masked_indices=None, probability_matrix=None):
        """BERT masking with the reference's quirks (modules/modeling.py:181-205): never masks EOT (its "pad") or SOT;
        80% [MASK]=49394, 10% random id; the random draws are made on the CPU like the reference's."""
        if self._mlm_draws is not None:
            masked, replaced, randsel, words = [t.to(input_ids.device) for t in self._mlm_draws]
            masked, replaced, randsel = masked.bool(), replaced.bool(), randsel.bool()
        else:
            shape = input_ids.shape
            masked = torch.bernoulli(probability_matrix).bool().to(input_ids.device)
            replaced = torch.bernoulli(torch.full(shape, 0.8)).bool().to(input_ids.device)
            randsel = torch.bernoulli(torch.full(shape, 0.5)).bool().to(input_ids.device)
            words = torch.randint(vocab_size, shape, dtype=torch.long).to(input_ids.device)
        masked = masked.clone()
        masked[input_ids == self.PAD_ID] = False
        masked[input_ids == self.CLS_ID] = False
        if targets is not None:
            targets[~masked] = -100
        rep = replaced & masked
        input_ids[rep] = self.MASK_ID
        rnd = randsel & masked & ~rep
        input_ids[rnd] = words[rnd]
        return (input_ids, targets) if targets is not None else input_ids

    def _mlm_inputs(self, input_ids):
        """masked ids and labels of the MLM pass (the first half of get_mlm_loss, reference modules/modeling.py:160-165)"""
        ids = input_ids.clone()
        labels = ids.clone()
        prob = torch.full(labels.shape, self.mlm_probability)
        # random replacement ids are drawn from the rows the embedding table really has (VOCAB for CLIP's tokenizer)
        vocab = self.text_encoder.token_embedding.weight.shape[0]
        return self.mask(ids, vocab, input_ids.device, targets=labels, probability_matrix=prob)

    def get_mlm_loss(self, input_ids, input_mask):
        ids, labels = self._mlm_inputs(input_ids)
        hidden = self.text_encoder(ids, input_mask, return_hidden=True)
        return self.calculate_mlm_loss(hidden, labels, _label_density=float(self.mlm_probability))

    def calculate_mlm_loss(self, sequence_output_mlm, labels, _label_density=1.0):
        """reference modules/modeling.py:171-179.  `_label_density` sizes the head's row buffer: get_mlm_loss passes the
        Bernoulli rate its own mask() drew the labels with; labels from anywhere else (a direct call) may be dense, so
        every position gets a row (1.0: no compaction limit, nothing can be dropped)."""
        c = self.cls
        return Fn.MlmHeadFn.apply(sequence_output_mlm, labels, c.transform.dense.weight, c.transform.dense.bias,
                                  c.transform.LayerNorm.weight, c.transform.LayerNorm.bias, c.decoder.weight, c.bias,
                                  _label_density)

    def forward(self, video_data, video_frame, tag_ids, tag_mask, title_ids, title_mask, global_step):
        tag_ids = tag_ids.view(-1, tag_ids.shape[-1])
        tag_mask = tag_mask.view(-1, tag_mask.shape[-1])
        title_ids = title_ids.view(-1, title_ids.shape[-1])
        title_mask = title_mask.view(-1, title_mask.shape[-1])
        video = torch.as_tensor(video_data)
        if not self.training:
            return None
        bird = self.task_config.dataset == "bird"
        overlap = _OVERLAP_TOWERS and video.is_cuda
        # Text passes that share weights share ONE pass of the tower (TextEncoder.encode_many): the online title pass with the
        # MLM pass of the masked titles (the reference runs the latter after the enqueue, :413-416; the online weights do not
        # change in between and the mask's random draws are the only consumer of the host generator in this function), and the
        # two momentum passes.  The `bird` branches (unreachable from main_pretrain.py) keep separate passes.
        mlm_hidden = mlm_labels = None
        if not bird:
            mlm_ids, mlm_labels = self._mlm_inputs(title_ids)

        def online_text():
            if bird:
                return self.text_encoder(tag_ids, tag_mask), self.text_encoder(title_ids, title_mask), None
            title, hidden = self.text_encoder.encode_many([title_ids, mlm_ids], ["feat", "hidden"])
            return None, title, hidden

        def key_text():
            if bird:
                return self.text_encoder_k(tag_ids, tag_mask), self.text_encoder_k(title_ids, title_mask)
            return self.text_encoder_k.encode_many([tag_ids, title_ids], ["feat", "feat"])
        if overlap:
            # The text towers (online title / tag features, momentum text encoder) run on a side stream beside the two frame
            # towers; the momentum pass waits for the EMA update, which stays on the main stream.  Autograd replays each
            # backward on its forward stream.  The MLM HEAD is not moved (only the masked titles' pass through the text tower
            # rides with the title pass): its 49 408-way fp32 GEMM is a long kernel of many small workgroups which, run beside
            # the frame tower, keeps the persistent 128 KiB-LDS GEMM workgroups from becoming resident (measured: 105.6 vs
            # 101.7 ms per step).
            cur = torch.cuda.current_stream(video.device)
            side = _side_stream(video.device)
            side.wait_stream(cur)
            for t in (tag_ids, tag_mask, title_ids, title_mask) + (() if bird else (mlm_ids,)):
                t.record_stream(side)
            with torch.cuda.stream(side):
                tag_fea, title_fea, mlm_hidden = online_text()
        # the momentum tower below encodes the same frames: one im2col for both (the context manager drops the shared patch
        # matrix on every exit path, so a failed step cannot leave a stale one behind for the next caller)
        with Fn.share_patches():
            v_fea, frame_fea = self.visual_encoder(video, video_frame)
            if not overlap:
                tag_fea, title_fea, mlm_hidden = online_text()
            bs, frame, hidden = frame_fea.shape
            frame_proj = self.v_projector(frame_fea.reshape(-1, hidden))
            frame_pred = self.v_predictor(frame_proj).view(bs, frame, hidden)
            frame_proj = frame_proj.view(bs, frame, hidden)
            with torch.no_grad():
                self._momentum_update()
                if overlap:
                    ema_done = torch.cuda.Event()
                    ema_done.record(cur)
                    side.wait_event(ema_done)
                    with torch.cuda.stream(side):
                        tag_fea_k, title_fea_k = key_text()
                else:
                    tag_fea_k, title_fea_k = key_text()
                v_fea_k, frame_fea_k = self.visual_encoder_k(video, video_frame)
        with torch.no_grad():
            frame_proj_k = self.v_projector_k(frame_fea_k.reshape(-1, hidden)).view(bs, frame, hidden)
        if overlap:
            cur.wait_stream(side)
            for t in (tag_fea, title_fea, tag_fea_k, title_fea_k, mlm_hidden):
                if t is not None:
                    t.record_stream(cur)
        # The losses (and their backward, which runs after the enqueue below) must see the OLD negatives: one
        # snapshot per queue per step (the reference clones the queue inside each of its 48 contrastive_loss calls).
        names = ["queue_frame_proj_ng", "queue_frame_cross_ng", "queue_title_cross_ng", "queue_v_cross_ng"] + (["queue_tag_cross_ng"] if bird else [])
        snaps = [torch.empty_like(getattr(self, n)) for n in names]
        torch._foreach_copy_(snaps, [getattr(self, n) for n in names])         # one multi-tensor launch for the four / five snapshots
        q_proj, q_cross, q_title, q_v = snaps[:4]
        q_tag = snaps[4] if bird else None
        loss_FAM = self.frame_self_loss(frame_pred, frame_proj_k, q_proj)
        v_title = self.contrastive_loss(v_fea, title_fea_k, q_title) + self.contrastive_loss(title_fea, v_fea_k, q_v)
        if bird:
            v_tag = self.contrastive_loss(v_fea, tag_fea_k, q_tag) + self.contrastive_loss(tag_fea, v_fea_k, q_v)
            loss_VTM = (v_tag + v_title) / 2
        else:
            loss_VTM = v_title
        loss_FTM = 0.0
        if self.task_config.use_frame_fea:
            ft = self.frame_cross_loss(frame_fea, frame_fea_k, q_cross, title_fea, title_fea_k, q_title)
            if bird:
                ftag = self.frame_cross_loss(frame_fea, frame_fea_k, q_cross, tag_fea, tag_fea_k, q_tag)
                loss_FTM = (ftag + ft) / 2
            else:
                loss_FTM = ft
        self._dequeue_and_enqueue(v_fea_k, tag_fea_k, title_fea_k, frame_fea_k, frame_proj_k)
        if bird:
            loss_MLM = (self.get_mlm_loss(tag_ids, tag_mask) + self.get_mlm_loss(title_ids, title_mask)) / 2
        else:
            loss_MLM = self.calculate_mlm_loss(mlm_hidden, mlm_labels, _label_density=float(self.mlm_probability))
        self.last_losses = (loss_FAM, loss_VTM, loss_FTM, loss_MLM)
        loss = self.weight_FAM * loss_FAM + self.weight_VTM * loss_VTM + self.weight_FTM * loss_FTM + self.weight_MLM * loss_MLM
        if self.rank == 0 and getattr(self.task_config, "logdir", None):
            self.task_config.writer.add_scalars("loss", {"loss": float(loss)}, global_step=global_step)
        # device-side checks (token id outside the embedding table, MLM row buffer overflow) surface at the reference's own
        # logging interval, where its loop reads the loss on the host anyway (main_pretrain.py:232-240)
        nd = getattr(self.task_config, "n_display", 0)
        if nd and global_step and global_step % nd == 0:
            ops.raise_on_device_errors(video.device)
        return loss

    def loose_similarity(self, sequence_output, visual_output):
        """100 * n(q) n(v)^T; visual may be [bv,512] or [bv,F,512] -> [bq,bv,F]
        (reference modules/modeling.py:207-229).  Differentiable when its inputs require grad; the training step itself goes
        through the fused InfoNCE head."""
        q = sequence_output.contiguous().float().view(-1, sequence_output.shape[-1])
        v = visual_output.contiguous().float()
        E = q.shape[-1]
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        if torch.is_grad_enabled() and (q.requires_grad or v.requires_grad):      # differentiable, as the reference's
            S = Fn.LooseSimFn.apply(q, v.view(-1, E), scale)
            return S.view(q.shape[0], v.shape[0], v.shape[1]) if v.dim() == 3 else S
        qn, _ = ops.l2norm_fwd(q)
        three_d = v.dim() == 3
        vn, _ = ops.l2norm_fwd(v.view(-1, E))
        S = ops.gemm_f32(qn, vn, qn.shape[0], vn.shape[0], E, (E, 1), (1, E), alpha=scale)
        return S.view(qn.shape[0], v.shape[0], v.shape[1]) if three_d else S

    def eval_scores(self, query_output, visual_output, frame_output, top_frames=None, packed=None):
        """(video-text logits, mean of the top-k frame-text logits), each [queries, videos]
        (main_task_retrieval.py:332-336: loose_similarity twice, torch.topk(frame_logits, k, dim=2)[0].mean(2)).
        One fused launch (hmmc_eval_score): the [queries, videos, F] logits tensor the reference materialises - 2.16 GB at
        VATEX's 15 000 x 1 500 x 24 - is never written.  `packed` = ops.eval_pack(visual, frames) of the candidates, for
        callers that score many query batches against the same videos."""
        k = top_frames or self.top_frames
        q = query_output.contiguous().float().view(-1, query_output.shape[-1])
        v = visual_output.contiguous().float().view(-1, q.shape[-1])
        u = frame_output.contiguous().float()
        nv, F, E = u.shape
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        if ops.eval_slots(F) == 0:                       # more than 31 frames per video: two launches, [bq, bv, F] in HBM
            sv = self.loose_similarity(q, v)
            sf = self.loose_similarity(q, u)
            return sv, ops.topk_mean(sf.view(q.shape[0], nv * F), q.shape[0], nv, F, k)
        qn, _ = ops.l2norm_fwd(q)
        if packed is None:
            packed = ops.eval_pack(v, u)
        sv, sf, _ = ops.eval_score(qn, packed, nv, F, k, scale)
        return sv, sf

    @torch.no_grad()
    def eval_similarity(self, batch_query_output_list, batch_visual_output_list, batch_frame_output_list,
                        use_frame_fea=True, query_chunk=4096):
        """The cached-feature scoring loop of eval_epoch (main_task_retrieval.py:321-357 _run_on_single_gpu + :512-513):
        lists of per-batch query / video / frame features -> the [all queries, all videos] score matrix
        sim_matrix (+ sim_matrix_frame when use_frame_fea), on the device.  The reference runs len(queries) x len(videos)
        small matmuls with a host copy each; here the candidates are packed once and every chunk of queries is one launch."""
        q = torch.cat([t.float() for t in batch_query_output_list], dim=0)
        v = torch.cat([t.float() for t in batch_visual_output_list], dim=0)
        u = torch.cat([t.float() for t in batch_frame_output_list], dim=0)
        nv, F, E = u.shape
        k = self.top_frames
        if ops.eval_slots(F) == 0:
            sv, sf = self.eval_scores(q, v, u)
            return sv + sf if use_frame_fea else sv
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        packed = ops.eval_pack(v.contiguous(), u.contiguous())
        qn, _ = ops.l2norm_fwd(q.contiguous())
        out = torch.empty((q.shape[0], nv), dtype=torch.float32, device=q.device)
        for s in range(0, q.shape[0], query_chunk):
            e = min(q.shape[0], s + query_chunk)
            want = ("score",) if use_frame_fea else ("video",)
            sv, sf, sc = ops.eval_score(qn[s:e], packed, nv, F, k, scale, want=want)
            out[s:e] = sc if use_frame_fea else sv
        return out


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

    def frame_loss(self, query_output, frame_output):
        """(1/F) sum_f [CrossEn(S_f) + CrossEn(S_f^T)], S_f = loose_similarity(query, frame[:, f, :]); differentiable
        (reference modules/modeling.py:665-672).  forward() does not call it: there the same F terms are part of one fused
        head together with the video-text term; this member evaluates them alone through the same kernels (the video
        columns of the fused logit matrix are given weight 0 and a stand-in operand)."""
        q = query_output.contiguous().float().view(-1, query_output.shape[-1])
        u = frame_output.contiguous().float()
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        return Fn.FinetuneHeadFn.apply(q, q.detach(), u, 0.0, 1.0, scale)

    def forward(self, query_ids, query_mask, video_data, video_frame, idx, global_step):
        query_ids = query_ids.view(-1, query_ids.shape[-1])
        video = torch.as_tensor(video_data)
        if not self.training:
            return None
        if _OVERLAP_TOWERS and video.is_cuda:
            # The text tower (small, latency-bound kernels) runs on a side stream beside the frame tower, whose
            # persistent GEMMs leave most CUs idle in their last, partial round.  Autograd replays each node's backward
            # on its forward stream, so the two backward passes overlap the same way.
            cur = torch.cuda.current_stream(video.device)
            side = _side_stream(video.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                query_output = self.text_encoder(query_ids, query_mask)
            visual_output, frame_output = self.visual_encoder(video, video_frame)
            cur.wait_stream(side)
            query_output.record_stream(cur)
            query_ids.record_stream(side)
        else:
            query_output = self.text_encoder(query_ids, query_mask)
            visual_output, frame_output = self.visual_encoder(video, video_frame)
        b, F, E = frame_output.shape
        if Fn.collectives_active():
            # one packed all-gather [b, (F+2)*E] instead of the reference's three (modules/modeling.py:698-700)
            packed = dist_collect(torch.cat([visual_output, query_output, frame_output.reshape(b, F * E)], dim=1))
            visual_output, query_output = packed[:, :E], packed[:, E:2 * E]
            frame_output = packed[:, 2 * E:].reshape(-1, F, E)
        # (one rank: the gather is the identity - no packing, slicing and re-packing of the features and of their gradients)
        scale = min(math.exp(float(self.text_encoder.logit_scale)), 100.0)
        use_frames = bool(self.task_config.use_frame_fea)
        loss = Fn.FinetuneHeadFn.apply(query_output, visual_output, frame_output if use_frames else None,
                                       self.weight_VTM_finetune, self.weight_FTM_finetune, scale)
        if self.task_config.local_rank == 0 and getattr(self.task_config, "logdir", None):
            self.task_config.writer.add_scalar("loss", float(loss), global_step=global_step)
        # a token id outside the embedding table raises IndexError in the reference; here the kernel sets a device flag, read at
        # the reference's own logging interval (main_task_retrieval.py:304-312 reads the loss on the host there anyway)
        nd = getattr(self.task_config, "n_display", 0)
        if nd and global_step and global_step % nd == 0:
            ops.raise_on_device_errors(video.device)
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
        self.num_layers = num_layers

    def forward(self, x):
        """linear_out(linear_hidden(x)) (reference modules/modeling.py:803-807) on the fused kernels: Linear -> BatchNorm1d ->
        ReLU -> Linear in one autograd node.  Training mode: batch statistics over every rank's rows (the reference converts
        the model to SyncBatchNorm, main_pretrain.py:199-204) and the running statistics are updated as nn.BatchNorm1d does;
        eval mode: the running statistics, forward only."""
        if self.num_layers != 2:
            raise NotImplementedError("the HMMC projector / predictor MLPs have two layers (proj_num_layers = pred_num_layers = 2)")
        lin1, bn, lin2 = self.linear_hidden[1], self.linear_hidden[2], self.linear_out
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).float()
        if not self.training:
            if torch.is_grad_enabled() and (x.requires_grad or lin1.weight.requires_grad):
                raise RuntimeError("MLP in eval mode is forward-only here: call it under torch.no_grad()")
            h = ops.linear_f32(x2.contiguous(), lin1.weight, bias=lin1.bias)
            y = ops.bn_apply_relu(h, bn.running_mean, torch.rsqrt(bn.running_var + bn.eps), bn.weight, bn.bias)
            return ops.linear_f32(y, lin2.weight, bias=lin2.bias).view(*lead, -1)
        # the running statistics are updated by the same launch that forms the batch statistics (hmmc_bn_finalize)
        out, mean, var = Fn.MlpFn.apply(x2, lin1.weight, lin1.bias, bn.weight, bn.bias, lin2.weight, lin2.bias, bn.eps,
                                        (bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum))
        return out.view(*lead, -1)
