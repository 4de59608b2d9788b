"""ORACLE — test infrastructure only, never the product path.

CPU restatement (plain torch ops on CPU tensors, no custom kernels) of the
reference's training hot path.  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this file; the shipped package
(hmmc_amd/) never does and fails loudly when its HIP library is missing.

Pinned: every function here is checked in tests/test_oracle_golden.py against
golden vectors produced by running the reference itself (imported from
/root/reference on CPU by tests/golden/make_golden.py, outputs committed as
plain arrays under tests/golden/).

Every function cites the reference lines it restates (paths relative to the
reference checkout).  `sd` is a flat state_dict {name: tensor} with the
reference's key names.  Two precision regimes:
  * mode="fp32"      — everything fp32 (the reference after model.float()).
  * mode="aswritten" — CLIP-tower weights and activations fp16, LayerNorm in
                        fp32, temporal transformer / heads fp32
                        (convert_weights, modules/module_clip.py:506-527).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- building blocks

def layer_norm_clip(x, w, b):
    """modules/module_clip.py:217-223 — fp32 math, eps 1e-5, cast back."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), 1e-5).to(x.dtype)


def layer_norm_tf(x, w, b, eps=1e-12):
    """modules/until_module.py:54-67 — TF-style, eps inside the sqrt."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def quick_gelu(x):
    """modules/module_clip.py:226-228."""
    return x * torch.sigmoid(1.702 * x)


def mha(x, in_w, in_b, out_w, out_b, n_head, mask=None):
    """nn.MultiheadAttention as used at modules/module_clip.py:235,251 and
    modules/module_cross.py:118,130.  x: [N, L, D] (batch-first here; the
    reference's LND layout is only a permutation).  Packed in-proj rows are
    Q | K | V; q is scaled by 1/sqrt(head_dim); additive float mask [L, L]."""
    N, L, D = x.shape
    dh = D // n_head
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(N, L, n_head, dh).transpose(1, 2) * (dh ** -0.5)
    k = k.reshape(N, L, n_head, dh).transpose(1, 2)
    v = v.reshape(N, L, n_head, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask.to(s.dtype)
    p = torch.softmax(s.float(), dim=-1).to(x.dtype)
    o = (p @ v).transpose(1, 2).reshape(N, L, D)
    return F.linear(o, out_w, out_b)


def causal_mask(L):
    """modules/module_clip.py:441-447."""
    return torch.full((L, L), float("-inf")).triu_(1)


def _get(sd, key, dt=None):
    t = sd[key]
    return t.to(dt) if dt is not None and t.is_floating_point() else t


def resblock(x, sd, p, n_head, mask, dt, tf_ln):
    """modules/module_clip.py:253-257 (CLIP LN) / modules/module_cross.py:132-139 (TF LN)."""
    ln = (lambda y, n: layer_norm_tf(y, sd[p + n + ".weight"].float(), sd[p + n + ".bias"].float())) if tf_ln else \
         (lambda y, n: layer_norm_clip(y, sd[p + n + ".weight"], sd[p + n + ".bias"]))
    x = x + mha(ln(x, "ln_1"), _get(sd, p + "attn.in_proj_weight", dt), _get(sd, p + "attn.in_proj_bias", dt),
                _get(sd, p + "attn.out_proj.weight", dt), _get(sd, p + "attn.out_proj.bias", dt), n_head, mask)
    h = F.linear(ln(x, "ln_2"), _get(sd, p + "mlp.c_fc.weight", dt), _get(sd, p + "mlp.c_fc.bias", dt))
    x = x + F.linear(quick_gelu(h), _get(sd, p + "mlp.c_proj.weight", dt), _get(sd, p + "mlp.c_proj.bias", dt))
    return x


def transformer(x, sd, prefix, n_head, mask, dt, tf_ln=False):
    n_layers = len({k.split(".resblocks.")[1].split(".")[0] for k in sd if k.startswith(prefix + ".resblocks.")})
    for i in range(n_layers):
        x = resblock(x, sd, f"{prefix}.resblocks.{i}.", n_head, mask, dt, tf_ln)
    return x


def _tower_dtype(mode):
    return torch.float16 if mode == "aswritten" else torch.float32


# ----------------------------------------------------------------------------- encoders

def vit_hidden(image, sd, prefix="visual_encoder.visual.", mode="fp32"):
    """VisualTransformer.forward, modules/module_clip.py:297-325 ('2d' branch).
    image [N,3,H,W] -> hidden [N, L, width] (tower dtype)."""
    dt = _tower_dtype(mode)
    w = _get(sd, prefix + "conv1.weight", dt)
    width, patch = w.shape[0], w.shape[-1]
    x = F.conv2d(image.to(dt), w, stride=patch)                       # :307-310
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)
    cls = sd[prefix + "class_embedding"].to(dt) + torch.zeros(x.shape[0], 1, width, dtype=dt)
    x = torch.cat([cls, x], dim=1)                                     # :311
    x = x + sd[prefix + "positional_embedding"].to(dt)                 # :312
    x = layer_norm_clip(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"])  # :313
    return transformer(x, sd, prefix + "transformer", width // 64, None, dt)


def encode_image(image, sd, prefix="visual_encoder.visual.", mode="fp32"):
    """VisualEncoder.encode_image, modules/module_cross.py:222-237: ln_post @ proj, CLS row, .float()."""
    dt = _tower_dtype(mode)
    hidden = vit_hidden(image, sd, prefix, mode)
    hidden = layer_norm_clip(hidden, sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"]) @ sd[prefix + "proj"].to(dt)
    return hidden[:, 0, :].float()


def visual_encoder(video, sd, prefix="visual_encoder.", mode="fp32", use_temp=True):
    """VisualEncoder.forward, modules/module_cross.py:178-216.
    video [b,F,3,H,W] -> (video_emb [b,E], frame_output [b,F,E]) fp32."""
    b, f = video.shape[:2]
    u = encode_image(video.reshape(b * f, *video.shape[2:]), sd, prefix + "visual.", mode).reshape(b, f, -1)
    h = u
    if use_temp:
        e = u.shape[-1]
        h = u + sd[prefix + "frame_position_embeddings.weight"].float()[:f]       # :195-199
        n_head = _temporal_heads(sd, prefix, e)
        h = transformer(h, sd, prefix + "temporal_transformer", n_head, torch.zeros(f, f), torch.float32, tf_ln=True)
        h = h + u                                                                   # :207
    h = h / h.norm(dim=-1, keepdim=True)                                            # :210
    return h.mean(dim=1), u                                                         # :212


def _temporal_heads(sd, prefix, e):
    return 8 if e % 8 == 0 and e >= 64 else max(1, e // 64)   # cross_config.json:4


def encode_text(ids, sd, prefix="text_encoder.", mode="fp32", return_hidden=False):
    """TextEncoder.encode_text, modules/module_cross.py:287-305."""
    dt = _tower_dtype(mode)
    L = ids.shape[1]
    x = F.embedding(ids, sd[prefix + "token_embedding.weight"].float()).to(dt)     # :288
    x = x + sd[prefix + "positional_embedding"][:L].to(dt)                          # :290-291
    width = x.shape[-1]
    x = transformer(x, sd, prefix + "transformer", width // 64, causal_mask(L), dt)
    hidden = layer_norm_clip(x, sd[prefix + "ln_final.weight"], sd[prefix + "ln_final.bias"]).to(dt) \
        @ sd[prefix + "text_projection"].to(dt)                                     # :296
    feat = hidden[torch.arange(ids.shape[0]), ids.argmax(dim=-1)].float()           # :300
    return (feat, hidden.float()) if return_hidden else feat


# ----------------------------------------------------------------------------- fine-tune head

def loose_similarity(q, v, logit_scale=math.log(100.0)):
    """modules/modeling.py:207-229."""
    v = v / v.norm(dim=-1, keepdim=True)
    q = q / q.norm(dim=-1, keepdim=True)
    s = min(math.exp(logit_scale), 100.0)
    if v.dim() == 2:
        return s * (q @ v.t())
    return (s * torch.matmul(q, v.permute(0, 2, 1))).permute(1, 0, 2)   # [bq, bv, F]


def cross_en(sim):
    """modules/until_module.py:196-205."""
    return -torch.diag(F.log_softmax(sim, dim=-1)).mean()


def frame_loss(q, frames):
    """modules/modeling.py:665-672."""
    f = frames.shape[1]
    loss = 0.0
    for i in range(f):
        s = loose_similarity(q, frames[:, i, :])
        loss = loss + (cross_en(s) + cross_en(s.t())) / f
    return loss


def finetune_head(q, v, frames, w_vtm=0.85, w_ftm=0.15, use_frame_fea=True):
    """modules/modeling.py:702-709 on globally gathered features."""
    loss = 0.0
    if use_frame_fea:
        loss = loss + w_ftm * frame_loss(q, frames)
    s = loose_similarity(q, v)
    return loss + w_vtm * (cross_en(s) + cross_en(s.t()))


def finetune_loss(ids, video, sd, mode="fp32", use_temp=True, use_frame_fea=True):
    """BirdModel.forward, modules/modeling.py:682-722, world size 1."""
    q = encode_text(ids, sd, mode=mode)
    v, u = visual_encoder(video, sd, mode=mode, use_temp=use_temp)
    return finetune_head(q, v, u, use_frame_fea=use_frame_fea), (q, v, u)


# ----------------------------------------------------------------------------- eval scorer + metrics

def eval_scores(q, v, frames, top_frames):
    """main_task_retrieval.py:332-336,512-513: video-text logits, frame-text top-k mean."""
    sv = loose_similarity(q, v)
    sf = loose_similarity(q, frames)
    sf = torch.topk(sf, k=top_frames, dim=2)[0].mean(dim=2)
    return sv, sf


def compute_metrics(x):
    """metrics.py:12-39 (numpy)."""
    x = np.asarray(x)
    sx = np.sort(-x, axis=1)
    d = np.diag(-x)[:, np.newaxis]
    ind = np.where(sx - d == 0)[1]
    return {"R1": float(np.sum(ind == 0)) * 100 / len(ind), "R5": float(np.sum(ind < 5)) * 100 / len(ind),
            "R10": float(np.sum(ind < 10)) * 100 / len(ind), "MR": float(np.median(ind) + 1),
            "MedianR": float(np.median(ind) + 1), "MeanR": float(np.mean(ind) + 1), "ranks": ind}


# ----------------------------------------------------------------------------- pre-train heads

def contrastive_loss(q, k, queue, T=0.07):
    """modules/modeling.py:286-313."""
    q = F.normalize(q, dim=1)
    k = F.normalize(k, dim=1)
    l_pos = torch.diag(q @ k.t()).reshape(q.shape[0], 1)
    l_neg = q @ queue.clone().detach()
    logits = torch.cat([l_pos, l_neg], dim=1) / T
    return F.cross_entropy(logits, torch.zeros(q.shape[0], dtype=torch.long))


def frame_self_loss(fr, fr_k, queue, T=0.07):
    """modules/modeling.py:315-323 (FAM)."""
    n = fr.shape[1]
    loss = 0.0
    for i in range(n - 1):
        loss = loss + contrastive_loss(fr[:, i], fr_k[:, i + 1], queue, T) + contrastive_loss(fr[:, i + 1], fr_k[:, i], queue, T)
    return loss / (n - 1)


def frame_cross_loss(fr, fr_k, q_frame, txt, txt_k, q_txt, T=0.07):
    """modules/modeling.py:325-332 (FTM)."""
    n = fr.shape[1]
    loss = 0.0
    for i in range(n):
        loss = loss + contrastive_loss(txt, fr_k[:, i], q_frame, T) + contrastive_loss(fr[:, i], txt_k, q_txt, T)
    return loss / n


def mlp_forward(x, sd, p, training=True, eps=1e-5):
    """MLP, modules/modeling.py:788-807: Linear -> BatchNorm1d(train stats) -> ReLU -> Linear.
    Returns (y, batch_mean, batch_var_unbiased) so running stats can be checked."""
    h = F.linear(x, sd[p + "linear_hidden.1.weight"], sd[p + "linear_hidden.1.bias"])
    if training:
        mean = h.mean(0)
        var = h.var(0, unbiased=False)
    else:
        mean, var = sd[p + "linear_hidden.2.running_mean"], sd[p + "linear_hidden.2.running_var"]
    hn = (h - mean) / torch.sqrt(var + eps) * sd[p + "linear_hidden.2.weight"] + sd[p + "linear_hidden.2.bias"]
    y = F.linear(torch.relu(hn), sd[p + "linear_out.weight"], sd[p + "linear_out.bias"])
    n = h.shape[0]
    return y, mean, var * n / max(n - 1, 1)


def momentum_update(p_k, p, m):
    """modules/modeling.py:238-242: three tensor ops in the parameter's dtype."""
    return p_k * m + p * (1.0 - m)


def enqueue(queues, ptr, v_k, tag_k, title_k, frame_k, frame_proj_k, K):
    """modules/modeling.py:244-284 on already-gathered keys.  queues: dict name->[E, *] (modified in place)."""
    v_k, tag_k, title_k = F.normalize(v_k, dim=1), F.normalize(tag_k, dim=1), F.normalize(title_k, dim=1)
    frame_k, frame_proj_k = F.normalize(frame_k, dim=2), F.normalize(frame_proj_k, dim=2)
    B, Fr = v_k.shape[0], frame_k.shape[1]
    queues["queue_v_cross_ng"][:, ptr:ptr + B] = v_k.T
    queues["queue_tag_cross_ng"][:, ptr:ptr + B] = tag_k.T
    queues["queue_title_cross_ng"][:, ptr:ptr + B] = title_k.T
    queues["queue_frame_proj_ng"][:, ptr * Fr:(ptr + B) * Fr] = frame_proj_k.reshape(-1, frame_k.shape[-1]).T
    queues["queue_frame_cross_ng"][:, ptr * Fr:(ptr + B) * Fr] = frame_k.reshape(-1, frame_k.shape[-1]).T
    return (ptr + B) % K


def gelu_erf(x):
    """modules/module_cross.py:33-39."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mlm_head(hidden, sd, p="cls."):
    """BertLMPredictionHead, modules/module_cross.py:308-357."""
    h = F.linear(hidden, sd[p + "transform.dense.weight"], sd[p + "transform.dense.bias"])
    h = layer_norm_tf(gelu_erf(h), sd[p + "transform.LayerNorm.weight"], sd[p + "transform.LayerNorm.bias"], 1e-12)
    return F.linear(h, sd[p + "decoder.weight"], sd[p + "bias"])


def mlm_apply_mask(ids, masked, replaced, rand_sel, random_words):
    """modules/modeling.py:181-205 with the four random draws passed in.
    masked/replaced/rand_sel: bool draws of bernoulli(p), bernoulli(0.8), bernoulli(0.5)."""
    ids = ids.clone()
    labels = ids.clone()
    masked = masked.clone()
    masked[ids == 49407] = False     # tokenizer.pad_token_id is EOT (tokenization_clip.py)
    masked[ids == 49406] = False     # cls_token_id = SOT
    labels[~masked] = -100
    rep = replaced & masked
    ids[rep] = 49394
    rnd = rand_sel & masked & ~rep
    ids[rnd] = random_words[rnd]
    return ids, labels


def pretrain_loss(batch, sd, queues, ptr, K, m=0.99, T=0.07, mode="fp32", mlm_draws=None,
                  weights=(0.05, 0.45, 0.45, 0.05), use_frame_fea=True):
    """BirdPreTrainedModel.forward, modules/modeling.py:334-436, dataset != 'bird', world size 1.
    Mutates sd (the *_k entries: EMA; BN running stats) and queues in place; returns
    (loss, parts, new_ptr)."""
    video, _, tag_ids, _, title_ids, _ = batch
    v_fea, frame_fea = visual_encoder(video, sd, "visual_encoder.", mode)
    title_fea = encode_text(title_ids, sd, "text_encoder.", mode)
    b, f, e = frame_fea.shape
    frame_proj, mu1, var1 = mlp_forward(frame_fea.reshape(-1, e), sd, "v_projector.")
    frame_pred, mu2, var2 = mlp_forward(frame_proj, sd, "v_predictor.")
    frame_proj, frame_pred = frame_proj.reshape(b, f, e), frame_pred.reshape(b, f, e)
    with torch.no_grad():
        for pair in ("visual_encoder", "text_encoder", "v_projector", "t_projector"):
            for k in list(sd.keys()):
                if k.startswith(pair + ".") and "running_" not in k and "num_batches" not in k:
                    kk = k.replace(pair + ".", pair + "_k.", 1)
                    if mode == "aswritten" and _stored_fp16(k):
                        sd[kk] = momentum_update(sd[kk].half(), sd[k].half(), m).float()
                    else:
                        sd[kk] = momentum_update(sd[kk], sd[k], m)
        tag_fea_k = encode_text(tag_ids, sd, "text_encoder_k.", mode)
        title_fea_k = encode_text(title_ids, sd, "text_encoder_k.", mode)
        v_fea_k, frame_fea_k = visual_encoder(video, sd, "visual_encoder_k.", mode)
        frame_proj_k, mu3, var3 = mlp_forward(frame_fea_k.reshape(-1, e), sd, "v_projector_k.")
        frame_proj_k = frame_proj_k.reshape(b, f, e)
        for p, mu, var in (("v_projector.", mu1, var1), ("v_predictor.", mu2, var2), ("v_projector_k.", mu3, var3)):
            sd[p + "linear_hidden.2.running_mean"] = 0.9 * sd[p + "linear_hidden.2.running_mean"] + 0.1 * mu.detach()
            sd[p + "linear_hidden.2.running_var"] = 0.9 * sd[p + "linear_hidden.2.running_var"] + 0.1 * var.detach()
            sd[p + "linear_hidden.2.num_batches_tracked"] = sd[p + "linear_hidden.2.num_batches_tracked"] + 1
    fam = frame_self_loss(frame_pred, frame_proj_k, queues["queue_frame_proj_ng"], T)
    vtm = contrastive_loss(v_fea, title_fea_k, queues["queue_title_cross_ng"], T) + \
        contrastive_loss(title_fea, v_fea_k, queues["queue_v_cross_ng"], T)
    ftm = frame_cross_loss(frame_fea, frame_fea_k, queues["queue_frame_cross_ng"], title_fea, title_fea_k,
                           queues["queue_title_cross_ng"], T) if use_frame_fea else 0.0
    with torch.no_grad():
        new_ptr = enqueue(queues, ptr, v_fea_k, tag_fea_k, title_fea_k, frame_fea_k, frame_proj_k, K)
    masked_ids, labels = mlm_apply_mask(title_ids, *mlm_draws)
    _, hidden = encode_text(masked_ids, sd, "text_encoder.", mode, return_hidden=True)
    scores = mlm_head(hidden, sd)
    mlm = F.cross_entropy(scores.reshape(-1, scores.shape[-1]), labels.reshape(-1), ignore_index=-100)
    w = weights
    loss = w[0] * fam + w[1] * vtm + w[2] * ftm + w[3] * mlm
    return loss, (fam, vtm, ftm, mlm), new_ptr


def _stored_fp16(key):
    """Which state_dict entries convert_weights casts to fp16 (modules/module_clip.py:506-527)."""
    if ".temporal_transformer." in key or "frame_position_embeddings" in key or "projector" in key or "predictor" in key:
        return False
    if key.endswith(("conv1.weight", ".proj", "text_projection")):
        return True
    if ".resblocks." in key and (".attn." in key or ".mlp." in key):
        return True
    return False


# ----------------------------------------------------------------------------- optimizer

def warmup_cosine(x, warmup=0.002):
    """modules/optimization.py:26-29."""
    return x / warmup if x < warmup else 0.5 * (1.0 + math.cos(math.pi * x))


def bert_adam_step(p, g, m, v, step, lr, t_total, warmup, wd, b1=0.9, b2=0.98, e=1e-6, max_grad_norm=1.0):
    """BertAdam.step for one tensor, modules/optimization.py:103-168, evaluated op by op in the
    tensor's own dtype exactly as the reference's tensor expressions do.  Returns new (p, m, v, g)."""
    if max_grad_norm > 0:
        # clip_grad_norm_ on this tensor alone (:135-136).  torch evaluates the norm and the clip
        # coefficient in the gradient's own dtype: for fp16 both are rounded to fp16.
        norm = torch.linalg.vector_norm(g, 2.0)
        coef = torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0)
        g = g * coef
    m = m.mul(b1).add(g, alpha=1 - b1)
    v = v.mul(b2).addcmul(g, g, value=1 - b2)
    update = m / (v.sqrt() + e)
    if wd > 0.0:
        update = update + wd * p
    lr_t = lr * warmup_cosine(step / t_total, warmup) if t_total != -1 else lr
    p = p + (-(lr_t * update))
    return p, m, v, g
