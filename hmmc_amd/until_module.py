"""Loss / LayerNorm / weight-loading utilities with the reference's names
(reference: modules/until_module.py:54-67 LayerNorm, :104-160 init_preweight, :196-205 CrossEn)."""
from __future__ import annotations

import logging

import torch
from torch import nn

logger = logging.getLogger(__name__)


class LayerNorm(nn.Module):
    """TF-style LayerNorm parameter holder (eps inside the sqrt, 1e-12); the temporal blocks run it
    through hmmc_layernorm_fwd/bwd (reference modules/until_module.py:54-67)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        from . import ops
        shape = x.shape
        y, _, _ = ops.layernorm_fwd(x.contiguous().view(-1, shape[-1]), self.weight, self.bias, self.variance_epsilon)
        return y.view(shape)


class CrossEn(nn.Module):
    """-mean(diag(log_softmax(S, -1))) (reference modules/until_module.py:196-205), evaluated by the
    fused InfoNCE kernels; used directly only outside the fused BirdModel.forward."""

    def forward(self, sim_matrix):
        from . import ops
        S = sim_matrix.contiguous().float()
        B = S.shape[0]
        assert S.shape[1] == B
        _, lse_row, _ = ops.infonce_fwd(S, B, 0, 1.0, 0.0)
        return (lse_row.view(-1) - torch.diagonal(S)).mean()


class PreTrainedModel(nn.Module):
    """Checkpoint -> module loading with the reference's semantics (modules/until_module.py:104-160):
    gamma/beta are renamed to weight/bias, tensors are copied into the existing parameters (so an
    fp32 checkpoint lands in fp16 tower weights), missing / unexpected keys are logged, not fatal."""

    def __init__(self, config=None, *inputs, **kwargs):
        super().__init__()
        self.config = config

    @classmethod
    def init_preweight(cls, model, state_dict, prefix=None, task_config=None):
        sd = {}
        for key, val in state_dict.items():
            nk = key.replace("gamma", "weight") if "gamma" in key else key
            nk = nk.replace("beta", "bias") if "beta" in nk else nk
            sd[(prefix or "") + nk] = val
        result = model.load_state_dict(sd, strict=False)
        if prefix is None and (task_config is None or getattr(task_config, "local_rank", 0) == 0):
            if result.missing_keys:
                logger.info("Weights of %s not initialized from pretrained model: %s", model.__class__.__name__,
                            "\n   ".join(result.missing_keys))
            if result.unexpected_keys:
                logger.info("Weights from pretrained model not used in %s: %s", model.__class__.__name__,
                            "\n   ".join(result.unexpected_keys))
        return model
