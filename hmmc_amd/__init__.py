"""hmmc_amd — MI355X-native (gfx950) implementation of the HMMC training hot path behind the
reference's own module / optimizer API.  Importing the package is CPU-safe; anything that computes
needs libhmmc_hip.so and a GPU and fails loudly otherwise (no CPU or eager fallback)."""

__all__ = ["BirdModel", "BirdPreTrainedModel", "MLP", "BertAdam", "clip_grad_norm_", "dist_collect"]


def __getattr__(name):
    if name in ("BirdModel", "BirdPreTrainedModel", "MLP", "dist_collect"):
        from . import modeling
        return getattr(modeling, name)
    if name in ("BertAdam", "clip_grad_norm_"):
        from . import optimization
        return getattr(optimization, name)
    raise AttributeError(name)
