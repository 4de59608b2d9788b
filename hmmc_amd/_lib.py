"""ctypes binding of libhmmc_hip.so — the C-ABI declared in include/hmmc_hip.h.

The product path has no CPU fallback: if the library is missing, `load()` raises.
PyTorch only supplies device memory (tensor.data_ptr()) and the current HIP stream.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMMC_LIB") or os.path.join(_HERE, "libhmmc_hip.so")     # HMMC_LIB: a variant build (A/B runs, scratch/)

_C = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_long, "f": ctypes.c_float, "z": ctypes.c_size_t, "s": ctypes.c_char_p}

# name -> (argument codes, return code); must match include/hmmc_hip.h
SIGNATURES = {
    "hmmc_gemm_f16_workspace": ("iii", "z"),
    "hmmc_gemm_f16_colsum_rows": ("iii", "z"),
    "hmmc_gemm_f16": ("pppiiiiiiiippppipzp", "i"),
    "hmmc_gemm_f16_fold": ("pppiiiiiiippppippppzp", "i"),
    "hmmc_layernorm_bwd_fold_rows": ("i", "i"),
    "hmmc_layernorm_bwd_fold": ("ppppppiiilp", "i"),
    "hmmc_fold_grad_finish": ("ppppppppppiip", "i"),
    "hmmc_fold_grad_scratch_floats": ("ii", "z"),
    "hmmc_attention_f16_bwd_scaled": ("pppppppiiiip", "i"),
    "hmmc_tower_bwd_fold": ("pppppppzpiiiiiiiipzpp", "i"),
    "hmmc_ln_fold_prep": ("pppppppiip", "i"),
    "hmmc_rowstat": ("ppiilfp", "i"),
    "hmmc_rowstat_finalize": ("ppiiifp", "i"),
    "hmmc_tower_fold_bytes": ("liii", "z"),
    "hmmc_tower_act_bytes_fold": ("liiii", "z"),
    "hmmc_tower_fwd_fused": ("pppppiiiiiiifiipzp", "i"),
    "hmmc_vit_embed_ln": ("pppppppppiiifip", "i"),
    "hmmc_gemm_f16_wgrad_group_workspace": ("ppii", "z"),
    "hmmc_gemm_f16_wgrad_group": ("ppppppiipzp", "i"),
    "hmmc_gemm_reserve_cus": ("i", "i"),
    "hmmc_gemm_profile_start": ("", "i"),
    "hmmc_gemm_profile_stop": ("pppp", "i"),
    "hmmc_layernorm_fwd": ("pppppppiilfip", "i"),
    "hmmc_layernorm_bwd_workspace": ("ii", "z"),
    "hmmc_layernorm_bwd": ("pppppppppppiilipzp", "i"),
    "hmmc_layernorm_bwd_rows": ("i", "i"),
    "hmmc_layernorm_bwd_partial": ("ppppppppipiilip", "i"),
    "hmmc_multi_colreduce": ("pip", "i"),
    "hmmc_colsum_workspace": ("ii", "z"),
    "hmmc_colsum": ("ppiiliiipzp", "i"),
    "hmmc_patchify_u8": ("pppiiiippip", "i"),
    "hmmc_patchify": ("ppiiiiip", "i"),
    "hmmc_vit_embed": ("pppliiip", "i"),
    "hmmc_text_embed": ("ppppliilpip", "i"),
    "hmmc_text_embed_bwd": ("ppplilip", "i"),
    "hmmc_eot_index": ("ppiillp", "i"),
    "hmmc_cast": ("pplip", "i"),
    "hmmc_attention_f16_fwd": ("pppiiiip", "i"),
    "hmmc_attention_f16_bwd": ("ppppppiiiip", "i"),
    "hmmc_attention_f16_fwd_lead": ("pppiiiip", "i"),
    "hmmc_attention_f16_bwd_lead": ("pppppppiiiip", "i"),
    "hmmc_gemm_f32": ("pppiiillllifppppip", "i"),
    "hmmc_l2norm_fwd": ("pppiifp", "i"),
    "hmmc_l2norm_bwd": ("ppppiip", "i"),
    "hmmc_infonce_fwd": ("ppppiiffp", "i"),
    "hmmc_infonce_bwd": ("pppppiiffp", "i"),
    "hmmc_retrieval_rank": ("pppiilip", "i"),
    "hmmc_topk_mean": ("pppiiiillp", "i"),
    "hmmc_segment_max": ("pppiilp", "i"),
    "hmmc_eval_slots": ("i", "i"),
    "hmmc_eval_pack": ("pppiiip", "i"),
    "hmmc_eval_score": ("pppppiiiiifp", "i"),
    "hmmc_temporal_pool_fwd": ("ppppiiip", "i"),
    "hmmc_temporal_pool_bwd": ("pppppiiip", "i"),
    "hmmc_add_rowbias": ("pppliip", "i"),
    "hmmc_temporal_attention_fwd": ("pppiiiip", "i"),
    "hmmc_temporal_attention_bwd": ("ppppiiip", "i"),
    "hmmc_mt_chunk_elems": ("", "i"),
    "hmmc_mt_sumsq": ("ppipip", "i"),
    "hmmc_mt_clip_grad_norm": ("ppipifpp", "i"),
    "hmmc_mt_bertadam": ("ppipipip", "i"),
    "hmmc_mt_clip_grad_norm_keep": ("ppipifppp", "i"),
    "hmmc_mt_bertadam_ext": ("ppipiippp", "i"),
    "hmmc_mt_ema": ("ppiffp", "i"),
    "hmmc_enqueue": ("ppiillp", "i"),
    "hmmc_bn_finalize": ("ppfffppppppip", "i"),
    "hmmc_bn_workspace": ("ii", "z"),
    "hmmc_bn_stats": ("ppiipzp", "i"),
    "hmmc_bn_apply_relu": ("ppppppli p".replace(" ", ""), "i"),
    "hmmc_bn_bwd_reduce": ("ppppppiipzp", "i"),
    "hmmc_bn_bwd_apply": ("pppppppplifp", "i"),
    "hmmc_rowdot": ("pppiip", "i"),
    "hmmc_moco_loss_fwd": ("pppppilffp", "i"),
    "hmmc_moco_loss_bwd": ("pppppilffp", "i"),
    "hmmc_row_axpy": ("ppplip", "i"),
    "hmmc_gelu_erf_fwd": ("pplp", "i"),
    "hmmc_gelu_erf_bwd": ("ppplp", "i"),
    "hmmc_ce_fwd": ("ppppppilp", "i"),
    "hmmc_ce_bwd": ("pppppilp", "i"),
    "hmmc_tower_act_bytes": ("liiiii", "z"),
    "hmmc_tower_bwd_scratch_bytes": ("lii", "z"),
    "hmmc_tower_workspace_bytes": ("liiii", "z"),
    "hmmc_tower_fwd": ("ppppiiiiiiifiipzp", "i"),
    "hmmc_tower_bwd": ("pppppppiiiiiiiipzpp", "i"),
    "hmmc_tower_release": ("pp", "i"),
    "hmmc_set_option": ("si", "i"),
    "hmmc_get_option": ("s", "i"),
}

# The library reads no environment variable itself (include/hmmc_hip.h: hmmc_set_option).  These A/B switches of scratch/ runs
# and tests are translated once, when the library is loaded; a variable that is PRESENT (any value, the empty string included)
# switches its option on - the same rule hmmc_amd/functional.py applies on its side.
ENV_OPTIONS = {"HMMC_NO_WGRAD_GROUP": "no_wgrad_group", "HMMC_NO_F32_WAVEK": "no_f32_wavek", "HMMC_NO_F32_DMA": "no_f32_dma",
               "HMMC_NO_LEAD_ATTN": "no_lead_attn"}


def set_option(key, value):
    """hmmc_set_option(key, value): process-wide A/B switch of the library (see ENV_OPTIONS for the keys)."""
    rc = load().hmmc_set_option(key.encode(), int(bool(value)))
    if rc != 0:
        raise KeyError(f"hmmc_set_option: unknown option {key!r}")


def get_option(key):
    rc = load().hmmc_get_option(key.encode())
    if rc < 0:
        raise KeyError(f"hmmc_get_option: unknown option {key!r}")
    return bool(rc)

ERRORS = {-1: "invalid argument", -2: "unsupported shape/alignment", -3: "workspace too small", -4: "kernel launch failed"}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m hmmc_amd.build` (hipcc, gfx950). "
                "hmmc_amd has no CPU or eager fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (args, ret) in SIGNATURES.items():
            fn = getattr(lib, name)            # AttributeError here = header/library mismatch
            fn.argtypes = [_C[c] for c in args]
            fn.restype = _C[ret]
        _lib = lib
        for var, key in ENV_OPTIONS.items():
            if var in os.environ:
                lib.hmmc_set_option(key.encode(), 1)
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """Call an int-returning entry point on the current stream; raise on a non-zero status."""
    rc = getattr(load(), name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(rc, rc)}")


def query(name, *args):
    return getattr(load(), name)(*args)
