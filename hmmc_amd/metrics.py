"""Retrieval rank metrics of the reference (metrics.py:12-39 compute_metrics, main_task_retrieval.py:512-525), with the
ranking itself on the device: the [Q, V] similarity matrix stays in HBM, `hmmc_retrieval_rank` counts per query how many
candidates beat the ground-truth item, and only the Q integer ranks come back for R@K / median / mean (host arithmetic on
Q numbers, as in the reference)."""
import numpy as np
import torch

from . import ops


def ranks(sim_matrix, transposed=False, target=None):
    """0-based rank of the ground-truth item per query (the `ind` array of metrics.py:20-28), int64 numpy."""
    if not torch.is_tensor(sim_matrix) or not sim_matrix.is_cuda:
        raise RuntimeError("hmmc_amd.metrics ranks a similarity matrix that lives on the GPU (no CPU fallback)")
    sim = sim_matrix.float().contiguous()
    return ops.retrieval_rank(sim, target=target, transposed=transposed).cpu().numpy().astype(np.int64)


def metrics_from_ranks(ind):
    ind = np.asarray(ind)
    return {"R1": float(np.sum(ind == 0)) * 100 / len(ind), "R5": float(np.sum(ind < 5)) * 100 / len(ind),
            "R10": float(np.sum(ind < 10)) * 100 / len(ind), "MR": float(np.median(ind) + 1),
            "MedianR": float(np.median(ind) + 1), "MeanR": float(np.mean(ind) + 1)}


def compute_metrics(sim_matrix):
    """Text -> video metrics of a [n_text, n_video] matrix whose ground truth is the diagonal (metrics.py:12-39)."""
    return metrics_from_ranks(ranks(sim_matrix))


def compute_metrics_t2v_v2t(sim_matrix):
    """(tv_metrics, vt_metrics) as main_task_retrieval.py:512-513 computes them (sim_matrix and sim_matrix.T)."""
    return metrics_from_ranks(ranks(sim_matrix)), metrics_from_ranks(ranks(sim_matrix, transposed=True))
