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


# ---- multi-sentence retrieval (several captions per video: MSVD, VATEX) -------------------------------------------------

def _groups(cut_off_points, n_sentences):
    """cut_off_points = index of the LAST sentence of every video, as eval_epoch hands them to logging_rank
    (main_task_retrieval.py:378-381 subtracts 1 from the dataset's running sentence count) -> (offsets [G + 1],
    video id per sentence)."""
    ends = [int(c) + 1 for c in cut_off_points]
    starts = [0] + ends[:-1]
    if not ends or ends[-1] != n_sentences or any(e <= s for s, e in zip(starts, ends)):
        raise ValueError("cut_off_points must be increasing and end at the last sentence")
    vid = np.repeat(np.arange(len(ends)), [e - s for s, e in zip(starts, ends)])
    return np.asarray(starts + [ends[-1]], dtype=np.int32), vid.astype(np.int32)


def tensor_text_to_video_metrics(sim_matrix, cut_off_points, top_k=(1, 5, 10)):
    """Text -> video metrics of multi-sentence retrieval (metrics.py:49-77).  The reference pads the [sentences, videos]
    matrix to [videos, max_captions, videos] with -inf and takes a double argsort; the rank of sentence s is the position of
    ITS video in s's row, which is what hmmc_retrieval_rank(target = video of s) counts on the unpadded matrix.
    MedianR is torch.median's (the lower middle element), as in the reference."""
    _, vid = _groups(cut_off_points, sim_matrix.shape[0])
    r = ranks(sim_matrix, target=torch.from_numpy(vid).to(sim_matrix.device))
    res = {f"R{k}": float(np.sum(r < k) * 100 / len(r)) for k in top_k}
    res["MedianR"] = float(np.sort(r + 1)[(len(r) - 1) // 2])
    res["MeanR"] = float(np.mean(r + 1))
    res["Std_Rank"] = float(np.std(r + 1))
    res["MR"] = res["MedianR"]
    return res


def video_to_text_metrics(sim_matrix, cut_off_points):
    """Video -> text metrics of multi-sentence retrieval: compute_metrics(tensor_video_to_text_sim(...)) (metrics.py:79-86,
    112): a video's score against caption group g is its best caption of g; ground truth is the diagonal."""
    off, _ = _groups(cut_off_points, sim_matrix.shape[0])
    sim = sim_matrix.float().contiguous()
    best = ops.segment_max(sim, torch.from_numpy(off).to(sim.device))          # [groups, videos]
    return metrics_from_ranks(ops.retrieval_rank(best, transposed=True).cpu().numpy().astype(np.int64))


def logging_rank(sim_matrix, multi_sentence_, cut_off_points_, logger=None):
    """(tv_metrics, vt_metrics) of a device-resident similarity matrix, as metrics.py:89-144 logs them; the reference
    returns tv_metrics only."""
    if multi_sentence_:
        tv = tensor_text_to_video_metrics(sim_matrix, cut_off_points_)
        vt = video_to_text_metrics(sim_matrix, cut_off_points_)
    else:
        tv, vt = compute_metrics_t2v_v2t(sim_matrix)
    if logger is not None:
        logger.info("Text-to-Video:")
        logger.info('\t>>>  R@1: {:.1f} - R@5: {:.1f} - R@10: {:.1f} - Median R: {:.1f} - Mean R: {:.1f}'.format(
            tv['R1'], tv['R5'], tv['R10'], tv['MR'], tv['MeanR']))
        logger.info("Video-to-Text:")
        logger.info('\t>>>  V2T$R@1: {:.1f} - V2T$R@5: {:.1f} - V2T$R@10: {:.1f} - V2T$Median R: {:.1f} - V2T$Mean R: {:.1f}'.format(
            vt['R1'], vt['R5'], vt['R10'], vt['MR'], vt['MeanR']))
    return tv, vt
