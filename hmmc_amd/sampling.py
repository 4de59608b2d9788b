"""Frame sampling of the reference's loaders (dataloaders/dataloader_msrvtt_retrieval.py:296-312,
dataloader_bird.py:170-188) with the selected frames gathered ON THE DEVICE: the clip's stored frames live in HBM as
uint8 [videos, stored, 3, H, W]; the policy picks `frames` indices per video on the host exactly as the reference does
(same calls into Python's `random` / numpy, so a seeded run draws the same indices), and hmmc_patchify_u8 reads the chosen
frames in place through a device index (no [videos, frames, 3, H, W] copy, no fp32 frames)."""
import random as _random

import numpy as np
import torch

POLICIES = ("uniform", "random", "uniform_random")


def frame_indices(frame_sample, stored_frames, frames, rng=_random):
    """The `sample_slice` of _get_rawvideo for one video: `frames` indices into its `stored_frames` stored frames."""
    if frame_sample == "uniform_random":
        video_index = list(np.arange(0, stored_frames))
        k = stored_frames // frames
        return [int(rng.sample(video_index[k * i:k * (i + 1)], 1)[0]) for i in np.arange(frames)]
    if frame_sample == "random":
        video_index = list(np.arange(0, stored_frames))
        return [int(i) for i in sorted(rng.sample(video_index, frames))]
    # every other value of --frame_sample falls through to uniform in the reference too
    return [int(i) for i in np.linspace(0, stored_frames, frames, endpoint=False, dtype=int)]


def batch_frame_index(frame_sample, videos, stored_frames, frames, device, rng=_random):
    """int32 [videos, frames] on `device`: index of every sampled frame within the flattened [videos * stored] frame list."""
    idx = np.asarray([[v * stored_frames + i for i in frame_indices(frame_sample, stored_frames, frames, rng)]
                      for v in range(videos)], dtype=np.int32)
    return torch.from_numpy(idx).to(device)
