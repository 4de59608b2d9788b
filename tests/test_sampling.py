"""CPU test of the frame-sampling policies against the indices the reference's own loader drew
(tests/golden/frame_sampling.npz: MSRVTT_TrainDataLoader._get_rawvideo driven under seeded `random`,
dataloaders/dataloader_msrvtt_retrieval.py:296-312)."""
import random

import numpy as np

from conftest import golden
from hmmc_amd import sampling


def test_frame_indices_equal_the_reference_loader():
    g = golden("frame_sampling")
    keys = [k for k in g.files if not k.startswith("_")]
    assert len(keys) == 36
    for key in keys:
        policy, stored, frames, seed = key.rsplit(".", 3)
        random.seed(int(seed))
        got = sampling.frame_indices(policy, int(stored), int(frames))
        assert got == [int(x) for x in g[key]], key
    # an unknown policy falls through to uniform, as in the reference
    assert sampling.frame_indices("whatever", 30, 12) == [int(x) for x in np.linspace(0, 30, 12, endpoint=False, dtype=int)]
