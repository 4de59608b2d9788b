"""Checkpoint compatibility with the reference's pytorch_model.bin.N (SURVEY.md 8(b) state-dict table, 8(f) rank 4):
the state_dict of the drop-in models must carry exactly the reference's keys, shapes and dtypes, and the same set of
trainable parameters.  The manifest was generated from the reference itself (tests/golden/make_golden.py --only manifest).
CPU-only: constructing the modules launches no kernel."""
import json
import os

import pytest
import torch

from hmmc_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
MANIFEST = json.load(open(os.path.join(HERE, "golden", "statedict_manifest.json")))


def _task_config(**kw):
    from argparse import Namespace
    d = dict(local_rank=0, rank=0, use_temp=True, language="english", top_frames=2, max_frames=4, n_display=100000,
             logdir=None, use_frame_fea=True, dataset="msrvtt", contrast_momentum=0.99, contrast_temperature=0.07,
             contrast_num_negative=16, pretrained_text=None, lr=1e-4, text_lr=3e-5, coef_lr=1e-3, weight_decay=0.2,
             warmup_proportion=0.1)
    d.update(kw)
    return Namespace(**d)


CASES = [("finetune_tiny", "BirdModel", lambda: synth.finetune_state(synth.TINY), {}),
         ("finetune_b32", "BirdModel", lambda: synth.finetune_state(synth.VIT_B32), {"max_frames": 12}),
         ("pretrain_tiny", "BirdPreTrainedModel", lambda: synth.pretrain_state(synth.TINY, 16, 4),
          {"dataset": "chvtt", "contrast_num_negative": 16})]


@pytest.mark.parametrize("tag,cls,make_sd,tc", CASES)
def test_state_dict_layout_matches_reference(tag, cls, make_sd, tc):
    import hmmc_amd.modeling as M
    model = getattr(M, cls).from_pretrained("cross-base", state_dict=make_sd(), task_config=_task_config(**tc))
    ref = MANIFEST[tag]
    mine = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()}
    assert set(mine) == set(ref), (sorted(set(mine) - set(ref))[:8], sorted(set(ref) - set(mine))[:8])
    bad = [(k, mine[k], ref[k]) for k in ref if mine[k] != ref[k]]
    assert not bad, bad[:8]
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    assert trainable == MANIFEST[tag + ":trainable"]


def test_checkpoint_file_round_trip(tmp_path):
    """torch.save(model.state_dict()) -> pytorch_model.bin.N -> from_pretrained(state_dict=torch.load(...)) restores
    every tensor bit for bit (main_task_retrieval.py:215-222,149-153)."""
    import hmmc_amd.modeling as M
    sd = synth.finetune_state(synth.TINY)
    model = M.BirdModel.from_pretrained("cross-base", state_dict=sd, task_config=_task_config())
    path = tmp_path / "pytorch_model.bin.3"
    torch.save(model.state_dict(), path)
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    again = M.BirdModel.from_pretrained("cross-base", state_dict=loaded, task_config=_task_config())
    a, b = model.state_dict(), again.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
