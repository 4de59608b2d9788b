"""Two ranks (processes) of the fine-tune step on the one available GPU, DDP over gloo: the packed feature
all-gather, its reduce-scatter backward and DDP's gradient averaging must reproduce the single-process
global-batch loss and gradients.  RCCL itself needs one GPU per rank; the collective
semantics exercised here are backend-independent."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from hmmc_amd import synth  # noqa: E402
from test_gpu_model import task_config  # noqa: E402

KEYS = ["text_encoder.text_projection", "visual_encoder.visual.conv1.weight", "visual_encoder.visual.proj",
        "visual_encoder.visual.transformer.resblocks.1.mlp.c_fc.weight", "text_encoder.token_embedding.weight",
        "visual_encoder.temporal_transformer.resblocks.0.attn.in_proj_weight", "visual_encoder.frame_position_embeddings.weight"]


def _run(rank, world, store, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if world > 1:
        dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    from hmmc_amd.modeling import BirdModel
    torch.cuda.set_device(0)
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY),
                                      task_config=task_config(rank=rank)).cuda().train()
    for m in model.modules():                  # exercise the per-run autograd nodes the towers use under DDP
        if hasattr(m, "ddp_layers_per_node"):
            m.ddp_layers_per_node = 1
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0]) if world > 1 else model
    B = 4
    b = B // world
    ids, mask, vid, vf, idx = [t[rank * b:(rank + 1) * b].cuda() for t in synth.finetune_batch(B, 4, 32, tag="ddp")]
    loss = net(ids, mask, vid, vf, idx, 1)
    loss.backward()
    torch.cuda.synchronize()
    P = dict(model.named_parameters())
    torch.save({"loss": loss.detach().cpu(), **{k: P[k].grad.float().cpu() for k in KEYS}}, os.path.join(out_dir, f"w{world}r{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run, args=(1, os.path.join(d, "s1"), d), nprocs=1, join=True)
        mp.spawn(_run, args=(2, os.path.join(d, "s2"), d), nprocs=2, join=True)
        ref = torch.load(os.path.join(d, "w1r0.pt"))
        outs = [torch.load(os.path.join(d, f"w2r{r}.pt")) for r in range(2)]
    for o in outs:
        assert abs(float(o["loss"]) - float(ref["loss"])) < 2e-3, (float(o["loss"]), float(ref["loss"]))
        for k in KEYS:
            a, b = o[k].flatten(), ref[k].flatten()
            cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20))
            ratio = float(a.norm() / (b.norm() + 1e-20))
            assert cos > 0.995 and abs(ratio - 1) < 0.03, (k, cos, ratio)
    for k in KEYS:      # DDP left identical gradients on both ranks
        assert torch.equal(outs[0][k], outs[1][k]), k
