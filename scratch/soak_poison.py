"""Uninitialised-memory soak: run training steps once on fresh memory, then again after the caching allocator's free blocks
have been filled with 0xFF bytes (fp16 / fp32 NaN, pointer -1), from identical state.  A kernel that reads a buffer before
writing it (scratch slabs, partial matrices, padding rows) shows up as a NaN or as a difference; equal results mean every byte
that is read was written in the same step.  usage: python scratch/soak_poison.py [ft|pt|b16] [gib to poison]"""
import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import task_config, prep_optimizer
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
from hmmc_amd.optimization import clip_grad_norm_
DEV = "cuda"
kind = sys.argv[1] if len(sys.argv) > 1 else "ft"
gib = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if kind == "pt":
    dims = synth.VIT_B32
    cfg = task_config(max_frames=4, pretrained_clip_name="ViT-B/32", dataset="chvtt", contrast_momentum=0.99, contrast_temperature=0.07,
                      contrast_num_negative=64, pretrained_text=None)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(dims, 64, 4), task_config=cfg).to(DEV).train()
    vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(16, 4, tag="soak")]
    args = lambda i: (vid, vf, tg, gm, ti, tm, i)
else:
    name = "ViT-B/16" if kind == "b16" else "ViT-B/32"
    dims = synth.NAMED[name]
    cfg = task_config(max_frames=6, pretrained_clip_name=name)
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(dims), task_config=cfg).to(DEV).train()
    batch = [t.to(DEV) for t in synth.finetune_batch(24 if kind == "ft" else 6, 6, 32, dims.image_res, tag="soak")]
    args = lambda i: (*batch, i)
sd0 = copy.deepcopy(model.state_dict())
params = [p for p in model.parameters() if p.requires_grad]

def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = []
    for _ in range(gib):
        b = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
        b.fill_(0xFF)
        blocks.append(b)
    for size in (512, 4096, 65536, 524288):        # the small-block pool (< 1 MiB requests) has its own cache
        blocks += [torch.full((size,), 0xFF, dtype=torch.uint8, device=DEV) for _ in range(768)]
    torch.cuda.synchronize()
    del blocks                      # the poisoned blocks stay in the allocator's cache and are split for the next requests

def run(poisoned):
    torch.manual_seed(0)
    model.load_state_dict(sd0)
    if hasattr(model, "_queue_ptr_host"): model._queue_ptr_host = None
    opt = prep_optimizer(model, cfg, t_total=100)
    losses = []
    for i in range(3):
        if poisoned:
            ops._ws_cache.clear()   # the grow-only workspaces too: they are re-created from poisoned memory
            poison()
        loss = model(*args(i + 1))
        loss.backward()
        clip_grad_norm_(params, 1.0)
        opt.step(); opt.zero_grad()
        losses.append(float(loss))
    torch.cuda.synchronize()
    ops.raise_on_device_errors()
    return losses, [p.detach().clone() for p in params]

ref_l, ref_w = run(False)
l, w = run(True)
bad = [i for i, (a, b) in enumerate(zip(w, ref_w)) if not torch.equal(a, b)]
nan = sum(int(torch.isnan(a.float()).any()) for a in w)
print(f"{kind}: fresh losses {ref_l}; poisoned losses {l}; tensors with NaN {nan}; tensors that differ {len(bad)} of {len(w)}")
