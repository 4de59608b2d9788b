"""Soak for stream-ordering races: the same fine-tune run (true ViT-B/32 dims, a few optimizer steps) repeated from identical
state under each stream mode; every repeat must end with bit-identical weights and losses, equal to the single-stream run.
usage: python scratch/soak_determinism.py [repeats] [steps] [pt]"""
import sys, os, copy, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import task_config, prep_optimizer
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
from hmmc_amd.optimization import clip_grad_norm_
import hmmc_amd.modeling as M
import hmmc_amd.functional as Fn
DEV = "cuda"
repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 12
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "ft"
dims = synth.VIT_B32
SB, SF = int(os.environ.get("SOAK_B", "24")), int(os.environ.get("SOAK_F", "6"))
if kind == "ft":
    cfg = task_config(max_frames=SF, pretrained_clip_name="ViT-B/32")
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(dims), task_config=cfg).to(DEV).train()
    batch = [t.to(DEV) for t in synth.finetune_batch(SB, SF, 32, tag="soak")]
    args = lambda i: (*batch, i)
else:
    cfg = task_config(max_frames=4, pretrained_clip_name="ViT-B/32", dataset="chvtt", contrast_momentum=0.99, contrast_temperature=0.07,
                      contrast_num_negative=64, pretrained_text=None)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(dims, 64, 4), task_config=cfg).to(DEV).train()
    vid, vf, tg, gm, ti, tm = [t.to(DEV) for t in synth.pretrain_batch(16, 4, tag="soak")]
    args = lambda i: (vid, vf, tg, gm, ti, tm, i)
sd0 = copy.deepcopy(model.state_dict())
params = [p for p in model.parameters() if p.requires_grad]
names = [n for n, p in model.named_parameters() if p.requires_grad]

def run(overlap, wgrad):
    M._OVERLAP_TOWERS, Fn._WGRAD_STREAM = overlap, wgrad
    torch.manual_seed(0)
    model.load_state_dict(sd0)
    if hasattr(model, "_queue_ptr_host"): model._queue_ptr_host = None
    opt = prep_optimizer(model, cfg, t_total=100)
    losses, gsnap = [], []
    for i in range(steps):
        loss = model(*args(i + 1))
        loss.backward()
        gsnap.append([p.grad.detach().clone() if p.grad is not None else None for p in params])
        clip_grad_norm_(params, 1.0)
        opt.step(); opt.zero_grad()
        losses.append(float(loss))
    torch.cuda.synchronize()
    ops.raise_on_device_errors()
    return losses, [p.detach().clone() for p in params], gsnap

ref_l, ref_w, ref_g = run(False, False)
print("single-stream losses", ref_l, flush=True)
for mode in ((False, False), (True, False), (False, True), (True, True)):
    bad = 0
    for r in range(repeats):
        l, w, gs = run(*mode)
        if l == ref_l and all(torch.equal(a, b) for a, b in zip(w, ref_w)):
            continue
        bad += 1
        # first step whose gradients differ, and which tensors
        for st in range(steps):
            d = [names[i] for i, (a, b) in enumerate(zip(gs[st], ref_g[st])) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b))]
            if d:
                print(f"  mode overlap={mode[0]} wgrad={mode[1]} repeat {r}: gradients first differ at step {st + 1} in {len(d)} tensors, e.g. {d[:6]}", flush=True)
                break
        else:
            print(f"  mode overlap={mode[0]} wgrad={mode[1]} repeat {r}: gradients equal at every step but weights / losses differ ({l})", flush=True)
    print(f"mode overlap={mode[0]} wgrad={mode[1]}: {bad} of {repeats} repeats differ from the single-stream run", flush=True)
