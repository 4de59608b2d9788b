"""Where a fine-tune step's wall time goes, un-profiled: CUDA events on the main stream around the sections of the step
(forward towers, temporal + head forward, backward of head + temporal, frame-tower backward + rest, clip, optimizer).
usage: python scratch/step_sections.py [batch] [reserve_cus]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import task_config, prep_optimizer
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
import hmmc_amd.functional as Fn
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if len(sys.argv) > 2:
    os.environ["HMMC_RCCL_CUS"] = sys.argv[2]; ops.reserve_cus_for_collectives()
dev = torch.device("cuda", 0)
cfg = task_config(max_frames=12, pretrained_clip_name="ViT-B/32")
torch.manual_seed(42)
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
opt = prep_optimizer(model, cfg, 1000)
g = torch.Generator(device=dev).manual_seed(1)
video = torch.randn((b, 12, 3, 224, 224), generator=g, device=dev)
vf = torch.full((b,), 12, dtype=torch.long, device=dev)
ids, mask = [t.to(dev) for t in synth.token_ids("sec.ids", b, 32)]
inputs = (ids, mask, video, vf, torch.arange(b, device=dev))
params = [p for p in model.parameters() if p.requires_grad]
marks = {}
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.setdefault(name, []).append(e)
# hooks: end of the frame tower forward (visual.hidden_tokens), end of visual_encoder (temporal), head
orig_hidden = model.visual_encoder.visual.hidden_tokens
def hidden(*a, **k):
    out = orig_hidden(*a, **k); mark("frame_tower_fwd_end"); return out
model.visual_encoder.visual.hidden_tokens = hidden
model.visual_encoder.register_forward_hook(lambda m, i, o: mark("temporal_fwd_end"))
def bwd_hook_frame(grad): mark("temporal_bwd_end")
def step(i):
    mark("start")
    loss = model(*inputs, i)
    mark("fwd_end")
    loss.backward()
    mark("bwd_end")
    clip_grad_norm_(params, 1.0)
    mark("clip_end")
    opt.step(); opt.zero_grad()
    mark("opt_end")
# gradient hook on frame_output to mark the end of head + temporal backward: patch VisualEncoder.forward output
orig_fwd = model.visual_encoder.encode_image
def enc(*a, **k):
    out = orig_fwd(*a, **k)
    if out.requires_grad: out.register_hook(bwd_hook_frame)
    return out
model.visual_encoder.encode_image = enc
for i in range(5): step(i)
torch.cuda.synchronize(); marks.clear()
N = 30
for i in range(N): step(5 + i)
torch.cuda.synchronize()
order = ["start", "frame_tower_fwd_end", "temporal_fwd_end", "fwd_end", "temporal_bwd_end", "bwd_end", "clip_end", "opt_end"]
tot = 0.0
for a, c in zip(order[:-1], order[1:]):
    ms = sum(x.elapsed_time(y) for x, y in zip(marks[a], marks[c])) / N
    tot += ms
    print(f"{a:22s} -> {c:22s} {ms:7.3f} ms")
nxt = sum(x.elapsed_time(y) for x, y in zip(marks["opt_end"][:-1], marks["start"][1:])) / (N - 1)
print(f"{'opt_end':22s} -> {'next start':22s} {nxt:7.3f} ms;  sum {tot + nxt:.3f} ms per step at b = {b}")
