"""Does anything grow with the step count?  Device memory (allocated / reserved) and host RSS over the bench loop at a small batch,
by stage (forward only / + backward / + clip / + optimizer) and stream mode.  usage: python scratch/leak_check.py [steps]"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import task_config, prep_optimizer
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
import hmmc_amd.modeling as M
import hmmc_amd.functional as Fn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
b = 16
dev = torch.device("cuda", 0)
cfg = task_config(max_frames=12, pretrained_clip_name="ViT-B/32")
torch.manual_seed(42)
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
opt = prep_optimizer(model, cfg, t_total=100000)
g = torch.Generator(device=dev).manual_seed(1234)
video = torch.randn((b, 12, 3, 224, 224), generator=g, device=dev)
vf = torch.full((b,), 12, dtype=torch.long, device=dev)
ids, mask = [t.to(dev) for t in synth.token_ids("bench.ids.0", b, 32)]
inputs = (ids, mask, video, vf, torch.arange(b, device=dev))
params = [p for p in model.parameters() if p.requires_grad]
def rss():
    with open("/proc/self/statm") as f: return int(f.read().split()[1]) * 4096 / 2**20
def loop(stage, steps):
    for i in range(steps):
        if stage == 0:
            with torch.no_grad(): model(*inputs, i)
            continue
        loss = model(*inputs, i)
        if stage >= 2: loss.backward()
        if stage >= 3: clip_grad_norm_(params, 1.0)
        if stage >= 4: opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
for overlap in (True, False):
    M._OVERLAP_TOWERS, Fn._WGRAD_STREAM = overlap, overlap
    for stage, name in ((0, "forward, no_grad"), (1, "forward"), (2, "+ backward"), (3, "+ clip"), (4, "+ optimizer")):
        loop(stage, 10)
        r0, d0 = rss(), torch.cuda.memory_reserved() / 2**20
        loop(stage, n)
        print(f"streams {'overlapped' if overlap else 'single'}, {name:18s}: host RSS {(rss() - r0) / n:+.2f} MiB per step, device reserved {(torch.cuda.memory_reserved() / 2**20 - d0) / n:+.2f} MiB per step", flush=True)
ops.raise_on_device_errors()
