"""Per-step wall time (synchronised after every step) of the bench loop: does the step time settle after the warm-up?"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import task_config, prep_optimizer
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
cfg = task_config(max_frames=12, pretrained_clip_name="ViT-B/32")
torch.manual_seed(42)
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
opt = prep_optimizer(model, cfg, t_total=1000)
g = torch.Generator(device=dev).manual_seed(1234)
video = torch.randn((b, 12, 3, 224, 224), generator=g, device=dev)
vf = torch.full((b,), 12, dtype=torch.long, device=dev)
ids, mask = [t.to(dev) for t in synth.token_ids("bench.ids.0", b, 32)]
inputs = (ids, mask, video, vf, torch.arange(b, device=dev))
params = [p for p in model.parameters() if p.requires_grad]
ts = []
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = model(*inputs, i); loss.backward(); clip_grad_norm_(params, 1.0); opt.step(); opt.zero_grad()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"b={b}: " + " ".join(f"{t:.1f}" for t in ts))
