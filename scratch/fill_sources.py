"""Which aten ops launch the fill / copy kernels of a fine-tune step, with their Python call sites (torch.profiler, with_stack).
usage: python scratch/fill_sources.py [batch]"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
from torch.profiler import profile, ProfilerActivity
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32")
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
batch = [t.cuda() for t in synth.finetune_batch(b, 12, 32, tag="bench")]
opt = bench.prep_optimizer(model, cfg, t_total=1000)
params = [p for p in model.parameters() if p.requires_grad]
def step(i):
    loss = model(*batch, i)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step(); opt.zero_grad()
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::add_", "aten::add", "aten::mul"):
        if ev.cpu_parent is not None and ev.cpu_parent.name in ("aten::zero_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::copy_"):
            continue                                   # count the outermost op only
        st = [f for f in (ev.stack or []) if "hmmc_amd" in f or "bench.py" in f or "scratch" in f]
        site = st[0].split("/")[-1] if st else ("(autograd / torch internals)" + (" <- " + ev.cpu_parent.name if ev.cpu_parent is not None else ""))
        shape = str(ev.input_shapes[0]) if ev.input_shapes else ""
        cnt[(ev.name, site, shape)] += 1
for (name, site, shape), n in cnt.most_common(45):
    print(f"{n:4d}  {name:18s} {shape:22s} {site}")
