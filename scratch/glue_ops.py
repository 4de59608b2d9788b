"""Which aten ops (fills, copies, adds ...) a training step still launches besides the library's kernels, by call site.
usage: python scratch/glue_ops.py [ft|pt] [batch]"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
from hmmc_amd.optimization import clip_grad_norm_
kind = sys.argv[1] if len(sys.argv) > 1 else "ft"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if kind == "ft":
    cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32")
    model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
    batch = [t.cuda() for t in synth.finetune_batch(b, 12, 32, tag="bench")]
else:
    cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32", dataset="chvtt", contrast_momentum=0.99,
                            contrast_temperature=0.07, contrast_num_negative=1024, pretrained_text=None)
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
    batch = [t.cuda() for t in synth.pretrain_batch(b, 12, tag="bench")]
opt = bench.prep_optimizer(model, cfg, t_total=1000)
params = [p for p in model.parameters() if p.requires_grad]
def step(i):
    loss = model(*batch, i)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step(); opt.zero_grad()
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step(3)
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name.split("::")[1] in ("zero_", "fill_", "copy_", "add_", "add", "clone", "contiguous", "to", "_to_copy", "zeros", "empty_like", "cat", "mul", "sum", "div"):
        st = [s for s in (e.stack or []) if "hmmc_amd" in s or "bench" in s or "glue_ops" in s]
        site = st[0].split("/")[-1][:90] if st else "(autograd engine / other)"
        cnt[(e.name, site)] += 1
for (name, site), n in cnt.most_common(45):
    print(f"{n:4d}  {name:18s} {site}")
