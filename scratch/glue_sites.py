"""Call sites of the torch fills / copies a fine-tune step still makes (monkeypatched counters).  usage: python scratch/glue_sites.py [batch]"""
import sys, os, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
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
cnt = collections.Counter()
ON = [False]
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "hmmc_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "(torch internals)"
def wrap_method(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        if ON[0] and isinstance(self, torch.Tensor) and self.is_cuda: cnt[(name, site())] += 1
        return orig(self, *a, **k)
    setattr(cls, name, f)
def wrap_fn(mod, name):
    orig = getattr(mod, name)
    def f(*a, **k):
        r = orig(*a, **k)
        if ON[0] and isinstance(r, torch.Tensor) and r.is_cuda: cnt[(name, site())] += 1
        return r
    setattr(mod, name, f)
for m in ("zero_", "fill_", "copy_", "clone", "contiguous", "float", "half", "to", "add_", "mul_"): wrap_method(torch.Tensor, m)
for m in ("zeros", "zeros_like", "full", "cat", "ones", "stack"): wrap_fn(torch, m)
ON[0] = True
step(3)
ON[0] = False
torch.cuda.synchronize()
for (name, s), n in cnt.most_common(40): print(f"{n:4d}  {name:12s} {s}")
