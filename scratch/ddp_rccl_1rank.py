"""One-rank RCCL rehearsal on the GPU box: ProcessGroupNCCL + DistributedDataParallel around BirdModel with the tower chunking,
the CU reservation and the three streams exactly as bench.py --gpus N sets them up (N ranks need N GPUs; this checks the plumbing)."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import bench
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32")
torch.manual_seed(42)
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
opt = bench.prep_optimizer(model, cfg, t_total=1000)
kw = dict(gradient_as_bucket_view=os.environ.get("BUCKET_VIEW", "0") == "1", bucket_cap_mb=int(os.environ.get("BUCKET_MB", "25")),
          static_graph=os.environ.get("STATIC", "0") == "1")
net = model if os.environ.get("NO_DDP") else torch.nn.parallel.DistributedDataParallel(
    model, device_ids=[0], output_device=0, find_unused_parameters=False, **kw)
batch = [t.to(dev) for t in synth.finetune_batch(b, 12, 32, tag="bench")]
params = [p for p in model.parameters() if p.requires_grad]
def step(i):
    loss = net(*batch, i)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step(); opt.zero_grad()
    return loss
# NOTE world_size 1: the towers do not chunk or reserve CUs by themselves; force both to rehearse that path
ops.reserve_cus_for_collectives()
for blk in (model.visual_encoder.visual.transformer, model.text_encoder.transformer):
    blk.ddp_layers_per_node = 3
for i in range(3): loss = step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): loss = step(3 + i)
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
print(f"1-rank RCCL DDP b={b} {kw} no_ddp={bool(os.environ.get('NO_DDP'))}: host {t_issue*1e3:.2f} ms, {(time.perf_counter()-t0)/10*1e3:.2f} ms/step, loss {float(loss.detach()):.4f}")
dist.destroy_process_group()
