"""Host time to ENQUEUE one fine-tune step (no synchronisation inside the loop) against the GPU time of the step:
how far the launch stream runs ahead of the GPU at a given per-GPU batch.   usage: python scratch/cpu_issue.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
from hmmc_amd.optimization import clip_grad_norm_
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
if os.environ.get("RESERVE"):
    from hmmc_amd import ops as _o
    os.environ["HMMC_RCCL_CUS"] = os.environ["RESERVE"]; _o.reserve_cus_for_collectives()
if os.environ.get("PG"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32")
torch.manual_seed(42)
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
opt = bench.prep_optimizer(model, cfg, t_total=1000)
batch = [t.cuda() for t in synth.finetune_batch(b, 12, 32, tag="bench")]
params = [p for p in model.parameters() if p.requires_grad]
def step(i):
    loss = model(*batch, i)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step(); opt.zero_grad()
for i in range(3): step(i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for i in range(n): step(3 + i)
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / n
print(f"b={b}: host enqueue {t_issue*1e3:.2f} ms/step, wall {t_total*1e3:.2f} ms/step")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n): step(20 + i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
