"""One pre-training step (ViT-B/32, B=128, F=12, title L=45 / tag L=25, K=1024) on the GPU: timing only."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from argparse import Namespace
from hmmc_amd import synth
from hmmc_amd.modeling import BirdPreTrainedModel
from hmmc_amd.optimization import BertAdam, clip_grad_norm_
B, F, K = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 12, 1024
cfg = Namespace(local_rank=0, rank=0, use_temp=True, language="english", top_frames=2, max_frames=F, n_display=10 ** 9, logdir=None,
                use_frame_fea=True, dataset="chvtt", contrast_momentum=0.99, contrast_temperature=0.07, contrast_num_negative=K,
                pretrained_text=None, lr=1e-4, text_lr=3e-5, coef_lr=1e-3, weight_decay=0.2, warmup_proportion=0.1,
                pretrained_clip_name="ViT-B/32")
model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
params = [p for p in model.parameters() if p.requires_grad]
opt = BertAdam([{"params": params, "weight_decay": 0.2}], lr=1e-4, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
               t_total=1000, weight_decay=0.2, max_grad_norm=1.0)
batch = [t.cuda() for t in synth.pretrain_batch(B, F, tag="bench")]
def step(i):
    loss = model(*batch, i + 1)
    loss.backward()
    clip_grad_norm_(params, 1.0)
    opt.step(); opt.zero_grad()
    return loss
for i in range(2): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for i in range(n): loss = step(2 + i)
t_issue = (time.perf_counter() - t0) / n          # host time to enqueue a step (includes the MLM row-list read)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"pre-train step B={B} F={F} K={K}: host enqueue {t_issue*1e3:.1f} ms/step; {dt*1e3:.1f} ms/step, {B/dt:.0f} pairs/s, loss {float(loss):.4f}, losses {[round(float(x), 3) for x in model.last_losses]}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
import os
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(5): step(10 + i)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
