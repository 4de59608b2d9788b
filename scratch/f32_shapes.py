"""Every fp32 GEMM of one training step (shape, orientation, epilogue, count) and its time alone on the GPU.
usage: python scratch/f32_shapes.py [pretrain|finetune] [batch]"""
import sys, os, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hmmc_amd import synth, ops
from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
from hmmc_amd.optimization import clip_grad_norm_
mode = sys.argv[1] if len(sys.argv) > 1 else "pretrain"
pre = mode == "pretrain"
b = int(sys.argv[2]) if len(sys.argv) > 2 else (128 if pre else 256)
extra = dict(dataset="chvtt", contrast_momentum=0.99, contrast_temperature=0.07, contrast_num_negative=1024, pretrained_text=None) if pre else {}
cfg = bench.task_config(local_rank=0, rank=0, max_frames=12, pretrained_clip_name="ViT-B/32", **extra)
torch.manual_seed(42)
model = (BirdPreTrainedModel if pre else BirdModel).from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
opt = bench.prep_optimizer(model, cfg, t_total=1000)
g = torch.Generator(device="cuda").manual_seed(1234)
video = torch.randn((b, 12, 3, 224, 224), generator=g, device="cuda")
vf = torch.full((b,), 12, dtype=torch.long, device="cuda")
if pre:
    title, tmask = [t.cuda() for t in synth.token_ids("bench.title.0", b, 45)]
    tag, gmask = [t.cuda() for t in synth.token_ids("bench.tag.0", b, 25)]
    inputs = (video, vf, tag, gmask, title, tmask)
else:
    ids, mask = [t.cuda() for t in synth.token_ids("bench.ids.0", b, 32)]
    inputs = (ids, mask, video, vf, torch.arange(b, device="cuda"))
params = [p for p in model.parameters() if p.requires_grad]
def step(i):
    loss = model(*inputs, i); loss.backward(); clip_grad_norm_(params, 1.0); opt.step(); opt.zero_grad()
for i in range(2): step(i)
torch.cuda.synchronize()
cnt = collections.Counter()
orig = ops.gemm_f32
def rec(a, b_, M, N, K, sa, sb, alpha=1.0, bias=None, resid=None, aux_in=None, epilogue=0, want_aux=False, out=None):
    cnt[(M, N, K, tuple(sa), tuple(sb), int(epilogue) | (1 if bias is not None else 0) | (2 if resid is not None else 0), bool(want_aux))] += 1
    return orig(a, b_, M, N, K, sa, sb, alpha=alpha, bias=bias, resid=resid, aux_in=aux_in, epilogue=epilogue, want_aux=want_aux, out=out)
ops.gemm_f32 = rec
import hmmc_amd.functional as Fn
step(2)
torch.cuda.synchronize()
ops.gemm_f32 = orig
del model, opt, video
torch.cuda.empty_cache()
tot = 0.0
rows = []
for (M, N, K, sa, sb, epi, aux), n in cnt.items():
    a = torch.randn(max((M - 1) * sa[0] + (K - 1) * sa[1] + 1, 1), device="cuda")
    bb = torch.randn(max((K - 1) * sb[0] + (N - 1) * sb[1] + 1, 1), device="cuda")
    f = lambda: orig(a, bb, M, N, K, sa, sb)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    rows.append((n * us, n, M, N, K, sa, sb, epi, us))
    tot += n * us
for t, n, M, N, K, sa, sb, epi, us in sorted(rows, reverse=True):
    lay = ("k" if sa[1] == 1 else "m") + ("k" if sb[0] == 1 else "m")
    print(f"{t:8.0f} us = {n:3d} x {us:7.1f} us  {M:6d} x {N:6d} x {K:6d} {lay} epi {epi:2d}  {2.0*M*N*K/us/1e6:6.1f} TF")
print(f"total {tot/1e3:.2f} ms in {sum(cnt.values())} launches")
