"""clip_grad_norm_ + BertAdam.step on a CLIP-sized parameter set: GPU time (events) and host time per pair, with the norms handed
from the clip to the optimizer (default) or not (HMMC_NO_SHARED_NORMS=1).  usage: python scratch/opt_bench.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import optimization
from hmmc_amd.optimization import BertAdam, clip_grad_norm_
g = torch.Generator(device="cuda").manual_seed(0)
sizes = []
for _ in range(24):                                   # 12 + 12 layers: 4 matrices + 8 vectors each
    sizes += [(2304 * 768, torch.float16), (768 * 768, torch.float16), (3072 * 768, torch.float16), (768 * 3072, torch.float16)] + [(768, torch.float16)] * 6 + [(2304, torch.float16), (3072, torch.float16)]
sizes += [(49408 * 512, torch.float16), (3072 * 768, torch.float16)] + [(512 * 512 * 3, torch.float32), (512 * 2048, torch.float32), (2048 * 512, torch.float32)] * 4 + [(512, torch.float32)] * 40
ps = [torch.nn.Parameter((torch.randn(n, device="cuda", generator=g) * 0.02).to(dt)) for n, dt in sizes]
print(len(ps), "tensors,", sum(p.numel() for p in ps) / 1e6, "M elements")
opt = BertAdam([{"params": ps[::2], "weight_decay": 0.01}, {"params": ps[1::2], "weight_decay": 0.0}], lr=1e-4, warmup=0.1, schedule="warmup_cosine", t_total=1000, max_grad_norm=1.0)
grads = [(torch.randn(p.shape, device="cuda", generator=g) * 0.01).to(p.dtype) for p in ps]
def pair():
    for p, gr in zip(ps, grads): p.grad = gr
    clip_grad_norm_(ps, 1.0)
    opt.step()
for mode in (True, False, False, True):
    optimization._NO_SHARED_NORMS = mode
    for _ in range(5): pair()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(20): pair()
    e1.record(); th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"NO_SHARED_NORMS={mode}: GPU {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per pair, host enqueue {th / 20 * 1e6:.0f} us per pair")
