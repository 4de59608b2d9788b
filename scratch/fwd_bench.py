"""Forward-only (eval) throughput of the two encoders at config 2 (B=256, F=12): ms and TFLOP/s."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from bench import task_config
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
cfg = task_config(max_frames=12, pretrained_clip_name="ViT-B/32")
model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().eval()
B, F = 256, 12
g = torch.Generator(device="cuda").manual_seed(1)
video = torch.randn((B, F, 3, 224, 224), generator=g, device="cuda")
ids, mask = synth.token_ids("fwd.ids", B, 32); ids, mask = ids.cuda(), mask.cuda()
vf = torch.full((B,), F, dtype=torch.long, device="cuda")
with torch.no_grad():
    for _ in range(2): model.visual_encoder(video, vf); model.text_encoder(ids, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): v, fr = model.visual_encoder(video, vf)
    torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): q = model.text_encoder(ids, mask)
    torch.cuda.synchronize(); tt = (time.perf_counter() - t0) / n
flop_v = B * (12 * 8.856e9 + 0.303e9)
flop_t = B * 2.458e9
print(f"visual encoder fwd: {tv*1e3:.2f} ms = {flop_v/tv/1e12:.0f} TFLOP/s ({flop_v/tv/2.5e15*100:.1f} % of MFMA peak); text encoder fwd: {tt*1e3:.2f} ms = {flop_t/tt/1e12:.0f} TFLOP/s; {B/(tv+tt):.0f} pairs/s forward-only")
