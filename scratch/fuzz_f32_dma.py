"""Random-shape check of the LDS-DMA fp32 GEMM (HMMC_F32_PICK=7 / 9 on a -DHMMC_SCRATCH build named by HMMC_LIB; shapes the pick
cannot take fall through to the dispatcher).  Integer operands: exact equality with the float64 product for the three operand
orientations and every epilogue.  usage: HMMC_F32_PICK=7 HMMC_LIB=... python scratch/fuzz_f32_dma.py [cases] [seed]"""
import sys, os, random, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(1)
bad = 0
for it in range(cases):
    M = rng.choice([1, 3, 17, 32, 33, 64, 100, 127, 128, 200, 333, 512, 1000]) * rng.choice([1, 1, 2, 4]) 
    N = rng.choice([4, 8, 36, 60, 64, 68, 128, 132, 256, 500, 512, 1024])
    K = 32 * rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 17, 33])
    if rng.random() < 0.3: M = (M + 3) // 4 * 4                     # row-contiguous A wants M % 4 == 0 for 16-byte rows
    a = torch.randint(-3, 4, (M, K), generator=g).float().cuda()
    b = torch.randint(-3, 4, (N, K), generator=g).float().cuda()
    ref = a.double() @ b.double().t()
    bias = torch.randint(-5, 6, (N,), generator=g).float().cuda()
    res = torch.randint(-5, 6, (M, N), generator=g).float().cuda()
    bt, at = b.t().contiguous(), a.t().contiguous()
    outs = {
        "kk": (ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K)), ref),
        "km+alpha+bias": (ops.gemm_f32(a, bt, M, N, K, (K, 1), (N, 1), alpha=2.0, bias=bias), 2 * ref + bias.double()),
        "mm+resid": (ops.gemm_f32(at, bt, M, N, K, (1, M), (N, 1), resid=res), ref + res.double()),
        "mk": (ops.gemm_f32(at, b, M, N, K, (1, M), (1, K)), ref),
        "kk+relu": (ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K), bias=bias, epilogue=ops.EPI_RELU), (ref + bias.double()).clamp_min(0)),
    }
    for name, (c, r) in outs.items():
        if not torch.equal(c.double(), r):
            bad += 1
            d = (c.double() - r).abs()
            print(f"MISMATCH {name} M={M} N={N} K={K}: max {d.max().item():.3g} at {torch.nonzero(d > 0)[:3].tolist()}", flush=True)
    # QuickGELU + saved pre-activation, and the DGELU data gradient: against torch in fp32 (not exact: transcendental)
    y, h = ops.gemm_f32(a, b, M, N, K, (K, 1), (1, K), bias=bias, epilogue=ops.EPI_QGELU, want_aux=True)
    hr = (ref + bias.double()).float()
    if not torch.equal(h, hr) or not torch.allclose(y, hr * torch.sigmoid(1.702 * hr), rtol=1e-5, atol=1e-5):
        bad += 1; print(f"MISMATCH qgelu M={M} N={N} K={K}", flush=True)
print(f"{cases} cases, {bad} mismatches")
