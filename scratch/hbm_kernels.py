"""The HBM-bound kernels of the default step, each launched 4 times at the step's shapes (153 600 tokens x 768, 3 072 x 50 x 12
heads, the model's 350 parameter tensors): the workload of the two `rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE` passes behind
profiles/r04_hbm_kernels_traffic.txt (scratch/hbm_kernels_summary.py)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
T, D, nseq, L, H = 153600, 768, 3072, 50, 12
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(T, D, device="cuda", generator=g).half(); dy = torch.randn(T, D, device="cuda", generator=g).half()
dres = torch.randn(T, D, device="cuda", generator=g).half()
gm, bt = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-5)
st = ops.rowstat(x)
qkv = torch.randn(T, 3 * D, device="cuda", generator=g).half()
att, lse = ops.attention_f16_fwd(qkv, nseq, L, H, False)
torch.cuda.synchronize()
for _ in range(4):
    ops.layernorm_fwd(x, gm, bt, 1e-5)
    ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=dres, want_colsum=True)
    ops.layernorm_bwd_fold(dy, x, st, dres=dres, want_colsum=True, reduce=False)
    ops.attention_f16_fwd(qkv, nseq, L, H, False)
    ops.attention_f16_bwd(qkv, att, lse, dy, nseq, L, H, False, want_dbias=True)
    ops.attention_f16_bwd(qkv, att, lse, dy, nseq, L, H, False, want_dbias=True, rowstat=st)
    ops.rowstat(x)
torch.cuda.synchronize()
