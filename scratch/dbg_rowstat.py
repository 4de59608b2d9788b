import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
M, N, K = 256, 128, 64
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.zeros(M, K, device="cuda").half()
W = torch.zeros(N, K, device="cuda").half()
b = torch.zeros(N, device="cuda").half()
r = (torch.arange(M, device="cuda").float()[:, None] * 1.0 + torch.zeros(N, device="cuda")[None, :]).half()   # row r has value r
y, part = ops.gemm_f16_fold(a, W, bias=b, resid=r, want_stat=True)
print(y[:4, :4], part.shape)
print("blk0 sums/64 rows 0..40:", (part[0, :40, 0] / 64).tolist())
print("blk1 sums/64 rows 0..40:", (part[1, :40, 0] / 64).tolist())
r2 = (torch.arange(N, device="cuda").float()[None, :] + torch.zeros(M, device="cuda")[:, None]).half()       # column n has value n
y, part = ops.gemm_f16_fold(a, W, bias=b, resid=r2, want_stat=True)
print("col-valued: blk0 row sums (expect 2016):", part[0, :8, 0].tolist(), "blk1 (expect 6112):", part[1, :8, 0].tolist())
