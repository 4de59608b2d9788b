import sys, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import ops
M, N, K = 256, 128, 64
for val in (1e-3, 1e-4, 3e-5, 1e-5, 1e-6):
    a = torch.full((M, K), val, device="cuda").half(); b = torch.ones(N, K, device="cuda").half()
    c = ops.gemm_f16(a, b, M, N, K)
    print(f"a={float(a[0,0]):.3e} (subnormal={float(a[0,0])<6.1e-5}) expect {float(a[0,0])*K:.4e} got {float(c[0,0]):.4e}")
# B operand subnormal too
a = torch.ones(M, K, device="cuda").half(); b = torch.full((N, K), 1e-5, device="cuda").half()
print("b subnormal:", float(ops.gemm_f16(a, b, M, N, K)[0, 0]), "expect", float(b[0,0])*K)
# output subnormal
a = torch.full((M, K), 1e-3, device="cuda").half(); b = torch.full((N, K), 1e-4, device="cuda").half()
print("output subnormal:", float(ops.gemm_f16(a, b, M, N, K)[0, 0]), "expect", float(a[0,0])*float(b[0,0])*K)
