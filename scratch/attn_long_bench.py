"""Long-sequence attention (ViT-B/16: 197 tokens) forward / backward timings.  usage: python scratch/attn_long_bench.py [nseq]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for L, H, causal in ((197, 12, False), (77, 8, True)):
    D = H * 64
    qkv = (torch.randn(nseq * L, 3 * D, device="cuda") * 0.5).half()
    dout = (torch.randn(nseq * L, D, device="cuda") * 0.5).half()
    out, lse = ops.attention_f16_fwd(qkv, nseq, L, H, causal)
    ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal)
    torch.cuda.synchronize()
    def t(f, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(lambda: ops.attention_f16_fwd(qkv, nseq, L, H, causal))
    tb = t(lambda: ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal))
    tbb = t(lambda: ops.attention_f16_bwd(qkv, out, lse, dout, nseq, L, H, causal, want_dbias=True))
    fl = 2.0 * 2 * nseq * H * L * L * 64 * (0.5 if causal else 1.0)
    print(f"L={L} H={H} causal={causal} nseq={nseq}: fwd {tf:8.1f} us ({fl/tf/1e6:6.1f} TF)   bwd {tb:8.1f} us ({2.5*fl/tb/1e6:6.1f} TF)   bwd + bias partials {tbb:8.1f} us")
