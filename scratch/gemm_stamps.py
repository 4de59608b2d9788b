"""Item-boundary timeline of the 256x256 GEMM (needs a -DHMMC_DBG=3 build: HMMC_LIB=scratch/_dbg/libhmmc_stamp.so).
usage: python scratch/gemm_stamps.py [N] [K]"""
import sys, os, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd._lib import call, ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = 65536
if os.environ.get('RESERVE'): call('hmmc_gemm_reserve_cus', int(os.environ['RESERVE']))
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).half()
b = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
c = torch.empty(M, N, device="cuda", dtype=torch.float16)
ws = torch.zeros(3 * 2 * 512, device="cuda", dtype=torch.int32)
for _ in range(3):
    call("hmmc_gemm_f16", ptr(a), ptr(b), ptr(c), M, N, K, K, K, N, 1, 1, None, None, None, None, 0, ptr(ws), ws.numel() * 4)
torch.cuda.synchronize()
st = ws.cpu().numpy().astype('uint32').reshape(3, 2, 512)
names = {1: "top", 2: "top(last)", 3: "w1", 4: "w2", 5: "w3", 6: "w4", 7: "epi>", 8: "epi<", 9: "s"}
for blk in range(3):
    for grp in range(2):
        row = st[blk, grp]
        ev = [(int(x >> 28), int(x & 0x0fffffff)) for x in row if x != 0]
        if not ev: continue
        t0 = ev[0][1]
        print(f"block sel {blk} group wm={grp}: {len(ev)} stamps (us since first; delta)")
        prev = t0; line = []
        for tag, t in ev[:160]:
            line.append(f"{names[tag]}@{(t - t0) / 100:.2f}(+{(t - prev) / 100:.2f})")
            prev = t
            if tag == 6 and len(line) > 6:
                print("   " + " ".join(line)); line = []
        if line: print("   " + " ".join(line))
