"""In-kernel timeline of one work item of the 256x256 GEMM (debug builds of gemm_f16.hip with -DHMMC_DBG=3/4/5: stamps of the
third item of every wave go to the workspace).  usage: HMMC_LIB=scratch/_dbg/libhmmc_st3.so python scratch/gemm_stamps.py [N] [K]
Prints, for a few workgroups, waves 0 (wm = 0) and 4 (wm = 1): phase durations of the item's K-tiles, the epilogue, and the phases after it."""
import sys, os, torch, numpy as np
sys.path.insert(0, '/root/repo')
from hmmc_amd import _lib
if os.environ.get('HMMC_LIB'): _lib.LIB_PATH = os.environ['HMMC_LIB']
from hmmc_amd import ops
from hmmc_amd._lib import call, ptr
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
K = int(sys.argv[2]) if len(sys.argv) > 2 else 768
M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(M, K, device="cuda", generator=g).half(); w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).half()
out = torch.empty(M, N, device="cuda", dtype=torch.float16)
ws = torch.zeros(256 * 8 * 128, device="cuda", dtype=torch.int32)
f = lambda: call("hmmc_gemm_f16", ptr(a), ptr(w), ptr(out), M, N, K, K, K, N, 1, 1, None, None, None, None, 0, ptr(ws), ws.numel() * 4)
for _ in range(20): f()
torch.cuda.synchronize()
st = ws.cpu().numpy().astype(np.int64).reshape(256, 8, 128)
nkt = K // 64
per_item = 4 * nkt + 2              # phase stamps + epilogue start / end
def fmt(x): return " ".join(f"{v / 100.0:5.2f}" for v in x)
tot = {0: [], 4: []}
for wgi in range(256):
    for wv in (0, 4):
        s = st[wgi, wv]
        n = int((s != 0).sum())
        if n < per_item + 9: continue
        d = np.diff(s[:n]) & 0xffffffff
        # layout from the start of item 2: [phase stamps of its K-tiles: 4 nkt] [epi start] [epi end] [phases of the next item ...]
        ph = d[:4 * nkt - 1]
        epi_wait = d[4 * nkt - 1]; epi = d[4 * nkt]; after = d[4 * nkt + 1: 4 * nkt + 9]
        tot[wv].append((ph.mean(), ph[-8:].mean(), epi_wait, epi, after[:4].sum(), after[4:8].sum(), (s[per_item] - s[0]) & 0xffffffff))
        if wgi in (3, 100):
            print(f"wg {wgi} wave {wv}: last 8 phases {fmt(ph[-8:])} | to epilogue {epi_wait/100:.2f} | epilogue {epi/100:.2f} | next phases {fmt(after)}")
for wv in (0, 4):
    t = np.array(tot[wv], dtype=np.float64) / 100.0
    print(f"wave {wv} (mean over {len(t)} workgroups): phase {t[:,0].mean():.3f} us, last 8 phases {t[:,1].mean():.3f}, barrier before epilogue {t[:,2].mean():.2f}, "
          f"epilogue {t[:,3].mean():.2f}, first K-tile after {t[:,4].mean():.2f}, second {t[:,5].mean():.2f}, whole item {t[:,6].mean():.2f} us")
