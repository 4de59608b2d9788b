import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from hmmc_amd import synth
from hmmc_amd.optimization import BertAdam
g = np.load('/root/repo/tests/golden/bertadam.npz')
shape, dt = (4, 32), torch.float16
p = torch.nn.Parameter(synth.normal("bertadam.d16.p", shape, 0.5).to(dt).cuda())
opt = BertAdam([{"params": [p], "weight_decay": 0.0, "lr": 1e-7}], lr=1e-4, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=20, weight_decay=0.2, max_grad_norm=1.0)
for step in range(3):
    p.grad = synth.normal(f"bertadam.d16.g{step}", shape, 0.05).to(dt).cuda()
    opt.step()
    st = opt.state[p]
    for nm, mine in (("m", st["next_m"]), ("v", st["next_v"]), ("p", p.data)):
        ref = g[f"d16.{nm}{step}"].ravel()
        mine = mine.float().cpu().numpy().ravel()
        bad = np.nonzero(mine != ref)[0]
        for i in bad[:4]:
            print(step, nm, i, "mine", repr(float(mine[i])), "ref", repr(float(ref[i])), "g", float(g[f"d16.g{step}"].ravel()[i]), "prev m", float(g[f"d16.m{max(step-1,0)}"].ravel()[i]))
print("done")
