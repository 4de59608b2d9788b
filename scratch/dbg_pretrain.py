import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from hmmc_amd import synth
from hmmc_amd.modeling import BirdPreTrainedModel
from oracle import hmmc_oracle as O
from test_gpu_model import task_config
g = np.load('/root/repo/tests/golden/moco_fp32.npz')
K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
raw = synth.pretrain_state(synth.TINY, K, Fr)
batch = synth.pretrain_batch(B, Fr, tag="moco.s0")
draws = [torch.from_numpy(g[f"mlm_{n}0"]) for n in ("masked", "replaced", "randsel", "words")]
keys = ["text_encoder.text_projection", "text_encoder.transformer.resblocks.0.mlp.c_fc.weight", "visual_encoder.visual.proj",
        "visual_encoder.visual.transformer.resblocks.0.mlp.c_fc.weight", "visual_encoder.temporal_transformer.resblocks.0.mlp.c_fc.weight",
        "v_projector.linear_out.weight", "v_predictor.linear_out.weight", "cls.decoder.weight"]
for wts in [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)]:
    cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt")
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=raw, task_config=cfg).cuda().train()
    model.weight_FAM, model.weight_VTM, model.weight_FTM, model.weight_MLM = wts
    model._mlm_draws = draws
    loss = model(*[t.cuda() for t in batch], 1); loss.backward()
    sd = {}
    for k, v in raw.items():
        tr = v.is_floating_point() and not any(s in k for s in ("_k.", "queue_", "running_", "num_batches"))
        sd[k] = v.clone().requires_grad_(tr)
    sd["cls.decoder.bias"] = sd["cls.bias"]
    queues = {k: sd[k] for k in sd if k.startswith("queue_") and k != "queue_ptr"}
    d2 = [d.bool() if i < 3 else d for i, d in enumerate(draws)]
    ref, parts, _ = O.pretrain_loss(batch, sd, queues, 0, K, mode="fp32", mlm_draws=d2, weights=wts)
    ref.backward()
    P = dict(model.named_parameters())
    print("weights", wts, "loss", float(loss), float(ref))
    for k in keys:
        a = P[k].grad; b = sd[k].grad
        if a is None or b is None: print("   ", k, "none", a is None, b is None); continue
        a = a.float().cpu().flatten(); b = b.flatten()
        print(f"    {k:70s} cos {float(torch.dot(a,b)/(a.norm()*b.norm()+1e-20)):.4f} ratio {float(a.norm()/(b.norm()+1e-20)):.3f} refnorm {float(b.norm()):.3e}")
