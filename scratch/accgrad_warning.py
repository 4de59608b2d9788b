"""When does torch's "AccumulateGrad node's stream does not match" warning fire on this path?  (round-3 review item 9)
Case A: the steady-state training loop (text tower always on the side stream).  Case B: the same after the text encoder was
first called directly on the main stream (what tests/test_gpu_model.py does before model(...)).  Case C: switching the stream
mode with the old graph still referenced (what bench.py's roofline steps did)."""
import sys, warnings, torch
sys.path.insert(0, '/root/repo')
from bench import task_config, prep_optimizer
from hmmc_amd import synth
from hmmc_amd.modeling import BirdModel
import hmmc_amd.modeling as md, hmmc_amd.functional as fn
from hmmc_amd.optimization import clip_grad_norm_

def run(case):
    cfg = task_config(max_frames=4, pretrained_clip_name="ViT-B/32")
    model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).cuda().train()
    opt = prep_optimizer(model, cfg, 100)
    ids, mask, vid, vf, idx = [t.cuda() for t in synth.finetune_batch(8, 4, 32, tag="w")]
    params = [p for p in model.parameters() if p.requires_grad]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        if case == "B":
            q = model.text_encoder(ids, mask)          # main stream first
        for i in range(4):
            if case == "C" and i == 2:
                md._OVERLAP_TOWERS, fn._WGRAD_STREAM = False, False
            loss = model(ids, mask, vid, vf, idx, i)
            loss.backward()
            clip_grad_norm_(params, 1.0); opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        md._OVERLAP_TOWERS, fn._WGRAD_STREAM = True, True
    n = sum("AccumulateGrad" in str(w.message) for w in rec)
    print(f"case {case}: {n} AccumulateGrad stream warnings")
for c in "ABC":
    run(c)
