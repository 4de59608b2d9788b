"""Two ranks (processes) of the fine-tune step on the one available GPU, DDP over gloo: the packed feature
all-gather, its reduce-scatter backward and DDP's gradient averaging must reproduce the single-process
global-batch loss and gradients.  RCCL itself needs one GPU per rank; the collective
semantics exercised here are backend-independent."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from hmmc_amd import synth  # noqa: E402
from test_gpu_model import task_config  # noqa: E402

KEYS = ["text_encoder.text_projection", "visual_encoder.visual.conv1.weight", "visual_encoder.visual.proj",
        "visual_encoder.visual.transformer.resblocks.1.mlp.c_fc.weight", "text_encoder.token_embedding.weight",
        "visual_encoder.temporal_transformer.resblocks.0.attn.in_proj_weight", "visual_encoder.frame_position_embeddings.weight"]


def _init(rank, world, store, backend):
    """Process group of `world` ranks; returns this rank's device index.  gloo: every rank on device 0 (what one GPU allows);
    nccl (= RCCL): one device per rank, initialised BEFORE anything in this fresh process touches the GPU
    (tests/test_gpu_rccl_multi.py, skipped below two devices)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = rank if backend == "nccl" else 0
    if world > 1 or backend == "nccl":
        kw = {"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}
        dist.init_process_group(backend, init_method=f"file://{store}", rank=rank, world_size=world, **kw)
    torch.cuda.set_device(dev)
    if backend == "nccl" and world > 1:
        from hmmc_amd import ops
        ops.reserve_cus_for_collectives()          # 16 CUs out of every GEMM grid, as the towers do under world_size > 1
    return dev


def _run(rank, world, store, out_dir, backend="gloo", B=4):
    dev = _init(rank, world, store, backend)
    from hmmc_amd.modeling import BirdModel
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY),
                                      task_config=task_config(rank=rank)).cuda().train()
    for m in model.modules():                  # exercise the per-run autograd nodes the towers use under DDP
        if hasattr(m, "ddp_layers_per_node"):
            m.ddp_layers_per_node = 1
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev]) if world > 1 else model
    b = B // world
    ids, mask, vid, vf, idx = [t[rank * b:(rank + 1) * b].cuda() for t in synth.finetune_batch(B, 4, 32, tag="ddp")]
    loss = net(ids, mask, vid, vf, idx, 1)
    loss.backward()
    torch.cuda.synchronize()
    P = dict(model.named_parameters())
    torch.save({"loss": loss.detach().cpu(), **{k: P[k].grad.float().cpu() for k in KEYS}}, os.path.join(out_dir, f"w{world}r{rank}.pt"))
    if dist.is_initialized():
        dist.destroy_process_group()


# Tolerances of the 2-rank = 1-process statement, MEASURED (round 5, printed by the test with -s): the two runs differ by fp16
# rounding only - each rank's towers see half the rows, so tile paths and the order of the partial sums in every weight
# gradient change.  Measured worst case over KEYS at the tiny dims (gloo, 2 ranks on one MI355X): 1 - cos = 2.0e-7,
# | norm ratio - 1 | = 2.0e-5, loss difference 0.0; the bounds below are 50 x / 10 x that.  A slip in the gather backward's
# scaling (x world, / world) moves the ratio by a factor of 2; a 2 % slip is 100 x the bound.
FT_COS, FT_RATIO, FT_LOSS = 1e-5, 2e-4, 1e-4


def check_finetune(ref, outs):
    worst = {"cos": 0.0, "ratio": 0.0, "loss": 0.0}
    for o in outs:
        worst["loss"] = max(worst["loss"], abs(float(o["loss"]) - float(ref["loss"])))
        for k in KEYS:
            a, b = o[k].flatten().double(), ref[k].flatten().double()
            cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
            ratio = float(a.norm() / (b.norm() + 1e-30))
            worst["cos"], worst["ratio"] = max(worst["cos"], 1 - cos), max(worst["ratio"], abs(ratio - 1))
    print("2 ranks vs 1 process (fine-tune):", worst)
    assert worst["cos"] < FT_COS and worst["ratio"] < FT_RATIO and worst["loss"] < FT_LOSS, worst
    for k in KEYS:      # DDP left identical gradients on every rank
        assert all(torch.equal(outs[0][k], o[k]) for o in outs[1:]), k


def test_two_ranks_equal_single_process():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run, args=(1, os.path.join(d, "s1"), d), nprocs=1, join=True)
        mp.spawn(_run, args=(2, os.path.join(d, "s2"), d), nprocs=2, join=True)
        ref = torch.load(os.path.join(d, "w1r0.pt"))
        outs = [torch.load(os.path.join(d, f"w2r{r}.pt")) for r in range(2)]
    check_finetune(ref, outs)


# ----------------------------------------------------------------------------- pre-training, two ranks

PT_KEYS = ["visual_encoder.visual.conv1.weight", "visual_encoder.visual.transformer.resblocks.0.attn.in_proj_weight",
           "text_encoder.text_projection", "v_projector.linear_hidden.1.weight", "v_projector.linear_hidden.2.weight",
           "v_predictor.linear_out.weight", "visual_encoder.temporal_transformer.resblocks.1.mlp.c_proj.weight",
           "visual_encoder.frame_position_embeddings.weight"]
PT_BUFFERS = ["queue_v_cross_ng", "queue_title_cross_ng", "queue_tag_cross_ng", "queue_frame_proj_ng", "queue_frame_cross_ng",
              "queue_ptr", "v_projector.linear_hidden.2.running_mean", "v_projector.linear_hidden.2.running_var",
              "v_predictor.linear_hidden.2.running_mean", "v_projector_k.linear_hidden.2.running_var",
              "visual_encoder_k.visual.proj", "text_encoder_k.ln_final.weight"]


def _run_pretrain(rank, world, store, out_dir, backend="gloo"):
    dev = _init(rank, world, store, backend)
    from conftest import golden
    from hmmc_amd.modeling import BirdPreTrainedModel
    g = golden("moco_aswritten")
    K, B, Fr = int(g["K"]), int(g["B"]), int(g["F"])
    cfg = task_config(rank=rank, contrast_num_negative=K, max_frames=Fr, dataset="chvtt")
    model = BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(synth.TINY, K, Fr),
                                                task_config=cfg).cuda().train()
    # t_projector is built, EMA'd and never used (reference modules/modeling.py:113-114): find_unused_parameters as main_pretrain.py:204
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev], find_unused_parameters=True) if world > 1 else model
    b = B // world
    sl = slice(rank * b, (rank + 1) * b)
    batch = synth.pretrain_batch(B, Fr, tag="moco.s0")
    draws = [torch.from_numpy(g[f"mlm_{n}0"]) for n in ("masked", "replaced", "randsel", "words")]
    model._mlm_draws = [d[sl] for d in draws]
    title = batch[4][sl]
    n_masked = int((draws[0][sl].bool() & (title != model.PAD_ID) & (title != model.CLS_ID)).sum())
    loss = net(*[t[sl].cuda() for t in batch], 1)
    loss.backward()
    torch.cuda.synchronize()
    P, S = dict(model.named_parameters()), model.state_dict()
    out = {"loss": loss.detach().cpu(), "parts": [float(x) for x in model.last_losses], "n_masked": n_masked}
    out.update({k: P[k].grad.float().cpu() for k in PT_KEYS})
    out.update({k: S[k].float().cpu() for k in PT_BUFFERS})
    torch.save(out, os.path.join(out_dir, f"p{world}r{rank}.pt"))
    if dist.is_initialized():
        dist.destroy_process_group()


def test_pretrain_two_ranks_equal_single_process():
    """BirdPreTrainedModel under DDP on two ranks (B = 4 split 2 + 2): the packed key all-gather keeps the five queues and
    queue_ptr identical on both ranks and equal to the single-process run, the BatchNorm statistics are those of all
    ranks' rows (SyncBatchNorm, reference modules/modeling.py:115-129,244-284), and the averaged gradients are the
    global-batch gradients."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_pretrain, args=(1, os.path.join(d, "s1"), d), nprocs=1, join=True)
        mp.spawn(_run_pretrain, args=(2, os.path.join(d, "s2"), d), nprocs=2, join=True)
        ref = torch.load(os.path.join(d, "p1r0.pt"))
        outs = [torch.load(os.path.join(d, f"p2r{r}.pt")) for r in range(2)]
    check_pretrain(ref, outs)


def check_pretrain(ref, outs):
    # FAM / VTM / FTM are means over a rank's rows: their average over the (equal-sized) ranks is the global mean
    for i, nm in enumerate(("FAM", "VTM", "FTM")):
        avg = 0.5 * (outs[0]["parts"][i] + outs[1]["parts"][i])
        assert abs(avg - ref["parts"][i]) < 3e-3 * max(1.0, abs(ref["parts"][i])), (nm, avg, ref["parts"][i])
    # MLM is a mean over a rank's masked tokens
    n = [o["n_masked"] for o in outs]
    assert sum(n) == ref["n_masked"] and min(n) > 0
    mlm = (outs[0]["parts"][3] * n[0] + outs[1]["parts"][3] * n[1]) / sum(n)
    assert abs(mlm - ref["parts"][3]) < 3e-3 * max(1.0, abs(ref["parts"][3])), (mlm, ref["parts"][3])
    for k in PT_BUFFERS:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k} differs between the ranks"
        tol = 0 if k == "queue_ptr" else 2e-3
        err = float((outs[0][k] - ref[k]).abs().max())
        assert err <= tol, f"{k}: two-rank run differs from the single process by {err}"
    for k in PT_KEYS:
        assert torch.equal(outs[0][k], outs[1][k]), f"DDP left different gradients for {k}"
        a, b = outs[0][k].flatten(), ref[k].flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-20))
        ratio = float(a.norm() / (b.norm() + 1e-20))
        assert cos > 0.99 and abs(ratio - 1) < 0.05, (k, cos, ratio)


# ----------------------------------------------------------------------------- RCCL, one rank

def _run_rccl(rank, out_dir, port):
    """ProcessGroupNCCL (= RCCL) is initialised BEFORE anything touches the GPU in this fresh process."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    from hmmc_amd import ops
    from hmmc_amd.modeling import BirdModel
    torch.cuda.set_device(0)
    ops.reserve_cus_for_collectives()                      # what the towers do when they see world_size > 1
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY),
                                      task_config=task_config(max_frames=6)).cuda().train()
    for m in model.modules():
        if hasattr(m, "ddp_layers_per_node"):
            m.ddp_layers_per_node = 1                      # gradients reach DDP's bucket hooks layer by layer
    batch = [t.cuda() for t in synth.finetune_batch(16, 6, 32, tag="det")]

    def grads(net):
        model.zero_grad(set_to_none=True)
        loss = net(*batch, 1)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().cpu(), {n: p.grad.clone().cpu() for n, p in model.named_parameters() if p.grad is not None}

    l0, g0 = grads(model)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, bucket_cap_mb=1)
    l1, g1 = grads(ddp)
    l2, g2 = grads(ddp)
    torch.save({"l": (l0, l1, l2), "g": (g0, g1, g2)}, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_rccl_one_rank_ddp_is_bit_identical():
    """DistributedDataParallel over the nccl backend (RCCL) with one rank: the bucketed all-reduces run on RCCL's stream
    beside the text-tower, frame-tower and weight-gradient streams, with 16 CUs kept out of the GEMM grids.  A missing
    dependency between a gradient's producer stream and its bucket's all-reduce would change bits here; with every
    dependency in place the loss and every gradient equal the plain module's."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_rccl, args=(d, port), nprocs=1, join=True)
        out = torch.load(os.path.join(d, "rccl.pt"))
    (l0, l1, l2), (g0, g1, g2) = out["l"], out["g"]
    assert float(l0) == float(l1) == float(l2), (float(l0), float(l1), float(l2))
    assert set(g0) == set(g1) == set(g2)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), f"{n}: DDP over RCCL changed the gradient"
        assert torch.equal(g1[n], g2[n]), f"{n}: not repeatable under DDP"


# ----------------------------------------------------------------------------- several optimizer steps under DDP

def _run_steps(rank, world, store, out_dir, backend="gloo"):
    dev = _init(rank, world, store, backend)
    from hmmc_amd.modeling import BirdModel
    from hmmc_amd.optimization import clip_grad_norm_
    from test_gpu_model import prep_optimizer
    cfg = task_config(rank=rank)
    model = BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY), task_config=cfg).cuda().train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev], gradient_as_bucket_view=True)     # as bench.py
    opt = prep_optimizer(model, cfg, 20)
    params = [p for p in model.parameters() if p.requires_grad]
    B = 8
    b = B // world
    losses = []
    for step in range(4):
        ids, mask, vid, vf, idx = [t[rank * b:(rank + 1) * b].cuda() for t in synth.finetune_batch(B, 4, 32, tag=f"ddp.s{step}")]
        loss = net(ids, mask, vid, vf, idx, step + 1)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    torch.cuda.synchronize()
    torch.save({"losses": losses, "weights": [p.detach().cpu() for p in params]}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_stay_identical_over_optimizer_steps():
    """Four steps of forward / backward / clip / BertAdam under DDP (gradients as bucket views, the bench's configuration): both
    ranks must hold bit-identical weights afterwards (same averaged gradients, deterministic norms and update) and see the
    same global losses."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_steps, args=(2, os.path.join(d, "s"), d), nprocs=2, join=True)
        a, b = [torch.load(os.path.join(d, f"s{r}.pt")) for r in range(2)]
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    bad = [i for i, (x, y) in enumerate(zip(a["weights"], b["weights"])) if not torch.equal(x, y)]
    assert not bad, f"{len(bad)} parameter tensors differ between the ranks after 4 steps"


# ----------------------------------------------------------------------------- the path's own RCCL collectives, one rank

def _run_rccl_collectives(rank, out_dir, port):
    """Every collective the product path issues, on the nccl backend (RCCL), in a process group of one rank with the
    world-size-1 shortcut switched off (hmmc_amd.functional._FORCE_COLLECTIVES): all_gather_into_tensor and its
    reduce_scatter_tensor backward (dist_collect, reference modules/modeling.py:25-36,698-700), the packed key gather of the
    enqueue (:249-258) and the SyncBatchNorm all-reduces (:115-129).  With one rank each is the identity, so the forced run
    must reproduce the unforced one bit for bit - what is exercised is that the calls are accepted by RCCL with these
    tensors, run on its stream, and are ordered correctly against the tower / side / weight-gradient streams."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    import hmmc_amd.functional as Fn
    import hmmc_amd.modeling as M
    from conftest import golden
    torch.cuda.set_device(0)
    res = {}
    Fn._FORCE_COLLECTIVES = True
    assert Fn.collectives_active() and M._AllGatherCat._flat()
    # (1) the differentiable all-gather itself
    x = torch.randn(6, 14 * 512, device="cuda", requires_grad=True)
    y = M._AllGatherCat.apply(x * 1.0)
    g = torch.randn_like(y)
    y.backward(g)
    torch.cuda.synchronize()
    res["gather_fwd"], res["gather_bwd"] = bool(torch.equal(y, x)), bool(torch.equal(x.grad, g))
    yc = M.dist_collect(x.detach())
    res["collect"] = bool(torch.equal(yc, x)) and yc.data_ptr() != x.data_ptr()      # went through the collective, not the shortcut
    # (2) SyncBatchNorm's sum
    t = torch.randn(2 * 4096 + 1, device="cuda")
    res["sync_sum"] = bool(torch.equal(Fn._sync_sum(t.clone()), t))

    # (3) fine-tune and (4) pre-train steps, forced vs not
    def finetune(force):
        Fn._FORCE_COLLECTIVES = force
        model = M.BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY),
                                            task_config=task_config(max_frames=6)).cuda().train()
        batch = [t.cuda() for t in synth.finetune_batch(16, 6, 32, tag="det")]
        loss = model(*batch, 1)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    def pretrain(force):
        Fn._FORCE_COLLECTIVES = force
        gm = golden("moco_aswritten")
        K, B, Fr = int(gm["K"]), int(gm["B"]), int(gm["F"])
        cfg = task_config(contrast_num_negative=K, max_frames=Fr, dataset="chvtt")
        model = M.BirdPreTrainedModel.from_pretrained("cross-base", state_dict=synth.pretrain_state(synth.TINY, K, Fr),
                                                      task_config=cfg).cuda().train()
        out = []
        for step in range(2):
            model._mlm_draws = [torch.from_numpy(gm[f"mlm_{n}{step}"]) for n in ("masked", "replaced", "randsel", "words")]
            loss = model(*[t.cuda() for t in synth.pretrain_batch(B, Fr, tag=f"moco.s{step}")], step + 1)
            model.zero_grad(set_to_none=True)
            loss.backward()
            out.append(float(loss))
        torch.cuda.synchronize()
        S = model.state_dict()
        return out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, \
            {k: S[k].clone() for k in PT_BUFFERS}

    calls = {"n": 0}
    for name in ("all_gather_into_tensor", "reduce_scatter_tensor", "all_reduce"):
        orig = getattr(dist, name)

        def counted(*a, _o=orig, _n=name, **k):
            calls[_n] = calls.get(_n, 0) + 1
            return _o(*a, **k)
        setattr(dist, name, counted)
    l0, g0 = finetune(False)
    assert not calls.get("all_gather_into_tensor")
    l1, g1 = finetune(True)
    res["ft_calls"] = (calls.get("all_gather_into_tensor", 0), calls.get("reduce_scatter_tensor", 0))
    res["ft_loss"] = (l0, l1)
    res["ft_bad"] = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    calls.clear()
    p0 = pretrain(False)
    assert not calls.get("all_gather_into_tensor") and not calls.get("all_reduce")
    p1 = pretrain(True)
    res["pt_calls"] = (calls.get("all_gather_into_tensor", 0), calls.get("all_reduce", 0))
    res["pt_loss"] = (p0[0], p1[0])
    res["pt_bad"] = [n for n in p0[1] if not torch.equal(p0[1][n], p1[1][n])] + [k for k in p0[2] if not torch.equal(p0[2][k], p1[2][k])]
    torch.save(res, os.path.join(out_dir, "rccl_coll.pt"))
    dist.destroy_process_group()


def test_rccl_collectives_of_the_path_execute_and_are_exact():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_run_rccl_collectives, args=(d, port), nprocs=1, join=True)
        r = torch.load(os.path.join(d, "rccl_coll.pt"))
    assert r["gather_fwd"] and r["gather_bwd"] and r["collect"] and r["sync_sum"], r
    assert r["ft_calls"] == (1, 1), r["ft_calls"]                 # one packed all-gather forward, one reduce-scatter backward
    assert r["ft_loss"][0] == r["ft_loss"][1] and not r["ft_bad"], (r["ft_loss"], r["ft_bad"][:5])
    # per pre-train step: one packed key gather; SyncBN: 3 forward sums (v_projector, v_predictor, v_projector_k) + 2 backward
    assert r["pt_calls"][0] == 2 and r["pt_calls"][1] == 2 * 5, r["pt_calls"]
    assert r["pt_loss"][0] == r["pt_loss"][1] and not r["pt_bad"], (r["pt_loss"], r["pt_bad"][:5])


# ----------------------------------------------------------------------------- bench.py starts its own ranks

def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts the N = 1 run) must start two ranks
    itself, as children, and print ONE JSON line from rank 0.  Rehearsed here on the one GPU over gloo (RCCL refuses two
    ranks on one device); the code path - self-launch, rendezvous on 127.0.0.1, DDP wrap, barriers, max-over-ranks timing,
    roofline steps - is the one the 8-GPU run takes (reference launch: README.md:83, main_task_retrieval.py:207-208)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HMMC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "8", "--frames", "4", "--steps", "2",
                        "--warmup", "1", "--roofline-steps", "1", "--vit-forward-iters", "1", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["per_gpu_batch"] == 4
    assert out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["final_loss"] == out["final_loss"] and abs(out["final_loss"]) < 50       # finite
    assert out["config"]["gemm_reserved_cus"] == 16 and out["roofline"]["launches_per_step"] > 0
    # the communication record a first scaling curve is read against (bench.py:comm_leg)
    comm = out["comm"]
    for key in ("allreduce_ms", "allgather_ms", "reduce_scatter_ms", "step_ms_no_sync", "grad_bytes", "buckets", "allreduce_bus_gb_per_s"):
        assert comm[key] > 0, (key, comm)
    assert comm["backend"] == "gloo" and comm["feature_bytes_per_rank"] == 4 * (4 + 2) * 512 * 4
    # weak scaling: --batch is per GPU, the line says so
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", "weak", "--batch", "4", "--frames", "4",
                        "--steps", "1", "--warmup", "1", "--roofline-steps", "1", "--vit-forward-iters", "0", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["config"]["per_gpu_batch"] == 4 and out["config"]["global_batch"] == 8
