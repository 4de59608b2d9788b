#!/usr/bin/env python3
"""HMMC hot-path benchmark: video-text pairs / second of one training step
(forward + backward + global grad clip + BertAdam step, reference main_task_retrieval.py:272-302 /
main_pretrain.py:213-245) on synthetic batches.  Default = BASELINE.json's headline config: MSR-VTT-shaped fine-tuning
[B=256, F=12, 3x224x224], ViT-B/32 + CLIP text transformer.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --mode pretrain --batch 128                       # SURVEY config 4 (FAM+VTM+FTM+MLM, MoCo K=1024)
    python bench.py --mode eval --frames 24                           # eval leg at VATEX size: 15 000 x 1 500 x 24
    python bench.py --clip ViT-B/16 --frames 24 --batch 16            # one rank's share of SURVEY config 5
    python bench.py --regime fp32                                     # the reference's model.float() recipe: the regime held to 1e-3
    python bench.py --gpus N --steps K --warmup W                     # starts its own N ranks (children, torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--batch` is the GLOBAL batch (the reference's --batch_size semantics, dataloaders/dataloader.py:84), split
over ranks: strong scaling.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0           # dense fp16/bf16, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.0        # exact-f32 MFMA (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md


def flop_model(dims, frames, text_len):
    """Algorithmic FLOPs (2 x MAC of GEMM-shaped work, the reference's formulation incl. its all-token final projection),
    SURVEY.md section 8(d): ViT-B/32, F=12, L=32 -> 8.856 GFLOP per frame, 324.33 GFLOP per pair per fine-tune step."""
    def tower(L, D, layers):
        return layers * (2 * L * D * 3 * D + 4 * L * L * D + 2 * L * D * D + 16 * L * D * D)
    g = dims.image_res // dims.patch
    Lv, D, E = g * g + 1, dims.vision_width, 512
    patch = 2 * (Lv - 1) * 3 * dims.patch * dims.patch * D
    frame = patch + tower(Lv, D, dims.vision_layers) + 2 * Lv * D * E
    text = tower(text_len, dims.text_width, dims.text_layers) + 2 * text_len * dims.text_width * E
    temporal = tower(frames, E, 4)
    fwd = frames * frame + text + temporal
    lead = 3 * frames * (2 * Lv * D * D + 16 * Lv * D * D) * (Lv - 1) / Lv    # pruned: last block's out_proj + MLP off the class token
    # forward work this path executes per frame: the last block's per-token half and ln_post / proj on the class token only
    frame_exec = frame - (2 * Lv * D * D + 16 * Lv * D * D + 2 * Lv * D * E) * (Lv - 1) / Lv
    return {"frame_fwd": frame, "frame_fwd_executed": frame_exec, "pair_fwd": fwd, "pair_train": 3 * fwd - frames * patch,
            "pair_train_executed": 3 * fwd - frames * patch - lead}


FLOP_PER_PAIR_PRETRAIN_C4 = 457.35e9     # B=128, F=12, title 45 / tag 25, K=1024


def task_config(**kw):
    d = dict(local_rank=0, rank=0, use_temp=True, language="english", top_frames=2, max_frames=12, n_display=10 ** 9,
             logdir=None, use_frame_fea=True, dataset="msrvtt", lr=1e-4, text_lr=3e-5, coef_lr=1e-3, weight_decay=0.2,
             warmup_proportion=0.1)
    d.update(kw)
    return Namespace(**d)


def eval_leg(args, dev):
    """The eval scorer over cached features (main_task_retrieval.py:321-357,512-513): queries x (video + F frames) logits,
    mean of the top-k frame logits, one [queries, videos] matrix.  A step = one pass over all queries and videos."""
    from hmmc_amd import ops
    from hmmc_amd.modeling import BirdModel
    F = args.frames
    nq, nv = args.eval_queries, args.eval_videos
    cfg = task_config(local_rank=0, rank=0, max_frames=F, pretrained_clip_name=args.clip)
    torch.manual_seed(42)
    model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(7)
    E = 512
    qs = [torch.randn(min(64, nq - s), E, generator=g, device=dev) for s in range(0, nq, 64)]          # loader batches
    vs = [torch.randn(min(64, nv - s), E, generator=g, device=dev) for s in range(0, nv, 64)]
    us = [torch.randn(min(64, nv - s), F, E, generator=g, device=dev) for s in range(0, nv, 64)]
    for _ in range(args.warmup):
        sim = model.eval_similarity(qs, vs, us)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim = model.eval_similarity(qs, vs, us)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.raise_on_device_errors()
    # the scorer alone, HIP events on the launch stream
    packed = ops.eval_pack(torch.cat(vs).float().contiguous(), torch.cat(us).float().contiguous())
    qn, _ = ops.l2norm_fwd(torch.cat(qs).float().contiguous())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launches = 0
    e0.record()
    for _ in range(args.steps):
        for s0 in range(0, nq, 4096):
            ops.eval_score(qn[s0:s0 + 4096], packed, nv, F, model.top_frames, 100.0, want=("score",))
            launches += 1
    e1.record()
    torch.cuda.synchronize()
    us_launch = e0.elapsed_time(e1) * 1e3 / launches
    flops = 2.0 * nq * nv * (F + 1) * E / (launches / args.steps)
    slots = ops.eval_slots(F)
    roof = {"bound": "mfma", "kernel": "gemm_f32_kernel<..., TOPK> (exact-f32 MFMA scorer with the top-k epilogue)",
            "achieved": round(flops / us_launch / 1e6, 1), "peak": 157.0, "unit": "TFLOP/s",
            "frac": round(flops / us_launch / 1e6 / 157.0, 4), "traffic": None,
            "avg_launch_us": round(us_launch, 1), "launches_per_step": launches // args.steps,
            "note": f"useful FLOPs 2 nq nv (F+1) 512; the kernel multiplies {slots} slots per video (padding: x{slots / (F + 1):.2f}); "
                    f"algorithmic HBM bytes {(nq * E + nv * (F + 1) * E + nq * nv) * 4 / 1e6:.0f} MB per step: not HBM-bound"}
    out = {"metric": f"query-video pairs scored/sec, eval leg {nq} x {nv} x F={F} (top-{model.top_frames} frames)",
           "value": round(nq * nv / dt, 1), "unit": "query-video pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": f"cached-feature retrieval scoring, {nq} queries x {nv} videos x {F} frames, 512-d fp32 features "
                                  f"(concatenate + normalise + pack + score)", "mode": "eval"},
           "roofline": roof, "checksum": round(float(sim.double().sum()), 3)}
    if not args.no_cpu_baseline:
        sys.path.insert(0, ROOT)
        import oracle.hmmc_oracle as O
        n_s = min(nq, 1500)
        q, v, u = torch.cat(qs)[:n_s].cpu(), torch.cat(vs).cpu(), torch.cat(us).cpu()
        torch.set_num_threads(_usable_cores())     # as cpu_baseline()
        ls = float(model.text_encoder.logit_scale)

        def cpu_scores(qq):
            sv = O.loose_similarity(qq, v, ls)
            return sv + torch.topk(O.loose_similarity(qq, u, ls), k=model.top_frames, dim=2)[0].mean(dim=2)
        cpu_scores(q[:64])
        t0 = time.perf_counter()
        ref = cpu_scores(q)
        t = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n_s * nv / t, 1), "unit": "query-video pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"oracle eval_scores on {n_s} of the queries x {nv} videos x {F} frames, {t:.2f} s, {_cpu_model()}"}
        out["max_abs_err_vs_oracle_sample"] = float((sim[:n_s] - ref.to(dev)).abs().max())
    print(json.dumps(out), flush=True)


def prep_optimizer(model, cfg, t_total):
    """Parameter groups exactly as main_task_retrieval.py:171-205."""
    from hmmc_amd.optimization import BertAdam
    named = list(model.named_parameters())
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    dec = [(n, p) for n, p in named if not any(nd in n for nd in no_decay)]
    nod = [(n, p) for n, p in named if any(nd in n for nd in no_decay)]
    wd, lrc = cfg.weight_decay, cfg.lr * cfg.coef_lr
    groups = [
        {"params": [p for n, p in dec if "visual_encoder.visual." in n], "weight_decay": wd, "lr": lrc},
        {"params": [p for n, p in dec if "text_encoder." in n], "weight_decay": wd, "lr": cfg.text_lr},
        {"params": [p for n, p in dec if "visual_encoder.visual." not in n and "text_encoder." not in n], "weight_decay": wd},
        {"params": [p for n, p in nod if "visual_encoder.visual." in n], "weight_decay": 0.0, "lr": lrc},
        {"params": [p for n, p in nod if "text_encoder." in n], "weight_decay": 0.0, "lr": cfg.text_lr},
        {"params": [p for n, p in nod if "visual_encoder.visual." not in n and "text_encoder." not in n], "weight_decay": 0.0},
    ]
    return BertAdam(groups, lr=cfg.lr, warmup=cfg.warmup_proportion, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6,
                    t_total=t_total, weight_decay=wd, max_grad_norm=1.0)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def _usable_cores():
    """Host cores this process can really use: the affinity mask capped by the cgroup's CPU quota.  The GPU boxes pin nothing
    (256 CPUs visible) but grant a quota of 16 cores (cpu.max = 1600000 100000): 64 threads on that quota ran the GEMM-bound
    oracle step SLOWER than 16 (1 135 vs 1 551 GFLOP/s on a 4096^3 matmul) - round 2's "64 threads" baseline was throttled."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                                           # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                pr = int(f.read())
            if q > 0:
                n = min(n, max(1, -(-q // pr)))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def _cpu_steps(B, frames, length, threads, warmups, min_steps, seconds, max_steps=5):
    """Timed oracle steps on `threads` host threads -> (seconds per step, n, {"fwd", "bwd", "clip+opt"} seconds per step)."""
    from hmmc_amd import synth
    from oracle import hmmc_oracle as O
    torch.set_num_threads(threads)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synth.finetune_state(synth.VIT_B32).items()}
    params = [v for v in sd.values() if v.requires_grad]
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    ids, mask, vid, vf, idx = synth.finetune_batch(B, frames, length, tag="cpu_baseline")
    split = {"fwd": 0.0, "bwd": 0.0, "clip+opt": 0.0}

    def step(i, timed):
        for p in params:
            p.grad = None
        t0 = time.time()
        loss, _ = O.finetune_loss(ids, vid, sd, mode="fp32")
        t1 = time.time()
        loss.backward()
        t2 = time.time()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        with torch.no_grad():
            for k, p in sd.items():
                if p.requires_grad and p.grad is not None:
                    m, v = state[k]
                    np_, m, v, _ = O.bert_adam_step(p.data, p.grad, m, v, i, 1e-4, 1000, 0.1, 0.2)
                    p.data.copy_(np_)
                    state[k] = (m, v)
        t3 = time.time()
        if timed:
            split["fwd"] += t1 - t0
            split["bwd"] += t2 - t1
            split["clip+opt"] += t3 - t2
    print(f"[bench] cpu_baseline: oracle step, B={B}, on {threads} host threads ...", file=sys.stderr, flush=True)
    for w in range(warmups):
        t0 = time.time()
        step(w, False)
        print(f"[bench] cpu_baseline: warm-up step {w + 1} took {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    t0 = time.time()
    n = 0
    while n < min_steps or (time.time() - t0 < seconds and n < max_steps):
        step(warmups + n, True)
        n += 1
        print(f"[bench] cpu_baseline: step {n} at {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = (time.time() - t0) / n
    return dt, n, {k: round(v / n, 3) for k, v in split.items()}


def cpu_baseline(frames, length, batch=16, warmups=2, min_steps=2, seconds=40.0):
    """The oracle's fp32 restatement of the same fine-tune step, timed on this host's cores (rank 0, N=1 only).
    BASELINE.md section 4: batch large enough for a step of 10-60 s (CPU pairs/s is flat in B past B ~ 8), >= 2 warm-up
    steps, CPU model and core count reported; the step is split into forward / backward / clip + BertAdam, and the same
    step is also timed on 8 threads - BASELINE.md section 4's cross-check against the reference itself, which ran 1.77
    pairs/s (fp32) on the 8 cores of the survey container."""
    cores = _usable_cores()
    B = batch
    dt, n, split = _cpu_steps(B, frames, length, cores, warmups, min_steps, seconds)
    out = {"value": round(B / dt, 4), "unit": "video-text pairs/s", "cores": cores, "kind": "port",
           "sample": f"oracle fp32 restatement of the fine-tune step, B={B} F={frames} L={length} ViT-B/32, {warmups} warm-up + {n} "
                     f"timed steps of {dt:.2f} s (fwd+bwd+clip+BertAdam), {cores} threads of {_cpu_model()}, torch CPU {torch.__version__}",
           "seconds_per_step_split": split}
    if cores > 8:
        b8 = 4
        dt8, n8, split8 = _cpu_steps(b8, frames, length, 8, 1, 2, 15.0, max_steps=3)
        out["cross_check_8_threads"] = {"value": round(b8 / dt8, 4), "unit": "video-text pairs/s", "cores": 8, "batch": b8,
                                        "seconds_per_step_split": split8,
                                        "note": "BASELINE.md section 4: the reference itself ran 1.77 pairs/s (fp32, B=4 F=4) / 0.64 (as "
                                                "written) on 8 threads of an Intel Xeon 2.1 GHz; this is the restatement on 8 threads of this host"}
        torch.set_num_threads(cores)
    return out


def _gemm_source_hash():
    import hashlib
    with open(os.path.join(ROOT, "hmmc_amd", "csrc", "gemm_f16.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def recorded_traffic(args, per_gpu_batch, launches_per_step):
    """HBM bytes per gemm_f16_kernel launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate
    rocprofv3 --pmc runs of this same command, scratch/pmc_traffic.py).  The counters cannot be read from inside the
    process, so the figure is reported only for the workload AND the kernel source it was collected on: the record carries
    the hash of gemm_f16.hip and the launches per step, and a mismatch reports the traffic as stale (None)."""
    if not (args.mode == "finetune" and per_gpu_batch == 256 and args.frames == 12 and args.length == 32 and args.clip == "ViT-B/32"):
        return None, "not collected for this workload"
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_f16_hbm_traffic.json")))
    if not paths:
        return None, "no PMC record"
    path = paths[-1]
    with open(path) as f:
        rec = json.load(f)
    rel = os.path.relpath(path, ROOT)
    if rec.get("gemm_f16_hip_sha256_16") != _gemm_source_hash() or rec.get("launches_per_step") != launches_per_step:
        return None, f"{rel} is stale (collected on another gemm_f16.hip / launch count); re-run scratch/pmc_traffic.py"
    return round(rec["hbm_traffic_per_launch_bytes"]), f"{rel} (rocprofv3 --pmc, separate passes)"


HBM_PEAK_TBS = 8.0                  # HBM3E, MI355X_MICROARCH.md (6.3 TB/s measured on a float4 copy)


def _time_us(fn, reps=10):
    """average microseconds of fn() over `reps` launches, HIP events on the current stream (after one untimed call)"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def hbm_rooflines(dev, b, frames, dims, params, optimizer, model, pretrain):
    """north_star: "rocprof HBM GB/s ... reported against peak" for the HBM-bound kernels of the step.  Each kernel alone, on
    this rank's shapes (one ViT layer's LayerNorm / attention; the whole model's optimizer), timed with events on the current
    stream; bytes = algorithmic (every operand once).  The clip + BertAdam figures include their small helper launches."""
    from hmmc_amd import ops
    from hmmc_amd.optimization import clip_grad_norm_
    g = dims.image_res // dims.patch
    L, D, H = g * g + 1, dims.vision_width, dims.vision_width // 64
    nseq = b * frames
    T = nseq * L
    gen = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn((T, D), generator=gen, device=dev).half()
    dy = torch.randn((T, D), generator=gen, device=dev).half()
    dres = torch.randn((T, D), generator=gen, device=dev).half()
    gm, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-5)
    qkv = torch.randn((T, 3 * D), generator=gen, device=dev).half()
    att, lse = ops.attention_f16_fwd(qkv, nseq, L, H, False)
    rows = []

    def add(kernel, nbytes, us):
        rows.append({"kernel": kernel, "bytes": int(nbytes), "us": round(us, 1), "tb_per_s": round(nbytes / us / 1e6, 2),
                     "frac_of_hbm_peak": round(nbytes / us / 1e6 / HBM_PEAK_TBS, 3)})
    add(f"ln_fwd [{T}, {D}] fp16", 2 * T * D * 2, _time_us(lambda: ops.layernorm_fwd(x, gm, bt, 1e-5)))
    add(f"ln_bwd [{T}, {D}] fp16 (+ residual gradient, dx column sums)", 4 * T * D * 2,
        _time_us(lambda: ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=dres, want_colsum=True)))
    st = ops.rowstat(x)
    add(f"ln_bwd_fold [{T}, {D}] fp16 (backward of a folded LayerNorm: + residual gradient, dx column sums)", 4 * T * D * 2,
        _time_us(lambda: ops.layernorm_bwd_fold(dy, x, st, dres=dres, want_colsum=True, reduce=False)))
    add(f"attn_fwd {nseq} x {L} tokens x {H} heads", 4 * T * D * 2, _time_us(lambda: ops.attention_f16_fwd(qkv, nseq, L, H, False)))
    add(f"attn_bwd {nseq} x {L} tokens x {H} heads (+ in_proj bias partials)", 7 * T * D * 2,
        _time_us(lambda: ops.attention_f16_bwd(qkv, att, lse, dy, nseq, L, H, False, want_dbias=True)))
    del x, dy, dres, y, qkv, att, lse
    pbytes = sum(p.numel() * p.element_size() for p in params)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    lr_state = [(grp, grp["lr"]) for grp in optimizer.param_groups]
    for grp, _ in lr_state:
        grp["lr"] = 0.0                                      # same kernels and traffic, the weights stay as they are
    add(f"clip_grad_norm_ ({len(params)} tensors, {pbytes / 1e6:.0f} MB: read twice, scaled in place)", 3 * pbytes,
        _time_us(lambda: clip_grad_norm_(params, 1.0), reps=5))
    add(f"BertAdam.step ({len(params)} tensors: p, g, m, v in; p, m, v out)", 7 * pbytes, _time_us(optimizer.step, reps=5))
    for grp, lr in lr_state:
        grp["lr"] = lr
    if pretrain:
        nb = sum(p.numel() * p.element_size() for pair in model.model_pairs for p in pair[0].parameters())
        m0 = model.contrast_momentum
        model.contrast_momentum = 1.0                        # p_k = p_k * 1 + p * 0: same launch, the keys stay as they are
        add("momentum update (mt_ema: p_k, p in; p_k out)", 3 * nb, _time_us(model._momentum_update, reps=5))
        model.contrast_momentum = m0
    return rows


def comm_leg(args, dev, model, net, params, step, b, frames, repeats=5):
    """world > 1: what the first scaling curve needs beside it - the gradient all-reduce in DDP's bucket sizes
    (main_task_retrieval.py:207-208), the packed feature all-gather and the reduce-scatter of its gradient (one collective
    each for the reference's three all_gathers, modules/modeling.py:698-700), each alone on an idle GPU, and a step without
    gradient synchronisation (DDP.no_sync)."""
    world = dist.get_world_size()
    flat = dist.get_backend() == "nccl"
    sizes = [p.numel() * p.element_size() for p in params]
    limits = [1024 * 1024, int(getattr(net, "bucket_bytes_cap", 25 * 1024 * 1024))]
    try:
        buckets, _ = dist._compute_bucket_assignment_by_size(list(reversed(params)), limits)
        bucket_bytes = [sum(sizes[len(params) - 1 - i] for i in idx) for idx in buckets]
    except Exception:                                                        # private helper: fall back to the cap alone
        total, bucket_bytes = sum(sizes), []
        while total > 0:
            bucket_bytes.append(min(total, limits[1]))
            total -= bucket_bytes[-1]
    bufs = [torch.zeros(max(nb // 2, 1), dtype=torch.float16, device=dev) for nb in bucket_bytes]

    def allreduce_all():
        works = [dist.all_reduce(t, async_op=True) for t in bufs]
        for w in works:
            w.wait()
    E = 512
    packed = torch.zeros((b, (frames + 2) * E), dtype=torch.float32, device=dev)
    gathered = torch.zeros((b * world, (frames + 2) * E), dtype=torch.float32, device=dev)

    def gather():
        if flat:
            dist.all_gather_into_tensor(gathered, packed)
        else:
            dist.all_gather(list(gathered.chunk(world, dim=0)), packed)

    def scatter():
        if flat:
            dist.reduce_scatter_tensor(packed, gathered, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(gathered, op=dist.ReduceOp.SUM)                  # the gloo form of _AllGatherCat.backward
    out = {"backend": dist.get_backend(), "grad_bytes": int(sum(sizes)), "buckets": len(bucket_bytes),
           "bucket_bytes_max": int(max(bucket_bytes)), "allreduce_ms": round(_time_us(allreduce_all, repeats) / 1e3, 3),
           "feature_bytes_per_rank": int(packed.numel() * 4), "allgather_ms": round(_time_us(gather, repeats) / 1e3, 3),
           "reduce_scatter_ms": round(_time_us(scatter, repeats) / 1e3, 3)}
    out["allreduce_bus_gb_per_s"] = round(2 * (world - 1) / world * out["grad_bytes"] / (out["allreduce_ms"] * 1e-3) / 1e9, 1)
    if hasattr(net, "no_sync"):
        with net.no_sync():
            step(10 ** 6)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                step(10 ** 6 + 1 + i)
            torch.cuda.synchronize()
            out["step_ms_no_sync"] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    t = torch.tensor([out["allreduce_ms"], out["allgather_ms"], out["reduce_scatter_ms"], out.get("step_ms_no_sync", 0.0)],
                     dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                 # slowest rank, like the timed region
    out["allreduce_ms"], out["allgather_ms"], out["reduce_scatter_ms"] = [round(float(v), 3) for v in t[:3]]
    if "step_ms_no_sync" in out:
        out["step_ms_no_sync"] = round(float(t[3]), 2)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--eval-queries", type=int, default=15000, help="--mode eval: captions (VATEX test: 1 500 videos x 10)")
    ap.add_argument("--eval-videos", type=int, default=1500, help="--mode eval: candidate videos")
    ap.add_argument("--mode", choices=("finetune", "pretrain", "eval"), default="finetune",
                    help="finetune: BirdModel (configs 2/3/5); pretrain: BirdPreTrainedModel, FAM+VTM+FTM+MLM, MoCo queues (config 4)")
    ap.add_argument("--batch", type=int, default=None, help="global batch (reference --batch_size); default 256 / 128 (pretrain); "
                                                            "with --scaling weak: the batch PER GPU")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): --batch is the global batch, split over ranks (the reference's --batch_size semantics, "
                         "dataloaders/dataloader.py:84); weak: --batch per GPU, global batch = batch x gpus (SURVEY 8d)")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--length", type=int, default=32, help="caption length (fine-tune)")
    ap.add_argument("--title-length", type=int, default=45)
    ap.add_argument("--tag-length", type=int, default=25)
    ap.add_argument("--negatives", type=int, default=1024, help="MoCo queue length K (pretrain)")
    ap.add_argument("--clip", default="ViT-B/32")
    ap.add_argument("--regime", choices=("f16", "fp32"), default="f16",
                    help="f16: the towers as the reference builds them (convert_weights: fp16 weights and activations); fp32: after "
                         "model.float() + text_encoder.dtype = float32, the reference's own fp32-upcast recipe "
                         "(modules/module_clip.py:566-577) - every stage in exact fp32, the regime whose logits are held to 1e-3 "
                         "(tests/test_gpu_fp32_regime.py); roofline against the 157 TFLOP/s exact-f32 MFMA peak")
    ap.add_argument("--unfolded-steps", type=int, default=3,
                    help="fine-tune / pre-train, f16 regime: extra timed steps with HMMC_FOLD_LN=0 HMMC_FOLD_LN_TRAIN=0 - the kernels "
                         "whose rounding points are the reference's - reported as ms_per_step_unfolded (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="run everything on one stream (no tower / weight-gradient overlap): the mode whose rocprofv3 "
                         "per-kernel averages are comparable with roofline.avg_launch_us")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units kept out of the GEMM grids for RCCL (default: 16 when --gpus > 1, else 0); "
                         "giving it with --gpus 1 measures what the reservation costs")
    ap.add_argument("--roofline-steps", type=int, default=3,
                    help="extra single-stream steps after the timed region over which the GEMM launches are timed with HIP events")
    ap.add_argument("--no-hbm-roofline", action="store_true",
                    help="skip the roofline_hbm leg (its stand-alone LayerNorm / attention / optimizer launches would otherwise sit in a "
                         "kernel trace of this command)")
    ap.add_argument("--vit-forward-iters", type=int, default=5, help="timed frame-encoder forward passes for the vit_forward record (0: skip)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 256 if args.mode == "finetune" else 128

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start one rank per GPU under torch.distributed.run (the reference is
        # launched the same way, README.md:83) as CHILD processes.  Nothing in this process has touched the GPU yet - and it
        # never will: it only forwards the children's output and exit code (no exec of a GPU-initialised process).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    local_dev = local_rank % max(torch.cuda.device_count(), 1)      # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HMMC_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; gloo only for rehearsals
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from hmmc_amd import ops, synth
    from hmmc_amd.modeling import BirdModel, BirdPreTrainedModel
    from hmmc_amd.optimization import clip_grad_norm_

    if args.mode == "eval":
        if world > 1:
            raise SystemExit("--mode eval scores cached features on one GPU (main_task_retrieval.py:321-357 runs it on rank 0)")
        return eval_leg(args, dev)

    if args.reserve_cus is not None:
        os.environ["HMMC_RCCL_CUS"] = str(args.reserve_cus)
    reserved = ops.reserve_cus_for_collectives() if (world > 1 or args.reserve_cus is not None) else 0
    if args.single_stream:
        import hmmc_amd.functional as _fn0
        import hmmc_amd.modeling as _md0
        _md0._OVERLAP_TOWERS, _fn0._WGRAD_STREAM = False, False
    if args.scaling == "weak":
        b, args.batch = args.batch, args.batch * world              # --batch was per GPU; from here on args.batch is global
    else:
        assert args.batch % world == 0
        b = args.batch // world
    dims = synth.NAMED[args.clip]
    pretrain = args.mode == "pretrain"
    extra = dict(dataset="chvtt", contrast_momentum=0.99, contrast_temperature=0.07, contrast_num_negative=args.negatives,
                 pretrained_text=None) if pretrain else {}
    cfg = task_config(local_rank=local_rank, rank=rank, max_frames=args.frames, pretrained_clip_name=args.clip, **extra)
    torch.manual_seed(42)
    cls = BirdPreTrainedModel if pretrain else BirdModel
    model = cls.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
    fp32 = args.regime == "fp32"
    if fp32:
        model.float()                                    # CLIP(...).float() of the reference (modules/module_clip.py:566-577) ...
        model.text_encoder.dtype = torch.float32         # ... and the stored dtype attribute the text path casts to (module_cross.py:256)
        if pretrain:
            model.text_encoder_k.dtype = torch.float32
    optimizer = prep_optimizer(model, cfg, t_total=1000)
    net = model
    if world > 1:
        # the pre-training model builds t_projector and never uses it (reference main_pretrain.py:204: find_unused_parameters)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_dev], output_device=local_dev,
                                                        find_unused_parameters=pretrain, gradient_as_bucket_view=True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    res = dims.image_res
    video = torch.randn((b, args.frames, 3, res, res), generator=g, device=dev)
    vf = torch.full((b,), args.frames, dtype=torch.long, device=dev)
    if pretrain:
        title, tmask = [t.to(dev) for t in synth.token_ids(f"bench.title.{rank}", b, args.title_length)]
        tag, gmask = [t.to(dev) for t in synth.token_ids(f"bench.tag.{rank}", b, args.tag_length)]
        inputs = (video, vf, tag, gmask, title, tmask)
    else:
        ids, mask = [t.to(dev) for t in synth.token_ids(f"bench.ids.{rank}", b, args.length)]
        inputs = (ids, mask, video, vf, torch.arange(b, device=dev))
    params = [p for p in model.parameters() if p.requires_grad]

    def step(i):
        loss = net(*inputs, i)
        loss.backward()
        clip_grad_norm_(params, 1.0)
        optimizer.step()
        optimizer.zero_grad()
        return loss

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.detach())
    ops.raise_on_device_errors()

    import hmmc_amd.functional as _fn
    import hmmc_amd.modeling as _md
    # The same step on the unfolded kernels (HMMC_FOLD_LN=0 HMMC_FOLD_LN_TRAIN=0: LayerNorm then GEMM, every rounding point where
    # the reference has it) - the default folds ln_1 / ln_2 of the frame tower into in_proj / c_fc.
    ms_unfolded = None
    if args.unfolded_steps > 0 and not fp32:
        pol = (_fn._FOLD_LN, _fn._FOLD_LN_TRAIN)
        _fn._FOLD_LN, _fn._FOLD_LN_TRAIN = "0", "0"
        step(args.warmup + args.steps)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        tu = time.perf_counter()
        for i in range(args.unfolded_steps):
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tun = torch.tensor([time.perf_counter() - tu], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tun, op=dist.ReduceOp.MAX)
        ms_unfolded = float(tun.item()) / args.unfolded_steps * 1e3
        _fn._FOLD_LN, _fn._FOLD_LN_TRAIN = pol
        step(args.warmup + args.steps)                   # back on the default kernels (workspaces, allocator state)

    # Roofline of the dominant kernel.  The timed region above runs the text tower, the frame tower and the weight
    # gradients on three streams, so a launch's HIP events there also bracket the time it waits for CUs held by another
    # stream's kernels.  The per-launch durations are therefore taken over `--roofline-steps` further steps of the same
    # loop with the overlap switched off (one stream; this is also what rocprofv3 --kernel-trace sees, it serialises
    # dispatches), events recorded on the launch stream around every hmmc_gemm_f16 call.
    ov = (_md._OVERLAP_TOWERS, _fn._WGRAD_STREAM)
    # The text tower's AccumulateGrad nodes were created under the side stream; as long as the last graph (`loss`) lives they are
    # reused, and torch warns ("AccumulateGrad node's stream does not match") when the single-stream steps below produce their
    # gradients on the main stream.  Dropping the graph first lets the next forward create them under the stream it runs on.
    # (scratch/accgrad_warning.py: the steady-state loop never warns; a direct text-encoder call on the main stream before the
    # overlapped forward - what several tests do - does, once.)
    del loss
    prof, dt_single = {}, 0.0
    if args.roofline_steps > 0:                          # (0: nothing after the timed region - kernel traces of the plain loop)
        import gc
        gc.collect()
        _md._OVERLAP_TOWERS, _fn._WGRAD_STREAM = False, False
        step(args.warmup + args.steps)                   # settle the single-stream workspaces
        torch.cuda.synchronize()
        ops.gemm_profile_start()
        t1 = time.perf_counter()
        for i in range(args.roofline_steps):
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        dt_single = time.perf_counter() - t1
        prof = ops.gemm_profile_stop()
        _md._OVERLAP_TOWERS, _fn._WGRAD_STREAM = ov

    # ViT forward alone (north_star: MFMA utilisation of the ViT forward): the frame encoder over this rank's b x F frames in
    # eval mode, timed with events on the current stream; FLOPs in the reference's formulation (all-token final projection)
    fm = flop_model(dims, args.frames, args.length)
    vit_forward = None
    if args.vit_forward_iters > 0 and not fp32:
        frames_flat = video.view(b * args.frames, 3, res, res)
        enc = model.visual_encoder

        def vit_ms():
            with torch.no_grad():
                return _time_us(lambda: enc.encode_image(frames_flat), args.vit_forward_iters) / 1e3
        policy = _fn._FOLD_LN
        vms = vit_ms()
        _fn._FOLD_LN = "0"                                  # the same pass on the unfolded kernels (HMMC_FOLD_LN=0), for the record
        vms_unfolded = vit_ms()
        _fn._FOLD_LN = policy
        vtf = b * args.frames * fm["frame_fwd"] / (vms * 1e-3) / 1e12
        vte = b * args.frames * fm["frame_fwd_executed"] / (vms * 1e-3) / 1e12
        vit_forward = {"ms": round(vms, 3), "frames": b * args.frames, "tflops_executed": round(vte, 1),
                       "frac_of_mfma_peak": round(vte / MFMA_PEAK_TFLOPS, 4),
                       "tflops_reference_formulation": round(vtf, 1),
                       "frac_of_mfma_peak_reference_formulation": round(vtf / MFMA_PEAK_TFLOPS, 4),
                       "ln_fold_policy": policy,
                       "ms_unfolded_kernels": round(vms_unfolded, 3),
                       "frac_of_mfma_peak_unfolded_kernels": round(vte * vms / vms_unfolded / MFMA_PEAK_TFLOPS, 4),
                       "note": "frame encoder forward (patch embed + blocks + ln_post/proj) of one rank's frames, no_grad, as the "
                               "product runs it by default: ln_1 / ln_2 folded into in_proj / c_fc (hmmc_tower_fwd_fused; same "
                               "FLOPs, no LayerNorm pass; as close to the reference's fp32 regime as the reference's own fp16 "
                               "regime, tests/test_gpu_fold.py); *_unfolded_kernels: HMMC_FOLD_LN=0, the kernels whose rounding "
                               "points follow the reference's. frac_of_mfma_peak counts the FLOPs this path executes (last block's "
                               "per-token half and the projection on the class token only), the reference-formulation figure "
                               "counts the reference's"}
    hbm = hbm_rooflines(dev, b, args.frames, dims, params, optimizer, model, pretrain) if world == 1 and not args.no_hbm_roofline else None
    comm = comm_leg(args, dev, model, net, params, step, b, args.frames) if world > 1 else None

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * args.steps / dt
        # the dominant kernel family: the fp16 GEMM (three operand layouts) as built, the exact-fp32 GEMM in the fp32 regime
        f32_prof = prof.pop("f32", None)
        if fp32:
            prof = {"f32": f32_prof} if f32_prof else {}
        peak = MFMA_F32_PEAK_TFLOPS if fp32 else MFMA_PEAK_TFLOPS
        flops = sum(p["flops"] for p in prof.values())
        secs = sum(p["seconds"] for p in prof.values())
        launches = sum(p["launches"] for p in prof.values())
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        lps = launches // max(args.roofline_steps, 1)
        roof = {"bound": "mfma", "kernel": ("gemm_f32_*kernel (exact-f32 MFMA 16x16x4 GEMM, all orientations and tile kernels)" if fp32 else
                                            "gemm_f16_kernel (fp16 MFMA GEMM, all operand layouts)"),
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None, "traffic_unit": "bytes/launch",
                "launches_per_step": lps,
                "avg_launch_us": round(secs / max(launches, 1) * 1e6, 2),
                "measured_over": f"{args.roofline_steps} single-stream steps after the timed region "
                                 f"({dt_single / max(args.roofline_steps, 1) * 1e3:.2f} ms/step without the stream overlap)",
                "gemm_share_of_step": round(secs / dt_single, 4) if dt_single > 0 else None,
                "algorithmic_bytes_per_launch": round(sum(p.get("bytes", 0) for p in prof.values()) / max(launches, 1)) or None,
                "by_layout": {k: {"tflops": round(p["flops"] / p["seconds"] / 1e12, 1), "launches": p["launches"],
                                  "avg_us": round(p["seconds"] / p["launches"] * 1e6, 2)} for k, p in prof.items()}}
        if fp32:
            roof["traffic_source"] = "not collected for the fp32 regime"
        else:
            roof["traffic"], roof["traffic_source"] = recorded_traffic(args, b, lps)
            if f32_prof:
                roof["gemm_f32_beside_it"] = {"tflops": round(f32_prof["flops"] / f32_prof["seconds"] / 1e12, 1), "launches": f32_prof["launches"] // max(args.roofline_steps, 1),
                                              "ms_per_step": round(f32_prof["seconds"] / max(args.roofline_steps, 1) * 1e3, 3),
                                              "note": "hmmc_gemm_f32 launches of the same steps (temporal transformer, heads): exact-f32 MFMA, peak 157"}
        if pretrain:
            c4 = (args.clip == "ViT-B/32" and args.frames == 12 and args.title_length == 45 and args.tag_length == 25)
            per_pair, per_pair_exec = (FLOP_PER_PAIR_PRETRAIN_C4 if c4 else None), None
            workload = (f"{args.clip} CHVTT-shaped pre-train step (FAM+VTM+FTM+MLM, MoCo m=0.99 K={args.negatives}), global B={args.batch} "
                        f"F={args.frames} title L={args.title_length} tag L={args.tag_length}, {res}x{res}, fwd+bwd+clip+BertAdam, random-init weights")
        else:
            # (the fp32 tower has no class-token pruning of the last block: it executes the reference's formulation)
            per_pair, per_pair_exec = fm["pair_train"], (fm["pair_train"] if fp32 else fm["pair_train_executed"])
            workload = (f"{args.clip} english fine-tune step, global B={args.batch} F={args.frames} L_text={args.length}, "
                        f"{res}x{res}, fwd+bwd+clip+BertAdam, random-init weights")
        headline = (not pretrain and not fp32 and args.batch == 256 and args.frames == 12 and args.clip == "ViT-B/32" and res == 224)
        metric = ("video-text pairs/sec (whole node), B=256 F=12 224^2" if headline else
                  f"video-text pairs/sec (whole node), {args.mode} {args.clip} B={args.batch} F={args.frames} {res}^2"
                  + (", fp32 regime (model.float())" if fp32 else ""))
        if fp32:
            workload += "; fp32 regime: model.float() + text_encoder.dtype = float32 (the reference's fp32-upcast recipe), every stage exact fp32"
        out = {"metric": metric, "value": round(value, 2),
               "unit": "video-text pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "f32" if fp32 else "f16", "data": "synthetic",
               "config": {"workload": workload, "mode": args.mode, "global_batch": args.batch, "per_gpu_batch": b, "frames": args.frames,
                          "parallelism": f"dp{world}", "streams": "single" if args.single_stream else "overlapped",
                          "gemm_reserved_cus": reserved},
               "gflop_per_pair_reference_formulation": round(per_pair / 1e9, 2) if per_pair else None,
               "step_tflops": round(value * per_pair_exec / 1e12, 1) if per_pair_exec else None,
               "step_tflops_reference_formulation": round(value * per_pair / 1e12, 1) if per_pair else None,
               "mfma_frac_whole_step": round(value * (per_pair_exec or per_pair) / 1e12 / (world * peak), 4) if per_pair else None,
               "ms_per_step_unfolded": round(ms_unfolded, 2) if ms_unfolded else None,
               "ms_per_step_unfolded_note": (f"{args.unfolded_steps} further steps of the same loop with HMMC_FOLD_LN=0 HMMC_FOLD_LN_TRAIN=0: LayerNorm "
                                             "then GEMM with the reference's rounding points everywhere; `value` / `ms_per_step` are the default "
                                             "(frame tower's ln_1 / ln_2 folded into in_proj / c_fc)") if ms_unfolded else None,
               "final_loss": round(final_loss, 4), "roofline": roof, "roofline_hbm": hbm, "comm": comm, "vit_forward": vit_forward,
               "peak_device_memory_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(12, 32)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
