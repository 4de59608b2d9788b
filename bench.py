#!/usr/bin/env python3
"""HMMC hot-path benchmark: video-text pairs / second of one fine-tuning step
(forward + backward + global grad clip + BertAdam step, reference main_task_retrieval.py:272-302)
on synthetic MSR-VTT-shaped batches [B=256, F=12, 3x224x224], ViT-B/32 + CLIP text transformer.

    python bench.py --gpus 1 --steps 10 --warmup 3
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

FLOP_PER_PAIR_TRAIN = 324.33e9      # ViT-B/32, F=12, L_text=32, the reference's formulation
# what this implementation executes: the last ViT block's out_proj + MLP run on the class token only (49 of 50 tokens pruned,
# 3 x 12 frames x (0.0590 + 0.4719) GFLOP x 49/50), see hmmc_tower_fwd's lead_only
FLOP_PER_PAIR_EXECUTED = FLOP_PER_PAIR_TRAIN - 3 * 12 * (0.0590e9 + 0.4719e9) * 49 / 50
MFMA_PEAK_TFLOPS = 2500.0           # dense fp16/bf16, MI355X_MICROARCH.md


def task_config(**kw):
    d = dict(local_rank=0, rank=0, use_temp=True, language="english", top_frames=2, max_frames=12, n_display=10 ** 9,
             logdir=None, use_frame_fea=True, dataset="msrvtt", lr=1e-4, text_lr=3e-5, coef_lr=1e-3, weight_decay=0.2,
             warmup_proportion=0.1)
    d.update(kw)
    return Namespace(**d)


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


def cpu_baseline(frames, length, seconds=20.0):
    """The oracle's fp32 restatement of the same step, timed on this host's cores (rank 0, N=1 only)."""
    from hmmc_amd import synth
    from oracle import hmmc_oracle as O
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))     # the cores this process may actually use
    torch.set_num_threads(cores)
    B = 4
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in synth.finetune_state(synth.VIT_B32).items()}
    params = [v for v in sd.values() if v.requires_grad]
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in sd.items() if v.requires_grad}
    ids, mask, vid, vf, idx = synth.finetune_batch(B, frames, length, tag="cpu_baseline")

    def step(i):
        for p in params:
            p.grad = None
        loss, _ = O.finetune_loss(ids, vid, sd, mode="fp32")
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        with torch.no_grad():
            for k, p in sd.items():
                if p.requires_grad and p.grad is not None:
                    m, v = state[k]
                    np_, m, v, _ = O.bert_adam_step(p.data, p.grad, m, v, i, 1e-4, 1000, 0.1, 0.2)
                    p.data.copy_(np_)
                    state[k] = (m, v)
    print(f"[bench] cpu_baseline: oracle step on {cores} host threads ...", file=sys.stderr, flush=True)
    step(0)                      # warm-up
    print("[bench] cpu_baseline: warm-up step done", file=sys.stderr, flush=True)
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds and n < 8):
        step(n + 1)
        n += 1
        print(f"[bench] cpu_baseline: step {n} at {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "video-text pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32 restatement, B={B} F={frames} L={length} ViT-B/32, {n} steps of {dt:.2f} s "
                      f"(fwd+bwd+clip+BertAdam), torch CPU {torch.__version__}"}


def recorded_traffic(args, per_gpu_batch):
    """HBM bytes per gemm_f16_kernel launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB units,
    separate rocprofv3 --pmc runs of this same command; profiles/r01_gemm_f16_hbm_traffic.json).  The counters cannot
    be read from inside the process, so the figure is only reported for the workload it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_gemm_f16_hbm_traffic.json")
    if not (per_gpu_batch == 256 and args.frames == 12 and args.length == 32 and args.clip == "ViT-B/32"
            and os.path.exists(path)):
        return None
    with open(path) as f:
        rec = json.load(f)
    return round(rec["hbm_traffic_per_launch_bytes"]), "profiles/r01_gemm_f16_hbm_traffic.json (rocprofv3 --pmc, separate passes)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="global batch (reference --batch_size)")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--length", type=int, default=32)
    ap.add_argument("--clip", default="ViT-B/32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="run everything on one stream (no tower / weight-gradient overlap): the mode whose rocprofv3 "
                         "per-kernel averages are comparable with roofline.avg_launch_us")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="compute units kept out of the GEMM grids for RCCL (default: 16 when --gpus > 1, else 0); "
                         "giving it with --gpus 1 measures what the reservation costs")
    ap.add_argument("--roofline-steps", type=int, default=3,
                    help="extra single-stream steps after the timed region over which the GEMM launches are timed with HIP events")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    local_dev = local_rank % max(torch.cuda.device_count(), 1)      # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HMMC_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; gloo only for rehearsals
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from hmmc_amd import ops, synth
    from hmmc_amd.modeling import BirdModel
    from hmmc_amd.optimization import clip_grad_norm_

    if args.reserve_cus is not None:
        os.environ["HMMC_RCCL_CUS"] = str(args.reserve_cus)
    reserved = ops.reserve_cus_for_collectives() if (world > 1 or args.reserve_cus is not None) else 0
    if args.single_stream:
        import hmmc_amd.functional as _fn0
        import hmmc_amd.modeling as _md0
        _md0._OVERLAP_TOWERS, _fn0._WGRAD_STREAM = False, False
    assert args.batch % world == 0
    b = args.batch // world
    cfg = task_config(local_rank=local_rank, rank=rank, max_frames=args.frames, pretrained_clip_name=args.clip)
    torch.manual_seed(42)
    model = BirdModel.from_pretrained("cross-base", state_dict=None, task_config=cfg).to(dev).train()
    optimizer = prep_optimizer(model, cfg, t_total=1000)
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_dev], output_device=local_dev,
                                                        find_unused_parameters=False, gradient_as_bucket_view=True)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    res = synth.NAMED[args.clip].image_res
    video = torch.randn((b, args.frames, 3, res, res), generator=g, device=dev)
    ids, mask = synth.token_ids(f"bench.ids.{rank}", b, args.length)
    ids, mask = ids.to(dev), mask.to(dev)
    vf = torch.full((b,), args.frames, dtype=torch.long, device=dev)
    idx = torch.arange(b, device=dev)

    def step(i):
        loss = net(ids, mask, video, vf, idx, i)
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0)
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

    # Roofline of the dominant kernel.  The timed region above runs the text tower, the frame tower and the weight
    # gradients on three streams, so a launch's HIP events there also bracket the time it waits for CUs held by another
    # stream's kernels.  The per-launch durations are therefore taken over `--roofline-steps` further steps of the same
    # loop with the overlap switched off (one stream; this is also what rocprofv3 --kernel-trace sees, it serialises
    # dispatches), events recorded on the launch stream around every hmmc_gemm_f16 call.
    import hmmc_amd.functional as _fn
    import hmmc_amd.modeling as _md
    ov = (_md._OVERLAP_TOWERS, _fn._WGRAD_STREAM)
    _md._OVERLAP_TOWERS, _fn._WGRAD_STREAM = False, False
    step(args.warmup + args.steps)                       # settle the single-stream workspaces
    torch.cuda.synchronize()
    ops.gemm_profile_start()
    t1 = time.perf_counter()
    for i in range(args.roofline_steps):
        step(args.warmup + args.steps + 1 + i)
    torch.cuda.synchronize()
    dt_single = time.perf_counter() - t1
    prof = ops.gemm_profile_stop()
    _md._OVERLAP_TOWERS, _fn._WGRAD_STREAM = ov

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = args.batch * args.steps / dt
        flops = sum(p["flops"] for p in prof.values())
        secs = sum(p["seconds"] for p in prof.values())
        launches = sum(p["launches"] for p in prof.values())
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        roof = {"bound": "mfma", "kernel": "gemm_f16_kernel (fp16 MFMA GEMM, all operand layouts)",
                "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None, "traffic_unit": "bytes/launch",
                "launches_per_step": launches // max(args.roofline_steps, 1),
                "avg_launch_us": round(secs / max(launches, 1) * 1e6, 2),
                "measured_over": f"{args.roofline_steps} single-stream steps after the timed region "
                                 f"({dt_single / max(args.roofline_steps, 1) * 1e3:.2f} ms/step without the stream overlap)",
                "gemm_share_of_step": round(secs / dt_single, 4),
                "by_layout": {k: {"tflops": round(p["flops"] / p["seconds"] / 1e12, 1), "launches": p["launches"],
                                  "avg_us": round(p["seconds"] / p["launches"] * 1e6, 2)} for k, p in prof.items()}}
        rec = recorded_traffic(args, b)
        if rec is not None:
            roof["traffic"], roof["traffic_source"] = rec
            roof["algorithmic_bytes_per_launch"] = round(sum(p.get("bytes", 0) for p in prof.values()) / max(launches, 1)) or None
        out = {"metric": "video-text pairs/sec (whole node), B=256 F=12 224^2", "value": round(value, 2),
               "unit": "video-text pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f16", "data": "synthetic",
               "config": {"workload": f"{args.clip} english MSR-VTT fine-tune step, global B={args.batch} F={args.frames} "
                                      f"L_text={args.length}, 224x224, fwd+bwd+clip+BertAdam, random-init weights",
                          "global_batch": args.batch, "per_gpu_batch": b, "frames": args.frames,
                          "parallelism": f"dp{world}", "streams": "single" if args.single_stream else "overlapped",
                          "gemm_reserved_cus": reserved},
               "step_tflops": round(value * FLOP_PER_PAIR_EXECUTED / 1e12, 1) if args.clip == "ViT-B/32" and args.frames == 12 else None,
               "step_tflops_reference_formulation": round(value * FLOP_PER_PAIR_TRAIN / 1e12, 1)
               if args.clip == "ViT-B/32" and args.frames == 12 else None,
               "mfma_frac_whole_step": round(value * FLOP_PER_PAIR_EXECUTED / 1e12 / (world * MFMA_PEAK_TFLOPS), 4)
               if args.clip == "ViT-B/32" and args.frames == 12 else None,
               "final_loss": round(final_loss, 4), "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.frames, args.length)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
