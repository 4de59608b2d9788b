"""More than one RCCL rank, one process per GPU - collected everywhere, SKIPPED below two devices.

Everything tests/test_gpu_ddp.py asserts with two ranks over gloo on one GPU is asserted here over the nccl backend (= RCCL
over xGMI) with one device per rank: fine-tune and pre-train steps equal the single-process global-batch run (loss,
gradients, queues, queue_ptr, BatchNorm statistics identical on every rank), the weights stay bit-identical on all ranks over
optimizer steps, and `_AllGatherCat` (all_gather_into_tensor forward, reduce_scatter_tensor backward) is the EXACT
concatenation / sum on the device - the 1e-6 statement of tests/test_dist_gloo.py, made exact with integer-valued data.
The workers are the gloo tests' own (`backend="nccl"`): each child is spawned fresh, joins the process group with its
device id BEFORE its first GPU call, reserves 16 CUs for RCCL (hmmc_gemm_reserve_cus) and runs with the stream overlap on.
`bench.py --gpus 2` is run the way the driver runs it.

Reference: modules/modeling.py:25-36 (dist_collect), :249-258 (key gather before the enqueue), :698-700 (the three feature
gathers), main_task_retrieval.py:28,207-208 and main_pretrain.py:204-205 (init_process_group("nccl"), DistributedDataParallel).
"""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NDEV = torch.cuda.device_count()            # counting devices does not initialise the GPU
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NDEV < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")]

import test_gpu_ddp as D  # noqa: E402


def _gather_worker(rank, world, store, out_dir, backend="nccl"):
    """backend "gloo": the same statements on CPU tensors (tests/test_dist_gloo.py runs that here, so the expected values of the
    RCCL test are themselves tested where no second GPU exists)."""
    if backend == "nccl":
        D._init(rank, world, store, "nccl")
    else:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    dev = "cuda" if backend == "nccl" else "cpu"
    import hmmc_amd.functional as Fn
    import hmmc_amd.modeling as M
    assert Fn.collectives_active() and M._AllGatherCat._flat() == (backend == "nccl")
    b, C = 6, 14 * 512                                            # the packed feature row of F = 12: (F + 2) * 512
    g = torch.Generator().manual_seed(77)
    xs = [torch.randint(-64, 64, (b, C), generator=g).float() for _ in range(world)]        # integer data: fp32 sums are exact
    ws = [torch.randint(-8, 8, (world * b, C), generator=g).float() for _ in range(world)]
    x = xs[rank].to(dev).requires_grad_()
    y = M._AllGatherCat.apply(x * 1.0)
    (y * ws[rank].to(dev)).sum().backward()
    if dev == "cuda":
        torch.cuda.synchronize()
    want_y = torch.cat(xs, 0)
    want_g = sum(w[rank * b:(rank + 1) * b] for w in ws)          # d/dx_r of sum_r' <gather(x), w_r'>: every rank's weights on my rows
    res = {"fwd": bool(torch.equal(y.detach().cpu(), want_y)), "bwd": bool(torch.equal(x.grad.cpu(), want_g))}
    yc = M.dist_collect(xs[rank].to(dev))
    res["collect"] = bool(torch.equal(yc.cpu(), want_y))
    t = torch.arange(2 * 4096 + 1, dtype=torch.float32) * (rank + 1)
    res["sync_sum"] = bool(torch.equal(Fn._sync_sum(t.to(dev)).cpu(), torch.arange(2 * 4096 + 1, dtype=torch.float32) * (world * (world + 1) // 2)))
    torch.save(res, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", sorted({2, min(NDEV, 8)}) if NDEV >= 2 else [2])
def test_allgathercat_is_exact_on_rccl(world):
    """`_AllGatherCat` on RCCL: forward = the ranks' rows in rank order, backward = the exact sum over ranks of the gradient
    rows that belong to this rank (reduce_scatter_tensor); `dist_collect` and SyncBatchNorm's `_sync_sum` likewise."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_gather_worker, args=(world, os.path.join(d, "s"), d), nprocs=world, join=True)
        for r in range(world):
            res = torch.load(os.path.join(d, f"g{r}.pt"))
            assert all(res.values()), (r, res)


def test_finetune_two_rccl_ranks_equal_single_process():
    """BirdModel.forward under DDP, two RCCL ranks: global loss and averaged gradients equal the single-process run of the
    global batch within the measured fp16 bounds of tests/test_gpu_ddp.py, identical on both ranks."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(D._run, args=(1, os.path.join(d, "s1"), d, "nccl"), nprocs=1, join=True)
        mp.spawn(D._run, args=(2, os.path.join(d, "s2"), d, "nccl"), nprocs=2, join=True)
        ref = torch.load(os.path.join(d, "w1r0.pt"))
        outs = [torch.load(os.path.join(d, f"w2r{r}.pt")) for r in range(2)]
    D.check_finetune(ref, outs)


def test_pretrain_two_rccl_ranks_equal_single_process():
    """BirdPreTrainedModel.forward under DDP, two RCCL ranks: the packed key gather leaves the five queues and queue_ptr
    identical on both ranks and equal to the single-process run, BatchNorm statistics are those of all ranks' rows."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(D._run_pretrain, args=(1, os.path.join(d, "s1"), d, "nccl"), nprocs=1, join=True)
        mp.spawn(D._run_pretrain, args=(2, os.path.join(d, "s2"), d, "nccl"), nprocs=2, join=True)
        ref = torch.load(os.path.join(d, "p1r0.pt"))
        outs = [torch.load(os.path.join(d, f"p2r{r}.pt")) for r in range(2)]
    D.check_pretrain(ref, outs)


def test_rccl_ranks_stay_identical_over_optimizer_steps():
    """Four optimizer steps under DDP over RCCL (gradients as bucket views, 16 reserved CUs, stream overlap): every rank holds
    bit-identical weights afterwards and saw the same global losses."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(D._run_steps, args=(world, os.path.join(d, "s"), d, "nccl"), nprocs=world, join=True)
        a, b = [torch.load(os.path.join(d, f"s{r}.pt")) for r in range(2)]
    assert a["losses"] == b["losses"], (a["losses"], b["losses"])
    bad = [i for i, (x, y) in enumerate(zip(a["weights"], b["weights"])) if not torch.equal(x, y)]
    assert not bad, f"{len(bad)} parameter tensors differ between the ranks after 4 steps"


def test_bench_two_rccl_ranks():
    """`python bench.py --gpus 2` as the driver starts it (no launcher): two ranks on RCCL, ONE JSON line, the `comm` record."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "HMMC_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "8", "--frames", "4", "--steps", "2",
                        "--warmup", "1", "--roofline-steps", "1", "--vit-forward-iters", "1", "--no-cpu-baseline"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["gemm_reserved_cus"] == 16
    assert out["final_loss"] == out["final_loss"] and abs(out["final_loss"]) < 50
    assert out["comm"]["backend"] == "nccl" and out["comm"]["allreduce_ms"] > 0
