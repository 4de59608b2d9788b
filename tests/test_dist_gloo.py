"""world_size-2 gloo tests (CPU) of the multi-rank plumbing of the hot path: the packed differentiable
all-gather (reference dist_collect / diffdist, modules/modeling.py:25-36,698-700), the effective-gradient rule of
SURVEY.md section 8(e) (every rank evaluates the global loss; gather-backward sums over ranks; DDP averages) and the
BatchNorm statistics exchange."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hmmc_amd import synth
from oracle import hmmc_oracle as O


def _worker(rank, world, store, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    from hmmc_amd.modeling import dist_collect
    from hmmc_amd.functional import _sync_sum
    B, F, E = 4, 3, 512
    b = B // world
    q = synth.normal("ddp.q", (B, E))[rank * b:(rank + 1) * b].clone().requires_grad_()
    v = synth.normal("ddp.v", (B, E))[rank * b:(rank + 1) * b].clone().requires_grad_()
    u = synth.normal("ddp.u", (B, F, E))[rank * b:(rank + 1) * b].clone().requires_grad_()
    packed = dist_collect(torch.cat([v, q, u.reshape(b, F * E)], dim=1))
    assert packed.shape == (B, (F + 2) * E)
    vg, qg, ug = packed[:, :E], packed[:, E:2 * E], packed[:, 2 * E:].reshape(B, F, E)
    loss = O.finetune_head(qg, vg, ug)
    loss.backward()
    # DDP would now average parameter gradients over ranks; features are per-rank, so emulate with 1/world
    torch.save({"loss": loss.detach(), "dq": q.grad / world, "dv": v.grad / world, "du": u.grad / world,
                "gathered_q": qg.detach()}, os.path.join(out_dir, f"r{rank}.pt"))
    s = _sync_sum(torch.tensor([1.0 + rank, 10.0]))
    assert torch.equal(s, torch.tensor([3.0, 20.0]))
    dist.destroy_process_group()


def test_packed_allgather_and_effective_gradient():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        store = os.path.join(d, "store")
        mp.spawn(_worker, args=(world, store, d), nprocs=world, join=True)
        outs = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    B, F, E = 4, 3, 512
    q = synth.normal("ddp.q", (B, E)).requires_grad_()
    v = synth.normal("ddp.v", (B, E)).requires_grad_()
    u = synth.normal("ddp.u", (B, F, E)).requires_grad_()
    loss = O.finetune_head(q, v, u)
    loss.backward()
    for r, o in enumerate(outs):
        assert torch.allclose(o["loss"], loss.detach(), atol=1e-6)          # every rank evaluates the global loss
        assert torch.equal(o["gathered_q"], q.detach())                      # rank order
        sl = slice(r * 2, r * 2 + 2)
        # gather-backward sums W identical contributions, the DDP average divides by W: exactly dL/dx of the global loss
        assert torch.allclose(o["dq"], q.grad[sl], atol=1e-6)
        assert torch.allclose(o["dv"], v.grad[sl], atol=1e-6)
        assert torch.allclose(o["du"], u.grad[sl], atol=1e-6)


@pytest.mark.parametrize("world", [2, 3])
def test_allgathercat_exact_on_integer_data(world):
    """The worker of tests/test_gpu_rccl_multi.py::test_allgathercat_is_exact_on_rccl on the gloo backend with CPU tensors: forward =
    the ranks' rows in rank order, backward = the exact sum over ranks of the gradient rows that belong to this rank (integer-valued
    data: no rounding), `dist_collect` and the SyncBatchNorm sum likewise.  Keeps the RCCL test's expected values honest on a
    box without a second GPU."""
    import test_gpu_rccl_multi as R
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(R._gather_worker, args=(world, os.path.join(d, "s"), d, "gloo"), nprocs=world, join=True)
        for r in range(world):
            res = torch.load(os.path.join(d, f"g{r}.pt"))
            assert all(res.values()), (r, res)
