"""GPU tests of the eval leg next to the training hot path: the fused scorer
(main_task_retrieval.py:321-357 _run_on_single_gpu: loose_similarity x 2, top-k over frames, mean), its batched driver and
the multi-sentence rank metrics (metrics.py:49-144), against the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden  # noqa: E402
from hmmc_amd import metrics as M  # noqa: E402
from hmmc_amd import ops, synth  # noqa: E402
from oracle import hmmc_oracle as O  # noqa: E402
from test_gpu_model import close, task_config  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def model():
    from hmmc_amd.modeling import BirdModel
    return BirdModel.from_pretrained("cross-base", state_dict=synth.finetune_state(synth.TINY), task_config=task_config()).to(DEV).eval()


def _reference_scores(q, v, u, k):
    """the reference's arithmetic (modules/modeling.py:207-229 + main_task_retrieval.py:332-336) in the oracle's plain torch"""
    sv = O.loose_similarity(q, v)
    sf = O.loose_similarity(q, u)
    return sv, torch.topk(sf, k=k, dim=2)[0].mean(dim=2)


@pytest.mark.parametrize("nq,nv,F,k", [(200, 70, 24, 3), (65, 5, 12, 2), (33, 129, 31, 31), (7, 3, 1, 1), (130, 40, 15, 4)])
def test_fused_scorer_vs_oracle_and_unfused(model, nq, nv, F, k):
    """both slot widths (F + 1 <= 16 and <= 32), ragged tile edges, exact ties between frames (torch.topk counts equal
    values separately), k = F"""
    q = synth.normal(f"ev.q{nq}", (nq, 512))
    v = synth.normal(f"ev.v{nv}", (nv, 512))
    u = synth.normal(f"ev.u{nv}.{F}", (nv, F, 512))
    if F >= 4:
        u[1, 2] = u[1, 0]                     # two identical frames: the same logit twice
        u[nv - 1, F - 1] = u[nv - 1, F - 2]
    rv, rf = _reference_scores(q, v, u, k)
    qg, vg, ug = q.to(DEV), v.to(DEV), u.to(DEV)
    with torch.no_grad():
        sv, sf = model.eval_scores(qg, vg, ug, top_frames=k)
        close(sv, rv, 1e-3, what="video logits")
        close(sf, rf, 1e-3, what="top-k frame mean")
        # the two-launch path (hmmc_gemm_f32 + hmmc_topk_mean on the materialised [nq, nv, F] tensor): same arithmetic
        s3 = model.loose_similarity(qg, ug)
        uf = ops.topk_mean(s3.view(nq, nv * F), nq, nv, F, k)
        close(sf, uf, 2e-5, what="fused vs unfused")
        close(sv, model.loose_similarity(qg, vg), 2e-5, what="fused vs unfused video")


def test_fused_scorer_refuses_what_it_cannot_take(model):
    q, v, u = torch.randn(4, 512, device=DEV), torch.randn(3, 512, device=DEV), torch.randn(3, 40, 512, device=DEV)
    assert ops.eval_slots(40) == 0 and ops.eval_slots(31) == 32 and ops.eval_slots(15) == 16
    with torch.no_grad():
        sv, sf = model.eval_scores(q, v, u, top_frames=3)           # 40 frames: the two-launch path, same contract
    rv, rf = _reference_scores(q.cpu(), v.cpu(), u.cpu(), 3)
    close(sv, rv, 1e-3)
    close(sf, rf, 1e-3)
    packed = ops.eval_pack(v, torch.randn(3, 12, 512, device=DEV))
    qn, _ = ops.l2norm_fwd(q)
    with pytest.raises(RuntimeError, match="invalid argument"):      # k > F
        ops.eval_score(qn, packed, 3, 12, 13, 100.0)


def test_eval_similarity_driver_vs_reference_golden(model):
    """the cached-feature loop of eval_epoch: uneven batches of queries and videos in, one score matrix out; identical
    retrieval ranks and R@K to the reference's golden (tests/golden/head_eval.npz), whatever the query chunking"""
    g = golden("head_eval")
    q = synth.normal("head_eval.q", (48, 512))
    v = synth.normal("head_eval.v", (48, 512))
    u = synth.normal("head_eval.u", (48, 12, 512))
    q = (q + 0.7 * v).to(DEV)
    v, u = v.to(DEV), u.to(DEV)
    cut_q, cut_v = [0, 16, 17, 40, 48], [0, 5, 37, 48]
    ql = [q[a:b] for a, b in zip(cut_q[:-1], cut_q[1:])]
    vl = [v[a:b] for a, b in zip(cut_v[:-1], cut_v[1:])]
    ul = [u[a:b] for a, b in zip(cut_v[:-1], cut_v[1:])]
    for k in (1, 2, 3, 12):
        model.top_frames = k
        ref = g["S_video"] + g[f"topk{k}"]
        sim = model.eval_similarity(ql, vl, ul, use_frame_fea=True)
        close(sim, ref, 2e-3, what=f"score k={k}")
        assert np.array_equal(np.argsort(-sim.cpu().numpy(), 1), np.argsort(-ref, 1))
        assert torch.equal(sim, model.eval_similarity(ql, vl, ul, use_frame_fea=True, query_chunk=7))
        tv, vt = M.logging_rank(sim, False, [])
        close([tv["R1"], tv["R5"], tv["R10"], tv["MR"], tv["MeanR"]], g[f"metrics{k}"], 1e-9, what="metrics")
    sim_v = model.eval_similarity(ql, vl, ul, use_frame_fea=False)
    close(sim_v, g["S_video"], 1e-3)
    tv, vt = M.logging_rank(sim_v, False, [])
    close([tv["R1"], tv["R5"], tv["R10"], tv["MR"], tv["MeanR"]], g["metrics_video"], 1e-9)
    close([vt["R1"], vt["R5"], vt["R10"], vt["MR"], vt["MeanR"]], g["metrics_video_v2t"], 1e-9)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_multi_sentence_metrics_vs_reference_golden(tag):
    """metrics.py:49-144 with uneven captions per video (1..20): ranks on the device, no -inf padded
    [videos, max_captions, videos] tensor"""
    g = golden("multisent")
    sim = torch.from_numpy(g[f"{tag}.sim"]).to(DEV)
    cut = [int(c) for c in g[f"{tag}.cut"]]
    tv, vt = M.logging_rank(sim, True, cut)
    close([tv["R1"], tv["R5"], tv["R10"], tv["MedianR"], tv["MeanR"], tv["Std_Rank"]], g[f"{tag}.tv"], 1e-5, 1e-6, "text->video")
    close([vt["R1"], vt["R5"], vt["R10"], vt["MR"], vt["MeanR"]], g[f"{tag}.vt"], 1e-9, what="video->text")
    with pytest.raises(ValueError):
        M.logging_rank(sim, True, cut[:-1])


def test_multi_sentence_metrics_with_exact_ties():
    """Equal logits: the reference ranks by a double torch.argsort, which leaves the ground truth at an
    implementation-defined position among its equals; the device kernel counts strictly greater candidates (the first of
    the tied positions, as compute_metrics' `sx - d == 0` does for single-sentence retrieval).  The reference's numbers must
    lie between ours and ours with every tie resolved against the ground truth."""
    g = golden("multisent")
    sim = torch.from_numpy(g["t.sim"]).to(DEV)
    cut = [int(c) for c in g["t.cut"]]
    tv, _ = M.logging_rank(sim, True, cut)
    _, vid = M._groups(cut, sim.shape[0])
    s = g["t.sim"]
    gt = s[np.arange(len(vid)), vid]
    worst = (s >= gt[:, None]).sum(1) - 1
    ref = g["t.tv"]
    assert tv["MeanR"] <= ref[4] + 1e-6 <= float(np.mean(worst + 1)) + 2e-6
    assert tv["R1"] >= ref[0] - 1e-4 and tv["R10"] >= ref[2] - 1e-4


def test_scorer_at_vatex_size_properties(model):
    """15 000 captions x 1 500 videos x 24 frames (SURVEY C5's test set): no oracle at this size; size-independent
    properties instead - rows equal the two-launch path on a sample, a duplicated caption scores identically, permuting
    the videos permutes the columns, and the multi-sentence metrics of a planted solution are perfect."""
    nq, nv, F, k = 15000, 1500, 24, 3
    gen = torch.Generator(device=DEV).manual_seed(5)
    v = torch.randn(nv, 512, device=DEV, generator=gen)
    u = torch.randn(nv, F, 512, device=DEV, generator=gen) + v[:, None, :]
    vid = torch.arange(nq, device=DEV) // 10
    q = v[vid] + 0.8 * torch.randn(nq, 512, device=DEV, generator=gen)
    q[7] = q[3]
    model.top_frames = k
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim = model.eval_similarity([q], [v], [u])
    e0.record()
    sim = model.eval_similarity([q], [v], [u])
    e1.record()
    torch.cuda.synchronize()
    print(f"eval_similarity 15000 x 1500 x 24: {e0.elapsed_time(e1):.2f} ms")
    assert sim.shape == (nq, nv) and bool(torch.isfinite(sim).all())
    assert torch.equal(sim[7], sim[3])
    rows = torch.tensor([0, 3, 4999, 9999, 14999], device=DEV)
    with torch.no_grad():
        s3 = model.loose_similarity(q[rows], u)
        ref = model.loose_similarity(q[rows], v) + ops.topk_mean(s3.view(5, nv * F), 5, nv, F, k)
    close(sim[rows], ref, 2e-4, what="sampled rows")      # logits up to ~70: a few fp32 ulps between the two summation orders
    perm = torch.randperm(nv, device=DEV, generator=gen)
    sim_p = model.eval_similarity([q[:256]], [v[perm]], [u[perm]])
    assert torch.equal(sim_p, sim[:256][:, perm])
    cut = list(range(9, nq, 10))
    tv, vt = M.logging_rank(sim, True, cut)
    assert tv["R1"] > 99.0 and vt["R1"] > 99.0 and tv["MedianR"] == 1.0
