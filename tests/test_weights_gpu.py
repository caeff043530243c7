"""Incremental weight re-pack (VERDICT r2 missing 2): fine-tuning updates the model between rollouts
(finetune.py:306, learner.py:506-538); ``attach`` follows the parameters per tensor and ``m3pc_load_weights`` re-derives only
what depends on the tensors that changed -- their bf16 copies, the fragment stream of the layer they belong to, the embedding
tables, the cached decoder tables."""
import time
import types

import pytest
import torch

from m3pc_amd import synth
from m3pc_amd.planner import HipPlanner, attach

pytestmark = pytest.mark.gpu


def _setup():
    from fake_learner import make_learner
    dims = synth.Dims(11, 3, 32)
    cfg = types.SimpleNamespace(traj_length=32, action_samples=512, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    learner = make_learner(dims, cfg, with_critic=False)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    eps = synth.make_eps(512, dims, 4).cuda()
    return dims, cfg, learner, hist, eps


def _fresh_scores(dims, cfg, sd, hist, eps):
    p = HipPlanner(cfg, {k: v.detach().clone() for k, v in sd.items()}, synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                   rescore="topk", rescore_topk=8)
    p._eps = lambda shape: eps
    p.action_sample(hist, plan=True, eval=True, rtg=3.0)
    out = p.last["expect_return"].clone(), p.last["expect_return_bf16"].clone()
    p.handle.close()
    return out


def test_one_changed_tensor_repacks_only_its_layer():
    dims, cfg, learner, hist, eps = _setup()
    planner = attach(learner, precision="bf16", rescore="topk", rescore_topk=8)
    planner._eps = lambda shape: eps
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    er0 = planner.last["expect_return_bf16"].clone()
    assert planner.handle.load_stats()["tensors"] == len(learner.mtm.names)  # the first load brought everything
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        w = learner.mtm.param("encoder.layers.1.linear1.weight")
        w.add_(0.02 * torch.randn(w.shape, device="cuda", generator=g))
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    assert planner.last_sync == ["encoder.layers.1.linear1.weight"]
    st = planner.handle.load_stats()
    assert st == dict(tensors=1, tail_streams=1, kv_streams=0, tables_invalidated=False), st
    er1, er1b = planner.last["expect_return"].clone(), planner.last["expect_return_bf16"].clone()
    assert float((er1b - er0).abs().max()) > 1e-3, "the updated weight was not picked up"
    ref, refb = _fresh_scores(dims, cfg, learner.mtm.state_dict(), hist, eps)
    assert torch.equal(er1b, refb) and torch.equal(er1, ref)  # == a planner built from the whole new state_dict
    # nothing changed -> nothing sent
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    assert planner.last_sync == []
    # the sync of one tensor alone (version scan + upload + re-pack of one 4.5 MiB stream) stays within a few milliseconds
    ts = []
    for _ in range(5):
        with torch.no_grad():
            learner.mtm.param("encoder.layers.0.linear2.weight").mul_(1.0001)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        planner.sync()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print("one-tensor sync ms:", [round(t, 3) for t in ts])
    assert sorted(ts)[len(ts) // 2] <= 3.0, ts
    planner.handle.close()


def test_decoder_side_tensors_invalidate_the_cached_tables():
    dims, cfg, learner, hist, eps = _setup()
    planner = attach(learner, precision="bf16", rescore="topk", rescore_topk=8)
    planner._eps = lambda shape: eps
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    with torch.no_grad():
        learner.mtm.param("decoder.layers.0.self_attn.in_proj_weight").mul_(1.01)
        learner.mtm.param("mask_token_dict.rewards").add_(0.05)
        learner.mtm.param("encoder_embed_dict.states.bias").add_(0.01)
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    st = planner.handle.load_stats()
    assert sorted(planner.last_sync) == sorted(["decoder.layers.0.self_attn.in_proj_weight", "mask_token_dict.rewards",
                                                "encoder_embed_dict.states.bias"])
    assert st["tensors"] == 3 and st["tail_streams"] == 0 and st["kv_streams"] == 4 and st["tables_invalidated"]
    ref, refb = _fresh_scores(dims, cfg, learner.mtm.state_dict(), hist, eps)
    assert torch.equal(planner.last["expect_return_bf16"], refb) and torch.equal(planner.last["expect_return"], ref)
    planner.handle.close()


def test_a_full_optimizer_step_resends_everything_and_matches():
    dims, cfg, learner, hist, eps = _setup()
    planner = attach(learner, precision="bf16", rescore="topk", rescore_topk=8)
    planner._eps = lambda shape: eps
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    with torch.no_grad():
        for p in learner.mtm.ps:
            if p.dim() >= 1 and p.numel() > 1:
                p.mul_(0.999)
    t0 = time.perf_counter()
    planner.sync()
    torch.cuda.synchronize()
    full_ms = 1e3 * (time.perf_counter() - t0)
    print("full re-sync ms:", round(full_ms, 2))
    assert len(planner.last_sync) >= len(learner.mtm.names) - 8
    learner.action_sample(hist, plan=True, eval=True, rtg=3.0)
    ref, refb = _fresh_scores(dims, cfg, learner.mtm.state_dict(), hist, eps)
    assert torch.equal(planner.last["expect_return_bf16"], refb)
    planner.handle.close()
