"""The policy pass of a plan step through the exactly pruned decoder (M3PC_PLAN_PRUNED_POLICY): the
reference reads the policy distribution at the last h steps only (learner.py:285-287: dist.sample((N,))[:, 0, T-h:]), and under the
rcbc mask those action tokens are masked -- shared query rows and masked-token K|V from the plan tables, the kept tokens alone through
decoder-embed / K|V, out-proj / FFN / actor head on h rows instead of 4T.  The head at those rows equals the full pass's (fp32
re-association apart) and the reference's goldens; rows below T - h are zero."""
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner

pytestmark = pytest.mark.gpu
GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cfg(T, N, H, tau=0.01, guidance="rtg_guiding"):
    return types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=tau, lmbda=0.6,
                                 plan_guidance=guidance, device="cuda")


@pytest.mark.parametrize("env,guidance,T,H,pl", [("hopper", "rtg_guiding", 32, 16, 500), ("hopper", "rtg_guiding", 32, 16, 7),
                                                 ("walker2d", "critic_lambda_guiding", 32, 16, 500), ("halfcheetah", "rtg_guiding", 64, 32, 500),
                                                 ("hopper", "rtg_guiding", 8, 4, 500), ("hopper", "noise_adding_lambda", 16, 8, 300),
                                                 ("hopper", "rtg_guiding", 16, 16, 500)])
def test_pruned_policy_head_equals_the_full_pass(env, guidance, T, H, pl):
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    qsd, om, os_ = synth.make_critic(dims, 0)
    tau = 0.01 if guidance == "rtg_guiding" else 1.0
    mk = lambda head: HipPlanner(_cfg(T, 64, H, tau, guidance), synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_,
                                 precision="fp32", policy_head=head, generator=torch.Generator(device="cuda").manual_seed(3))
    pp, pf = mk("pruned"), mk("full")
    hist = synth.make_history(dims, 1)
    hist["path_length"] = pl
    ev_p = pp.action_sample(hist, plan=True, eval=True, rtg=3.0)
    ev_f = pf.action_sample(hist, plan=True, eval=True, rtg=3.0)
    h = pp.last["horizon"]
    idx = T - h
    loc_p, loc_f, sd_p, sd_f = pp.last["loc"], pf.last["loc"], pp.last["std"], pf.last["std"]
    assert float((loc_p[idx:] - loc_f[idx:]).abs().max()) <= 2e-5 * max(1.0, float(loc_f.abs().max()))
    assert float((sd_p[idx:] / sd_f[idx:] - 1).abs().max()) <= 2e-5
    if idx > 0:
        assert float(loc_p[:idx].abs().max()) == 0.0 and float(sd_p[:idx].abs().max()) == 0.0  # (not computed: zero, said in the header)
    # the candidates, hence the whole step, follow
    assert float((pp.last["sample_actions"] - pf.last["sample_actions"]).abs().max()) <= 2e-5
    scale = float(pf.last["expect_return"].abs().max())
    assert float((pp.last["expect_return"] - pf.last["expect_return"]).abs().max()) <= 5e-5 * max(scale, 1.0)
    assert float((ev_p - ev_f).abs().max()) <= 1e-4
    pp.handle.close()
    pf.handle.close()


def test_pruned_policy_against_the_reference_golden_c2():
    """BASELINE config 2: loc / std at t >= T - h and the sampled candidates against the reference's stored outputs."""
    g = np.load(os.path.join(GD, "g2_c2.npz"))
    dims = synth.Dims(11, 3, 32)
    p = HipPlanner(_cfg(32, 1024, 16), synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="fp32",
                   policy_head="pruned")
    eps = synth.make_eps(1024, dims, 1).cuda()
    p._eps = lambda shape: eps
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    p.action_sample(hist, plan=True, eval=True, rtg=3.0)
    loc, std = g["loc"].reshape(32, 3), g["std"].reshape(32, 3)
    assert np.abs(p.last["loc"].cpu().numpy()[16:] - loc[16:]).max() <= 2e-5 * max(1.0, float(np.abs(loc).max()))
    assert np.abs(p.last["std"].cpu().numpy()[16:] / std[16:] - 1).max() <= 2e-5
    assert int(p.last["argmax"].item()) == int(g["argmax"])
    rows = g["rows"].astype(np.int64)
    assert np.abs(p.last["sample_actions"].cpu().numpy()[rows] - g["sample_actions_rows"]).max() <= 2e-5
    p.handle.close()
