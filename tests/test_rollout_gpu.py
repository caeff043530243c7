"""m3pc_amd.rollout.evaluate_plan (the counterpart of Learner.evaluate_plan, learner.py:645-741, for several environments with
their plan steps pipelined on the device) against the same loop with one step in flight: same round-robin order of windows,
same draws, hence the same actions, rewards and episode statistics -- bit for bit."""
import types

import numpy as np
import pytest
import torch

from m3pc_amd import synth
from m3pc_amd.planner import HipPlanner
from m3pc_amd.rollout import evaluate_plan

pytestmark = pytest.mark.gpu


def _planner(dims, N, H, precision, seed):
    cfg = types.SimpleNamespace(traj_length=dims.traj_length, action_samples=N, horizon=H, discount=0.99, temperature=0.01,
                                lmbda=0.6, plan_guidance="rtg_guiding", device="cuda")
    return HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision=precision,
                      generator=torch.Generator(device="cuda").manual_seed(seed), pipeline_depth=3)


@pytest.mark.parametrize("precision,eval_", [("bf16", True), ("bf16", False), ("fp32", True)])
def test_pipelined_rollout_equals_one_step_in_flight(precision, eval_):
    from fake_learner import ToyEnv

    dims = synth.Dims(11, 3, 16)
    rtg_ref = np.linspace(3.0, 1.0, 1000)
    lengths = [9, 14, 11, 14]   # episodes end at different times: the round-robin order must not depend on what is in flight

    def run(in_flight):
        p = _planner(dims, 256, 8, precision, seed=21)
        envs = [ToyEnv(11, 3, i, length=lengths[i]) for i in range(4)]
        out = evaluate_plan(p, envs, rtg_ref, max_steps=14, in_flight=in_flight, eval=eval_)
        torch.cuda.synchronize()
        p.handle.close()
        return out

    a, b = run(1), run(3)
    assert a["plan_steps"] == b["plan_steps"] == sum(lengths)
    assert a["lengths"] == b["lengths"] == [float(v) for v in lengths]
    for ta, tb in zip(a["trajectories"], b["trajectories"]):
        assert np.array_equal(ta["actions"], tb["actions"]) and np.array_equal(ta["rewards"], tb["rewards"])
        assert np.array_equal(ta["observations"], tb["observations"])
    assert a["return_mean"] == b["return_mean"] and np.isfinite(a["return_mean"])
    assert float(np.abs(a["trajectories"][0]["actions"][: lengths[0]]).max()) <= 1.0


def test_rollout_matches_the_reference_shaped_serial_loop():
    """One environment, one step in flight == the reference's loop body spelled out (learner.py:675-697) on action_sample."""
    from fake_learner import ToyEnv

    dims = synth.Dims(11, 3, 16)
    rtg_ref = np.linspace(3.0, 1.0, 1000)
    p = _planner(dims, 128, 8, "fp32", seed=5)
    out = evaluate_plan(p, [ToyEnv(11, 3, 3, length=10)], rtg_ref, max_steps=10)
    p.handle.close()
    q = _planner(dims, 128, 8, "fp32", seed=5)
    env = ToyEnv(11, 3, 3, length=10)
    traj = {"observations": np.zeros((1000, 11), dtype=np.float32), "actions": np.zeros((1000, 3), dtype=np.float32),
            "rewards": np.zeros((1000, 1), dtype=np.float32), "values": np.zeros((1000, 1), dtype=np.float32), "total_return": 0,
            "path_length": 0}
    obs, done, t = env.reset(), False, 0
    while not done and t < 1000:
        traj["observations"][t] = obs
        action = q.action_sample(traj, percentage=1.0, plan=True, eval=True, rtg=rtg_ref[t] * 1.0)
        action = np.clip(action.cpu().numpy(), -1, 1)
        obs, reward, done, info = env.step(action)
        traj["actions"][t] = action
        traj["rewards"][t] = reward
        t += 1
        traj["path_length"] += 1
    q.handle.close()
    assert np.array_equal(out["trajectories"][0]["actions"][:10], traj["actions"][:10]) and t == 10
    assert out["returns"][0] == float(traj["rewards"].sum())


def test_lockstep_rollout_steps_all_environments_together():
    """evaluate_plan(lockstep=True): one batched plan call per round over the live environments; same episode lengths and step
    count as the pipelined loop, every action a valid plan (inside the box, finite returns), fp32 mode close to the per-window
    planner on the first round (same windows; the policy head of a batch agrees to fp32 rounding)."""
    from fake_learner import ToyEnv

    dims = synth.Dims(11, 3, 16)
    rtg_ref = np.linspace(3.0, 1.0, 1000)
    lengths = [6, 9, 7]
    cfg = types.SimpleNamespace(traj_length=16, action_samples=256, horizon=8, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    for precision in ("bf16", "fp32"):
        p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision=precision,
                       generator=torch.Generator(device="cuda").manual_seed(3), max_batch=3, max_windows=3)
        envs = [ToyEnv(11, 3, i, length=lengths[i]) for i in range(3)]
        rounds = []
        out = evaluate_plan(p, envs, rtg_ref, max_steps=9, lockstep=True, on_step=lambda i, t, a, r, d: rounds.append((i, t)))
        p.handle.close()
        assert out["plan_steps"] == sum(lengths) and out["lengths"] == [float(v) for v in lengths]
        assert rounds[:3] == [(0, 1), (1, 1), (2, 1)]  # round-robin inside a round, all environments at the same timestep
        assert np.isfinite(out["return_mean"])
        for tr, n in zip(out["trajectories"], lengths):
            assert float(np.abs(tr["actions"][:n]).max()) <= 1.0 and np.abs(tr["actions"][:n]).sum() > 0
