"""Rollout-level caller of the pipelined planner: the counterpart of ``Learner.evaluate_plan``
(research/finetune_omtm/learner.py:645-741: ``num_episodes`` independent evaluation episodes, one after the other, one
``action_sample`` per environment step with the action read back each time) and of ``ReplayBuffer.online_rollout``'s env
loop (replay_buffer.py:204-232) for SEVERAL environments at once.

The reference's loops are strictly sequential because one environment's next window needs its last action.  Across
environments nothing couples the plan steps, so ``evaluate_plan`` here steps E environments round-robin and keeps their
plan steps in flight on the device (``HipPlanner.plan_async``): while environment i's candidate pass runs, environment
i+1's policy pass and environment i-1's re-score + select run on the planner's other streams, and the host steps the
simulator of the environment whose action has just arrived.  Per environment, the sequence of windows, draws and actions
is what the reference's loop produces for that environment with the same generator order (tests/test_rollout_gpu.py).

The simulator is the caller's: any object with ``reset() -> obs`` and ``step(action) -> (obs, reward, done, info)``
(gym's API, which is what jaxrl's env wrappers hand to the reference; gym / d4rl / mujoco are not needed here).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np


def new_trajectory(state_dim: int, action_dim: int, max_steps: int = 1000) -> Dict[str, Any]:
    """The episode buffer of learner.py:660-674 / replay_buffer.py:189-203."""
    return {"observations": np.zeros((max_steps, state_dim), dtype=np.float32),
            "actions": np.zeros((max_steps, action_dim), dtype=np.float32),
            "rewards": np.zeros((max_steps, 1), dtype=np.float32), "values": np.zeros((max_steps, 1), dtype=np.float32),
            "total_return": 0, "path_length": 0}


class _Episode:
    def __init__(self, env, index, traj):
        self.env, self.index, self.traj = env, index, traj
        self.observation = None
        self.timestep = 0
        self.done = False
        self.info: Dict[str, Any] = {}
        self.ticket = None


def evaluate_plan(planner, envs: Sequence, episode_rtg_ref: np.ndarray, ratio: float = 1.0, max_steps: int = 1000,
                  in_flight: Optional[int] = None, on_step: Optional[Callable] = None, eval: bool = True,
                  percentage: float = 1.0, lockstep: bool = False) -> Dict[str, Any]:
    """One evaluation episode in each of ``envs`` (learner.py:645-741 with num_episodes = len(envs)), planned through the
    pipelined planner.  ``episode_rtg_ref[t] * ratio`` is the return-to-go handed to the planner at timestep t (learner.py:
    688); with ``eval=False`` the planner explores as ``online_rollout`` does (sampled action, return-to-go from the
    ``percentage`` of the returns range, replay_buffer.py:206-212).
    in_flight: how many environments have a plan step on the device at the same time (default: planner.pipeline_depth + 1,
    at most the number of slots - 1).
    lockstep: the environments step TOGETHER instead -- per round one ``action_sample_batch(lockstep=True)`` over the windows
    of all live environments (one policy pass at batch E, the candidate passes back to back, one batched fp32 re-score: the
    short launches of the fp32 chains are paid once per round, not once per environment; needs a planner built with
    ``max_batch >= len(envs)``; per environment the actions agree with the pipelined loop to fp32 rounding of the policy
    head, not bit for bit, and the draws are taken per round).  Returns {"return_mean", "return_std", "length_mean", "length_std", "returns",
    "lengths", "trajectories", "plan_steps"} -- the statistics evaluate_plan logs (learner.py:715-731)."""
    from . import capi

    E = len(envs)
    S, A = planner.S, planner.A
    cap = max(1, min(in_flight if in_flight is not None else planner.pipeline_depth + 1, capi.SLOTS - 1, E))
    eps_ = [_Episode(env, i, new_trajectory(S, A, max_steps)) for i, env in enumerate(envs)]
    for ep in eps_:
        ep.observation = ep.env.reset()
    waiting: List[_Episode] = list(eps_)   # environments that need a plan step issued, in round-robin order
    flying: List[_Episode] = []            # environments whose plan step is on the device, oldest first
    steps = 0

    def issue(ep: _Episode):
        ep.traj["observations"][ep.timestep] = ep.observation
        rtg = float(episode_rtg_ref[ep.timestep] * ratio) if eval else None
        ep.ticket = planner.plan_async(ep.traj, percentage=percentage, eval=eval, rtg=rtg)
        flying.append(ep)

    def land(ep: _Episode, action=None):
        nonlocal steps
        if action is None:
            action = ep.ticket.result().cpu().numpy()
            ep.ticket = None
        action = np.clip(action, -1, 1)  # learner.py:690 / replay_buffer.py:213-215
        obs, reward, done, info = ep.env.step(action)
        ep.traj["actions"][ep.timestep] = action
        ep.traj["rewards"][ep.timestep] = reward
        ep.traj["total_return"] += reward
        ep.observation = obs
        ep.timestep += 1
        ep.traj["path_length"] += 1
        ep.done, ep.info = bool(done), info
        steps += 1
        if on_step is not None:
            on_step(ep.index, ep.timestep, action, reward, done)
        if not ep.done and ep.timestep < max_steps:
            waiting.append(ep)

    while lockstep and waiting:
        live, waiting = waiting, []
        for ep in live:
            ep.traj["observations"][ep.timestep] = ep.observation
        rtgs = [float(episode_rtg_ref[ep.timestep] * ratio) for ep in live] if eval else None
        acts = planner.action_sample_batch([ep.traj for ep in live], percentage=percentage, eval=eval, rtg=rtgs, lockstep=True)
        for ep, action in zip(live, acts.cpu().numpy()):  # one read-back per round
            land(ep, action)
    while waiting or flying:
        while waiting and len(flying) < cap:
            issue(waiting.pop(0))
        land(flying.pop(0))
    stats: Dict[str, List[float]] = defaultdict(list)
    for ep in eps_:
        if isinstance(ep.info, dict) and "episode" in ep.info:       # jaxrl's EpisodeMonitor (learner.py:703-708)
            for k, v in ep.info["episode"].items():
                stats[k].append(float(v))
        else:
            stats["return"].append(float(ep.traj["rewards"].sum()))
            stats["length"].append(float(ep.traj["path_length"]))
    out: Dict[str, Any] = {}
    for k, v in stats.items():
        out[k + "_mean"] = float(np.mean(v))
        out[k + "_std"] = float(np.std(v))
    out["returns"], out["lengths"] = stats.get("return", []), stats.get("length", [])
    out["trajectories"] = [ep.traj for ep in eps_]
    out["plan_steps"] = steps
    return out
