"""Zero-shot goal reaching driver: what ``Learner.shot`` of the reference does around the per-step
``action_piid_sample`` / ``action_id_sample`` calls (research/zeroshot_omtm/learner.py:497-652) and how ``unseen.py``
picks the mode (unseen.py:146-148) -- minus the simulator, which the caller owns.

    follower = WaypointFollower(planner, "hopper-wiggle-f2.txt", index_jump=4, goal_mask="piid")   # or "piid_allout" / "id"
    traj = follower.new_trajectory()
    obs = env.reset()
    for t in range(1000):
        action = follower.act(traj, obs, t, rtg=episode_rtg_ref[t] * ratio)     # np.ndarray (1, A) as the reference's, clipped to [-1, 1]
        obs, reward, done, info = env.step(action)
        follower.record(traj, t, action, reward)

``act_batch`` does the same for E environments at once on ``planner.action_piid_sample_batch``."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch


def load_waypoints(path: str) -> np.ndarray:
    """The 1000 x S way-point files of research/zeroshot_omtm/waypoint_gen (np.savetxt text), as the reference reads
    them (learner.py:528): ``np.loadtxt``."""
    wp = np.loadtxt(path)
    if wp.ndim != 2:
        raise ValueError(f"{path}: expected a 2-D table of way-points, got shape {wp.shape}")
    return wp


def hold_waypoints(observations: np.ndarray, index_jump: int) -> np.ndarray:
    """learner.py:530-539: every (index_jump + 1)-th row is a goal; the index_jump rows before it are overwritten with
    it, so the goal is held while the agent approaches.  Works in place and returns its argument."""
    father = index_jump
    n = observations.shape[0]
    while father < n - 1:
        for i in range(index_jump):
            observations[father - 1 - i] = observations[father]
        father += index_jump + 1
    return observations


def goal_mode(goal_mask: str) -> str:
    """unseen.py:146-148: "piid" -> two-stage path inference + inverse dynamics; "piid_allout" -> the action-list
    variant; anything else ("id") -> single-stage."""
    if goal_mask == "piid":
        return "two_stage"
    if goal_mask == "piid_allout":
        return "list_stage"
    return "single"


class WaypointFollower:
    def __init__(self, planner, way_points_path: str, index_jump: int = None, goal_mask: str = "piid"):
        self.planner = planner
        self.index_jump = int(index_jump if index_jump is not None else planner.cfg.index_jump)
        self.mode = goal_mode(goal_mask)
        self.waypoints = hold_waypoints(np.array(load_waypoints(way_points_path), dtype=np.float64), self.index_jump)

    def new_trajectory(self) -> Dict[str, np.ndarray]:
        """learner.py:515-539: the episode buffer; its observation rows start out as the (held) way-points."""
        S, A = self.planner.S, self.planner.A
        if self.waypoints.shape[1] != S:
            raise ValueError(f"way-points have {self.waypoints.shape[1]} columns, the model has {S} state dims")
        return {"observations": self.waypoints.copy(), "actions": np.zeros((1000, A), dtype=np.float32),
                "rewards": np.zeros((1000, 1), dtype=np.float32), "values": np.zeros((1000, 1), dtype=np.float32),
                "total_return": 0, "path_length": 0}

    def _call(self, traj, rtg):
        if self.mode == "list_stage":
            # learner.py:559-568: refill the planner's action list when it is empty, then pop its head
            if len(self.planner.action_list) == 0:
                self.planner.action_piid_list_sample(traj, percentage=1.0, plan=False, eval=True, rtg=rtg)
            return self.planner.action_list.pop(0)
        fn = self.planner.action_piid_sample if self.mode == "two_stage" else self.planner.action_id_sample
        return fn(traj, percentage=1.0, plan=False, eval=True, rtg=rtg)

    def act(self, traj, observation, timestep: int, rtg: float) -> np.ndarray:
        """learner.py:548-582: write the observation, plan, clip (np.clip(action.cpu().numpy(), -1, 1))."""
        traj["observations"][timestep] = observation
        traj["path_length"] = timestep
        return np.clip(self._call(traj, rtg).cpu().numpy(), -1, 1)

    def act_batch(self, trajs: Sequence[dict], observations: Sequence[np.ndarray], timesteps: Sequence[int],
                  rtgs: Sequence[float]) -> np.ndarray:
        """E environments per call (two-stage mode): rows of the returned (E, A) array are what ``act`` returns per env.
        One rtg per call in the batched kernel path: windows are grouped by rtg value."""
        if self.mode != "two_stage":
            return np.stack([self.act(tr, o, t, g) for tr, o, t, g in zip(trajs, observations, timesteps, rtgs)])
        for tr, o, t in zip(trajs, observations, timesteps):
            tr["observations"][t] = o
            tr["path_length"] = t
        out = np.empty((len(trajs), self.planner.A), dtype=np.float32)
        groups: Dict[float, List[int]] = {}
        for i, g in enumerate(rtgs):
            groups.setdefault(float(g), []).append(i)
        for g, ids in groups.items():
            acts = self.planner.action_piid_sample_batch([trajs[i] for i in ids], percentage=1.0, eval=True, rtg=g)
            out[ids] = acts.cpu().numpy()
        return np.clip(out, -1, 1)

    @staticmethod
    def record(traj, timestep: int, action, reward) -> None:
        """learner.py:599-603."""
        traj["actions"][timestep] = action
        traj["rewards"][timestep] = reward
        traj["total_return"] += reward
        traj["path_length"] = timestep + 1
