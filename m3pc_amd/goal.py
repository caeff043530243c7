"""The zero-shot goal-reaching calls of the reference (research/zeroshot_omtm/learner.py: action_id_sample 60-149,
action_piid_sample 151-261, action_piid_list_sample 263-370) and the CEM refinement of a plan, as a mixin of
``HipPlanner`` (m3pc_amd/planner.py: the step pipeline; m3pc_amd/certificate.py: the certified re-score).  Everything here runs
on the planner's handle: ``m3pc_goal_step`` (both forwards of one or a few windows, fp32), ``m3pc_goal_step_batch`` (thousands of
windows per call, exactly pruned, BASELINE config 5), ``m3pc_forward`` and ``m3pc_score_actions``."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import capi
from .tokenizers import SquashedNormal


class GoalMixin:
    # ---------------------------------------------------------------------------------------- zero-shot
    def assemble_goal_window(self, sequence_history, rtg=None, percentage=1.0):
        """research/zeroshot_omtm/learner.py:164-223: the history window, with the observation rows of the
        WHOLE window taken from the buffer (future rows are way-points), shortened near the 1000-step end."""
        dev, (horizon, return_to_go) = self._stage_copy(lambda flat: self._goal_window_host(sequence_history, rtg, percentage, flat))
        states, actions, rewards = self._blocks(dev)
        return states, actions, rewards, horizon, return_to_go

    def _goal_window_host(self, sequence_history, rtg, percentage, flat):
        """Host half of ``assemble_goal_window``: fills the flat window buffer and returns (horizon, rtg)."""
        T = self.T
        horizon = int(self.cfg.horizon)
        end_idx = int(sequence_history["path_length"])
        if end_idx + horizon < T:
            horizon = T - end_idx
        smart = T
        if end_idx + horizon > 1000:
            smart = smart - (end_idx + horizon - 1000)
        hl = T - horizon + 1
        flat[:] = 0.0
        bs, ba, br = self._blocks(flat)
        lo = end_idx - hl + 1
        ba[:hl] = sequence_history["actions"][lo : end_idx + 1]
        br[:hl] = np.asarray(sequence_history["rewards"][lo : end_idx + 1]).reshape(hl, 1)
        bs[:hl] = sequence_history["observations"][lo : end_idx + 1]
        bs[:smart] = sequence_history["observations"][lo : lo + T]
        return horizon, self._rtg_value(rtg, percentage)

    def _goal_tokens(self, states, actions, rewards, rtg):
        T = self.T
        ret = torch.full((1, T, 1), rtg, dtype=torch.float64, device=self.device)
        return [self.handle.tokenize(capi.STATES, states[None]), actions[None].contiguous(),
                self.handle.tokenize(capi.REWARDS, rewards[None]), self.handle.tokenize(capi.RETURNS, ret)]

    def _policy_from(self, toks, masks, h, eval):
        from .masks import mask_rows
        mu, sd = self.handle.forward(toks, mask_rows(masks), want=("actions",))["actions"]
        self._mark_main()
        dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
        if eval:
            return dist_.mean[0, self.T - h]
        return dist_.sample(eps=self._eps(tuple(dist_.loc.shape)))[0, self.T - h]

    @torch.no_grad()
    def action_id_sample(self, sequence_history, percentage=1.0, horizon=4, plan=True, eval=False, rtg=None):
        """zeroshot learner.py:60-149: one forward under the goal inverse-dynamics mask."""
        if eval:
            assert rtg is not None
        from .masks import create_gid_mask
        s, a, r, h, rtg_v = self.assemble_goal_window(sequence_history, rtg, percentage)
        toks = self._goal_tokens(s, a, r, rtg_v)
        return self._policy_from(toks, create_gid_mask(self.T, "cpu", self.T - h), h, eval)

    @torch.no_grad()
    def action_piid_sample(self, sequence_history, percentage=1.0, horizon=4, plan=True, eval=False, rtg=None):
        """zeroshot learner.py:151-261: path inference (pi mask) -> write the inferred states into the
        window -> inverse dynamics (fid mask) -> action at T-h."""
        if eval:
            assert rtg is not None
        from .masks import create_fid_mask, create_pi_mask, mask_rows
        T = self.T
        s, a, r, h, rtg_v = self.assemble_goal_window(sequence_history, rtg, percentage)
        idx = T - h
        self._drain()  # (m3pc_goal_step runs in the policy workspace: no pipelined plan step may still be using it)
        # both forwards and the hand-over between them in one library call on the raw window (m3pc_goal_step)
        mu, sd, inferred, window = self.handle.goal_step(s[None], a[None], r[None], [rtg_v], mask_rows(create_pi_mask(T, "cpu", idx)),
                                                         mask_rows(create_fid_mask(T, "cpu", idx)), idx)
        self._mark_main()
        self.last = dict(state_inference=inferred, window_states=window[0])
        dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
        if eval:
            return dist_.mean[0, idx]
        return dist_.sample(eps=self._eps(tuple(dist_.loc.shape)))[0, idx]

    @torch.no_grad()
    def action_piid_list_sample(self, sequence_history, percentage=1.0, horizon=4, plan=True, eval=False, rtg=None):
        """zeroshot learner.py:263-370 (goal_mask "piid_allout", unseen.py:146-148): the piid arithmetic, but the result --
        always the MEAN of the action distribution at T-h, eval or not -- is left in ``self.action_list`` (one entry: the
        reference's further entries are commented out at 366-370) for the rollout loop to pop (learner.py:559-568).
        Returns None, as the reference does."""
        if eval:
            assert rtg is not None
        self.action_list = [self.action_piid_sample(sequence_history, percentage, horizon, plan, eval=True,
                                                    rtg=self._rtg_value(rtg, percentage))]
        return None

    @torch.no_grad()
    def goal_actions(self, states, actions, horizon: int, eval: bool = True, goal_mask: str = "piid", precision: Optional[str] = None,
                     want_window: bool = False):
        """The zero-shot action of E windows that are on the device already: states (E,T,S), actions (E,T,A) raw, one effective
        horizon for all of them (BASELINE config 5: thousands of goal-reaching windows per GPU).  Exactly pruned many-window path
        (m3pc_goal_step_batch): path inference reads the states head at the rows the overlay uses only, inverse dynamics reads
        ONE action token (zeroshot learner.py:240-256).  goal_mask "piid" (action_piid_sample) or "id" (action_id_sample).
        precision: "bf16" / "fp32"; default the planner's.  Returns (E, A): tanh(loc) when eval, a sample else."""
        prec = self.precision if precision is None else {"fp32": capi.PREC_FP32, "bf16": capi.PREC_BF16}[precision]
        idx = self.T - int(horizon)
        self._drain()  # (m3pc_goal_step_batch runs in the candidate workspace)
        res = self.handle.goal_step_batch(states, actions, idx, capi.GOAL_PIID if goal_mask == "piid" else capi.GOAL_ID, prec,
                                          want_window=want_window)
        mu, sd = res[0], res[1]
        if want_window:
            self.last = dict(window_states=res[2], loc=mu, std=sd)
        if eval:
            return torch.tanh(mu)
        # SquashedNormal.sample (mtm_model.py:263-269): the variates of the whole (E,T,1,A) distribution are drawn, as the
        # reference draws them, and the token's are used
        eps = self._eps((mu.shape[0], self.T, 1, self.A))[:, idx, 0]
        return torch.tanh(eps * sd + mu)

    @torch.no_grad()
    def action_piid_sample_batch(self, sequence_histories, percentage=1.0, eval=True, rtg=None, pruned: Optional[bool] = None):
        """E independent goal-reaching windows per launch (BASELINE config 5 / SURVEY §8 f1): the reference plans one
        env per call (zeroshot learner.py:151-261, unseen.py rollout loop); here the windows that share a horizon go
        through the pi and fid forwards as ONE batch of the same kernels.  Per window the arithmetic is that of
        ``action_piid_sample``.  Returns (E, A).
        pruned=False (default up to 64 windows): the fp32 few-row kernels of ``action_piid_sample`` on the whole batch
        (``max_batch >= E``; ``last["state_inference"]`` holds every window's full states head).
        pruned=True (default beyond, needs ``goal_batch >= E``): the exactly pruned many-window path in the planner's
        precision (``goal_actions``)."""
        if eval:
            assert rtg is not None
        from .masks import create_fid_mask, create_pi_mask, mask_rows
        T, E = self.T, len(sequence_histories)
        S, A = self.S, self.A
        if pruned is None:
            pruned = E > 64 or E > self._max_batch
        if pruned and E > self._goal_batch:
            raise ValueError(f"{E} windows through the pruned path need HipPlanner(..., goal_batch >= {E})"
                             + (f" (or, up to 64 windows, max_batch >= {E} for the un-pruned fp32 path: max_batch is {self._max_batch})"
                                if E <= 64 else ""))
        host = np.empty((E, T * (S + A + 1)), dtype=np.float32)
        meta = [self._goal_window_host(hst, rtg, percentage, host[i]) for i, hst in enumerate(sequence_histories)]
        dev = torch.from_numpy(host).to(self.device)  # one packed H2D copy for all windows
        out = torch.empty((E, self.A), dtype=torch.float32, device=self.device)
        infer = [None] * E
        self._drain()
        for h in sorted({m[0] for m in meta}):
            ids = [i for i, m in enumerate(meta) if m[0] == h]
            idx = T - h
            sel = dev if len(ids) == E else dev[torch.tensor(ids, device=self.device)]
            s = sel[:, : T * S].reshape(-1, T, S).contiguous()
            a = sel[:, T * S : T * (S + A)].reshape(-1, T, A).contiguous()
            if pruned:
                act = self.goal_actions(s, a, h, eval=eval)
                if len(ids) == E:
                    out = act
                else:
                    out[torch.tensor(ids, device=self.device)] = act
                continue
            r = sel[:, T * (S + A) :].reshape(-1, T, 1).contiguous()
            mu, sd, inferred, _ = self.handle.goal_step(s, a, r, [meta[i][1] for i in ids], mask_rows(create_pi_mask(T, "cpu", idx)),
                                                        mask_rows(create_fid_mask(T, "cpu", idx)), idx)
            dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
            act = dist_.mean if eval else dist_.sample(eps=self._eps(tuple(dist_.loc.shape)))
            out[torch.tensor(ids, device=self.device)] = act[:, idx, 0]
            for j, i in enumerate(ids):
                infer[i] = inferred[j]
        self._mark_main()
        # (the pruned path computes the states head on the rows the overlay reads only: there is no full per-window inference to
        # report -- said here instead of a list of None; goal_actions(want_window=True) gives the window rows)
        self.last = dict(state_inference=None if pruned else infer, pruned=bool(pruned))
        return out

    # ---------------------------------------------------------------------------------------- CEM refinement
    @torch.no_grad()
    def cem_guiding(self, trajectory: Dict[str, torch.Tensor], h: int, iterations: int = 2, top_k: int = 128, init_std: float = 0.1,
                    noise=None):
        """Cross-entropy refinement of the plan (the legacy ``sample_action_cem`` of
        research/omtm/datasets/sequence_dataset.py:919-1000 -- N=1024, top_k=128, 2 iterations -- restated on this model's
        plan step: that function predates the four-key omtm model and cannot run on it, so parity is pinned on the oracle's
        restatement of the same algorithm (tests/test_batch_gpu.py), not on the reference).
          candidates_0 = clamp(tanh(policy loc) + init_std * noise_0, -1, 1) over the last h steps
          repeat: score (TD(lambda) as rtg_guiding / critic_lambda_guiding) -> top_k -> mean / std per (t, a)
                  candidates = clamp(mean + std * noise_i, -1, 1)
        Returns (sample_action (1,A): first action of candidate 0 after the last refit, as the legacy code returns;
                 eval_action (A,): first action of the final mean).  ``noise``: optional (iterations+1, N, h, A) normals."""
        self._drain()
        s, a, r, rtg, ret = self._split(trajectory)
        cfg = self.cfg
        N, T, A = int(cfg.action_samples), self.T, self.A
        mode = capi.MODE_CRITIC if cfg.plan_guidance == "critic_lambda_guiding" else capi.MODE_RTG
        lmbda = 0.6 if mode == capi.MODE_RTG else float(cfg.lmbda)
        toks = [self.handle.tokenize(capi.STATES, s[None]), a[None], None, self._returns_tokens(rtg, ret)]
        from .masks import create_rcbc_mask, mask_rows
        mu, _ = self.handle.forward(toks, mask_rows(create_rcbc_mask(T, "cpu", T - h)), want=("actions",))["actions"]
        mean = torch.tanh(mu[0, T - h :])  # (h, A)
        std = torch.full_like(mean, float(init_std))
        if noise is None:
            noise = self._eps((iterations + 1, N, h, A))
        k = min(int(top_k), N)
        cand = torch.clamp(mean[None] + std[None] * noise[0], -1.0, 1.0)
        trace = []
        for it in range(iterations):
            er = self.handle.score_actions(mode, s, a, r, cand, None, h, lmbda, float(cfg.discount), precision=self.precision)
            top = torch.topk(er, k).indices
            elite = cand[top]
            mean = elite.mean(dim=0)
            std = elite.std(dim=0) if k > 1 else torch.zeros_like(mean)
            trace.append(dict(expect_return=er, top=top, mean=mean, std=std))
            cand = torch.clamp(mean[None] + std[None] * noise[it + 1], -1.0, 1.0)
        self._mark_main()
        self.last = dict(cem=trace, candidates=cand)
        return cand[0, 0][None], mean[0]

