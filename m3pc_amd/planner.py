"""The plan_guidance plug-in surface of the reference ``Learner`` (research/finetune_omtm/learner.py:
103-115 mtm_sampling, 142-208 noise_adding_lambda, 211-268 critic_lambda_guiding, 271-327 rtg_guiding,
329-417 action_sample) driving libm3pc_hip.so.  Same method names, argument meaning, return shapes
((1,A) sampled action, (A,) eval action) and string dispatch on ``cfg.plan_guidance``.

Two ways in:
  * ``HipPlanner(cfg, state_dict, tokenizer_manager | stats, q_state_dict, obs_mean, obs_std)``
  * ``attach(learner)``: build a planner from a live reference ``Learner`` (its ``mtm``, ``tokenizer_manager``,
    ``iql.qf``, ``cfg``) and rebind the guidance methods + ``action_sample`` on that object; weights are
    re-synchronised whenever a parameter's ``_version`` changes (fine-tuning updates them between rollouts,
    finetune.py:306).

Host work per step: assemble the window in NumPy (one packed H2D copy), draw eps / the multinomial index
with torch's device generator, and -- when candidates are sharded over ranks -- one all-gather.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import numpy as np
import os

import torch

from . import capi, dist as mdist
from .tokenizers import ContinuousTokenizer, DataStatistics, SquashedNormal, TokenizerManager

KEYS = capi.KEYS
_MODES = {"rtg_guiding": capi.MODE_RTG, "critic_lambda_guiding": capi.MODE_CRITIC,
          "noise_adding_lambda": capi.MODE_NOISE}


def _cfg_get(cfg, name, default=None):
    return getattr(cfg, name, default)


class HipPlanner:
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], tokenizer_manager, q_state_dict=None,
                 obs_mean=None, obs_std=None, n_embd: int = 512, n_head: int = 4, n_enc_layer: int = 2,
                 n_dec_layer: int = 1, precision: str = "fp32", rescore_topk: int = 16, device: Optional[int] = None,
                 group=None, generator: Optional[torch.Generator] = None, max_batch: int = 1,
                 rescore: str = "bound", rescore_min: int = 8, rescore_max: int = 64, rescore_delta: Optional[float] = None,
                 max_windows: int = 1):
        """cfg: any object with traj_length, action_samples, horizon, discount, temperature, lmbda,
        plan_guidance (finetune.py RunConfig fields read at learner.py:276,319,342).
        tokenizer_manager: a TokenizerManager (this package's) or {key: {"mean","std","min","max"}}.
        precision: "fp32" (reference-accurate) or "bf16" (bf16 MFMA candidate pass followed by an fp32 re-score of the
        candidates that can still be the arg-max, so that the arg-max does not depend on bf16 rounding):
          rescore="bound" (default): every candidate whose bf16 score lies within 2*delta of the bf16 maximum is
            re-scored (at least ``rescore_min``, at most ``rescore_max``).  delta bounds the bf16 error of score
            DIFFERENCES: |(b_i - f_i) - median(b - f)| <= delta; then f_argmax >= f_j for all j implies
            b_argmax >= b_max - 2 delta.  delta is calibrated per weight load on 64 candidates scored in both
            arithmetics (1.5 x the largest deviation seen), or fixed by ``rescore_delta``.  ``planner.last`` reports
            n_rescored, min_margin_outside (distance from the bf16 maximum to the best candidate NOT re-scored: the
            bound held with room when it exceeds 2*delta) and delta.  One 16-byte device-to-host read per step.
          rescore="topk": the fixed ``rescore_topk`` best candidates (round-1 behaviour, no host read)."""
        self.cfg = cfg
        self.group = group
        self.rank, self.world = mdist.world_info(group)
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        S = state_dict["encoder_embed_dict.states.weight"].shape[1]
        A = state_dict["encoder_embed_dict.actions.weight"].shape[1]
        T = int(cfg.traj_length)
        N = int(cfg.action_samples)
        _, n_local = mdist.shard_range(N, 0, self.world)
        hidden = 0 if q_state_dict is None else q_state_dict["q1.net.0.weight"].shape[0]
        self.handle = capi.Handle(S, A, T, n_embd, n_head, n_enc_layer, n_dec_layer,
                                  max_candidates=max(n_local * max(int(max_windows), 1), rescore_topk,
                                                     int(rescore_max) * max(int(max_windows), 1) if precision == "bf16" else 1, 1),
                                  max_batch=max(int(max_batch), int(max_windows), 1),
                                  critic_hidden=hidden,
                                  device=device)
        self.device = self.handle.device
        self.S, self.A, self.T = S, A, T
        self.precision = {"fp32": capi.PREC_FP32, "bf16": capi.PREC_BF16}[precision]
        self.rescore_topk = int(rescore_topk) if self.precision == capi.PREC_BF16 else 0
        assert rescore in ("bound", "topk")
        self.rescore = rescore if self.precision == capi.PREC_BF16 else "none"
        self.rescore_min, self.rescore_max = int(rescore_min), int(rescore_max)
        self._delta_fixed = None if rescore_delta is None else float(rescore_delta)
        self._delta: Optional[float] = self._delta_fixed
        self._kspec, self._exceed = int(rescore_min), 0.05  # size of the unconditional first re-score, share of steps that needed more
        self.generator = generator
        if self.world > 1:
            # every rank draws eps / the multinomial variates itself: the streams must be the same ones
            if generator is None:
                raise ValueError("a sharded planner needs an explicit torch.Generator seeded identically on every rank")
            mdist.check_same_generator(generator, group)
        if isinstance(tokenizer_manager, dict):
            # raw dataset statistics: build the tokenizers as ContinuousTokenizer.create does (std < 0.1 -> 1)
            tokenizer_manager = TokenizerManager({k: ContinuousTokenizer.from_statistics(k, tokenizer_manager[k]) for k in KEYS})
        self.tokenizer_manager = tokenizer_manager.bind(self.handle)
        self.load_state_dict(state_dict)
        if q_state_dict is not None:
            self.load_critic(q_state_dict, obs_mean, obs_std)
        self._host = np.zeros((T, S + A + 1), dtype=np.float32)
        self.last: Dict[str, torch.Tensor] = {}

    # ---------------------------------------------------------------------------------------- weights
    def load_state_dict(self, state_dict):
        self.handle.load_weights(state_dict)
        self._delta = getattr(self, "_delta_fixed", None)  # the bf16 error bound belongs to the weights: re-calibrate
        self._kspec, self._exceed = int(getattr(self, "rescore_min", 8)), 0.05

    def load_critic(self, q_state_dict, obs_mean, obs_std):
        self.handle.set_critic(q_state_dict, obs_mean, obs_std)

    # ---------------------------------------------------------------------------------------- window
    def assemble_window(self, sequence_history, rtg=None, percentage=1.0):
        """learner.py:342-385.  Returns (states (T,S), actions (T,A), rewards (T,1) on device, horizon, rtg)."""
        T = self.T
        horizon = int(self.cfg.horizon)
        end_idx = int(sequence_history["path_length"])
        if end_idx + horizon < T:
            horizon = T - end_idx
        hl = T - horizon + 1
        buf = self._host
        buf[:] = 0.0
        lo, hi = end_idx - hl + 1, end_idx + 1
        buf[:hl, : self.S] = sequence_history["observations"][lo:hi]
        buf[:hl, self.S : self.S + self.A] = sequence_history["actions"][lo:hi]
        buf[:hl, self.S + self.A :] = np.asarray(sequence_history["rewards"][lo:hi]).reshape(hl, 1)
        dev = torch.from_numpy(buf).to(self.device)  # one packed H2D copy
        states = dev[:, : self.S].contiguous()
        actions = dev[:, self.S : self.S + self.A].contiguous()
        rewards = dev[:, self.S + self.A :].contiguous()
        if rtg is not None:
            return_to_go = float(rtg)
        else:
            st = self.tokenizer_manager.tokenizers["returns"].stats
            return_to_go = float(np.asarray(st.min + (st.max - st.min) * percentage).reshape(-1)[0])
        return states, actions, rewards, horizon, return_to_go

    # ---------------------------------------------------------------------------------------- guidance
    def _eps(self, shape):
        return torch.randn(shape, device=self.device, dtype=torch.float32, generator=self.generator)

    def _guide(self, mode: int, states, actions, rewards, rtg: float, h: int, lmbda: float, eps=None):
        cfg = self.cfg
        N, T, A = int(cfg.action_samples), self.T, self.A
        if eps is None:
            # same shapes the reference draws: dist.sample((N,)) over loc (1,T,1,A) (learner.py:285) /
            # randn((N,h,A)) for the fixed-variance variant (learner.py:157-163); identical on every rank
            eps = self._eps((N, h, A)) if mode == capi.MODE_NOISE else self._eps((N, 1, T, 1, A))
        eps = eps.reshape(N, -1, A)
        begin, count = mdist.shard_range(N, self.rank, self.world)
        res = self.handle.plan_step(mode, states, actions, rewards, eps, h, rtg, float(lmbda), float(cfg.discount),
                                    N, begin, count, precision=self.precision)
        er, a0 = res["expect_return"], res["sample_actions"][:, 0]
        er, a0 = mdist.gather_candidates(er, a0, N, self.group)
        top = sel = None
        extra = {}
        # the re-score is replicated on every rank (identical inputs => identical result): candidates chosen from the
        # gathered scores, fp32 candidate pass on them, scores written back in place
        if self.rescore == "topk" and self.rescore_topk > 0:
            top = self.handle.rescore_topk(mode, states, actions, rewards, eps, er, min(self.rescore_topk, N), h, rtg,
                                           float(lmbda), float(cfg.discount))
        elif self.rescore == "bound":
            rs = (mode, states, actions, rewards, eps)
            tail = (h, rtg, float(lmbda), float(cfg.discount))
            if self._delta is None:
                self._delta = self._calibrate(er, rs, tail, N)
            kmax, kmin = min(self.rescore_max, N - 1 if N > 1 else 1), min(max(self.rescore_min, self._kspec), N)
            kmin = max(min(kmin, kmax), 1)
            # The candidates come sorted by bf16 score, so the set inside the window is a prefix of the list and its first
            # kmin entries are re-scored whatever the count turns out to be: that re-score and the select are enqueued
            # BEFORE the one host read of the step (how many candidates are inside the window; pinned-memory spin, no
            # stream sync), so the device never waits for the host.  Only when more than kmin candidates are inside the
            # window the rest is re-scored and the select repeated on the same variates.
            cand, ticket = self.handle.topk_window_issue(er, max(kmax, 1), kmin, 2.0 * self._delta)
            if os.environ.get("M3PC_NO_SPEC"):  # A/B switch: the host read first (the device idles for the round trip)
                self.handle.topk_window_wait(ticket)
            self.handle.rescore_listed(mode, states, actions, rewards, eps, er, cand[:kmin], *tail)
            expo = torch.empty((N,), dtype=torch.float32, device=self.device).exponential_(1, generator=self.generator)
            sel = self.handle.select(er, a0, float(cfg.temperature), expo)
            st = self.handle.topk_window_wait(ticket)
            n_re = int(st[0])
            if n_re > kmin:
                self.handle.rescore_listed(mode, states, actions, rewards, eps, er, cand[kmin:n_re].contiguous(), *tail)
                sel = self.handle.select(er, a0, float(cfg.temperature), expo)
            top = cand[:n_re]
            # the size of the first (unconditional) re-score follows the workload: a second pass costs a whole fp32 chain
            # (~0.3 ms), four more candidates in the first ~0.02-0.05 ms -- grown when more than a fifth of the recent steps
            # needed the second pass, shrunk again when (almost) none did.  Same decisions on every rank (same counts).
            self._exceed += (float(n_re > kmin) - self._exceed) / 16.0
            if self._exceed > 0.2 and kmin < kmax:
                self._kspec, self._exceed = min(kmax, kmin + 4), 0.05
            elif self._exceed < 0.005 and self._kspec > self.rescore_min:
                self._kspec, self._exceed = max(self.rescore_min, self._kspec - 4), 0.05
            extra = dict(n_rescored=n_re, n_in_window=int(st[3]), min_margin_outside=float(st[1]), delta=self._delta, n_first=kmin)
        # torch.multinomial(p, 1) == argmax(p / q), q ~ Exp(1) from the same generator (ATen's
        # multinomial fast path); drawing q here and finishing inside the select kernel gives the same index
        if sel is None:
            expo = torch.empty((N,), dtype=torch.float32, device=self.device).exponential_(1, generator=self.generator)
            sel = self.handle.select(er, a0, float(cfg.temperature), expo)
        p, eval_action, argmax, sample_idx, sample_action = sel
        self.last = dict(expect_return=er, p=p, argmax=argmax, sample_idx=sample_idx, loc=res["loc"], std=res["std"],
                         sample_actions=res["sample_actions"], eps=eps, topk=top, **extra)
        return sample_action, eval_action

    def _calibrate(self, er_bf16, rs, tail, N, n_cal: int = 64) -> float:
        """delta of the bound-driven re-score: fp32 scores of n_cal candidates (a fixed pseudo-random subset, the same
        on every rank) against their bf16 scores; 1.5 x the largest deviation of (bf16 - fp32) from its median."""
        g = torch.Generator().manual_seed(0x5eed)
        ids = torch.randperm(N, generator=g)[: min(n_cal, N)].to(torch.int32).to(self.device)
        f32, _ = self.handle.rescore(*rs, ids, tail[0], tail[1], tail[2], tail[3], N)
        d = er_bf16[ids.long()] - f32
        dev = float((d - d.median()).abs().max())
        return max(1.5 * dev, 1e-6 * float(f32.abs().max()), 1e-30)

    @staticmethod
    def _split(trajectory):
        s, a, r = trajectory["states"][0], trajectory["actions"][0], trajectory["rewards"][0]
        rtg = trajectory.get("_rtg")
        if rtg is None:
            ret = trajectory["returns"].reshape(-1)  # device sync; action_sample passes _rtg
            rtg = float(ret[0])
            # the library conditions on ONE return-to-go (what action_sample builds, learner.py:368-385); a window with
            # varying returns is legal in the reference but not representable here: refuse it instead of mis-planning
            if not bool((ret == ret[0]).all()):
                raise ValueError("trajectory['returns'] must be constant over the window (rtg_guiding is called by "
                                 "action_sample with a constant return-to-go, learner.py:368-385)")
        return s.float().contiguous(), a.float().contiguous(), r.float().contiguous(), float(rtg)

    @torch.no_grad()
    def rtg_guiding(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float = 0.6):
        """learner.py:271-327."""
        s, a, r, rtg = self._split(trajectory)
        return self._guide(capi.MODE_RTG, s, a, r, rtg, h, lmbda)

    @torch.no_grad()
    def critic_lambda_guiding(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float):
        """learner.py:211-268."""
        s, a, r, rtg = self._split(trajectory)
        return self._guide(capi.MODE_CRITIC, s, a, r, rtg, h, lmbda)

    @torch.no_grad()
    def noise_adding_lambda(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float):
        """learner.py:142-208."""
        s, a, r, rtg = self._split(trajectory)
        return self._guide(capi.MODE_NOISE, s, a, r, rtg, h, lmbda)

    @torch.no_grad()
    def mtm_sampling(self, trajectory: Dict[str, torch.Tensor], h: int):
        """learner.py:103-115: one return-conditioned policy pass, no planning."""
        s, a, r, rtg = self._split(trajectory)
        T = self.T
        toks = [self.handle.tokenize(capi.STATES, s[None]), a[None], None,
                self.handle.tokenize(capi.RETURNS, torch.full((1, T, 1), rtg, dtype=torch.float64, device=self.device))]
        from .masks import create_rcbc_mask, mask_rows
        out = self.handle.forward(toks, mask_rows(create_rcbc_mask(T, "cpu", T - h)), want=("actions",))
        mu, sd = out["actions"]
        dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
        eps = self._eps(tuple(dist_.loc.shape))
        sample_action = dist_.sample(eps=eps)[0, T - h]
        eval_action = dist_.mean[0, T - h]
        return sample_action, eval_action

    # ---------------------------------------------------------------------------------------- zero-shot
    def assemble_goal_window(self, sequence_history, rtg=None, percentage=1.0):
        """research/zeroshot_omtm/learner.py:164-223: the history window, with the observation rows of the
        WHOLE window taken from the buffer (future rows are way-points), shortened near the 1000-step end."""
        horizon, return_to_go = self._goal_window_host(sequence_history, rtg, percentage, self._host)
        dev = torch.from_numpy(self._host).to(self.device)
        states = dev[:, : self.S].contiguous()
        actions = dev[:, self.S : self.S + self.A].contiguous()
        rewards = dev[:, self.S + self.A :].contiguous()
        return states, actions, rewards, horizon, return_to_go

    def _goal_window_host(self, sequence_history, rtg, percentage, buf):
        """Host half of ``assemble_goal_window``: fills ``buf`` (T, S+A+1) and returns (horizon, rtg)."""
        T = self.T
        horizon = int(self.cfg.horizon)
        end_idx = int(sequence_history["path_length"])
        if end_idx + horizon < T:
            horizon = T - end_idx
        smart = T
        if end_idx + horizon > 1000:
            smart = smart - (end_idx + horizon - 1000)
        hl = T - horizon + 1
        buf[:] = 0.0
        lo = end_idx - hl + 1
        buf[:hl, self.S : self.S + self.A] = sequence_history["actions"][lo : end_idx + 1]
        buf[:hl, self.S + self.A :] = np.asarray(sequence_history["rewards"][lo : end_idx + 1]).reshape(hl, 1)
        buf[:hl, : self.S] = sequence_history["observations"][lo : end_idx + 1]
        buf[:smart, : self.S] = sequence_history["observations"][lo : lo + T]
        if rtg is not None:
            return_to_go = float(rtg)
        else:
            st = self.tokenizer_manager.tokenizers["returns"].stats
            return_to_go = float(np.asarray(st.min + (st.max - st.min) * percentage).reshape(-1)[0])
        return horizon, return_to_go

    def _goal_tokens(self, states, actions, rewards, rtg):
        T = self.T
        ret = torch.full((1, T, 1), rtg, dtype=torch.float64, device=self.device)
        return [self.handle.tokenize(capi.STATES, states[None]), actions[None].contiguous(),
                self.handle.tokenize(capi.REWARDS, rewards[None]), self.handle.tokenize(capi.RETURNS, ret)]

    def _policy_from(self, toks, masks, h, eval):
        from .masks import mask_rows
        mu, sd = self.handle.forward(toks, mask_rows(masks), want=("actions",))["actions"]
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
        toks = self._goal_tokens(s, a, r, rtg_v)
        raw = self.handle.forward(toks, mask_rows(create_pi_mask(T, "cpu", idx)), want=("states",))["states"]
        inferred = self.handle.detokenize(capi.STATES, raw)  # (1,T,S)
        s = s.clone()
        s[idx + 2 : T - 1] = inferred[0, idx + 2 : T - 1]
        s[: idx + 1] = inferred[0, : idx + 1]
        toks[0] = self.handle.tokenize(capi.STATES, s[None])
        self.last = dict(state_inference=inferred, window_states=s)
        return self._policy_from(toks, create_fid_mask(T, "cpu", idx), h, eval)

    @torch.no_grad()
    def action_piid_sample_batch(self, sequence_histories, percentage=1.0, eval=True, rtg=None):
        """E independent goal-reaching windows per launch (BASELINE config 5 / SURVEY §8 f1): the reference plans one
        env per call (zeroshot learner.py:151-261, unseen.py rollout loop); here the windows that share a horizon go
        through the pi and fid forwards as ONE batch of the same kernels.  Per window the arithmetic is that of
        ``action_piid_sample``.  Returns (E, A).  The planner must have been built with ``max_batch >= E``."""
        if eval:
            assert rtg is not None
        from .masks import create_fid_mask, create_pi_mask, mask_rows
        T, E = self.T, len(sequence_histories)
        host = np.empty((E, T, self.S + self.A + 1), dtype=np.float32)
        meta = [self._goal_window_host(hst, rtg, percentage, host[i]) for i, hst in enumerate(sequence_histories)]
        dev = torch.from_numpy(host).to(self.device)  # one packed H2D copy for all windows
        out = torch.empty((E, self.A), dtype=torch.float32, device=self.device)
        infer = [None] * E
        for h in sorted({m[0] for m in meta}):
            ids = [i for i, m in enumerate(meta) if m[0] == h]
            idx = T - h
            sel = dev if len(ids) == E else dev[torch.tensor(ids, device=self.device)]
            s = sel[:, :, : self.S].contiguous()
            a = sel[:, :, self.S : self.S + self.A].contiguous()
            r = sel[:, :, self.S + self.A :].contiguous()
            ret = torch.tensor([meta[i][1] for i in ids], dtype=torch.float64, device=self.device)[:, None, None].expand(-1, T, 1)
            toks = [self.handle.tokenize(capi.STATES, s), a, self.handle.tokenize(capi.REWARDS, r),
                    self.handle.tokenize(capi.RETURNS, ret.contiguous())]
            raw = self.handle.forward(toks, mask_rows(create_pi_mask(T, "cpu", idx)), want=("states",))["states"]
            inferred = self.handle.detokenize(capi.STATES, raw)  # (B,T,S)
            s = s.clone()
            s[:, idx + 2 : T - 1] = inferred[:, idx + 2 : T - 1]
            s[:, : idx + 1] = inferred[:, : idx + 1]
            toks[0] = self.handle.tokenize(capi.STATES, s)
            mu, sd = self.handle.forward(toks, mask_rows(create_fid_mask(T, "cpu", idx)), want=("actions",))["actions"]
            dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
            act = dist_.mean if eval else dist_.sample(eps=self._eps(tuple(dist_.loc.shape)))
            out[torch.tensor(ids, device=self.device)] = act[:, idx, 0]
            for j, i in enumerate(ids):
                infer[i] = inferred[j]
        self.last = dict(state_inference=infer)
        return out

    # ---------------------------------------------------------------------------------------- batched planning
    @torch.no_grad()
    def action_sample_batch(self, sequence_histories, percentage=1.0, eval=False, rtg=None):
        """``action_sample(history, plan=True)`` for E environments in ONE pass of the kernels.  The
        reference steps one environment at a time (replay_buffer.py:204-232, learner.py:681-691: one action_sample per env
        step); here the E windows share the policy pass (batch E), one candidate pass over E x N rows, one fp32 re-score
        pass over every window's re-score set, and E select calls.  Per window the arithmetic is that of the single-window
        call (same kernels on more rows; split-K decisions of the fp32 policy pass may differ in the last bits).
        ``rtg``: None, a float, or one value per window.  Windows are grouped by effective horizon (early-episode windows
        plan with horizon T - end_idx, learner.py:342-345).  Returns (E, A).  Needs max_batch >= E and
        max_candidates >= E * N at construction (``HipPlanner(..., max_batch=E, max_windows=E)``)."""
        cfg = self.cfg
        guidance = cfg.plan_guidance
        assert guidance in _MODES, guidance
        mode = _MODES[guidance]
        lmbda = 0.6 if guidance == "rtg_guiding" else float(cfg.lmbda)  # learner.py:405-407
        E, T, A, N = len(sequence_histories), self.T, self.A, int(cfg.action_samples)
        rtgs = [rtg] * E if (rtg is None or np.isscalar(rtg)) else list(rtg)
        if eval:
            assert all(r is not None for r in rtgs)
        host = np.empty((E, T, self.S + self.A + 1), dtype=np.float32)
        meta = []
        for i, hst in enumerate(sequence_histories):
            meta.append(self._window_host(hst, rtgs[i], percentage, host[i]))
        dev = torch.from_numpy(host).to(self.device)  # one packed H2D copy for all windows
        out = torch.empty((E, A), dtype=torch.float32, device=self.device)
        info = [None] * E
        for h in sorted({m[0] for m in meta}):
            ids = [i for i, m in enumerate(meta) if m[0] == h]
            sel = dev if len(ids) == E else dev[torch.tensor(ids, device=self.device)]
            s = sel[:, :, : self.S].contiguous()
            a = sel[:, :, self.S : self.S + self.A].contiguous()
            r = sel[:, :, self.S + self.A :].contiguous()
            Eg = len(ids)
            eps = self._eps((Eg, N, h, A)) if mode == capi.MODE_NOISE else self._eps((Eg, N, T, A))
            res = self.handle.plan_step_batch(mode, s, a, r, [meta[i][1] for i in ids], eps, h, lmbda, float(cfg.discount), N,
                                              precision=self.precision)
            er, acts = res["expect_return"], res["sample_actions"]
            stats_h = None
            if self.rescore != "none":
                smode = capi.MODE_RTG if mode == capi.MODE_RTG else capi.MODE_CRITIC
                if self.rescore == "bound":
                    if self._delta is None:  # calibrate on window 0 of the group: 64 of its candidates in fp32
                        g = torch.Generator().manual_seed(0x5eed)
                        cid = torch.randperm(N, generator=g)[: min(64, N)].to(self.device)
                        f32 = self.handle.score_actions(smode, s[0], a[0], r[0], acts[0, cid], None, h, lmbda, float(cfg.discount))
                        d = er[0, cid] - f32
                        self._delta = max(1.5 * float((d - d.median()).abs().max()), 1e-6 * float(f32.abs().max()), 1e-30)
                    kmax, kmin = max(min(self.rescore_max, N - 1), 1), max(min(self.rescore_min, N), 1)
                    tops, stats = zip(*[self.handle.topk_window(er[w], kmax, min(kmin, kmax), 2.0 * self._delta) for w in range(Eg)])
                    stats_h = torch.stack(stats).cpu()  # the one host read of the group
                    counts = [int(stats_h[w, 0]) for w in range(Eg)]
                else:
                    k = min(self.rescore_topk, N)
                    tops = [torch.topk(er[w], k).indices.to(torch.int32) for w in range(Eg)]
                    counts = [k] * Eg
                pick = torch.cat([tops[w][: counts[w]].long() for w in range(Eg)])
                wsel = torch.cat([torch.full((counts[w],), w, dtype=torch.int32, device=self.device) for w in range(Eg)])
                f32 = self.handle.score_actions(smode, s, a, r, acts[wsel.long(), pick], wsel, h, lmbda, float(cfg.discount))
                er[wsel.long(), pick] = f32
            for j, i in enumerate(ids):
                expo = torch.empty((N,), dtype=torch.float32, device=self.device).exponential_(1, generator=self.generator)
                p, ev, am, si, sa = self.handle.select(er[j], acts[j, :, 0], float(cfg.temperature), expo)
                out[i] = ev if eval else sa[0]
                info[i] = dict(expect_return=er[j], argmax=am, sample_idx=si, eval_action=ev, sample_action=sa, horizon=h,
                               n_rescored=None if stats_h is None else int(stats_h[j, 0]),
                               min_margin_outside=None if stats_h is None else float(stats_h[j, 1]))
        self.last = dict(windows=info, delta=self._delta)
        return out

    def _window_host(self, sequence_history, rtg, percentage, buf):
        """Host half of ``assemble_window`` into ``buf`` (T, S+A+1); returns (horizon, rtg)."""
        T = self.T
        horizon = int(self.cfg.horizon)
        end_idx = int(sequence_history["path_length"])
        if end_idx + horizon < T:
            horizon = T - end_idx
        hl = T - horizon + 1
        buf[:] = 0.0
        lo, hi = end_idx - hl + 1, end_idx + 1
        buf[:hl, : self.S] = sequence_history["observations"][lo:hi]
        buf[:hl, self.S : self.S + self.A] = sequence_history["actions"][lo:hi]
        buf[:hl, self.S + self.A :] = np.asarray(sequence_history["rewards"][lo:hi]).reshape(hl, 1)
        if rtg is not None:
            return horizon, float(rtg)
        st = self.tokenizer_manager.tokenizers["returns"].stats
        return horizon, float(np.asarray(st.min + (st.max - st.min) * percentage).reshape(-1)[0])

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
        s, a, r, rtg = self._split(trajectory)
        cfg = self.cfg
        N, T, A = int(cfg.action_samples), self.T, self.A
        mode = capi.MODE_CRITIC if cfg.plan_guidance == "critic_lambda_guiding" else capi.MODE_RTG
        lmbda = 0.6 if mode == capi.MODE_RTG else float(cfg.lmbda)
        toks = [self.handle.tokenize(capi.STATES, s[None]), a[None], None,
                self.handle.tokenize(capi.RETURNS, torch.full((1, T, 1), rtg, dtype=torch.float64, device=self.device))]
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
        self.last = dict(cem=trace, candidates=cand)
        return cand[0, 0][None], mean[0]

    @torch.no_grad()
    def action_sample(self, sequence_history, percentage=1.0, horizon=4, plan=True, eval=False, rtg=None):
        """learner.py:329-417 (the ``horizon`` argument is ignored there too: cfg.horizon rules)."""
        if eval:
            assert rtg is not None
        states, actions, rewards, h, return_to_go = self.assemble_window(sequence_history, rtg, percentage)
        traj = {"states": states[None], "actions": actions[None], "rewards": rewards[None], "_rtg": return_to_go}
        if plan:
            guidance = self.cfg.plan_guidance
            assert guidance in _MODES, guidance
            if guidance == "rtg_guiding":
                sample_action, eval_action = self.rtg_guiding(traj, h)  # default lmbda=0.6, learner.py:405-407
            else:
                sample_action, eval_action = getattr(self, guidance)(traj, h, lmbda=self.cfg.lmbda)
        else:
            sample_action, eval_action = self.mtm_sampling(traj, h)
        return eval_action if eval else sample_action


# ------------------------------------------------------------------------------------------------------
def _param_version(module) -> int:
    return sum(int(p._version) for p in module.parameters())


def attach(learner, precision: str = "fp32", rescore_topk: int = 16, group=None):
    """Rebind the plan path of a reference-style ``Learner`` onto the HIP library.

    Reads: learner.cfg, learner.mtm (state_dict + config), learner.tokenizer_manager.tokenizers[k]
    (._data_mean, ._data_std, .normalize, .stats), learner.iql.qf (state_dict, obs_mean, obs_std).
    Afterwards learner.action_sample / rtg_guiding / critic_lambda_guiding / noise_adding_lambda /
    mtm_sampling run on the GPU; everything else on the object is untouched."""
    mtm = learner.mtm
    mc = mtm.config
    toks = {}
    for k in KEYS:
        t = learner.tokenizer_manager.tokenizers[k]
        st = t.stats
        toks[k] = ContinuousTokenizer(t._data_mean.detach().cpu().numpy(), t._data_std.detach().cpu().numpy(),
                                      DataStatistics(st.mean, st.std, st.min, st.max), normalize=bool(t.normalize))
    qf = getattr(getattr(learner, "iql", None), "qf", None)
    planner = HipPlanner(learner.cfg, mtm.state_dict(), TokenizerManager(toks),
                         q_state_dict=None if qf is None else qf.state_dict(),
                         obs_mean=None if qf is None else qf.obs_mean, obs_std=None if qf is None else qf.obs_std,
                         n_embd=mc.n_embd, n_head=mc.n_head, n_enc_layer=mc.n_enc_layer, n_dec_layer=mc.n_dec_layer,
                         precision=precision, rescore_topk=rescore_topk, group=group)
    state = {"mtm": _param_version(mtm), "qf": None if qf is None else _param_version(qf)}

    def _sync():
        v = _param_version(mtm)
        if v != state["mtm"]:
            planner.load_state_dict(mtm.state_dict())
            state["mtm"] = v
        if qf is not None:
            vq = _param_version(qf)
            if vq != state["qf"]:
                planner.load_critic(qf.state_dict(), qf.obs_mean, qf.obs_std)
                state["qf"] = vq

    def _wrap(name):
        fn = getattr(planner, name)

        def method(self, *a, **kw):
            _sync()
            return fn(*a, **kw)

        method.__name__ = name
        return types.MethodType(method, learner)

    for name in ("action_sample", "rtg_guiding", "critic_lambda_guiding", "noise_adding_lambda", "mtm_sampling",
                 "action_piid_sample", "action_id_sample"):
        setattr(learner, name, _wrap(name))
    learner._hip_planner = planner
    return planner
