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

Pipelining (``plan_async`` / ``PlanTicket.result``, ``action_sample_batch``, ``rollout.PipelinedPlanner``): the reference
plans one window per call and reads the action back before the next one (replay_buffer.py:204-232, learner.py:645-741).
For INDEPENDENT windows (several environments, evaluation episodes) the three parts of a plan step need not wait for each
other across steps: the candidate passes run back to back on the caller's stream, while the latency-bound fp32 chains --
the policy passes and the re-scores + selects of the neighbouring steps -- run on two more ("chain") streams in the library's
chain workspaces (``chain_mode``: a step's two chains on the stream of its slot's parity).  Every step owns a slot (policy head,
returns tokens, re-score scratch); its results are bit-identical to the serial order.

This file is the step pipeline (``HipPlanner``, ``PlanTicket``, ``attach``); the certified re-score's protocol lives in
``certificate.py``, the zero-shot calls and CEM in ``goal.py``, environments that step together in ``lockstep.py``.
"""
from __future__ import annotations

import types
from typing import Dict, Optional

import contextlib
import warnings

import numpy as np

import torch

from . import capi, dist as mdist
from .certificate import RACE_MAX, _TicketOps, resolve as _resolve_certificate
from .goal import GoalMixin
from .lockstep import LockstepMixin
from .tokenizers import ContinuousTokenizer, DataStatistics, SquashedNormal, TokenizerManager

KEYS = capi.KEYS
_MODES = {"rtg_guiding": capi.MODE_RTG, "critic_lambda_guiding": capi.MODE_CRITIC,
          "noise_adding_lambda": capi.MODE_NOISE}


_CHAIN_STREAMS = {}  # (device, priority) -> (policy-pass stream, re-score + select stream), shared by all planners


def _cfg_get(cfg, name, default=None):
    return getattr(cfg, name, default)


class _Slot:
    """Host side of one step slot of the handle (capi.SLOTS of them): the re-score scratch of the step that owns it, the
    host-mapped statistics buffers, the staging buffer of its window and the events that order its parts across streams."""

    def __init__(self, i: int, device):
        self.i = i
        self.owner = None
        self.device = device
        self.b_top = self.f_top = self.b_lst = self.f_lst = self.stats = self.mstats = None
        self.hs_win = self.hs_mrg = None
        self.win = self.win_np = self.win_dev = self.eps_buf = self.expo_buf = None
        self.ev_in = self.ev_pol = self.ev_cand = self.ev_done = self.ev_h2d = None

    def ready(self, planner):
        if self.b_top is None:
            dev = self.device
            # the re-score lists of the step (m3pc_topk_race_window): R race entries in front of the score entries
            R = planner._R
            self.b_lst = torch.empty((R + 1024,), dtype=torch.float32, device=dev)  # their bf16 scores
            self.f_lst = torch.empty((R + 1024,), dtype=torch.float32, device=dev)  # their fp32 re-scores
            self.b_top, self.f_top = self.b_lst[R:], self.f_lst[R:]
            self.stats = torch.empty((4,), dtype=torch.float32, device=dev)
            self.mstats = torch.empty((8,), dtype=torch.float32, device=dev)
            self.hs_win, self.hs_mrg = capi.HostStats(), capi.HostStats()
            self.win = torch.zeros((planner.T * (planner.S + planner.A + 1),), dtype=torch.float32).pin_memory()
            self.win_np = self.win.numpy()
            # Device buffers a pipelined step fills on the policy stream and reads on the other streams (window, variates): owned
            # by the SLOT, not allocated per step -- the caching allocator recycles a block within the stream it was allocated on
            # without knowing about the other streams' readers, so a per-step tensor freed when its ticket goes could be
            # overwritten by the next step's draw while this step's select still reads it.  A slot's buffers are re-filled only
            # behind ev_done of the slot's previous owner.
            n = int(planner.cfg.action_samples)
            self.win_dev = torch.empty((planner.T * (planner.S + planner.A + 1),), dtype=torch.float32, device=dev)
            self.eps_buf = torch.empty((n * planner.T * planner.A,), dtype=torch.float32, device=dev)
            self.expo_buf = torch.empty((n,), dtype=torch.float32, device=dev)
            self.ev_in, self.ev_pol, self.ev_cand, self.ev_done, self.ev_h2d = (torch.cuda.Event() for _ in range(5))
        return self


class PlanTicket:
    """One plan step in flight (``HipPlanner.plan_async``).  ``result()`` -> what ``action_sample`` returns for the window:
    the eval action (A,) when the step was issued with eval=True, else the sampled action (1, A); ``pair()`` -> both;
    ``info`` (after the result): the step's ``planner.last`` record."""

    def __init__(self, planner, slot, mode, states, actions, rewards, rtg, h, lmbda, returns):
        self.planner, self.slot, self.mode = planner, slot, mode
        self.states, self.actions, self.rewards, self.rtg, self.h, self.lmbda, self.returns = states, actions, rewards, rtg, h, lmbda, returns
        self.chain = self.tchain = None
        self.eps = self.expo = self.res = self.er_b = self.er = self.a0 = self.sel = self.top = None
        self.tail_enqueued = False
        self.deferred = False
        self.kmin = self.kmax = self.n_done = self.r_done = self.index = 0
        self.grow_in, self.kfirst_in, self.rfirst_in = 0.0, 0, 0
        self.lst, self.R, self.wset = None, 0, None
        self.seq_mrg = 0.0
        self.delta = None
        self.seq_win = 0.0
        self.shift = self.deviation = None
        self.out = None
        self.info = None
        self.eval = None
        self.keep = self.keep_window = self.outbuf = None

    def pair(self):
        """(sample_action (1, A), eval_action (A,))"""
        return self.planner._finish(self)

    def result(self):
        sa, ev = self.planner._finish(self)
        if self.eval is None:
            return sa, ev
        return ev if self.eval else sa


class HipPlanner(GoalMixin, LockstepMixin):
    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], tokenizer_manager, q_state_dict=None,
                 obs_mean=None, obs_std=None, n_embd: int = 512, n_head: int = 4, n_enc_layer: int = 2,
                 n_dec_layer: int = 1, precision: str = "fp32", rescore_topk: int = 16, device: Optional[int] = None,
                 group=None, generator: Optional[torch.Generator] = None, max_batch: int = 1,
                 rescore: str = "bound", rescore_min: int = 8, rescore_max: int = 128, rescore_delta: Optional[float] = None,
                 max_windows: int = 1, pipeline_depth: int = 3, chain_priority: int = -1, tail_stream: bool = True,
                 defer_join: bool = True, goal_batch: int = 0, race_min: int = 2, calibration_windows: Optional[int] = None, certify_sample: bool = True,
                 chain_mode: str = "alternate", policy_head: str = "full", auto_fp32: bool = True,
                 calibration_factor: float = 1.6, rescore_round: int = 4):
        """cfg: any object with traj_length, action_samples, horizon, discount, temperature, lmbda,
        plan_guidance (finetune.py RunConfig fields read at learner.py:276,319,342).
        tokenizer_manager: a TokenizerManager (this package's) or {key: {"mean","std","min","max"}}.
        precision: "fp32" (reference-accurate) or "bf16" (bf16 MFMA candidate pass followed by an fp32 re-score of the
        candidates that can still be the arg-max, so that the arg-max does not depend on bf16 rounding):
          rescore="bound" (default): a CERTIFIED re-score (m3pc_amd/certificate.py).  Model: bf16 score b_j = f_j + c + e_j with
            a common shift c and a deviation |e_j| <= delta.  Two lists are re-scored in fp32 -- the best candidates by bf16
            score (the arg-max / eval action) and the best by race key tau b_j - log q_j (the multinomial draw / sampled action,
            ``certify_sample``) -- and the library counts who can still beat the best re-scored fp32 score (``need``) or win the
            draw (``need_race``); a certificate that asks for more gets a second pass (at most ``rescore_max`` / 32 through the
            lists; beyond that the whole set goes through a slow path with a warning, ``last["saturated"]``).  delta is
            calibrated per weight load on FULL fp32 candidate passes over the first steps (``calibration_windows``, default 16,
            8 from N = 2048 on; ``calibration_factor`` = 1.6 x the largest deviation from the median over all of them), checked on every step's
            re-scored set and raised when 1.5 x what that step saw is more (``delta_grown``), or fixed by ``rescore_delta``.
            ``planner.last`` reports n_rescored / n_race (score / race entries re-scored), n_in_window / need_race (what the
            first certificates asked for), min_margin_outside, shift, deviation and delta.  One 32-byte device-to-host read per
            step, after the select has been enqueued.
          rescore="topk": the fixed ``rescore_topk`` best candidates (round-1 behaviour, no certificate, no host read).
          Either way the select runs on the MERGED vector: fp32 scores for the re-scored candidates, bf16 scores minus the
          estimated shift for the rest, so an un-re-scored candidate cannot win the arg-max through a constant bf16 offset.
        auto_fp32 (bf16 + rescore="bound"): the certificate keeps a bf16 step's arg-max and draw the fp32 ones whatever the weights
          are, but its PRICE depends on them -- where the bf16 deviation delta is not small against the spread of the scores the
          certificate asks for most of the candidates in fp32 (tests/test_certificate_gpu.py, "trained-like" weights: every Linear
          x 2 already re-scores a third of them) and the step is slower than a plain fp32 step.  With auto_fp32 the planner watches
          for that -- half of the candidates or more re-scored in fp32, on average over the last (up to 16, at least 4) steps -- and
          plans in fp32 from then on (one warning; ``planner.fp32_fallback``), until the next weight load gives bf16 another try.
        pipeline_depth: how many plan steps ``action_sample_batch`` / ``rollout`` keep in flight (<= capi.SLOTS - 1).
        goal_batch: the largest number of zero-shot windows one ``action_piid_sample_batch`` / ``goal_actions`` call plans
        through the pruned many-window path (m3pc_goal_step_batch; BASELINE config 5: 8192 per GPU); 0 = that path is off."""
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
        nw = max(int(max_windows), 1)
        # chain workspace: the fp32 re-score sets (all windows of a lock-step batch together): a first pass is at most rescore_max
        # score entries + RACE_MAX race entries
        # race entries a step lists (m3pc_topk_race_window); certify_sample=False: none -- the arg-max (hence eval_action) stays
        # certified, the multinomial index is then a near-winner of the reference's draw only (round-4 behaviour, A/B switch)
        self._R = max(min(RACE_MAX, N), 1) if certify_sample else 0
        max_rescore = max((int(rescore_max) + self._R) * nw, int(rescore_topk), 1) if precision == "bf16" else 1
        self._max_batch = max(int(max_batch), nw, 1)
        self.handle = capi.Handle(S, A, T, n_embd, n_head, n_enc_layer, n_dec_layer,
                                  max_candidates=max(n_local * nw, 1), max_batch=self._max_batch,
                                  critic_hidden=hidden, device=device, max_rescore=max_rescore, max_goal_batch=int(goal_batch))
        self._goal_batch = int(goal_batch)
        self.device = self.handle.device
        self.S, self.A, self.T = S, A, T
        self.precision = {"fp32": capi.PREC_FP32, "bf16": capi.PREC_BF16}[precision]
        self.rescore_topk = int(rescore_topk) if self.precision == capi.PREC_BF16 else 0
        assert rescore in ("bound", "topk")
        self.rescore = rescore if self.precision == capi.PREC_BF16 else "none"
        self.rescore_min, self.rescore_max = int(rescore_min), int(rescore_max)
        self.race_min = max(1, int(race_min))  # race entries of a first pass (the winner of the bf16 race + one runner-up)
        self._auto_fp32 = bool(auto_fp32) and self.precision == capi.PREC_BF16 and self.rescore == "bound"
        self._configured = (self.precision, self.rescore)  # what the caller asked for (the fp32 fallback returns to it)
        self._sat_recent: list = []  # saturated flags of the last 16 resolved steps
        self._want_fp32 = False      # set by _finish, acted on by the next _issue (steps in flight are drained first)
        self.fp32_fallback = False
        self._delta_fixed = None if rescore_delta is None else float(rescore_delta)
        # Adaptive state of the certified re-score.  It must not depend on how many steps are in flight (a pipelined run has
        # to reproduce the serial one bit for bit), so it is LAGGED: step t uses what the steps up to t - capi.SLOTS saw --
        # exactly the steps that are certain to be resolved when step t is issued (its slot's previous owner is step
        # t - SLOTS) -- whatever has been resolved since.  _delta0: the calibrated bound; _hist[t] = (deviation, need) of step t.
        self._delta0: Optional[float] = self._delta_fixed
        # full-pass calibrations behind a weight load: 16 windows, 8 from N = 2048 on (one fp32 candidate pass each, ~10 ms at
        # N = 1024: ~0.16 s per weight load).  Round 5 took enough windows for ~4096 candidates (4 at N = 1024): fine on the init
        # recipe (0 of 3216 long-sweep trials above the bound, maximum 0.994), but on the "trained-like" weight families of round 6
        # some candidate exceeded the bound in 14 of 1248 trials, by up to 1.28 x -- a window's deviation SCALE depends on its
        # history, and four windows do not see that.  Raising the factor to 2.1 left 3 of 1248 (<= 1.04) and cost every step
        # (bench -2 %, the shipped N = 625 config +31 % per pipelined step: delta 6.8 -> 8.7, 18 -> 27 candidates re-scored); 16
        # windows at the old factor leave 1 of 1248 (1.003) and cost only the weight load (profiles/r06_certificate_sweep*,
        # r06_ab_calibration_factor.txt, r06_ab_shipped_calibration.txt)
        self._cal_windows = max(1, int(calibration_windows)) if calibration_windows is not None else max(8, min(16, -(-16384 // max(N, 1))))
        self._cal_left = self._cal_windows  # full-pass calibrations still to run behind the last weight load
        # delta = calibration_factor x the largest deviation the calibration passes saw (see _cal_windows above for what round 6
        # measured: more WINDOWS, not a larger factor, is what the "trained-like" weight families needed)
        self.calibration_factor = float(calibration_factor)
        self.rescore_round = max(1, int(rescore_round))  # a first pass's size is rounded up to a multiple of this (_adapt)
        self._hist: Dict[int, tuple] = {}
        self._step_index = 0
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
        self._host = np.zeros((T * (S + A + 1),), dtype=np.float32)  # [states (T,S) | actions (T,A) | rewards (T,1)] blocks
        self.last: Dict[str, torch.Tensor] = {}
        self.pipeline_depth = max(1, min(int(pipeline_depth), capi.SLOTS - 1))
        self._slots = [_Slot(i, self.device).ready(self) for i in range(capi.SLOTS)]
        self._next_slot = 0
        self._chain = None          # the policy-pass stream (created on first pipelined use)
        self._tchain = None         # the re-score + select stream (tail_stream=False: the policy-pass stream)
        self._tail_stream = bool(tail_stream)
        self._defer_join = bool(defer_join)  # pipelined steps: the step's tail, not the current stream, joins the candidate parts
        self._chain_priority = int(chain_priority)  # -1: high priority -- its short dependent launches go first when a CU frees up
        self._pending = []          # pipelined tickets whose tail (re-score + select) is not enqueued yet, oldest first
        # How the two chain streams of the process are used by pipelined steps.
        #   "alternate" (default): a step runs BOTH its fp32 chains -- its policy pass and, two issues later, its re-score +
        #     select -- on the stream of its slot's parity, in the library's chain workspaces of that parity: per stream
        #     policy(t+2) | tail(t) | policy(t+4) | tail(t+2) ...; the chains of consecutive steps overlap.
        #   "split" (rounds 3-4): every policy pass on one stream, every tail on the other.  Measured (r5, pipeline_phases.py): a
        #     tail takes ~1.15 ms beside the candidate passes (0.37 ms alone), the tails of consecutive steps queue on their one
        #     stream back to back, and that stream -- not the candidate passes (1.04 ms + gaps) -- set the step rate.
        assert chain_mode in ("alternate", "split")
        self._alternate = chain_mode == "alternate"
        # "full" (default): the policy pass returns every row of the policy head, as omtm.forward does.  "pruned": it computes the head
        # at the h action tokens the candidates are sampled from only (learner.py:285-287 reads [T-h:] of the distribution) through
        # the exactly pruned decoder -- out-proj / FFN / actor head on h rows instead of 4T; last["loc"] / ["std"] are zero below
        # T - h.  Measured (r5, same box, three runs each): one step alone 1.715-1.723 ms pruned against 1.702-1.709 full, pipelined
        # 844-852 against 848-856 plan-steps/s -- the batch-1 chain is bound by its ~30 dependent launches, not by its rows, and the
        # pruned decoder has as many; kept as an option (tests/test_policy_pruned_gpu.py), not the default.
        assert policy_head in ("pruned", "full")
        self._policy_pruned = policy_head == "pruned"
        self._warned_saturated = False
        self.delta_grown = 0        # how often the per-step deviation check raised delta since the last weight load
        self.action_list = []       # zero-shot "piid_allout" (action_piid_list_sample)
        self._force_collective = False  # test hook: run the all-gather even in a world of one
        self._bf16_offset = 0.0     # test hook: a constant added to the bf16 scores before the re-score
        self._stage, self._stage_i = None, 0  # two pinned window staging buffers (+ the event of the copy that read each last)
        self._ev_main = None        # recorded behind the last use of the chain workspaces on the CALLER's stream (_mark_main)

    # ---------------------------------------------------------------------------------------- weights
    def load_state_dict(self, state_dict):
        """All tensors of ``omtm.state_dict()``, or -- after the first load -- any subset of them: the library re-packs only
        what depends on the tensors given (``handle.load_stats()`` tells what that was)."""
        for tk in [sl.owner for sl in getattr(self, "_slots", []) if sl.owner is not None]:
            self._finish(tk)  # steps in flight were issued against the old weights: resolve them first
        self.handle.load_weights(state_dict)
        self._reset_calibration()

    def _reset_calibration(self):
        """The bf16 error bound delta belongs to the weights it was measured on -- the model's AND, under critic guidance, the Q
        networks' (the deviation contains min(q1, q2) evaluated on bf16-decoded states, learner.py:250-252, and fine-tuning
        updates them between rollouts, finetune.py:288-290): the next steps run the full-pass calibrations again."""
        self._delta0 = getattr(self, "_delta_fixed", None)
        self._cal_left = getattr(self, "_cal_windows", 3)
        self._hist = {}
        self.delta_grown = 0
        if getattr(self, "fp32_fallback", False):  # new weights: bf16 gets another try
            self.precision, self.rescore = self._configured
            self.fp32_fallback = False
        self._sat_recent, self._want_fp32 = [], False

    # -- adaptive re-score state ------------------------------------------------------------------------
    @property
    def _delta(self) -> Optional[float]:
        """The bound as the next step would see it if everything resolved so far counted (reporting; lock-step batches)."""
        if self._delta0 is None:
            return None
        if self._delta_fixed is not None:
            return self._delta0
        return max([self._delta0] + [1.5 * v[0] for v in self._hist.values()])

    @_delta.setter
    def _delta(self, value):
        self._delta0 = value
        self._hist = {}
        self._cal_left = 0  # (an explicit bound stands: no further calibration passes until the next weight load)

    def _adapt(self, index: int):
        """(growth of delta, size of the first re-score pass by score, by race key) for step `index`, from the steps up to
        index - SLOTS."""
        seen = [(i, v) for i, v in self._hist.items() if i <= index - capi.SLOTS]
        grow = max([1.5 * v[0] for _, v in seen], default=0.0) if self._delta_fixed is None else 0.0
        # first pass: what the 80th percentile of the recent steps' certificates asked for -- by score and by race key together,
        # in fours (a second pass costs a whole fp32 chain, ~0.3 ms; four more candidates in the first ~0.02-0.05 ms; the few-row
        # fp32 kernels work in 64-row tiles, 8 candidates = 392 encoder / 256 decoder rows fill 7 / 4 of them), at least
        # rescore_min in all: the race entries take their share of that floor instead of adding to it
        recent = sorted(seen)[-16:]
        q80 = lambda vals: sorted(vals)[min(len(vals) - 1, int(0.8 * len(vals)))]
        rfirst = min(max(self.race_min, q80([v[2] for _, v in recent]) if recent else 0), self._R)
        rr = self.rescore_round
        total = max(self.rescore_min, -(-(q80([v[1] for _, v in recent]) + rfirst) // rr) * rr) if recent else self.rescore_min
        return grow, max(total - rfirst, 1), rfirst

    def load_critic(self, q_state_dict, obs_mean, obs_std):
        """TwinQ weights + observation statistics (model.py:157-161).  Steps in flight were issued against the old Q networks:
        they are resolved first; the calibrated bound of the certified re-score is reset like after a model weight load."""
        for tk in [sl.owner for sl in getattr(self, "_slots", []) if sl.owner is not None]:
            self._finish(tk)
        self.handle.set_critic(q_state_dict, obs_mean, obs_std)
        self._reset_calibration()

    # ---------------------------------------------------------------------------------------- window
    def _blocks(self, flat):
        """(states (T,S), actions (T,A), rewards (T,1)) views of a flat [states | actions | rewards] window buffer."""
        T, S, A = self.T, self.S, self.A
        return flat[: T * S].reshape(T, S), flat[T * S : T * (S + A)].reshape(T, A), flat[T * (S + A) :].reshape(T, 1)

    def _window_host(self, sequence_history, rtg, percentage, flat):
        """Host half of ``assemble_window`` (learner.py:342-385) into the flat window buffer; returns (horizon, rtg)."""
        T = self.T
        horizon = int(self.cfg.horizon)
        end_idx = int(sequence_history["path_length"])
        if end_idx + horizon < T:
            horizon = T - end_idx
        hl = T - horizon + 1
        flat[:] = 0.0
        bs, ba, br = self._blocks(flat)
        lo, hi = end_idx - hl + 1, end_idx + 1
        bs[:hl] = sequence_history["observations"][lo:hi]
        ba[:hl] = sequence_history["actions"][lo:hi]
        br[:hl] = np.asarray(sequence_history["rewards"][lo:hi]).reshape(hl, 1)
        return horizon, self._rtg_value(rtg, percentage)

    def _rtg_value(self, rtg, percentage):
        if rtg is not None:
            return float(rtg)
        st = self.tokenizer_manager.tokenizers["returns"].stats
        return float(np.asarray(st.min + (st.max - st.min) * percentage).reshape(-1)[0])

    def _stage_copy(self, fill):
        """One packed H2D copy of a window through one of two pinned staging buffers (asynchronous: the host goes on enqueuing
        while the copy runs; a buffer is re-used only after the copy that read it last has completed).  fill(flat) -> meta."""
        if self._stage is None:
            n = self.T * (self.S + self.A + 1)
            self._stage = [(torch.zeros((n,), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
        pin, ev = self._stage[self._stage_i]
        self._stage_i ^= 1
        ev.synchronize()
        meta = fill(pin.numpy())
        dev = pin.to(self.device, non_blocking=True)
        ev.record()
        return dev, meta

    def assemble_window(self, sequence_history, rtg=None, percentage=1.0):
        """learner.py:342-385.  Returns (states (T,S), actions (T,A), rewards (T,1) on device, horizon, rtg)."""
        dev, (horizon, return_to_go) = self._stage_copy(lambda flat: self._window_host(sequence_history, rtg, percentage, flat))
        states, actions, rewards = self._blocks(dev)  # one packed H2D copy; the three blocks are contiguous views
        return states, actions, rewards, horizon, return_to_go

    # ---------------------------------------------------------------------------------------- guidance
    def _eps(self, shape):
        return torch.randn(shape, device=self.device, dtype=torch.float32, generator=self.generator)

    def _draw_eps(self, mode, h, buf=None):
        """The candidate variates, in the shapes the reference draws: dist.sample((N,)) over loc (1,T,1,A) (learner.py:285) /
        randn((N,h,A)) for the fixed-variance variant (learner.py:157-163); identical on every rank.  buf: a flat device
        buffer to draw into (same variates as a fresh tensor of the shape)."""
        N, T, A = int(self.cfg.action_samples), self.T, self.A
        shape = (N, h, A) if mode == capi.MODE_NOISE else (N, 1, T, 1, A)
        if buf is None or type(self)._eps is not HipPlanner._eps or "_eps" in self.__dict__:  # (tests substitute _eps)
            return self._eps(shape)
        return buf[: int(np.prod(shape))].view(shape).normal_(generator=self.generator)

    def _draw_expo(self):
        """The multinomial's exponentials (ATen's own algorithm: argmax(p / q), q ~ Exp(1))."""
        return torch.empty((int(self.cfg.action_samples),), dtype=torch.float32, device=self.device).exponential_(1, generator=self.generator)

    def _chain_streams(self):
        """(stream 0, stream 1) of this device and priority, shared by every planner of the process."""
        if self._chain is None:
            # one pair of streams per (device, priority) for every planner of the process: the device gives a process four
            # hardware queues (current stream, the library's stream for the second candidate half, these two) -- with more
            # streams than queues two of them share a queue and no longer overlap
            key = (str(self.device), self._chain_priority)
            if key not in _CHAIN_STREAMS:
                _CHAIN_STREAMS[key] = (torch.cuda.Stream(device=self.device, priority=self._chain_priority),
                                       torch.cuda.Stream(device=self.device, priority=self._chain_priority))
            self._chain, t = _CHAIN_STREAMS[key]
            self._tchain = t if self._tail_stream else self._chain
        return self._chain, self._tchain

    def _streams_of(self, sl):
        """(policy-pass stream, tail stream) of the pipelined step that owns slot ``sl`` (see ``chain_mode``)."""
        a, b = self._chain_streams()
        if self._alternate and self._tail_stream:
            c = (a, b)[sl.i & 1]
            return c, c
        return a, b

    def _acquire_slot(self):
        sl = self._slots[self._next_slot]
        self._next_slot = (self._next_slot + 1) % len(self._slots)
        if sl.owner is not None:  # a step still lives in this slot: finish it first (its buffers are about to be reused)
            self._finish(sl.owner)
        return sl

    def _drain(self):
        """Resolve every pipelined step still in flight (the current stream is then ordered behind the chain stream's work:
        what follows may use the chain workspace on the current stream)."""
        for tk in [sl.owner for sl in self._slots if sl.owner is not None and sl.owner.chain is not None]:
            self._finish(tk)

    def _mark_main(self):
        """The current stream has just used the library's chain workspaces (a serial plan step's fp32 chains, a generic
        forward, the zero-shot calls): the chain streams of the pipelined steps that follow must not start in them before it
        is through.  (Found by the soak test: in fp32 mode nothing makes the host wait per step, and the policy pass of the
        next plan_async on the chain stream overwrote the workspace of a serial step still running on the caller's stream.)"""
        if self._ev_main is None:
            self._ev_main = torch.cuda.Event()
        self._ev_main.record(torch.cuda.current_stream(self.device))

    def _guide(self, mode: int, states, actions, rewards, rtg: float, h: int, lmbda: float, eps=None, returns=None):
        """One plan step, serial: everything on the current stream, results when the call returns (device-resident)."""
        return self._issue(mode, states, actions, rewards, rtg, h, lmbda, eps=eps, returns=returns, pipelined=False).result()

    def _issue(self, mode: int, states, actions, rewards, rtg: float, h: int, lmbda: float, eps=None, returns=None,
               pipelined: bool = True, slot=None, inputs_ready: bool = False) -> "PlanTicket":
        """Enqueue one plan step and return its ticket.
        serial (pipelined=False): policy pass, candidate pass, re-score + select on the current stream.
        pipelined: the policy pass goes to the chain stream at once, the candidate pass to the current stream behind it, and
        the step's tail (re-score + select, chain stream) is enqueued with the NEXT step -- behind that step's policy pass --
        or when the ticket is resolved: the chain stream then runs policy(t+1) while the current stream still runs
        candidates(t), and tail(t) while it runs candidates(t+1).
        inputs_ready (pipelined): the window tensors are complete already (written before earlier work of the current stream was
        enqueued, or on the chain stream as ``plan_async`` does); otherwise the chain stream first waits for the current
        stream -- i.e. for the previous step's candidate pass, which costs the overlap of the policy pass."""
        cfg = self.cfg
        N, T, A = int(cfg.action_samples), self.T, self.A
        if self._auto_fp32 and not self.fp32_fallback:
            self._maybe_fall_back_to_fp32()
        if not pipelined:
            self._drain()  # a serial step runs its fp32 chains on the current stream: nothing pipelined may still be using them
        sl = (slot if slot is not None else self._acquire_slot()).ready(self)
        tk = PlanTicket(self, sl, mode, states, actions, rewards, float(rtg), int(h), float(lmbda), returns)
        sl.owner = tk
        tk.index, self._step_index = self._step_index, self._step_index + 1
        tk.grow_in, tk.kfirst_in, tk.rfirst_in = self._adapt(tk.index)
        main = torch.cuda.current_stream(self.device)
        chain, tchain = self._streams_of(sl) if pipelined else (None, None)
        tk.chain = chain
        tk.tchain = tchain
        hd = self.handle
        # what the caller gets back (and the re-score's merged vector / candidate list) lives in the CURRENT stream's memory
        # pool: the caller consumes it there, so that is where its blocks must be recycled
        tk.outbuf = hd.select_buffers(N)
        if self.rescore != "none":
            tk.er = torch.empty((N,), dtype=torch.float32, device=self.device)
            tk.R = self._R
            tk.lst = torch.empty((tk.R + 1024,), dtype=torch.int32, device=self.device)  # race entries | score entries
            tk.top = tk.lst[tk.R :]
        with (torch.cuda.stream(chain) if chain is not None else contextlib.nullcontext()):
            if chain is not None and not inputs_ready:
                sl.ev_in.record(main)
                chain.wait_event(sl.ev_in)
            # the variates of the step in the serial order of draws: eps, then the multinomial's exponentials.  (Pipelined:
            # drawn on the chain stream -- the generator's state advances on the host in issue order either way -- so that
            # nothing but the candidate pass sits on the current stream.)
            if chain is not None:
                chain.wait_event(sl.ev_done)  # (see _Slot.ready: the slot's buffers are free once its previous owner is done)
                if self._ev_main is not None:
                    chain.wait_event(self._ev_main)  # (_mark_main: the caller's stream was in the policy workspace)
            # the policy pass consumes no variate: its ~30 launches go out first, the draws are enqueued while the device runs them
            hd.policy_pass(mode, states, actions, rewards, h, tk.rtg, slot=sl.i, returns=returns, pruned=self._policy_pruned)
            if eps is None:
                eps = self._draw_eps(mode, h, sl.eps_buf if chain is not None else None)
            tk.eps = eps = eps.reshape(N, -1, A)
            tk.expo = sl.expo_buf.exponential_(1, generator=self.generator) if chain is not None else self._draw_expo()
            if chain is not None:
                sl.ev_pol.record(chain)
        if chain is not None and self._alternate:
            # alternate mode: this stream's order is policy(t) | tail(t-2) | policy(t+2) | tail(t): the tail of the step two
            # issues back (same parity, its candidate pass is enqueued long since) goes behind this step's policy pass, so that
            # the policy pass -- which the candidate pass of THIS step waits for -- is not held up by it
            for old in [o for o in self._pending if o.tchain is chain]:
                self._pending.remove(old)
                self._enqueue_tail(old)
        if chain is not None:
            main.wait_event(sl.ev_pol)
        begin, count = mdist.shard_range(N, self.rank, self.world)
        # pipelined, one rank: the current stream does not wait for the candidate parts that run on the library's own streams
        # -- the step's tail joins them on its stream (_enqueue_tail) -- so the next step's first part starts right behind this
        # step's first part while the last part still runs.  (A sharded run gathers the scores on the current stream: joined.)
        tk.deferred = bool(chain is not None and self.world == 1 and not self._force_collective and not self._bf16_offset
                           and self._defer_join)
        res = hd.candidate_pass(mode, states, actions, rewards, eps, h, tk.lmbda, float(cfg.discount), N, begin, count,
                                precision=self.precision, slot=sl.i, defer_join=tk.deferred)
        er, a0 = res["expect_return"], res["sample_actions"][:, 0]
        er, a0 = mdist.gather_candidates(er, a0, N, self.group, force=self._force_collective)
        if self._bf16_offset and self.precision == capi.PREC_BF16:
            er = er + self._bf16_offset
        tk.res, tk.er_b, tk.a0 = res, er, a0
        if chain is not None:
            sl.ev_cand.record(main)
            if not self._alternate:  # split mode: step t's tail goes out with step t+1, behind that step's policy pass
                for old in self._pending:
                    self._enqueue_tail(old)
                self._pending = []
            self._pending.append(tk)
        else:
            self._enqueue_tail(tk)
            self._mark_main()
        return tk

    def _maybe_fall_back_to_fp32(self):
        """auto_fp32: the certified bf16 step costs more than an fp32 step once its certificates keep asking for the whole window
        set (delta not small against the score spread: a property of the weights).  Decided on the steps up to index - SLOTS, like
        every adaptive quantity (_adapt): the same decision at the same step at any pipeline depth."""
        seen = [v[3] for _, v in sorted((i, v) for i, v in self._hist.items() if i <= self._step_index - capi.SLOTS)[-16:] if len(v) > 3]
        N = int(self.cfg.action_samples)
        if len(seen) < 4 or sum(seen) < 0.5 * N * len(seen):  # (re-scoring k of N candidates costs ~k / N of an fp32 step on top of the bf16 one)
            return
        for tk in [sl.owner for sl in self._slots if sl.owner is not None]:
            self._finish(tk)  # (issued as bf16 steps: resolved as such)
        warnings.warn(f"m3pc_amd: the bf16 certificate re-scored {sum(seen) / len(seen):.0f} of {N} candidates in fp32 on average over "
                      f"the last {len(seen)} plan steps (delta = {self._delta:.3g}): planning in fp32 until the next weight load "
                      f"(HipPlanner(auto_fp32=False) keeps bf16)")
        self.precision, self.rescore, self.fp32_fallback = capi.PREC_FP32, "none", True

    def _rescore_args(self, tk):
        return (tk.mode, tk.states, tk.actions, tk.rewards, tk.eps), (tk.h, tk.rtg, tk.lmbda, float(self.cfg.discount))

    def _on(self, tk):
        """Context of the stream the step's tail runs on."""
        return torch.cuda.stream(tk.tchain) if tk.tchain is not None else contextlib.nullcontext()

    def _enqueue_tail(self, tk):
        """Re-score + select of a step (replicated on every rank: identical inputs => identical result)."""
        cfg, hd, sl = self.cfg, self.handle, tk.slot
        N = tk.er_b.numel()
        tk.tail_enqueued = True
        if tk.tchain is not None:
            tk.tchain.wait_event(sl.ev_cand)  # (the candidate pass waited for the policy pass: ordered behind both)
            if self._ev_main is not None:
                tk.tchain.wait_event(self._ev_main)  # (_mark_main: the caller's stream was in the re-score workspace)
        with self._on(tk):
            if tk.deferred:
                hd.candidate_join(sl.i)
            if self.rescore == "none":
                tk.er = tk.er_b
                tk.sel = hd.select(tk.er, tk.a0, float(cfg.temperature), tk.expo, out=tk.outbuf)
            else:
                rs, tail = self._rescore_args(tk)
                if self.rescore == "bound":
                    if self._delta_fixed is None and self._cal_left > 0:
                        self._cal_left -= 1
                        d = self._calibrate(tk)
                        self._delta0 = d if self._delta0 is None else max(self._delta0, d)
                    # (the merge kernels list r + n <= 1024 entries: a larger rescore_max takes the window-set path instead)
                    kmax = max(min(self.rescore_max, N - 1 if N > 1 else 1, 1024 - self._R - 1), 1)
                    kmin = max(min(tk.kfirst_in, N, kmax), 1)
                    tk.delta = max(self._delta0, tk.grow_in)
                else:
                    kmax = kmin = max(min(self.rescore_topk, N, hd.max_rescore), 1)
                    tk.delta = 0.0
                tk.kmin, tk.kmax = kmin, kmax
                # The kmax + 1 best candidates by bf16 score, best first, and (bound mode) the R best by race key.  The sets that
                # have to be re-scored are prefixes of these lists whose lengths only the re-score itself can tell
                # (m3pc_rescore_merge_race's certificates), so the first kmin / rfirst entries are re-scored, merged and the
                # select is enqueued BEFORE anybody reads anything: the device never waits for the host.  _finish reads the
                # certificates (host-mapped statistics, no stream synchronisation) and only when they ask for more the rest is
                # re-scored and merge + select are repeated on the same variates.
                R = tk.R
                if self.rescore == "bound" and R > 0:
                    rfirst = max(min(tk.rfirst_in, R), 1)
                    hd.topk_race_window(tk.er_b, tk.expo, float(cfg.temperature), kmax, kmin, R, lst=tk.lst, list_scores=sl.b_lst,
                                        want_stats=False)  # (the certificates come from the merge: no window statistics launch)
                else:
                    rfirst = 0
                    hd.topk_window(tk.er_b, kmax, kmin, 0.0, top=tk.top, stats=sl.stats, top_scores=sl.b_top)
                hd.rescore(*rs, tk.lst[R - rfirst : R + kmin], *tail, N, slot=sl.i, out=sl.f_lst[R - rfirst : R + kmin],
                           want_actions=False)
                tk.n_done, tk.r_done = kmin, rfirst
                self._merge(tk, kmin, rfirst, select=True)
            if tk.tchain is not None:
                sl.ev_done.record(tk.tchain)

    def _merge(self, tk, n, r=0, select=False):
        """Merge + certificates over the r race entries and n score entries re-scored so far (score entries: the step's list
        buffer, or the window-set path's own tensors ``tk.wset`` behind the buffer's race entries).  select: the select on the
        merged vector too (-> tk.sel) -- with the race certificate in the same launch (m3pc_merge_race_select)."""
        sl = tk.slot
        tk.seq_mrg = sl.hs_mrg.next_seq()
        o = tk.R - r
        lists = (tk.lst[o:], sl.b_lst[o:], sl.f_lst[o:])
        if tk.wset is not None:
            tk.keep = lists = tuple(torch.cat([a[:r], b]).contiguous() for a, b in zip(lists, tk.wset))
        temp = float(self.cfg.temperature)
        if self.rescore == "bound" and tk.R > 0:
            if select:
                _, _, tk.sel = self.handle.merge_race_select(tk.er_b, tk.expo, temp, lists[0], r, n, lists[1], lists[2], tk.a0,
                                                             delta=tk.delta, merged=tk.er, stats=sl.mstats,
                                                             host_stats=sl.hs_mrg.buf, seq=tk.seq_mrg, out=tk.outbuf)
                return
            self.handle.rescore_merge_race(tk.er_b, tk.expo, temp, lists[0], r, n, lists[1], lists[2],
                                           delta=tk.delta, merged=tk.er, stats=sl.mstats, host_stats=sl.hs_mrg.buf, seq=tk.seq_mrg)
        else:
            self.handle.rescore_merge(tk.er_b, lists[0], n, lists[1], lists[2], delta=tk.delta, merged=tk.er, stats=sl.mstats,
                                      host_stats=sl.hs_mrg.buf, seq=tk.seq_mrg)
        if select:
            tk.sel = self.handle.select(tk.er, tk.a0, temp, tk.expo, out=tk.outbuf)

    def _finish(self, tk):
        """Resolve a ticket: enqueue what is still missing, read the re-score's certificate, finish the re-score if it asks
        for more candidates, and order the current stream behind the step."""
        if tk.out is not None:
            return tk.out
        cfg, hd, sl = self.cfg, self.handle, tk.slot
        if not tk.tail_enqueued:
            self._pending = [o for o in self._pending if o is not tk]
            self._enqueue_tail(tk)
        extra = {}
        top = None
        if self.rescore == "bound":
            N = tk.er_b.numel()
            ops = _TicketOps(self, tk)
            extra = _resolve_certificate(self, N, tk.kmax, tk.R, tk.n_done, tk.r_done, tk.delta, ops)
            tk.delta = extra["delta"]
            extra["n_first"], extra["n_race_first"] = tk.kmin, min(max(tk.rfirst_in, 1), tk.R)
            top = ops.top if ops.top is not None else tk.top[: extra["n_rescored"]]
            extra["n_rescored"] = int(top.numel())
            extra["race"] = tk.lst[tk.R - extra["n_race"] : tk.R]  # the re-scored racers (a view: best bf16 race key LAST)
            # what this step saw feeds the steps from SLOTS later on (_adapt): the bound, and the sizes of the first pass
            self._hist[tk.index] = (float(extra["deviation"]), min(int(extra["n_in_window"]), tk.kmax),
                                    min(int(extra["need_race"]), tk.R), int(extra["n_rescored"]) + int(extra["n_race"]))
            for i in [i for i in self._hist if i < tk.index - 64]:
                # (old enough that every step still to come would count it anyway: fold its deviation into the base bound)
                if self._delta_fixed is None:
                    self._delta0 = max(self._delta0, 1.5 * self._hist[i][0])
                del self._hist[i]
        elif self.rescore == "topk":
            top = tk.top[: tk.kmin]
        if tk.chain is not None:
            torch.cuda.current_stream(self.device).wait_event(sl.ev_done)
        else:
            self._mark_main()  # (a serial step's second re-score passes ran on the caller's stream)
        p, eval_action, argmax, sample_idx, sample_action = tk.sel
        res = tk.res
        self.last = dict(expect_return=tk.er, expect_return_bf16=tk.er_b if self.rescore != "none" else None, p=p, argmax=argmax,
                         sample_idx=sample_idx, loc=res["loc"], std=res["std"], sample_actions=res["sample_actions"], eps=tk.eps,
                         topk=top, eval_action=eval_action, sample_action=sample_action, horizon=tk.h, expo=tk.expo, **extra)
        tk.info = self.last
        tk.out = (sample_action, eval_action)
        if sl.owner is tk:
            sl.owner = None
        return tk.out

    def _rescore_window_set(self, tk, need, everything=False):
        """A certificate asks for more candidates than the step's lists hold: re-score the whole set -- the `need` best
        candidates by bf16 score -- in chunks of the chain workspace (or, beyond the merge kernel's capacity or when the race
        list is exhausted, every candidate in fp32).  Slow path, taken only when the bf16 noise exceeds the score spread.
        Called inside the tail's stream context."""
        cfg, hd, sl = self.cfg, self.handle, tk.slot
        N = tk.er_b.numel()
        rs, tail = self._rescore_args(tk)
        cnt = min(need, N)
        if cnt <= 1024 - RACE_MAX and not everything:
            vals, idx = torch.topk(tk.er_b, cnt)
            idx = idx.to(torch.int32).contiguous()
            f = torch.empty((cnt,), dtype=torch.float32, device=self.device)
            cap = hd.max_rescore
            for c0 in range(0, cnt, cap):
                c1 = min(cnt, c0 + cap)
                hd.rescore(*rs, idx[c0:c1], *tail, N, slot=sl.i, out=f[c0:c1], want_actions=False)
            tk.wset = (idx, vals.contiguous(), f)  # (stands in for the score entries; the race entries stay and may still grow)
            self._merge(tk, cnt, tk.r_done)
            tk.top[:cnt].copy_(idx)
            idx = tk.top[:cnt]
            tk.n_done = cnt
        else:
            # everything: one fp32 candidate pass in the candidate workspace (nothing else may be using it)
            torch.cuda.synchronize(self.device)
            begin, count = mdist.shard_range(N, self.rank, self.world)
            r32 = hd.candidate_pass(tk.mode, tk.states, tk.actions, tk.rewards, tk.eps, tk.h, tk.lmbda, float(cfg.discount), N,
                                    begin, count, precision=capi.PREC_FP32, slot=sl.i)
            er32, _ = mdist.gather_candidates(r32["expect_return"], r32["sample_actions"][:, 0], N, self.group)
            idx = torch.arange(N, dtype=torch.int32, device=self.device)
            tk.n_done = N
            # (a merge of the best entry with itself: shift 0, deviation 0 -- keeps the statistics protocol of _finish alive)
            best = torch.argmax(er32).to(torch.int32).reshape(1)
            bval = er32.max().reshape(1).contiguous()
            tk.keep = (er32, best, bval)
            self.handle.rescore_merge(er32.contiguous(), best, 1, bval, bval, delta=0.0, merged=tk.er, stats=sl.mstats,
                                      host_stats=sl.hs_mrg.buf, seq=self._next_mrg_seq(tk))
            torch.cuda.synchronize(self.device)  # (the candidate workspace is free again before any later step's pass)
        tk.sel = hd.select(tk.er, tk.a0, float(cfg.temperature), tk.expo, out=tk.outbuf)
        if tk.tchain is not None:
            sl.ev_done.record(tk.tchain)
        return idx

    def _next_mrg_seq(self, tk):
        tk.seq_mrg = tk.slot.hs_mrg.next_seq()
        return tk.seq_mrg

    def _calibrate(self, tk) -> float:
        """delta of the certified re-score from ONE FULL fp32 candidate pass over this step's candidates (the same on every
        rank): calibration_factor (1.6) x the largest deviation of (bf16 - fp32) from its median over all N.  Run on each of the first
        ``calibration_windows`` steps behind a weight load (the bound is their maximum) -- per-weight-load setup like the
        weight re-pack, ~10 ms each at N = 1024: the device is synchronised around the pass, which runs in the candidate
        workspace.  (Round 4 calibrated on 256 candidates of one window: a sample maximum of 256 under-estimates the maximum
        over N x every later window, and some candidate exceeded the bound in 1.75 % of 1200 steps.)"""
        cfg, hd = self.cfg, self.handle
        N = tk.er_b.numel()
        torch.cuda.synchronize(self.device)
        begin, count = mdist.shard_range(N, self.rank, self.world)
        r32 = hd.candidate_pass(tk.mode, tk.states, tk.actions, tk.rewards, tk.eps, tk.h, tk.lmbda, float(cfg.discount), N,
                                begin, count, precision=capi.PREC_FP32, slot=tk.slot.i)
        f32, _ = mdist.gather_candidates(r32["expect_return"], r32["sample_actions"][:, 0], N, self.group)
        d = tk.er_b - f32
        dev = float((d - d.median()).abs().max())
        out = max(self.calibration_factor * dev, 1e-6 * float(f32.abs().max()), 1e-30)
        torch.cuda.synchronize(self.device)
        return out

    def _split(self, trajectory):
        """(states, actions, rewards, rtg, returns_row) of a reference-style trajectory dict of (1,T,D) tensors.  A window
        whose returns are constant (what action_sample builds, learner.py:368-385) is passed as one float; any other returns
        row goes to the library as it is (rtg_guiding consumes whatever trajectory["returns"] holds, learner.py:272-293)."""
        s, a, r = trajectory["states"][0], trajectory["actions"][0], trajectory["rewards"][0]
        rtg = trajectory.get("_rtg")
        returns = None
        if rtg is None:
            ret = trajectory["returns"].reshape(-1)
            if ret.dtype not in (torch.float32, torch.float64):
                ret = ret.float()
            returns = ret.to(self.device).contiguous()
            rtg = 0.0
        return s.float().contiguous(), a.float().contiguous(), r.float().contiguous(), float(rtg), returns

    @torch.no_grad()
    def rtg_guiding(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float = 0.6):
        """learner.py:271-327."""
        s, a, r, rtg, ret = self._split(trajectory)
        return self._guide(capi.MODE_RTG, s, a, r, rtg, h, lmbda, returns=ret)

    @torch.no_grad()
    def critic_lambda_guiding(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float):
        """learner.py:211-268."""
        s, a, r, rtg, ret = self._split(trajectory)
        return self._guide(capi.MODE_CRITIC, s, a, r, rtg, h, lmbda, returns=ret)

    @torch.no_grad()
    def noise_adding_lambda(self, trajectory: Dict[str, torch.Tensor], h: int, lmbda: float):
        """learner.py:142-208."""
        s, a, r, rtg, ret = self._split(trajectory)
        return self._guide(capi.MODE_NOISE, s, a, r, rtg, h, lmbda, returns=ret)

    def _returns_tokens(self, rtg, returns=None):
        """(1,T,1) returns tokens: the caller's row, or the constant return-to-go in float64 (learner.py:371-374)."""
        if returns is None:
            returns = torch.full((1, self.T, 1), rtg, dtype=torch.float64, device=self.device)
        return self.handle.tokenize(capi.RETURNS, returns.reshape(1, self.T, 1))

    @torch.no_grad()
    def mtm_sampling(self, trajectory: Dict[str, torch.Tensor], h: int):
        """learner.py:103-115: one return-conditioned policy pass, no planning."""
        s, a, r, rtg, ret = self._split(trajectory)
        T = self.T
        toks = [self.handle.tokenize(capi.STATES, s[None]), a[None], None, self._returns_tokens(rtg, ret)]
        from .masks import create_rcbc_mask, mask_rows
        out = self.handle.forward(toks, mask_rows(create_rcbc_mask(T, "cpu", T - h)), want=("actions",))
        self._mark_main()
        mu, sd = out["actions"]
        dist_ = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
        eps = self._eps(tuple(dist_.loc.shape))
        sample_action = dist_.sample(eps=eps)[0, T - h]
        eval_action = dist_.mean[0, T - h]
        return sample_action, eval_action

    # ---------------------------------------------------------------------------------------- batched planning
    @torch.no_grad()
    def plan_async(self, sequence_history, percentage=1.0, eval=False, rtg=None) -> "PlanTicket":
        """``action_sample(history, plan=True)`` without waiting for it: enqueues the plan step of one window and returns a
        ticket; ``ticket.result()`` gives what ``action_sample`` returns (on the device, ordered on the current stream).
        Up to ``capi.SLOTS - 1`` tickets may be outstanding; the windows must be independent of each other's results (several
        environments / evaluation episodes: learner.py:645-741 runs 20 of them one after the other).  Every step's result
        is bit-identical to the serial call; the draws (eps, multinomial) are taken from the generator in issue order."""
        if eval:
            assert rtg is not None
        guidance = self.cfg.plan_guidance
        assert guidance in _MODES, guidance
        lmbda = 0.6 if guidance == "rtg_guiding" else float(self.cfg.lmbda)  # learner.py:405-407
        sl = self._acquire_slot().ready(self)
        # pinned per-slot staging, refilled only once the copy that read it last has EXECUTED: resolving the slot's previous
        # owner blocks the host in the certified bf16 mode only -- in fp32 / top-k mode the host runs many steps ahead of
        # the device, and the copy of this slot's previous window may still be queued
        sl.ev_h2d.synchronize()
        h, return_to_go = self._window_host(sequence_history, rtg, percentage, sl.win_np)
        # copied on the CHAIN stream: the current stream is in order behind the previous step's candidate pass, the chain
        # stream is not
        chain = self._streams_of(sl)[0]
        with torch.cuda.stream(chain) as _:
            chain.wait_event(sl.ev_done)  # (the slot's previous owner has finished with the slot's device buffers)
            dev = sl.win_dev.copy_(sl.win, non_blocking=True)
            sl.ev_h2d.record(chain)
        states, actions, rewards = self._blocks(dev)
        tk = self._issue(_MODES[guidance], states, actions, rewards, return_to_go, h, lmbda, pipelined=True, slot=sl,
                         inputs_ready=True)
        tk.eval = bool(eval)
        tk.keep_window = dev
        return tk

    def flush(self):
        """Enqueue whatever a pipelined step still holds back (the tail of the last ticket)."""
        pend, self._pending = self._pending, []
        for tk in pend:
            if not tk.tail_enqueued:
                self._enqueue_tail(tk)

    @torch.no_grad()
    def action_sample_batch(self, sequence_histories, percentage=1.0, eval=False, rtg=None, lockstep: bool = False):
        """``action_sample(history, plan=True)`` for E environments.  The reference steps one environment at a
        time (replay_buffer.py:204-232, learner.py:681-691: one action_sample per env step, the action read back each time).
        Default: the E windows are E plan steps issued back to back through ``plan_async`` -- ``pipeline_depth`` of them in
        flight, staggered: window i+1's policy pass and window i-1's re-score + select run on the chain stream under window
        i's candidate pass.  Per window the result is bit-identical to the single-window call (same kernels, same rows, same
        draws in window order), first-layer history sharing included.
        ``lockstep=True``: the environments step together -- ONE policy pass at batch E (m3pc_policy_pass_batch), every window's own
        candidate pass back to back, ONE fp32 re-score pass over all windows' sets: the ~75 short launches of the fp32 chains are
        paid once per call instead of once per window (needs ``HipPlanner(..., max_batch=E, max_windows=E)``; windows grouped by
        effective horizon; the few-row fp32 kernels choose their tiling by the row count, so a window's result agrees with the
        single-window call to fp32 rounding, not bit for bit).  ``lockstep="onepass"``: the round-2 form, one candidate pass
        over E x N rows with a per-candidate window index (no history sharing between windows).
        ``rtg``: None, a float, or one value per window.  Returns (E, A)."""
        if lockstep:
            return self._action_sample_lockstep(sequence_histories, percentage, eval, rtg, onepass=lockstep == "onepass")
        E, A = len(sequence_histories), self.A
        rtgs = [rtg] * E if (rtg is None or np.isscalar(rtg)) else list(rtg)
        out = torch.empty((E, A), dtype=torch.float32, device=self.device)
        info = [None] * E
        flight = []

        def resolve():
            i, tk = flight.pop(0)
            sa, ev = tk.pair()
            out[i] = ev if eval else sa[0]
            info[i] = tk.info

        for i, hst in enumerate(sequence_histories):
            flight.append((i, self.plan_async(hst, percentage, eval, rtgs[i])))
            if len(flight) > self.pipeline_depth:
                resolve()
        while flight:
            resolve()
        self.last = dict(windows=info, delta=self._delta)
        return out

    @torch.no_grad()
    def action_sample(self, sequence_history, percentage=1.0, horizon=4, plan=True, eval=False, rtg=None):
        """learner.py:329-417 (the ``horizon`` argument is ignored there too: cfg.horizon rules)."""
        if eval:
            assert rtg is not None
        if plan:
            guidance = self.cfg.plan_guidance
            assert guidance in _MODES, guidance
            # (the step's variates are drawn behind the policy pass's launches: _issue)
            self._drain()
        states, actions, rewards, h, return_to_go = self.assemble_window(sequence_history, rtg, percentage)
        traj = {"states": states[None], "actions": actions[None], "rewards": rewards[None], "_rtg": return_to_go}
        if plan:
            if guidance == "rtg_guiding":
                sample_action, eval_action = self.rtg_guiding(traj, h)  # default lmbda=0.6, learner.py:405-407
            else:
                sample_action, eval_action = getattr(self, guidance)(traj, h, lmbda=self.cfg.lmbda)
        else:
            sample_action, eval_action = self.mtm_sampling(traj, h)
        return eval_action if eval else sample_action


# ------------------------------------------------------------------------------------------------------
def _versions(module):
    """(name -> version counter of every state_dict entry, total over parameters + buffers).  In-place optimizer updates bump
    a tensor's counter; the total catches modules whose state_dict hands out copies (their per-name counters never move)."""
    try:
        sd = module.state_dict(keep_vars=True)
    except TypeError:
        sd = module.state_dict()
    per = {k: int(v._version) for k, v in sd.items()}
    total = sum(int(p._version) for p in module.parameters()) + sum(int(b._version) for b in module.buffers())
    return per, total


def attach(learner, precision: str = "bf16", rescore_topk: int = 16, group=None, generator: Optional[torch.Generator] = None,
           **planner_kw):
    """Rebind the plan path of a reference-style ``Learner`` onto the HIP library.

    precision: "bf16" (default since round 6: the configuration the headline is measured on) -- the bf16 candidate pass with the
    certified fp32 re-score: BOTH returned actions follow the fp32 path's decisions (the arg-max behind ``eval_action`` and
    the multinomial index behind ``sample_action`` are certified, ``certify_sample=True``; ``eval_action``'s weights p agree
    to ~1e-4), and ``auto_fp32`` falls back to plain fp32 steps for weights on which the certificate is expensive.
    "fp32": every tolerance of the reference's own fp32 arithmetic, ~8 x the time per step.

    Reads: learner.cfg, learner.mtm (state_dict + config), learner.tokenizer_manager.tokenizers[k]
    (._data_mean, ._data_std, .normalize, .stats), learner.iql.qf (state_dict, obs_mean, obs_std).
    Afterwards learner.action_sample / rtg_guiding / critic_lambda_guiding / noise_adding_lambda /
    mtm_sampling (and the zero-shot calls) run on the GPU; everything else on the object is untouched.
    ``group`` + ``generator``: shard the candidates over the ranks of a process group (every rank attaches its own
    learner replica and passes a generator seeded identically); further keywords go to ``HipPlanner``.
    Weights are followed per tensor: before each call the version counters of ``mtm`` / ``iql.qf`` are compared with the
    ones uploaded last, and only the tensors that changed are sent (m3pc_load_weights re-packs what depends on them)."""
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
                         precision=precision, rescore_topk=rescore_topk, group=group, generator=generator, **planner_kw)
    state = {"mtm": _versions(mtm), "qf": None if qf is None else _versions(qf)}

    def _sync():
        per, total = _versions(mtm)
        old_per, old_total = state["mtm"]
        changed = [k for k, v in per.items() if old_per.get(k) != v]
        if not changed and total != old_total:
            changed = list(per.keys())  # something moved that the per-name counters do not see: send everything
        if changed:
            sd = mtm.state_dict()
            planner.load_state_dict({k: sd[k] for k in changed})
            state["mtm"] = (per, total)
        if qf is not None:
            vq = _versions(qf)
            if vq != state["qf"]:
                planner.load_critic(qf.state_dict(), qf.obs_mean, qf.obs_std)
                state["qf"] = vq
        planner.last_sync = changed

    def _wrap(name):
        fn = getattr(planner, name)

        def method(self, *a, **kw):
            _sync()
            out = fn(*a, **kw)
            if name == "action_piid_list_sample":  # the reference's rollout loop pops learner.action_list (learner.py:559-568)
                self.action_list = planner.action_list
            return out

        method.__name__ = name
        return types.MethodType(method, learner)

    for name in ("action_sample", "rtg_guiding", "critic_lambda_guiding", "noise_adding_lambda", "mtm_sampling",
                 "action_piid_sample", "action_id_sample", "action_piid_list_sample", "plan_async", "action_sample_batch"):
        setattr(learner, name, _wrap(name))
    learner._hip_planner = planner
    planner.sync = _sync
    return planner
