"""Environments that step together (``HipPlanner.action_sample_batch(lockstep=True)``): ONE policy pass at batch E
(m3pc_policy_pass_batch), every window's own candidate pass back to back, ONE fp32 re-score pass over all windows' re-score
sets (m3pc_score_actions) and one host read of all certificates -- the fp32 chains are paid once per call instead of once
per window.  A mixin of ``HipPlanner``; the certificate protocol itself is m3pc_amd/certificate.py."""
from __future__ import annotations

import types

import numpy as np
import torch

from . import capi
from .certificate import _WindowOps, resolve as _resolve_certificate

_MODES = {"rtg_guiding": capi.MODE_RTG, "critic_lambda_guiding": capi.MODE_CRITIC, "noise_adding_lambda": capi.MODE_NOISE}


class LockstepMixin:
    def _action_sample_lockstep(self, sequence_histories, percentage=1.0, eval=False, rtg=None, onepass: bool = False):
        self._drain()
        cfg = self.cfg
        guidance = cfg.plan_guidance
        assert guidance in _MODES, guidance
        mode = _MODES[guidance]
        lmbda = 0.6 if guidance == "rtg_guiding" else float(cfg.lmbda)  # learner.py:405-407
        E, T, S, A, N = len(sequence_histories), self.T, self.S, self.A, int(cfg.action_samples)
        rtgs = [rtg] * E if (rtg is None or np.isscalar(rtg)) else list(rtg)
        if eval:
            assert all(r is not None for r in rtgs)
        host = np.empty((E, T * (S + A + 1)), dtype=np.float32)
        meta = []
        for i, hst in enumerate(sequence_histories):
            meta.append(self._window_host(hst, rtgs[i], percentage, host[i]))
        dev = torch.from_numpy(host).to(self.device)  # one packed H2D copy for all windows
        out = torch.empty((E, A), dtype=torch.float32, device=self.device)
        info = [None] * E
        groups = []
        for h in sorted({m[0] for m in meta}):  # windows of one effective horizon, at most max_batch of them per group
            same = [i for i, m in enumerate(meta) if m[0] == h]
            groups += [(h, same[c0 : c0 + self._max_batch]) for c0 in range(0, len(same), self._max_batch)]
        for h, ids in groups:
            sel = dev if len(ids) == E else dev[torch.tensor(ids, device=self.device)]
            s = sel[:, : T * S].reshape(-1, T, S).contiguous()
            a = sel[:, T * S : T * (S + A)].reshape(-1, T, A).contiguous()
            r = sel[:, T * (S + A) :].reshape(-1, T, 1).contiguous()
            Eg = len(ids)
            eps = self._eps((Eg, N, h, A)) if mode == capi.MODE_NOISE else self._eps((Eg, N, T, A))
            if onepass:
                res = self.handle.plan_step_batch(mode, s, a, r, [meta[i][1] for i in ids], eps, h, lmbda, float(cfg.discount), N,
                                                  precision=self.precision)
            else:
                # ONE policy pass at batch Eg, then every window's own candidate pass (first-layer history sharing and the two
                # candidate halves as in the single-window step), back to back: the caller's stream joins the halves once, at the end
                f32 = dict(dtype=torch.float32, device=self.device)
                res = dict(expect_return=torch.empty((Eg, N), **f32), sample_actions=torch.empty((Eg, N, h, A), **f32),
                           loc=torch.empty((Eg, T, A), **f32), std=torch.empty((Eg, T, A), **f32))
                self.handle.policy_pass_batch(mode, s, a, r, h, [meta[i][1] for i in ids], slot=0)
                for w in range(Eg):
                    self.handle.candidate_pass(mode, s[w], a[w], r[w], eps[w], h, lmbda, float(cfg.discount), N, precision=self.precision,
                                               slot=0, window=w, defer_join=True,
                                               out={k: v[w] for k, v in res.items()})
                self.handle.candidate_join(0)
            er, acts = res["expect_return"], res["sample_actions"]
            stats_h = None
            merged = [er[w] for w in range(Eg)]
            # the multinomial's exponentials of every window (the same order of draws as before the race lists needed them early)
            expos = [torch.empty((N,), dtype=torch.float32, device=self.device).exponential_(1, generator=self.generator)
                     for _ in range(Eg)]
            temp = float(cfg.temperature)
            bound = self.rescore == "bound"
            if self.rescore != "none":
                smode = capi.MODE_RTG if mode == capi.MODE_RTG else capi.MODE_CRITIC
                R = self._R if bound else 0
                if bound:
                    if self._delta_fixed is None and (self._delta0 is None or self._cal_left > 0):
                        # calibrate on this group's windows -- all of a window's candidates in fp32 -- until the weight load's
                        # calibration windows are used up (round 6: the same count as the single-window steps take, certificate
                        # sweeps of DESIGN section 5; round 5 took window 0 of the first group only)
                        d0 = self._delta0 or 0.0
                        n_cal = max(1, min(Eg, self._cal_left))
                        for w in range(n_cal):
                            f32 = self.handle.score_actions(smode, s[w], a[w], r[w], acts[w], None, h, lmbda, float(cfg.discount))
                            d = er[w] - f32
                            d0 = max(d0, self.calibration_factor * float((d - d.median()).abs().max()), 1e-6 * float(f32.abs().max()), 1e-30)
                        self._delta0 = d0
                        self._cal_left = max(0, self._cal_left - n_cal)
                    # The certified re-score of certificate.py, for all windows of the group at once: the kmin best candidates
                    # by score and the rfirst best by race key of every window in ONE fp32 pass, merge + select enqueued for
                    # every window, THEN one host read of the certificates; windows that ask for more get passes of their own.
                    kmax = max(min(self.rescore_max, N - 1 if N > 1 else 1, 1024 - self._R - 1), 1)  # (merge kernels: r + n <= 1024)
                    rfirst = max(min(self.race_min, R), 1) if R > 0 else 0
                    kmin = max(min(self.rescore_min - rfirst, N, kmax), 1)  # (the race entries share the floor of the first pass)
                    lsts, blst = [], []
                    for w in range(Eg):
                        bt = torch.empty((R + kmax + 1,), dtype=torch.float32, device=self.device)
                        if R > 0:
                            lsts.append(self.handle.topk_race_window(er[w], expos[w], temp, kmax, kmin, R, list_scores=bt)[0])
                        else:
                            lsts.append(self.handle.topk_window(er[w], kmax, kmin, 0.0, top_scores=bt)[0])
                        blst.append(bt)
                else:
                    kmax = kmin = min(self.rescore_topk, N)
                    rfirst = 0
                    lsts = [torch.topk(er[w], kmin).indices.to(torch.int32) for w in range(Eg)]
                    blst = [er[w][lsts[w].long()].contiguous() for w in range(Eg)]
                delta = float(self._delta) if bound else 0.0
                m0 = rfirst + kmin
                pick = torch.cat([lsts[w][R - rfirst : R + kmin].long() for w in range(Eg)])
                wsel = torch.arange(Eg, dtype=torch.int32, device=self.device).repeat_interleave(m0)
                f32 = self.handle.score_actions(smode, s, a, r, acts[wsel.long(), pick], wsel, h, lmbda, float(cfg.discount))
                flst, mstats = [], []
                for w in range(Eg):  # fp32 scores for the sets, shift-corrected bf16 scores for the rest (m3pc_rescore_merge[_race])
                    fl = torch.empty_like(blst[w])
                    fl[R - rfirst : R + kmin] = f32[w * m0 : (w + 1) * m0]
                    flst.append(fl)
                    o = R - rfirst
                    if bound and R > 0:
                        merged[w], st_w = self.handle.rescore_merge_race(er[w], expos[w], temp, lsts[w][o:], rfirst, kmin, blst[w][o:],
                                                                        fl[o:], delta=delta)
                    else:
                        merged[w], st_w = self.handle.rescore_merge(er[w], lsts[w], kmin, blst[w], fl, delta=delta)
                    mstats.append(st_w)
            sels = [self.handle.select(merged[j], acts[j, :, 0], temp, expos[j]) for j in range(Eg)]
            certs = [None] * Eg
            counts = [kmin if self.rescore != "none" else 0] * Eg
            if bound:
                stats_h = torch.stack(mstats).cpu()  # the one host read of the group: the certificates' statistics per window
                ctx = types.SimpleNamespace(hdl=self.handle, cap=max(self.handle.max_rescore, 1), disc=float(cfg.discount),
                                            temp=temp, stats_h=stats_h, smode=smode, s=s, a=a, r=r, acts=acts, h=h,
                                            lmbda=lmbda, N=N, er=er, lsts=lsts, blst=blst, flst=flst, merged=merged, sels=sels,
                                            expos=expos, R=R, wset=[None] * Eg, nd=[kmin] * Eg, rd=[rfirst] * Eg)
                delta_first = delta
                for w in range(Eg):
                    ops_w = _WindowOps(ctx, w)
                    if delta > delta_first:  # an earlier window of the group raised the bound: this window's certificate again, under it
                        ops_w.merge_select(ctx.nd[w], ctx.rd[w], delta)
                    certs[w] = _resolve_certificate(self, N, kmax, R, ctx.nd[w], ctx.rd[w], delta, ops_w)
                    if certs[w]["delta"] > delta:  # this window saw a larger deviation than the bound: raised for everybody from here on
                        delta = certs[w]["delta"]
                        self._delta0 = max(self._delta0, delta)  # (the base bound, not the setter: calibration windows still to come stay)
                    counts[w] = certs[w]["n_rescored"]
            for j, i in enumerate(ids):
                p, ev, am, si, sa = sels[j]
                out[i] = ev if eval else sa[0]
                info[i] = dict(expect_return=merged[j], argmax=am, sample_idx=si, eval_action=ev, sample_action=sa, horizon=h,
                               n_rescored=None if certs[j] is None else counts[j],
                               n_race=None if certs[j] is None else certs[j]["n_race"],
                               min_margin_outside=None if certs[j] is None else certs[j]["min_margin_outside"],
                               saturated=None if certs[j] is None else certs[j]["saturated"],
                               delta=self._delta)
        self._mark_main()
        self.last = dict(windows=info, delta=self._delta)
        return out

