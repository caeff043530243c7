"""The certified bf16 -> fp32 re-score of a plan step (``HipPlanner(precision="bf16", rescore="bound")``): what the step's
arg-max (learner.py:318-323, the eval action) AND its multinomial draw (learner.py:324-325, the sampled action every online
rollout step executes, replay_buffer.py:206-216) need in order not to depend on bf16 rounding.

Model: bf16 score b_j = f_j + c + e_j with a common shift c and a deviation |e_j| <= delta (delta: ``HipPlanner._calibrate``,
checked on every step's re-scored set).  Two lists of candidates are re-scored in fp32 (m3pc_topk_race_window):

  * score list -- the best candidates by bf16 score.  An un-re-scored j can only hold the fp32 arg-max if
    b_j > f* + c - delta (f* = the best re-scored fp32 score): ``need`` counts them over the whole vector, a prefix of the
    descending order; need <= n re-scored certifies the arg-max.
  * race list -- torch.multinomial(p, 1) is arg-max_j p_j / q_j, q ~ Exp(1) (m3pc_select draws with the caller's q), i.e. the
    race arg-max_j (tau E_j - log q_j).  An un-re-scored j can only win it if tau (b_j - c + delta) - log q_j >= K* (K* = the
    best re-scored candidate's exact key): ``need_race`` counts them over the whole vector, a prefix of the descending
    race-key order; need_race <= r re-scored certifies the draw.

``resolve`` is the protocol both callers share (the pipelined ticket, ``_TicketOps``; a window of a lock-step batch,
``_WindowOps``): a first pass -- kmin by score, rfirst by race key, merge, select -- is enqueued before anybody reads
anything; then read the certificate from host-mapped memory and re-score / merge / select again only when it asks for more.
"""
from __future__ import annotations

import warnings

import torch

RACE_MAX = 32  # race entries a step may list (m3pc_topk_race_window rmax); beyond: the whole-set fp32 slow path


class _TicketOps:
    """The device work of ``resolve`` for one plan step in flight (its re-scores run through the step's slot, on the step's
    tail stream).  A plain object per resolution -- NOT a class defined per call: a class object sits in reference cycles,
    and a cycle that reaches the ticket keeps the step's device tensors alive until the cyclic collector runs; the caching
    allocator then has to hipMalloc fresh blocks (a device synchronisation each) and the step pipeline falls apart
    (measured: 780 -> 400 plan-steps/s in most runs)."""
    __slots__ = ("planner", "tk", "top")

    def __init__(self, planner, tk):
        self.planner, self.tk, self.top = planner, tk, None

    def read(self):
        sl = self.tk.slot
        return sl.hs_mrg.wait(self.tk.seq_mrg, sl.mstats)

    def _rescore(self, lo, hi):
        """fp32 re-score of list entries [lo, hi) (positions in the step's list buffer)."""
        pl, tk = self.planner, self.tk
        sl, N = tk.slot, tk.er_b.numel()
        with pl._on(tk):
            rs, tail = pl._rescore_args(tk)
            pl.handle.rescore(*rs, tk.lst[lo:hi], *tail, N, slot=sl.i, out=sl.f_lst[lo:hi], want_actions=False)

    def extend(self, lo, hi):
        R = self.tk.R
        self._rescore(R + lo, R + hi)
        self.tk.n_done = hi

    def extend_race(self, lo, hi):
        R = self.tk.R
        self._rescore(R - hi, R - lo)
        self.tk.r_done = hi

    def window_set(self, need, delta, everything=False):
        pl, tk = self.planner, self.tk
        tk.delta = delta
        with pl._on(tk):
            self.top = pl._rescore_window_set(tk, need, everything)
        return tk.n_done

    def merge_select(self, n, r, delta):
        pl, tk = self.planner, self.tk
        tk.delta = delta
        with pl._on(tk):
            pl._merge(tk, n, r, select=True)
            if tk.tchain is not None:
                tk.slot.ev_done.record(tk.tchain)


class _WindowOps:
    """The device work of ``resolve`` for window w of a lock-step group (fp32 re-scores through m3pc_score_actions on the
    window's own rows).  ``c``: the group's state (a plain namespace; see _TicketOps on why this is not a class defined
    inside the call).  Per window: lsts / blst / flst = the list buffer of m3pc_topk_race_window, its bf16 scores and fp32
    re-scores (R race entries in front of the score entries); nd / rd = how many score / race entries are re-scored; wset =
    the (ids, bf16 scores, fp32 re-scores) of the window-set slow path, which then stand in for the score entries."""
    __slots__ = ("c", "w", "pending")

    def __init__(self, c, w):
        self.c, self.w, self.pending = c, w, None

    def read(self):
        c, w = self.c, self.w
        if self.pending is not None:
            c.stats_h[w], self.pending = self.pending.cpu(), None
        v = [float(x) for x in c.stats_h[w]]
        return v[:4] + (v[5:8] if len(v) >= 8 else [0.0, 0.0, 0.0])

    def _score(self, ix):
        c, w = self.c, self.w
        return torch.cat([c.hdl.score_actions(c.smode, c.s[w], c.a[w], c.r[w], c.acts[w, ix[c0 : c0 + c.cap].long()], None, c.h,
                                              c.lmbda, c.disc) for c0 in range(0, ix.numel(), c.cap)])

    def extend(self, lo, hi):
        c, w = self.c, self.w
        c.flst[w][c.R + lo : c.R + hi] = self._score(c.lsts[w][c.R + lo : c.R + hi])
        c.nd[w] = hi

    def extend_race(self, lo, hi):
        c, w = self.c, self.w
        c.flst[w][c.R - hi : c.R - lo] = self._score(c.lsts[w][c.R - hi : c.R - lo])
        c.rd[w] = hi

    def window_set(self, need, dlt, everything=False):
        c, w = self.c, self.w
        cnt = min(need, c.N)
        if cnt <= 1024 - RACE_MAX and not everything:
            # the `need` best candidates by bf16 score, re-scored in chunks of the chain workspace; they replace the score part
            # of the window's lists (the race part stays, and may still grow)
            vals, idx = torch.topk(c.er[w], cnt)
            idx = idx.to(torch.int32).contiguous()
            c.wset[w] = (idx, vals.contiguous(), self._score(idx).contiguous())
            c.nd[w] = cnt
            self.merge_select(cnt, c.rd[w], dlt)
            return cnt
        # beyond what the merge kernel lists: EVERY candidate of the window in fp32 -- the select then runs on fp32 scores
        # alone (a merge of the best entry with itself keeps the statistics protocol alive)
        er32 = c.hdl.score_actions(c.smode, c.s[w], c.a[w], c.r[w], c.acts[w], None, c.h, c.lmbda, c.disc).contiguous()
        best = torch.argmax(er32).to(torch.int32).reshape(1)
        bval = er32.max().reshape(1).contiguous()
        c.wset[w], c.nd[w], c.rd[w] = (best, bval, bval), c.N, 0
        c.merged[w], self.pending = c.hdl.rescore_merge(er32, best, 1, bval, bval, delta=0.0)
        c.sels[w] = c.hdl.select(c.merged[w], c.acts[w, :, 0], c.temp, c.expos[w])
        return c.N

    def merge_select(self, n, r, dlt):
        c, w = self.c, self.w
        o = c.R - r
        lists = (c.lsts[w][o:], c.blst[w][o:], c.flst[w][o:])
        if c.wset[w] is not None:  # (the window-set path's own score entries behind the race entries of the list buffer)
            lists = tuple(torch.cat([a[:r], b]).contiguous() for a, b in zip(lists, c.wset[w]))
        if c.R > 0:
            c.merged[w], self.pending = c.hdl.rescore_merge_race(c.er[w], c.expos[w], c.temp, lists[0], r, n, lists[1], lists[2], delta=dlt)
        else:
            c.merged[w], self.pending = c.hdl.rescore_merge(c.er[w], lists[0], n, lists[1], lists[2], delta=dlt)
        c.sels[w] = c.hdl.select(c.merged[w], c.acts[w, :, 0], c.temp, c.expos[w])


def resolve(planner, N, kmax, rmax, n_done, r_done, delta, ops):
    """The certified re-score's protocol.  A first pass has been enqueued already: the ``n_done`` best candidates by bf16
    score and the ``r_done`` best by race key re-scored in fp32, merged, selected.  ``ops`` does the device work:
        read()                    -> (shift, deviation, need, margin, need_race, K*, its threshold) of the LAST merge (blocks
                                  the host until they are there)
        extend(lo, hi)            fp32 re-score of the entries [lo, hi) of the score list
        extend_race(lo, hi)       fp32 re-score of the entries [lo, hi) of the race list
        window_set(need, delta, everything)  the lists are too short -- re-score the `need` best candidates by score (or every
                                  candidate: beyond the list capacity, or when the race list is exhausted) and leave the merged
                                  vector + select enqueued;  -> n_done
        merge_select(n, r, delta) merge + select again over the n + r re-scored entries
    Loop: read the certificates; raise delta when this step's re-scored set deviates by more than it allows (then merge
    again: both counts depend on delta); re-score what a certificate asks for; stop when both are satisfied, when everything
    has been re-scored, or after the whole-set slow path.  Returns the step's record."""
    saturated, everything = False, False
    first_need = first_race = None
    while True:
        shift, dev, need, margin, need_race = ops.read()[:5]
        need, need_race = int(need), int(need_race)
        if first_need is None:
            first_need, first_race = need, need_race
        redo = False
        # delta bounds the deviation of (bf16 - fp32) from the common shift: every step checks it on its re-scored set and
        # raises it when 1.5 x what it saw is more (the same numbers, hence the same decision, on every rank and at any
        # pipeline depth)
        if planner._delta_fixed is None and 1.5 * dev > delta and not everything:
            delta = 1.5 * dev
            planner.delta_grown += 1
            redo = n_done < N
        if everything or n_done >= N:
            break
        if not redo:
            if need > n_done and saturated:
                # the window set was re-scored and its certificate STILL asks for more (the median shift is taken over the new
                # list, so the threshold moves): nothing short of every candidate in fp32 settles it
                n_done = ops.window_set(N, delta, everything=True)
                everything = True
                continue
            if need > n_done:
                if need <= kmax:
                    ops.extend(n_done, need)
                    n_done = need
                    redo = True
                else:
                    if not planner._warned_saturated:
                        planner._warned_saturated = True
                        warnings.warn(f"m3pc_amd: {need} candidates may still hold the fp32 arg-max (delta={delta:.3g}, rescore_max="
                                      f"{planner.rescore_max}); re-scoring the whole window set in fp32 (slow path)")
                    n_done = ops.window_set(need, delta)
                    saturated = True
                    everything = n_done >= N
                    continue
            if need_race > r_done:
                if need_race <= rmax:
                    ops.extend_race(r_done, need_race)
                    r_done = need_race
                    redo = True
                else:
                    # more racers than the race list holds (tau * delta is large against the Exp(1) spread of the keys): every
                    # candidate in fp32
                    if not planner._warned_saturated:
                        planner._warned_saturated = True
                        warnings.warn(f"m3pc_amd: {need_race} candidates may still win the multinomial draw (delta={delta:.3g}); "
                                      f"re-scoring every candidate in fp32 (slow path)")
                    n_done = ops.window_set(N, delta, everything=True)
                    saturated = everything = True
                    continue
        if not redo:
            break
        ops.merge_select(n_done, r_done, delta)
    # certified: the loop ended on satisfied certificates (or on fp32 scores for every candidate) -- it cannot end otherwise,
    # the record says so for callers and tests
    certified = bool(everything or n_done >= N or (need <= n_done and need_race <= r_done))
    return dict(n_rescored=n_done, n_in_window=first_need, min_margin_outside=float(margin), delta=delta, saturated=saturated,
                shift=shift, deviation=dev, n_race=r_done, need_race=first_race, certified=certified)
