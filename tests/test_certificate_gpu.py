"""The certified bf16 -> fp32 re-score, quantified, and a soak of the step pipeline (4b).

Sweep: for many (weight seed, window, eps) trials the bf16 planner's step is compared with a FULL fp32 pass over the same
candidates: the arg-max AND the multinomial index must be the fp32 planner's in every trial (learner.py:318-325: the eval
action is weighted by p, the sampled action -- what every online rollout step executes -- is a0[multinomial(p)]; "argmax
indices bit-exact" is the north star's bar), for the headline shape, critic guidance at temperature 1 (BASELINE config 3), the
T=64 shard of config 4 and the reference's shipped N=625/H=4/T=8, and the largest deviation of (bf16 - fp32) from the common shift over ALL candidates -- not only the
re-scored set -- is recorded relative to the bound delta the step used.  The table goes to gpurun_out/ (copied to profiles/).

Soak: a few hundred plan steps issued through plan_async / action_sample / load_state_dict in random order and at random
pipeline depths, with allocator churn on the caller's stream between the issues, against the same sequence planned serially:
every result bit-identical (the test that would have caught the recycled per-step tensor of commit db93b75)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(T, N, H, tau=0.01, guidance="rtg_guiding"):
    return types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=tau, lmbda=0.6,
                                 plan_guidance=guidance, device="cuda")


def _window(dims, i):
    h = synth.make_history(dims, i % 11)
    h["path_length"] = [500, 37, 321, 998, 640, 77, 250, 123, 864][i % 9] if i < 9 else 33 + (i * 37) % 960
    return h


SWEEPS = [  # env, guidance, temperature, N, T, H, weight seeds, trials per seed
    ("hopper", "rtg_guiding", 0.01, 1024, 32, 16, 5, 40),             # BASELINE config 2 (the headline)
    ("hopper", "rtg_guiding", 0.01, 256, 32, 16, 4, 30),
    ("walker2d", "critic_lambda_guiding", 1.0, 4096, 32, 16, 3, 16),  # BASELINE config 3: Q-value scale, temperature 1
    ("halfcheetah", "rtg_guiding", 0.01, 2048, 64, 32, 3, 16),        # one rank's share of BASELINE config 4
    ("hopper", "rtg_guiding", 0.01, 625, 8, 4, 4, 30),                # the reference's shipped config (config.yaml:5,77-79)
]


def _plain_weights(dims, seeds):
    return [(f"seed{ws}", synth.make_state_dict(dims, ws), synth.make_tokenizer_stats(dims, ws)) for ws in range(seeds)]


@pytest.mark.slow
@pytest.mark.parametrize("env,guidance,tau,N,T,H,seeds,per_seed", SWEEPS, ids=[f"{s[0]}-{s[1].split('_')[0]}-N{s[3]}-T{s[4]}" for s in SWEEPS])
def test_certificate_sweep_argmax_sample_index_and_deviation(env, guidance, tau, N, T, H, seeds, per_seed):
    S, A = synth.ENV_DIMS[env]
    _sweep(env, guidance, tau, N, T, H, _plain_weights(synth.Dims(S, A, T), seeds), per_seed, "")


# "Trained-like" weights: delta is a calibrated statistic, and every sweep above draws its weights from
# ONE init recipe.  synth.trained_like moves that recipe towards a trained model -- every Linear x 2 or x 4, LayerNorm gains
# ~ U(0.5, 2), the returns tokenizer's std x 0.1 or x 10 (the x 1000 returns term of learner.py:305 then sits on another
# sigma / delta ratio) -- six variants per shape; the bar is the same: 0 wrong arg-maxes, 0 wrong multinomial indices.
TRAINED = [(1.0, 10.0), (1.5, 1.0), (2.0, 0.1), (2.0, 10.0), (4.0, 0.1), (4.0, 10.0)]


@pytest.mark.slow
@pytest.mark.parametrize("N,T,H,per_variant", [(1024, 32, 16, 8), (625, 8, 4, 8)], ids=["N1024-T32", "N625-T8"])
def test_certificate_sweep_trained_like_weights(N, T, H, per_variant):
    dims = synth.Dims(11, 3, T)
    sets = []
    for vi, (ls, rs) in enumerate(TRAINED):
        sd, st = synth.trained_like(synth.make_state_dict(dims, vi), synth.make_tokenizer_stats(dims, vi), seed=vi,
                                    linear_scale=ls, returns_std_scale=rs)
        sets.append((f"linear_x{ls:g}_retstd_x{rs:g}", sd, st))
    _sweep("hopper", "rtg_guiding", 0.01, N, T, H, sets, per_variant, "_trained_like", fp32_tol=1e-3)


def _sweep(env, guidance, tau, N, T, H, weight_sets, per_seed, family, fp32_tol=5e-5):
    per_seed *= int(os.environ.get("M3PC_SWEEP_SCALE", "1"))  # (a longer sweep for the record: profiles/r06_certificate_sweep*_long_*)
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    mode = capi.MODE_RTG if guidance == "rtg_guiding" else capi.MODE_CRITIC
    rows = []
    mismatches = sample_mismatches = near_ties = 0
    for ws, (label, sd, st) in enumerate(weight_sets):
        qsd, om, os_ = synth.make_critic(dims, ws) if mode == capi.MODE_CRITIC else (None, None, None)
        kw = json.loads(os.environ.get("M3PC_SWEEP_KW", "{}"))  # (for the record runs: e.g. '{"calibration_factor": 1.6, "calibration_windows": 16}')
        mk = lambda prec: HipPlanner(_cfg(T, N, H, tau, guidance), sd, st, qsd, om, os_, precision=prec, auto_fp32=False, **kw,
                                     generator=torch.Generator(device="cuda").manual_seed(1))  # (auto_fp32 off: the certificate itself is what is measured)
        pb, pf = mk("bf16"), mk("fp32")  # (same generator seed: both draw the same Exp(1) variates step for step)
        for t in range(per_seed):
            hist = _window(dims, t)
            eps = synth.make_eps(N, dims, 1000 * ws + t).cuda()
            rtg = 3.0 + 0.25 * (t % 5)
            sb, ab, rb, h, g = pb.assemble_window(hist, rtg=rtg)
            sab, _ = pb._guide(mode, sb, ab, rb, g, h, 0.6, eps=eps)
            lb = pb.last
            b, merged = lb["expect_return_bf16"].clone(), lb["expect_return"].clone()
            am_b, si_b = int(lb["argmax"].item()), int(lb["sample_idx"].item())
            saf, _ = pf._guide(mode, sb, ab, rb, g, h, 0.6, eps=eps)
            f = pf.last["expect_return"]
            am_f, si_f = int(pf.last["argmax"].item()), int(pf.last["sample_idx"].item())
            d = b - f
            c = float(lb["shift"])
            ratio = float((d - c).abs().max()) / float(lb["delta"])
            gap = torch.topk(f, 2).values
            # A disagreement counts unless it is an fp32 NEAR-TIE: the fp32 planner's own top two scores (race keys) closer than
            # what two fp32 evaluation orders differ by (fp32_tol of the score scale: the re-score chain's few-row kernels and the
            # full pass's sum the same products in different orders) -- there the "fp32 arg-max" is not defined to the last bit by
            # any implementation, the reference's included (seen once in 9648 trials: gap 9e-4 at scale 200, round 6)
            tol_abs = fp32_tol * float(f.abs().max())
            tie_a = am_b != am_f and float(f[am_f] - f[am_b]) <= tol_abs
            keys = float(tau) * f - torch.log(lb["expo"])  # (the draw is the arg-max of these: learner.py:318-323 as an exponential race)
            tie_s = si_b != si_f and float(keys[si_f] - keys[si_b]) <= float(tau) * tol_abs + 1e-6
            near_ties += int(tie_a) + int(tie_s)
            mismatches += int(am_b != am_f and not tie_a)
            sample_mismatches += int(si_b != si_f and not tie_s)
            # the re-scored entries of the merged vector ARE the fp32 scores
            top = torch.cat([lb["topk"].long(), lb["race"].long()])
            # (fp32_tol: two fp32 passes through different kernels -- few-row re-score against the full candidate pass -- agree to
            # fp32 rounding x the model's conditioning: 5e-5 of the score scale on the init recipe, up to 5.1e-4 seen with every Linear x 4)
            assert float((merged[top] - f[top]).abs().max()) <= fp32_tol * float(f.abs().max())
            assert si_b != si_f or torch.equal(sab, saf)
            assert lb["certified"]
            rows.append(dict(weight_seed=ws, weights=label, trial=t, argmax_match=am_b == am_f, sample_idx_match=si_b == si_f, fp32_near_tie=bool(tie_a or tie_s),
                             ratio=round(ratio, 4),
                             delta=round(float(lb["delta"]), 4), n_rescored=int(lb["n_rescored"]), need_first=int(lb["n_in_window"]),
                             n_race=int(lb["n_race"]), need_race_first=int(lb["need_race"]), saturated=bool(lb["saturated"]),
                             second_pass=bool(lb["n_rescored"] > lb["n_first"] or lb["n_race"] > lb["n_race_first"]),
                             top1_top2_gap=round(float(gap[0] - gap[1]), 4), score_sigma=round(float(f.std()), 3)))
        pb.handle.close()
        pf.handle.close()
    ratios = np.array([r["ratio"] for r in rows])
    tot = [r["n_rescored"] + r["n_race"] for r in rows]
    summary = dict(config=f"{env} {guidance} N={N} T={T} H={H} temperature={tau}", trials=len(rows), argmax_mismatches=mismatches,
                   sample_idx_mismatches=sample_mismatches, fp32_near_ties=near_ties,
                   ratio_max=float(ratios.max()), ratio_p99=float(np.quantile(ratios, 0.99)), ratio_median=float(np.median(ratios)),
                   trials_with_ratio_above_1=int((ratios > 1).sum()),
                   n_rescored_mean=float(np.mean(tot)), n_rescored_max=int(max(tot)),
                   n_by_score_mean=float(np.mean([r["n_rescored"] for r in rows])), n_by_race_mean=float(np.mean([r["n_race"] for r in rows])),
                   need_race_first_max=int(max(r["need_race_first"] for r in rows)),
                   second_pass_trials=int(sum(r["second_pass"] for r in rows)),
                   saturated_trials=int(sum(r["saturated"] for r in rows)),
                   per_weights={lb_: dict(trials=len(rr), ratio_max=float(max(r["ratio"] for r in rr)),
                                          delta_median=float(np.median([r["delta"] for r in rr])),
                                          score_sigma_median=float(np.median([r["score_sigma"] for r in rr])),
                                          n_rescored_mean=float(np.mean([r["n_rescored"] + r["n_race"] for r in rr])),
                                          n_rescored_max=int(max(r["n_rescored"] + r["n_race"] for r in rr)),
                                          saturated=int(sum(r["saturated"] for r in rr)))
                                for lb_ in dict.fromkeys(r["weights"] for r in rows) for rr in [[r for r in rows if r["weights"] == lb_]]},
                   what="ratio = max over ALL candidates of |(bf16 - fp32) - shift| / delta of the step; > 1 means a candidate outside "
                        "the bound existed in that trial (the arg-max / draw may still be right: it needs such a candidate inside the "
                        "gap); n_rescored = score-list + race-list candidates re-scored in fp32")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        tag = ("_long" if os.environ.get("M3PC_SWEEP_SCALE") else "") + os.environ.get("M3PC_SWEEP_TAG", "")
        with open(os.path.join(out_dir, f"r06_certificate_sweep{family}{tag}_{env}_{guidance.split('_')[0]}_N{N}_T{T}.json"), "w") as fh:
            json.dump(dict(summary=summary, rows=rows), fh, indent=0)
    print(json.dumps(summary))
    assert len(rows) >= 45
    assert mismatches == 0, summary
    assert sample_mismatches == 0, summary


def _churn(rng, keep):
    """Allocator traffic on the CALLER's stream between two issues: blocks of random sizes allocated, written, some kept for
    a while, most freed at once (what a rollout loop's own tensor code does around the planner)."""
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.choice([256, 4096, 65536, 1 << 20, 3 << 20]))
        x = torch.empty((n,), device="cuda", dtype=torch.float32)
        x.fill_(float(rng.random()))
        y = x * 2.0 + 1.0
        if rng.random() < 0.3:
            keep.append(y)
    while len(keep) > 6:
        keep.pop(int(rng.integers(0, len(keep))))


@pytest.mark.slow
@pytest.mark.parametrize("precision,N,T,H,steps", [("bf16", 1024, 32, 16, 120), ("fp32", 256, 16, 8, 120), ("bf16", 256, 16, 8, 120)])
def test_pipeline_soak_with_allocator_churn(precision, N, T, H, steps):
    dims = synth.Dims(11, 3, T)
    rng = np.random.default_rng(7)
    script = []
    for i in range(steps):
        u = rng.random()
        if u < 0.03 and i > 5:
            script.append(("load", int(rng.integers(0, 3))))
        elif u < 0.25:
            script.append(("sync", int(rng.integers(0, 50))))
        else:
            script.append(("async", int(rng.integers(0, 50)), int(rng.integers(1, 4))))  # window, depth limit
    sds = [synth.make_state_dict(dims, k) for k in range(3)]

    def planner():
        return HipPlanner(_cfg(T, N, H), sds[0], synth.make_tokenizer_stats(dims, 0), None, precision=precision,
                          generator=torch.Generator(device="cuda").manual_seed(21), pipeline_depth=3)

    # the serial order: every step one blocking action_sample
    ps = planner()
    serial = []
    for op in script:
        if op[0] == "load":
            ps.load_state_dict(sds[op[1]])
            serial.append(None)
        else:
            serial.append(ps.action_sample(_window(dims, op[1]), plan=True, eval=(op[1] % 2 == 0), rtg=3.0).clone())
    ps.handle.close()
    # the same script, pipelined at varying depth, with allocator churn on the caller's stream
    pp = planner()
    got = [None] * len(script)
    flight = []
    keep = []
    crng = np.random.default_rng(11)

    def resolve_one():
        i, tk = flight.pop(0)
        got[i] = tk.result().clone()

    for i, op in enumerate(script):
        _churn(crng, keep)
        if op[0] == "load":
            pp.load_state_dict(sds[op[1]])  # resolves what is in flight first (the tickets keep their results)
        elif op[0] == "sync":
            got[i] = pp.action_sample(_window(dims, op[1]), plan=True, eval=(op[1] % 2 == 0), rtg=3.0).clone()
        else:
            flight.append((i, pp.plan_async(_window(dims, op[1]), eval=(op[1] % 2 == 0), rtg=3.0)))
            while len(flight) > op[2]:
                resolve_one()
    while flight:
        resolve_one()
    torch.cuda.synchronize()
    bad = [i for i, (a, b) in enumerate(zip(serial, got)) if a is not None and not torch.equal(a, b)]
    assert not bad, (bad[:10], len(bad))
    pp.handle.close()


def test_load_critic_recalibrates_the_bound():
    """critic_lambda_guiding's bf16 - fp32 deviation contains min(q1, q2) on bf16-decoded states (learner.py:250-252), and
    fine-tuning updates the Q networks between rollouts (finetune.py:288-290 -> attach._sync -> load_critic): the calibrated
    bound must not outlive the Q networks it was measured on."""
    dims = synth.Dims(17, 6, 32)
    N = 512
    sd, st = synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0)
    qsd, om, os_ = synth.make_critic(dims, 0)
    mk = lambda prec: HipPlanner(_cfg(32, N, 16, 1.0, "critic_lambda_guiding"), sd, st, qsd, om, os_, precision=prec,
                                 generator=torch.Generator(device="cuda").manual_seed(3))
    pb, pf = mk("bf16"), mk("fp32")
    assert pb._cal_left == pb._cal_windows and pb._delta0 is None

    def steps(n, first):
        for t in range(first, first + n):
            hist = _window(dims, t)
            eps = synth.make_eps(N, dims, 50 + t).cuda()
            s_, a_, r_, h, g = pb.assemble_window(hist, rtg=3.0)
            pb._guide(capi.MODE_CRITIC, s_, a_, r_, g, h, 0.6, eps=eps)
            lb = pb.last
            pf._guide(capi.MODE_CRITIC, s_, a_, r_, g, h, 0.6, eps=eps)
            assert int(lb["argmax"].item()) == int(pf.last["argmax"].item())
            assert int(lb["sample_idx"].item()) == int(pf.last["sample_idx"].item())
            assert lb["certified"]

    steps(pb._cal_windows + 1, 0)
    assert pb._cal_left == 0
    d_old = pb._delta0
    # the Q networks after "fine-tuning": output layers x 8 -- Q values, hence the deviation of the bf16 scores, scale with them
    q_new = {k: (v * 8.0 if ".net.4." in k else v.clone()) for k, v in qsd.items()}
    pb.load_critic(q_new, om, os_)
    pf.load_critic(q_new, om, os_)
    assert pb._cal_left == pb._cal_windows and pb._delta0 is None and pb._hist == {}, "load_critic must reset the calibration"
    steps(pb._cal_windows + 1, 100)
    assert pb._cal_left == 0
    assert pb._delta0 > 3.0 * d_old, (pb._delta0, d_old)  # the bound follows the new Q scale (measured: ~8 x)
    pb.handle.close()
    pf.handle.close()


def test_auto_fp32_fallback_on_weights_that_make_the_certificate_expensive():
    """HipPlanner(auto_fp32=True), the default: with every Linear x 4 the bf16 deviation exceeds the score spread, every step's
    certificate re-scores most of the candidates in fp32, and a certified bf16 step costs more than an fp32 step -- the planner says so once
    and plans in fp32 until the next weight load.  The decision is lagged by capi.SLOTS steps (same step at any pipeline depth)."""
    N, T, H = 1024, 32, 16  # (the headline shape: profiles/r06_certificate_sweep_trained_like_*: 1046 of 1024 + 32 list entries re-scored)
    dims = synth.Dims(11, 3, T)
    sd0, st0 = synth.make_state_dict(dims, 2), synth.make_tokenizer_stats(dims, 2)
    sd4, st4 = synth.trained_like(sd0, st0, seed=2, linear_scale=4.0, returns_std_scale=0.1)
    pb = HipPlanner(_cfg(T, N, H), sd4, st4, None, precision="bf16", generator=torch.Generator(device="cuda").manual_seed(1))
    pf = HipPlanner(_cfg(T, N, H), sd4, st4, None, precision="fp32", generator=torch.Generator(device="cuda").manual_seed(1))
    switched_at = None
    with pytest.warns(UserWarning, match="planning in fp32"):
        for t in range(4 + capi.SLOTS + 2):
            hist = _window(dims, t)
            eps = synth.make_eps(N, dims, 300 + t).cuda()
            pb._eps = pf._eps = lambda shape: eps
            ab = pb.action_sample(hist, plan=True, eval=True, rtg=3.0)
            af = pf.action_sample(hist, plan=True, eval=True, rtg=3.0)
            assert int(pb.last["argmax"].item()) == int(pf.last["argmax"].item())
            assert int(pb.last["sample_idx"].item()) == int(pf.last["sample_idx"].item())
            if pb.fp32_fallback:
                switched_at = t if switched_at is None else switched_at
                assert torch.equal(ab, af) and torch.equal(pb.last["expect_return"], pf.last["expect_return"])
            else:
                assert pb.last["certified"] and pb.last["n_rescored"] + pb.last["n_race"] >= N // 2
    assert switched_at == 3 + capi.SLOTS  # four expensive steps seen, SLOTS steps late
    pb.load_state_dict(sd0)   # new weights: bf16 again (the tokenizer statistics stay: the point is the switch)
    assert not pb.fp32_fallback and pb.precision == capi.PREC_BF16 and pb.rescore == "bound"
    pb.action_sample(_window(dims, 0), plan=True, eval=True, rtg=3.0)
    assert pb.last.get("expect_return_bf16") is not None
    pb.handle.close()
    pf.handle.close()


def test_auto_fp32_fallback_is_the_same_step_at_any_pipeline_depth():
    """The fallback is decided on the steps up to index - SLOTS (like every adaptive quantity of the certified re-score), so a
    pipelined run switches at the same step as the serial one and returns the same actions, bit for bit."""
    N, T, H = 1024, 32, 16
    dims = synth.Dims(11, 3, T)
    sd0, st0 = synth.make_state_dict(dims, 2), synth.make_tokenizer_stats(dims, 2)
    sd4, st4 = synth.trained_like(sd0, st0, seed=2, linear_scale=4.0, returns_std_scale=0.1)
    n_steps = 4 + capi.SLOTS + 4

    def planner():
        return HipPlanner(_cfg(T, N, H), sd4, st4, None, precision="bf16", generator=torch.Generator(device="cuda").manual_seed(5),
                          pipeline_depth=3)

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ps = planner()
        serial, sw_serial = [], None
        for t in range(n_steps):
            serial.append(ps.action_sample(_window(dims, t), plan=True, eval=(t % 2 == 0), rtg=3.0).clone())
            if ps.fp32_fallback and sw_serial is None:
                sw_serial = t
        ps.handle.close()
        pp = planner()
        got, flight, sw_pipe = [None] * n_steps, [], None
        for t in range(n_steps):
            flight.append((t, pp.plan_async(_window(dims, t), eval=(t % 2 == 0), rtg=3.0)))
            if pp.fp32_fallback and sw_pipe is None:
                sw_pipe = t  # (decided when step t was issued)
            while len(flight) > 3:
                i, tk = flight.pop(0)
                got[i] = tk.result().clone()
        for i, tk in flight:
            got[i] = tk.result().clone()
        torch.cuda.synchronize()
        pp.handle.close()
    assert sw_serial == 3 + capi.SLOTS and sw_pipe == sw_serial, (sw_serial, sw_pipe)
    bad = [i for i, (a, b) in enumerate(zip(serial, got)) if not torch.equal(a, b)]
    assert not bad, bad
