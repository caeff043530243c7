#!/usr/bin/env python3
"""bench.py -- MPC plan-steps/sec of the m3pc test-time planner on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one complete plan step of rtg_guiding (research/finetune_omtm/learner.py:271-327): draw eps,
policy pass (batch 1), sample N candidates, batched candidate pass, TD(lambda) scoring, all-gather of the
shards (N > 1 GPU), fp32 re-score of every candidate within 2*delta of the bf16 maximum (the bound-driven set,
m3pc_amd/planner.py; one 16-byte host read per step), softmax / weighted mean / argmax, multinomial draw.
Inputs (window, weights) are resident in HBM when the timed region starts.

Besides the contract fields the line carries (rank 0, 1 GPU): `rescore` (set sizes), `fp32_ms_per_step` (the same step
fp32 end to end: the mode that meets the fp32 tolerance everywhere), `closed_loop` (action_sample with the H2D window
copy and the action read back every step), `latency_ms_shipped` (the reference's shipped N=625/H=4/T=8 config and the
zero-shot B=1 call: the launch-latency-bound operating points), `batched` (E windows per launch).

Workload at 1 GPU: BASELINE configs[1] = hopper-medium-v2 shapes (S=11, A=3), rtg_guiding, N=1024
candidates, H=16, T=32 (=2H, the reference's shipped T/H ratio), bf16 candidate pass, synthetic data.
Multi-GPU: weak scaling -- 1024 candidates PER GPU (global N = 1024*G, sharded by candidate, one RCCL
all-gather of scores + first actions per step); value counts 1024-candidate plan-step units:
value = G * K / t.  `--strong` keeps the global N at 1024 instead.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # /opt/skills/guides/MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0


def alg_flops(N, T, H, S, A, d=512, n_enc=2, mode="rtg"):
    """Algorithmic (exactly pruned) FLOPs of one plan step, SURVEY.md section 8(d)."""
    idx = T - H
    Le = (idx + 1) + T
    Ld = 4 * T
    c, q, e = 24 * d * d, 4 * d, 2 * d * d
    nq = 2 * H
    feats = {"states": S, "actions": A}
    embed = 2 * d * (S * (idx + 1) + A * T)
    dbar = 1.0 if mode == "rtg" else (S + 1) / 2.0
    per_cand = (n_enc * (c * Le + q * Le * Le) + embed + Le * e + Le * 2 * e + nq * Ld * q + nq * e + nq * 16 * d * d
                + nq * (e + 2 * d * dbar))
    Le1 = (idx + 1) + idx + T
    pass1 = n_enc * (c * Le1 + q * Le1 * Le1) + Ld * e + (c * Ld + q * Ld * Ld) + T * (e + 2 * d * A)
    critic = 0 if mode == "rtg" else N * H * 2 * 2 * ((S + A) * 256 + 256 * 256 + 256)
    return N * per_cand + pass1 + critic


def alg_bytes(N, T, S, A, n_params=11_326_995):
    """Algorithmic HBM bytes of one plan step: bf16 weights once + window + eps + outputs."""
    return 2 * n_params + 4 * T * (S + A + 2) + 4 * N * T * A + 4 * N + 4 * N * A


def cpu_baseline(dims, cfg_kw, hist, rtg):
    """The oracle (fp32 PyTorch-CPU restatement of the reference path) timed on this host's cores: the full
    N-candidate plan step of the bench workload (value), and BASELINE config 1 (hopper N=64 H=8 T=16) beside it."""
    from m3pc_amd import synth
    from oracle import mtm_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = min(avail, 64)  # MKL stops scaling on these GEMM shapes well before 64 threads
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(dims, 0)
    stats = O.make_stats(synth.make_tokenizer_stats(dims, 0))
    N = cfg_kw["action_samples"]
    n_s = N  # the whole plan step (about 5 s on 64 threads at N = 1024): no extrapolation
    cfg = O.PlanCfg(dims.traj_length, cfg_kw["horizon"], n_s, 0.99, 0.01, 0.6)
    win, h = O.assemble_window(cfg, hist, 500, rtg)
    eps = synth.make_eps(N, dims, 1)[:n_s]
    small = O.PlanCfg(dims.traj_length, cfg_kw["horizon"], 32, 0.99, 0.01, 0.6)
    O.guiding(sd, stats, small, win, h, 0.6, eps[:32], "rtg")  # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    reps = 0
    while reps < 1 or (time.perf_counter() - t0 < 12.0 and reps < 4):
        O.guiding(sd, stats, cfg, win, h, 0.6, eps, "rtg")
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    # BASELINE configs[0]: hopper rtg_guiding N=64 H=8 T=16 (the reference's own CPU-runnable case)
    d1 = synth.Dims(dims.state_dim, dims.action_dim, 16)
    c1 = O.PlanCfg(16, 8, 64, 0.99, 0.01, 0.6)
    h1 = synth.make_history(d1, 0)
    w1, hh1 = O.assemble_window(c1, h1, 500, rtg)
    sd1, st1, e1 = synth.make_state_dict(d1, 0), O.make_stats(synth.make_tokenizer_stats(d1, 0)), synth.make_eps(64, d1, 1)
    O.guiding(sd1, st1, c1, w1, hh1, 0.6, e1, "rtg")
    t1 = time.perf_counter()
    r1 = 0
    while r1 < 3 or (time.perf_counter() - t1 < 3.0 and r1 < 20):
        O.guiding(sd1, st1, c1, w1, hh1, 0.6, e1, "rtg")
        r1 += 1
    dt1 = (time.perf_counter() - t1) / r1
    return {"value": 1.0 / dt, "unit": "plan-steps/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x the whole plan step ({N} candidates, rtg_guiding H={h} T={dims.traj_length}, fp32 torch-CPU "
                      f"oracle, {cores} threads, {dt:.2f}s each)",
            "config1": {"value": round(1.0 / dt1, 3), "unit": "plan-steps/s",
                        "sample": f"{r1} x hopper rtg_guiding N=64 H=8 T=16 ({1e3 * dt1:.0f} ms each)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--candidates", type=int, default=1024, help="candidates per GPU")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--traj-length", type=int, default=32)
    ap.add_argument("--rescore", default="bound", choices=["bound", "topk"],
                    help="fp32 re-score set: every candidate within 2*delta of the bf16 maximum (default) or a fixed top-k")
    ap.add_argument("--rescore-topk", type=int, default=16)
    ap.add_argument("--rescore-min", type=int, default=None, help="smallest bound-driven re-score set (default: the planner's)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / closed-loop / shipped-config side measurements")
    ap.add_argument("--strong", action="store_true", help="keep the global candidate count fixed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone-pass", action="store_true",
                    help="skip the extra instrumented pass with the candidate halves serialised (roofline.alone); "
                         "tools/profile_round.sh: every launch of the profiled command then runs as in the timed region")
    ap.add_argument("--env", default="hopper", choices=["hopper", "walker2d", "halfcheetah"],
                    help="state/action dims of the D4RL family (BASELINE configs 3-4 use walker2d / halfcheetah)")
    ap.add_argument("--guidance", default="rtg_guiding", choices=["rtg_guiding", "critic_lambda_guiding"])
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner
    import types

    S, A = synth.ENV_DIMS[args.env]
    critic_mode = args.guidance == "critic_lambda_guiding"
    T, H = args.traj_length, args.horizon
    n_global = args.candidates if args.strong else args.candidates * world
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=n_global, horizon=H, discount=0.99, temperature=0.01,
                                lmbda=0.6, plan_guidance=args.guidance)
    cfg.temperature = 1.0 if critic_mode else 0.01  # config.yaml:79
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)  # same seed on every rank: identical eps and multinomial draws
    qsd, om, os_ = synth.make_critic(dims, 0) if critic_mode else (None, None, None)
    planner = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_,
                         precision=args.precision, rescore_topk=args.rescore_topk, device=local_rank, generator=gen,
                         rescore=args.rescore, **({"rescore_min": args.rescore_min} if args.rescore_min else {}), group=torch.distributed.group.WORLD if world > 1 else None)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    states, actions, rewards, h, rtg = planner.assemble_window(hist, rtg=3.0)
    assert h == H

    def step():
        return planner._guide(capi.MODE_CRITIC if critic_mode else capi.MODE_RTG, states, actions, rewards, rtg, h, 0.6)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    n_re = []
    for _ in range(args.steps):
        step()
        n_re.append(planner.last.get("n_rescored", args.rescore_topk))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    units = args.steps * (n_global / float(args.candidates))
    value = units / elapsed

    # roofline of the dominant kernel class (the MFMA GEMM): a second, instrumented pass of the same K
    # steps with hipEvents around every GEMM launch on the launch stream (events perturb the step time,
    # so they are kept out of the timed region above).
    # per-step latency distribution: one event pair per step on the launch stream, outside the timed region
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 200))]
    for e0, e1 in pairs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
    latency = {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 4),
               "min": round(lat[0], 4), "steps": len(lat)}

    n_local = mdist_count(n_global, rank, world)
    prec = capi.PREC_BF16 if args.precision == "bf16" else capi.PREC_FP32

    def instrumented(mode):
        planner.handle.profile_enable(mode)
        planner.handle.profile_read(reset=True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        cls_ = planner.handle.profile_read(prec, reset=False)
        tail_ = planner.handle.profile_read(capi.PROF_LAYER_TAIL, reset=False)
        all_ = planner.handle.profile_read(-1, reset=True)
        planner.handle.profile_enable(False)
        return cls_, tail_, all_

    # as run: the two candidate halves overlapped on two streams exactly as in the timed region (a bracket also holds what
    # the other half does on the chip meanwhile) ...
    (launches, gemm_ms, gemm_flops), (t_launches, t_ms, t_flops), (all_launches, all_ms, all_flops) = instrumented(True)
    # ... and with the halves one after the other on one stream: the same launches, each alone on the chip
    (a_launches, a_ms, a_flops), (at_launches, at_ms, at_flops), _ = ((0, 0.0, 0.0), (0, 0.0, 0.0), None) if args.no_alone_pass else instrumented(2)
    peak = MFMA_PEAK_TFLOPS[args.precision]
    f_step = alg_flops(n_local, T, H, S, A, mode="critic" if critic_mode else "rtg")
    cls = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    if t_launches:   # the dominant kernel: the fused layer tail (block_fused.hip), ~40 % of the step's kernel time
        achieved, dom_l, dom_ms, dom_fl = t_flops / (t_ms * 1e-3) / 1e12, t_launches, t_ms, t_flops
        al_l, al_ms, al_fl = at_launches, at_ms, at_flops
        kname = ("m3pc::block_fused_kernel (layer tail: out-proj + residual + LayerNorm + Linear/GELU/Linear + residual + LayerNorm, "
                 "one launch per layer and candidate half)")
    else:            # fp32 mode / shapes the fused kernel does not cover: the GEMM class
        achieved, dom_l, dom_ms, dom_fl = cls, launches, gemm_ms, gemm_flops
        al_l, al_ms, al_fl = a_launches, a_ms, a_flops
        kname = f"the {args.precision} MFMA GEMM launches (gemm_line_kernel / gemm_glds_ring3_kernel / gemm_kernel)"
    alone = al_fl / (al_ms * 1e-3) / 1e12 if al_ms > 0 else None
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": pmc_traffic(args.precision),
                "peak_measured": 1800.0 if args.precision == "bf16" else None,  # register-resident v_mfma loop at the 1.75 GHz the chip holds
                "kernel": kname,
                "flops_per_launch": dom_fl / max(dom_l, 1), "avg_launch_us": 1e3 * dom_ms / max(dom_l, 1),
                "launches_per_step": dom_l / args.steps, "kernel_ms_per_step": dom_ms / args.steps,
                "note": "HIP-event brackets on the stream of each launch in an instrumented pass of the same K steps, run as the timed "
                        "region runs: the two candidate halves on two streams, so a launch shares the chip with the other half's "
                        "kernels for part of its bracket; `alone` = the same launches with the halves one after the other on one "
                        "stream (m3pc_profile_enable(h, 2)), each alone on the chip",
                "alone": None if alone is None else {"avg_launch_us": 1e3 * al_ms / max(al_l, 1), "achieved": round(alone, 2),
                                                      "frac": round(alone / peak, 4)},
                # every MFMA launch of the compute dtype (fused tails, fused decoder input, Q|K|V / head GEMMs)
                "mfma_class": {"achieved": round(cls, 2), "frac": round(cls / peak, 4), "launches_per_step": launches / args.steps,
                               "ms_per_step": gemm_ms / args.steps},
                "all_gemm_ms_per_step": all_ms / args.steps, "all_gemm_launches_per_step": all_launches / args.steps,
                "step_alg_tflop": round(f_step / 1e12, 4),
                "step_mfma_frac": round(f_step / (elapsed / args.steps) / 1e12 / peak, 4),
                "step_alg_bytes": alg_bytes(n_local, T, S, A),
                "step_hbm_frac_alg": round(alg_bytes(n_local, T, S, A) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}

    if rank == 0:
        out = {"metric": "MPC plan-steps/sec (N=1024, H=16, hopper-medium-v2)", "value": round(value, 2),
               "unit": "plan-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
               "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": args.precision,
               "data": "synthetic",
               "config": {"workload": f"{args.env}-medium-v2 shapes (S={S},A={A}) {args.guidance} N={args.candidates}/GPU H={H} "
                                      f"T={T} {args.precision} candidate pass + fp32 policy pass + fp32 re-score ("
                                      + (f"bound-driven set, {sum(n_re) / len(n_re):.1f} candidates on average" if args.rescore == "bound"
                                         else f"top-{args.rescore_topk}") + "); pipelined throughput, window resident in HBM",
                          "candidates_per_gpu": n_local, "global_candidates": n_global, "horizon": H, "traj_length": T,
                          "parallelism": f"candidate-shard x{world}"},
               "latency_ms": latency, "roofline": roofline,
               "rescore": {"mode": args.rescore, "n_mean": round(sum(n_re) / len(n_re), 2), "n_max": max(n_re),
                           "delta": planner.last.get("delta"), "min_margin_outside": planner.last.get("min_margin_outside")}}
        if world == 1 and not args.no_extras:
            try:
                out.update(extras(args, dims, cfg, hist, planner, S, A))
            except Exception as e:  # the side measurements never take the headline down with them
                out["extras_error"] = repr(e)[:300]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, dict(horizon=H, action_samples=args.candidates), hist, 3.0)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def _time_calls(fn, n, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return {"p50": round(ts[len(ts) // 2], 4), "min": round(ts[0], 4), "p99": round(ts[min(len(ts) - 1, int(0.99 * len(ts)))], 4), "calls": n}


def extras(args, dims, cfg, hist, planner, S, A):
    """Side measurements on rank 0 at 1 GPU, outside the timed region (each a few hundred milliseconds)."""
    import types

    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner

    out = {}
    # closed loop: what a rollout loop sees per env step -- host window assembly + H2D, the plan step, the action read back
    out["closed_loop"] = {"what": "action_sample(history, eval=True, rtg) incl. window H2D copy and .cpu() of the action, one call at a time",
                          "ms": _time_calls(lambda: planner.action_sample(hist, plan=True, eval=True, rtg=3.0).cpu(), 40)}
    out["closed_loop"]["steps_per_s"] = round(1e3 / out["closed_loop"]["ms"]["p50"], 2)
    if args.precision == "bf16":
        qsd, om, os_ = synth.make_critic(dims, 0) if "critic" in cfg.plan_guidance else (None, None, None)
        p32 = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_, precision="fp32",
                         generator=torch.Generator(device="cuda").manual_seed(1))
        s_, a_, r_, h_, rtg_ = p32.assemble_window(hist, rtg=3.0)
        mode = capi.MODE_CRITIC if "critic" in cfg.plan_guidance else capi.MODE_RTG
        for _ in range(2):
            p32._guide(mode, s_, a_, r_, rtg_, h_, 0.6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            p32._guide(mode, s_, a_, r_, rtg_, h_, 0.6)
        torch.cuda.synchronize()
        out["fp32_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / 8, 4)
        p32.handle.close()
    # the reference's shipped planning config (finetune_omtm/config.yaml:5,77-79): N=625, H=4, T=8 -- launch-latency-bound
    d8 = synth.Dims(S, A, 8)
    c8 = types.SimpleNamespace(traj_length=8, action_samples=625, horizon=4, discount=0.99, temperature=0.01, lmbda=0.6,
                               plan_guidance="rtg_guiding")
    h8 = synth.make_history(d8, 0)
    h8["path_length"] = 500
    ship = {}
    for prec in ("bf16", "fp32"):
        p8 = HipPlanner(c8, synth.make_state_dict(d8, 0), synth.make_tokenizer_stats(d8, 0), None, precision=prec,
                        generator=torch.Generator(device="cuda").manual_seed(1))
        ship[prec] = _time_calls(lambda: p8.action_sample(h8, plan=True, eval=True, rtg=3.0).cpu(), 40)
        if prec == "fp32":  # zero-shot goal reaching, one env per call (zeroshot_omtm/learner.py:151-261, config_hopper.yaml)
            ship["zeroshot_piid_B1"] = _time_calls(lambda: p8.action_piid_sample(h8, eval=True, rtg=2.5).cpu(), 40)
        p8.handle.close()
    out["latency_ms_shipped"] = {"config": "hopper rtg_guiding N=625 H=4 T=8 (closed loop, per call)", **ship}
    # batched multi-env planning: E windows x N candidates per call, pipelined like the headline
    if args.precision == "bf16" and cfg.plan_guidance == "rtg_guiding":
        bat = {}
        for E in (1, 4, 8):
            pb = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                            generator=torch.Generator(device="cuda").manual_seed(1), max_batch=E, max_windows=E)
            hs = []
            for i in range(E):
                hi = synth.make_history(dims, i)
                hi["path_length"] = 500
                hs.append(hi)
            for _ in range(3):
                pb.action_sample_batch(hs, eval=True, rtg=3.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 12
            for _ in range(reps):
                pb.action_sample_batch(hs, eval=True, rtg=3.0)
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t0) / reps
            bat[f"E{E}"] = {"ms_per_call": round(1e3 * dt_, 4), "plan_steps_per_s": round(E / dt_, 2)}
            pb.handle.close()
        out["batched"] = {"what": "action_sample_batch: E env windows x N=%d candidates per call (incl. window H2D copies)" % cfg.action_samples,
                          **bat}
    return out


def pmc_traffic(precision):
    """HBM bytes per launch of the dominant kernel (block_fused_kernel), from the committed rocprofv3 PMC passes of this same
    command (profiles/pmc_traffic.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950-corrected); PMC counters
    cannot be read from inside the process, so this is the launch-weighted mean of that profile, or null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if precision != "bf16" or not os.path.exists(path):
        return None
    try:
        ks = json.load(open(path))["kernels"]
    except Exception:
        return None
    n = b = 0
    for name, v in ks.items():
        if "block_fused_kernel" in name:
            n += v["launches_profiled"]
            b += v["launches_profiled"] * v["hbm_bytes_per_launch"]
    return int(b / n) if n else None


def mdist_count(n, rank, world):
    from m3pc_amd.dist import shard_range
    return shard_range(n, rank, world)[1]


if __name__ == "__main__":
    main()
