#!/usr/bin/env python3
"""bench.py -- MPC plan-steps/sec of the m3pc test-time planner on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher: starts `python -m torch.distributed.run --nproc-per-node N ... bench.py ...` itself (before
        anything touches a GPU), relays rank 0's JSON line and exits with the children's code.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one complete plan step of rtg_guiding (research/finetune_omtm/learner.py:271-327): draw eps, policy pass
(batch 1, fp32), sample N candidates, batched candidate pass (bf16), TD(lambda) scoring, [all-gather of the shards when the
candidates are sharded], certified fp32 re-score of the candidates that can still hold the arg-max (m3pc_amd/planner.py),
softmax / weighted mean / argmax, multinomial draw.  Inputs (window, weights) are resident in HBM when the timed region
starts.  SURVEY 8(d) defines the metric as completed plan_step calls per second with the result resident on the device:
the steps of the timed region are INDEPENDENT plan steps (the same window planned again and again, as before) issued
through HipPlanner's pipeline -- `--depth` of them in flight (default 2): step t+1's policy pass and step t-1's re-score +
select run on two more streams beside step t's candidate pass; every step's result is bit-identical to the serial call
(tests/test_pipeline_gpu.py).  `--depth 0` is the serial order; `latency_ms` (one step alone) and `closed_loop` stay serial.

Workload at 1 GPU: BASELINE configs[1] = hopper-medium-v2 shapes (S=11, A=3), rtg_guiding, N=1024 candidates, H=16, T=32
(=2H, the reference's shipped T/H ratio), bf16 candidate pass, synthetic data.

Multi-GPU (`--shard`):
  env (default)   every rank plans its own windows with all N=1024 candidates -- the metric's N, one environment per rank
                  (independent plan steps: no collective on the data path); value = G * K / t.  Weak scaling.
  candidates      the candidates of ONE plan step are sharded over the ranks with one RCCL all-gather of scores + first
                  actions per step (m3pc_amd/dist.py): weak (N = 1024 per rank, value counts 1024-candidate units) or
                  `--strong` (N = 1024 in total; `replicated_ms` tells what every rank repeats).
  `--config c4`   BASELINE configs[3]: halfcheetah shapes, rtg_guiding, N=16384, H=32, T=64, candidate-sharded over the
                  ranks (strong scaling: 16384 candidates in total), value = plan steps of 16384 candidates per second.
  `--config c5`   BASELINE configs[4]: zero-shot goal reaching, 8192 windows (64 environments x 1024 / 8 GPUs) per GPU and call
                  through the pruned many-window path, environment-sharded (no collective), value = windows/s.
With --shard env at N > 1 GPUs (the default multi-GPU run) the line also carries `c4` (BASELINE configs[3], candidate-sharded
through RCCL's all-gather on the same ranks, strong scaling), `c2_candidates_strong` (the headline shape with its 1024 candidates
sharded, `replicated_ms` beside it) and `rccl_ranks_seen`; these legs run behind the headline measurement under a wall-clock
watchdog (`--collective-timeout`): a leg that hangs costs its own entry ({"error": "timeout"}), never the headline.
At 1 GPU the line carries the legs `c5`, `c3` (walker2d critic N=4096) and `c4_shard` (2048 halfcheetah candidates at H=32/T=64).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time
from collections import deque

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # /opt/skills/guides/MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0



def usable_cpus():
    """CPUs this process can really run on at once: the affinity mask cut to the cgroup's CFS quota.  A GPU box of this pool shows 256
    CPUs and gives a job `cpu.max = 1600000 100000` = 16 of them; torch sizes its intra-op pool by the former (128 threads), and a
    pool that wakes -- its workers spin for a while behind every parallel CPU op -- burns the 100-ms quota of the whole process in
    ~12 ms: every thread, the one that feeds the GPU included, is then frozen until the period ends.  Seen as single 10-30 ms gaps
    inside 20-step legs (`step_gap_ms.max`)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2: "max 100000" | "1600000 100000"
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:                                                                           # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


class _NoGcPause:
    """No pass of Python's cyclic collector inside a timed region: on this process's heap (170 k tracked objects) a generation-2
    pass takes 30-40 ms -- measured (tools/stall_probe.py) as the one 20-step leg in four that came out a quarter slower than its
    neighbours.  The collector is switched off from the settle / warm-up steps to the end of the timed steps (what `timeit` does),
    after one full collection IN FRONT of the warm-up: a collection between warm-up and timed steps leaves the GPU idle for those
    30-40 ms, its clocks drop, and the first timed steps pay for the ramp (measured: 20 timed steps 845-862 -> 783-791 plan-steps/s)."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


def alg_flops(N, T, H, S, A, d=512, n_enc=2, mode="rtg"):
    """Algorithmic (exactly pruned) FLOPs of one plan step, SURVEY.md section 8(d)."""
    idx = T - H
    Le = (idx + 1) + T
    Ld = 4 * T
    c, q, e = 24 * d * d, 4 * d, 2 * d * d
    nq = 2 * H
    embed = 2 * d * (S * (idx + 1) + A * T)
    dbar = 1.0 if mode == "rtg" else (S + 1) / 2.0
    per_cand = (n_enc * (c * Le + q * Le * Le) + embed + Le * e + Le * 2 * e + nq * Ld * q + nq * e + nq * 16 * d * d
                + nq * (e + 2 * d * dbar))
    Le1 = (idx + 1) + idx + T
    pass1 = n_enc * (c * Le1 + q * Le1 * Le1) + Ld * e + (c * Ld + q * Ld * Ld) + T * (e + 2 * d * A)
    critic = 0 if mode == "rtg" else N * H * 2 * 2 * ((S + A) * 256 + 256 * 256 + 256)
    return N * per_cand + pass1 + critic


def alg_flops_goal(E, T, H, S, A, d=512, n_enc=2):
    """Algorithmic (exactly pruned) FLOPs of E zero-shot piid windows (zeroshot_omtm/learner.py:151-261): path inference under
    the pi mask with the states head on the rows the overlay reads, inverse dynamics under the fid mask with ONE action token.
    Same accounting as alg_flops: encoder layers on the kept tokens, decoder embedding + K|V of the kept tokens,
    Q of the un-masked query tokens, attention / out-proj / FFN / head on the query tokens only."""
    idx = T - H
    c, q, e = 24 * d * d, 4 * d, 2 * d * d
    Ld = 4 * T
    ns_a = T if idx == 0 else idx + 2          # pi mask: states 0..idx and T-1 (all of them when idx == 0)
    na = idx                                   # actions[:idx]
    nq_a = (idx + 1) + max(0, T - idx - 3)     # rows t <= idx and idx+2 .. T-2
    nq_a_unmasked = (idx + 1) if idx > 0 else nq_a
    le_a, le_b = ns_a + na, T + na
    pass_a = (n_enc * (c * le_a + q * le_a * le_a) + 2 * d * (S * ns_a + A * na) + le_a * 3 * e + nq_a_unmasked * e
              + nq_a * Ld * q + nq_a * e + nq_a * 16 * d * d + nq_a * (e + 2 * d * S))
    pass_b = (n_enc * (c * le_b + q * le_b * le_b) + 2 * d * (S * T + A * na) + le_b * 3 * e + Ld * q + e + 16 * d * d + 2 * 2 * d * A)
    return E * (pass_a + pass_b)


def goal_leg(local_rank, E, T=8, H=4, S=11, A=3, steps=20, warm=3, precisions=("bf16", "fp32"), world=1, fp32_steps=None):
    """BASELINE configs[4] on this GPU: E zero-shot goal-reaching windows (config_hopper: T=8, H=4; 64 environments x 1024 over
    8 GPUs = 8192 windows per GPU) per call through the pruned many-window path (m3pc_goal_step_batch), windows resident in
    HBM.  Per precision: ms per call (event-bracketed, the calls back to back), windows/s, the F_alg-based MFMA fraction."""
    import types

    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=1, horizon=H, discount=0.99, temperature=1.0, lmbda=0.6,
                                plan_guidance="rtg_guiding", index_jump=4)
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                   device=local_rank, goal_batch=E)
    g = torch.Generator().manual_seed(5 + int(os.environ.get("RANK", "0")))
    st = torch.randn((E, T, S), generator=g).cuda()
    ac = (torch.rand((E, T, A), generator=g) * 2 - 1).cuda()
    out_buf = (torch.empty((E, A), device="cuda"), torch.empty((E, A), device="cuda"))
    res = {"what": f"BASELINE configs[4]: zero-shot piid (pi mask -> overlay -> fid mask), hopper shapes T={T} H={H}, {E} windows per call"
                   f" (= 64 environments x 1024 / 8 GPUs), exactly pruned (path inference: {H + 2 if H < T else T} state rows of 4T; inverse "
                   "dynamics: one action token), windows resident in HBM",
           "alg_tflop_per_call": round(alg_flops_goal(E, T, H, S, A) / 1e12, 4), "windows": E}
    mus = {}
    for prec in precisions:
        pc = capi.PREC_BF16 if prec == "bf16" else capi.PREC_FP32
        n = steps if prec == "bf16" else (fp32_steps or max(3, steps // 4))
        for _ in range(warm):
            p.handle.goal_step_batch(st, ac, T - H, capi.GOAL_PIID, pc, out=out_buf)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            p.handle.goal_step_batch(st, ac, T - H, capi.GOAL_PIID, pc, out=out_buf)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        dt /= n
        mus[prec] = out_buf[0].clone()
        res[prec] = {"ms_per_call": round(1e3 * dt, 4), "windows_per_s": round(E / dt, 1), "calls": n,
                     "mfma_frac": round(alg_flops_goal(E, T, H, S, A) / dt / 1e12 / MFMA_PEAK_TFLOPS[prec], 4)}
    if "bf16" in mus and "fp32" in mus:
        res["bf16_vs_fp32_loc_err"] = float((mus["bf16"] - mus["fp32"]).abs().max())
    p.handle.close()
    return res


def alg_bytes(N, T, S, A, n_params=11_326_995):
    """Algorithmic HBM bytes of one plan step: bf16 weights once + window + eps + outputs."""
    return 2 * n_params + 4 * T * (S + A + 2) + 4 * N * T * A + 4 * N + 4 * N * A


def cpu_baseline(dims, cfg_kw, hist, rtg):
    """The oracle (fp32 PyTorch-CPU restatement of the reference path) timed on this host's cores: the full
    N-candidate plan step of the bench workload (value), and BASELINE config 1 (hopper N=64 H=8 T=16) beside it.
    Also returns the oracle's result for eps = synth.make_eps(N, dims, 1): the parity reference of the `parity` block."""
    from m3pc_amd import synth
    from oracle import mtm_oracle as O
    cores = min(usable_cpus(), 64)  # (what the cgroup lets run at once; MKL stops scaling on these GEMM shapes well before 64 threads)
    torch.set_num_threads(cores)
    sd = synth.make_state_dict(dims, 0)
    stats = O.make_stats(synth.make_tokenizer_stats(dims, 0))
    N = cfg_kw["action_samples"]
    cfg = O.PlanCfg(dims.traj_length, cfg_kw["horizon"], N, 0.99, 0.01, 0.6)
    win, h = O.assemble_window(cfg, hist, 500, rtg)
    eps = synth.make_eps(N, dims, 1)
    small = O.PlanCfg(dims.traj_length, cfg_kw["horizon"], 32, 0.99, 0.01, 0.6)
    O.guiding(sd, stats, small, win, h, 0.6, eps[:32], "rtg")  # warm-up (thread pools, allocator)
    t0 = time.perf_counter()
    reps = 0
    ref = None
    while reps < 1 or (time.perf_counter() - t0 < 12.0 and reps < 4):
        ref = O.guiding(sd, stats, cfg, win, h, 0.6, eps, "rtg")
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
    out = {"value": 1.0 / dt, "unit": "plan-steps/s", "cores": cores, "kind": "port",
           "sample": f"{reps} x the whole plan step ({N} candidates, rtg_guiding H={h} T={dims.traj_length}, fp32 torch-CPU "
                     f"oracle, {cores} threads, {dt:.2f}s each)",
           "config1": {"value": round(1.0 / dt1, 3), "unit": "plan-steps/s",
                       "sample": f"{r1} x hopper rtg_guiding N=64 H=8 T=16 ({1e3 * dt1:.0f} ms each)"}}
    return out, ref, eps


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(args):
    """`python bench.py --gpus N` as the driver runs it: start the N ranks as children of a fresh launcher process.  Nothing
    in THIS process has touched a GPU (importing torch does not), so no GPU-initialised process is replaced or forked."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if r.returncode != 0 or line is None:
        raise SystemExit(r.returncode or 1)
    raise SystemExit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--candidates", type=int, default=1024, help="candidates per plan step (per GPU when sharded weakly)")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--traj-length", type=int, default=32)
    ap.add_argument("--depth", type=int, default=3, help="independent plan steps in flight (0: the serial order)")
    ap.add_argument("--rescore", default="bound", choices=["bound", "topk"],
                    help="fp32 re-score set: certified (every candidate that can still hold the arg-max; default) or a fixed top-k")
    ap.add_argument("--rescore-topk", type=int, default=16)
    ap.add_argument("--rescore-min", type=int, default=None, help="smallest first re-score pass (default: the planner's)")
    ap.add_argument("--settle", type=int, default=24,
                    help="untimed plan steps right after the weights are loaded, before the W warm-up steps: the bf16 error bound of "
                         "the certified re-score is calibrated per weight load and rises over the first steps it sees (setup)")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / closed-loop / shipped-config / parity side measurements")
    ap.add_argument("--shard", default="env", choices=["env", "candidates"], help="what the ranks of a multi-GPU run divide")
    ap.add_argument("--strong", action="store_true", help="--shard candidates: keep the global candidate count fixed")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5"],
                    help="c4: BASELINE configs[3] (halfcheetah N=16384 H=32 T=64, candidate-sharded); c5: BASELINE configs[4] (zero-shot "
                         "goal reaching, 64 envs x 1024 = 8192 windows per GPU per call, environment-sharded, no collective)")
    ap.add_argument("--windows", type=int, default=8192, help="--config c5: zero-shot windows per GPU and call")
    ap.add_argument("--with-c4", action="store_true", help="(kept for old command lines: the collective legs now run by default)")
    ap.add_argument("--no-collective-legs", action="store_true",
                    help="N > 1 GPUs, --shard env: skip the candidate-sharded legs (c4, c2 strong) that run through RCCL's all-gather "
                         "behind the headline measurement")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="wall-clock limit of the collective legs: past it rank 0 prints the headline line with "
                         '"c4": {"error": "timeout"} and every rank exits with code 0')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alone-pass", action="store_true",
                    help="skip the extra instrumented pass with the candidate halves serialised (roofline.alone); "
                         "tools/profile_round.sh: every launch of the profiled command then runs as in the timed region")
    ap.add_argument("--env", default="hopper", choices=["hopper", "walker2d", "halfcheetah"],
                    help="state/action dims of the D4RL family (BASELINE configs 3-4 use walker2d / halfcheetah)")
    ap.add_argument("--guidance", default="rtg_guiding", choices=["rtg_guiding", "critic_lambda_guiding"])
    ap.add_argument("--serial-halves", action="store_true",
                    help="run the two candidate halves of a step one after the other on one stream (m3pc_profile_enable(h, 3)); with "
                         "--depth 0 every launch then runs alone on the chip: the arrangement of roofline.frac, for an external "
                         "kernel trace (profiles/r05_kernel_stats_alone.csv)")
    ap.add_argument("--no-certify-sample", action="store_true",
                    help="A/B switch: certify the arg-max only (round-4 behaviour), not the multinomial index of the sampled action")
    ap.add_argument("--chain-mode", default="alternate", choices=["alternate", "split"],
                    help="pipelined steps: a step's policy pass and re-score on the chain stream of its parity (alternate), or all "
                         "policy passes on one stream and all re-scores on the other (split, rounds 3-4)")
    ap.add_argument("--chain-priority", type=int, default=-1, help="stream priority of the two chain streams (-1: high, the default; 0: normal)")
    ap.add_argument("--policy-head", default="full", choices=["pruned", "full"],
                    help="policy pass: every row of the policy head (default) or the h sampled action tokens only, through the pruned decoder")
    ap.add_argument("--race-min", type=int, default=0, help="race entries of a first re-score pass (0: the planner's default)")
    ap.add_argument("--rescore-round", type=int, default=0, help="a first re-score pass's size is rounded up to a multiple of this (0: the planner's default)")
    ap.add_argument("--calibration-factor", type=float, default=0.0, help="delta = this x the calibration passes' largest deviation (0: the planner's default, 1.6)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU (tests/test_bench_launch_cpu.py): the ranks meet over gloo and rank 0 prints a stub line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(args)
    # the host threads of the GPU legs must never be frozen by this process's own CPU pool (usable_cpus): two CPUs stay free for the
    # thread that enqueues and the runtime's; cpu_baseline sizes the pool for itself
    torch.set_num_threads(max(1, min(usable_cpus(), 64) - 2))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.dry_run:
        import torch.distributed as dist
        out = {"metric": "MPC plan-steps/sec (N=1024, H=16, hopper-medium-v2)", "dry_run": True, "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup}
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.tensor([rank + 1.0])
            dist.all_reduce(t)
            assert float(t) == world * (world + 1) / 2
        if os.environ.get("M3PC_BENCH_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        if world > 1 and not args.no_collective_legs:
            finish_with_collective_legs(args, out, rank, local_rank, world)
            return
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner
    import types

    if args.config == "c5":
        return main_c5(args, rank, local_rank, world)
    if args.config == "c4":
        args.env, args.guidance, args.candidates, args.horizon, args.traj_length = "halfcheetah", "rtg_guiding", 16384, 32, 64
        args.shard, args.strong = "candidates", True
    S, A = synth.ENV_DIMS[args.env]
    critic_mode = args.guidance == "critic_lambda_guiding"
    T, H = args.traj_length, args.horizon
    shard_cand = world > 1 and args.shard == "candidates"
    n_global = args.candidates * world if (shard_cand and not args.strong) else args.candidates
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=n_global, horizon=H, discount=0.99, temperature=0.01,
                                lmbda=0.6, plan_guidance=args.guidance)
    cfg.temperature = 1.0 if critic_mode else 0.01  # config.yaml:79
    gen = torch.Generator(device="cuda")
    # candidate sharding: the same seed on every rank (identical eps and multinomial draws); env sharding: every rank its own
    gen.manual_seed(1 if (shard_cand or world == 1) else 1 + rank)
    qsd, om, os_ = synth.make_critic(dims, 0) if critic_mode else (None, None, None)
    group = torch.distributed.group.WORLD if shard_cand else None
    planner = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_,
                         precision=args.precision, rescore_topk=args.rescore_topk, device=local_rank, generator=gen,
                         rescore=args.rescore, **({"rescore_min": args.rescore_min} if args.rescore_min else {}), group=group,
                         certify_sample=not args.no_certify_sample, **({"race_min": args.race_min} if args.race_min else {}),
                         chain_mode=args.chain_mode, policy_head=args.policy_head, chain_priority=args.chain_priority,
                         **({"calibration_factor": args.calibration_factor} if args.calibration_factor else {}),
                         **({"rescore_round": args.rescore_round} if args.rescore_round else {}),
                         pipeline_depth=max(1, min(args.depth, capi.SLOTS - 1)))
    hist = synth.make_history(dims, 0 if (shard_cand or world == 1) else rank)  # env sharding: every rank its own environment
    hist["path_length"] = 500
    states, actions, rewards, h, rtg = planner.assemble_window(hist, rtg=3.0)
    assert h == H
    mode = capi.MODE_CRITIC if critic_mode else capi.MODE_RTG
    depth = max(0, min(args.depth, capi.SLOTS - 1))

    if args.serial_halves:
        planner.handle.profile_enable(3)

    def step():  # one plan step alone, serial
        return planner._guide(mode, states, actions, rewards, rtg, h, 0.6)

    n_re = []

    def run(k, record=False):
        """k independent plan steps, `depth` of them in flight."""
        if depth == 0:
            for _ in range(k):
                step()
                if record:
                    n_re.append(planner.last.get("n_rescored", args.rescore_topk) + planner.last.get("n_race", 0))
            return
        flight = deque()
        for _ in range(k):
            flight.append(planner._issue(mode, states, actions, rewards, rtg, h, 0.6, pipelined=True, inputs_ready=True))
            if len(flight) > depth:
                tk = flight.popleft()
                tk.pair()
                if record:
                    n_re.append(tk.info.get("n_rescored", args.rescore_topk) + tk.info.get("n_race", 0))
        while flight:
            tk = flight.popleft()
            tk.pair()
            if record:
                n_re.append(tk.info.get("n_rescored", args.rescore_topk) + tk.info.get("n_race", 0))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with _NoGcPause():
        run(max(args.settle, planner._cal_windows + capi.SLOTS))  # per-weight-load setup (calibration passes included): lets the re-score's error bound settle (a raise repeats that step's merge + select)
        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        run(args.steps, record=True)
        barrier()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if shard_cand:
        units = args.steps * (n_global / float(args.candidates))  # weak: 1024-candidate units; strong / c4: plan steps
    else:
        units = args.steps * world                                # every rank completed K plan steps of N candidates
    value = units / elapsed

    # per-step latency distribution: one step alone, serial, one event pair per step, outside the timed region
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 200))]
    for e0, e1 in pairs:
        e0.record()
        step()
        e1.record()
    torch.cuda.synchronize()
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
    latency = {"p50": round(lat[len(lat) // 2], 4), "p99": round(lat[min(len(lat) - 1, int(0.99 * len(lat)))], 4),
               "min": round(lat[0], 4), "steps": len(lat), "what": "one plan step alone (serial order), device time"}

    n_local = mdist_count(n_global, rank, world) if shard_cand else n_global
    prec = capi.PREC_BF16 if args.precision == "bf16" else capi.PREC_FP32

    # roofline of the dominant kernel: a further, instrumented pass of the same K steps with hipEvents around every MFMA launch
    # on the stream it runs on (events perturb the step time, so they are kept out of the timed region above)
    def instrumented(pmode):
        planner.handle.profile_enable(pmode)
        planner.handle.profile_read(reset=True)
        run(args.steps)
        torch.cuda.synchronize()
        cls_ = planner.handle.profile_read(prec, reset=False)
        tail_ = planner.handle.profile_read(capi.PROF_LAYER_TAIL, reset=False)
        all_ = planner.handle.profile_read(-1, reset=True)
        planner.handle.profile_enable(3 if args.serial_halves else False)
        return cls_, tail_, all_

    # as run: the steps pipelined and the two candidate halves on two streams exactly as in the timed region (a bracket also
    # holds what the other streams do on the chip meanwhile) ...
    (launches, gemm_ms, gemm_flops), (t_launches, t_ms, t_flops), (all_launches, all_ms, all_flops) = instrumented(True)
    # ... and serial with the halves one after the other on one stream: the same launches, each alone on the chip
    if args.no_alone_pass:
        (a_launches, a_ms, a_flops), (at_launches, at_ms, at_flops) = (0, 0.0, 0.0), (0, 0.0, 0.0)
    else:
        d_keep, depth = depth, 0
        (a_launches, a_ms, a_flops), (at_launches, at_ms, at_flops), _ = instrumented(2)
        depth = d_keep
    peak = MFMA_PEAK_TFLOPS[args.precision]
    f_step = alg_flops(n_local, T, H, S, A, mode="critic" if critic_mode else "rtg")
    cls = gemm_flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    if t_launches:   # the dominant kernel: the fused layer tail (block_fused.hip)
        achieved, dom_l, dom_ms, dom_fl = t_flops / (t_ms * 1e-3) / 1e12, t_launches, t_ms, t_flops
        al_l, al_ms, al_fl = at_launches, at_ms, at_flops
        kname = ("m3pc::block_fused_kernel (layer tail: out-proj + residual + LayerNorm + Linear/GELU/Linear + residual + LayerNorm "
                 "[+ the next encoder layer's Q|K|V projection | + the two scalar output heads], one launch per layer and "
                 "candidate half)")
    else:            # fp32 mode / shapes the fused kernel does not cover: the GEMM class
        achieved, dom_l, dom_ms, dom_fl = cls, launches, gemm_ms, gemm_flops
        al_l, al_ms, al_fl = a_launches, a_ms, a_flops
        kname = f"the {args.precision} MFMA GEMM launches (gemm_line_kernel / gemm_glds_ring3_kernel / gemm_kernel)"
    alone = al_fl / (al_ms * 1e-3) / 1e12 if al_ms > 0 else None
    step_s = elapsed / args.steps
    # `frac` is the kernel's own efficiency: every launch ALONE on the chip (serial order, the halves one after the other on one
    # stream, m3pc_profile_enable(h, 2)) -- the figure whose launches-per-step x duration fits inside ms_per_step and that
    # profiles/r05_kernel_stats_alone.csv reproduces.  `as_run` holds the brackets of the same launches as the timed region runs
    # (steps pipelined, halves on two streams): concurrent brackets overlap, so their sum exceeds the step -- sharing, not efficiency.
    as_run = {"avg_launch_us": 1e3 * dom_ms / max(dom_l, 1), "achieved": round(achieved, 2), "frac": round(achieved / peak, 4),
              "kernel_ms_per_step": dom_ms / args.steps,
              "note": "HIP-event brackets as the timed region runs: steps pipelined, the two candidate halves on two streams -- a bracket "
                      "also holds what the other streams run on the chip meanwhile; brackets of concurrent launches overlap (their sum "
                      "may exceed ms_per_step)"}
    if alone is not None:
        top = {"achieved": round(alone, 2), "frac": round(alone / peak, 4), "avg_launch_us": 1e3 * al_ms / max(al_l, 1),
               "kernel_ms_per_step": al_ms / args.steps, "arrangement": "alone: serial order, candidate halves one after the other on one stream"}
    else:
        top = {"achieved": as_run["achieved"], "frac": as_run["frac"], "avg_launch_us": as_run["avg_launch_us"],
               "kernel_ms_per_step": as_run["kernel_ms_per_step"], "arrangement": "as run (--no-alone-pass)"}
    roofline = {"bound": "mfma", "achieved": top["achieved"], "peak": peak, "unit": "TFLOP/s",
                "frac": top["frac"], "traffic": pmc_traffic(args.precision),
                "peak_measured": 1800.0 if args.precision == "bf16" else None,  # register-resident v_mfma loop at the 1.75 GHz the chip holds
                "kernel": kname,
                "flops_per_launch": dom_fl / max(dom_l, 1), "avg_launch_us": top["avg_launch_us"],
                "launches_per_step": dom_l / args.steps, "kernel_ms_per_step": top["kernel_ms_per_step"],
                "arrangement": top["arrangement"],
                "note": "HIP-event brackets on the stream of each launch in instrumented passes of the same K steps (outside the timed "
                        "region): achieved = algorithmic flops per launch / average launch duration with every launch alone on the chip",
                "as_run": as_run,
                "alone": None if alone is None else {"avg_launch_us": 1e3 * al_ms / max(al_l, 1), "achieved": round(alone, 2),
                                                      "frac": round(alone / peak, 4)},
                # every MFMA launch of the compute dtype (fused tails, fused decoder input, Q|K|V / head GEMMs)
                "mfma_class": {"achieved": round(cls, 2), "frac": round(cls / peak, 4), "launches_per_step": launches / args.steps,
                               "ms_per_step": gemm_ms / args.steps},
                "all_gemm_ms_per_step": all_ms / args.steps, "all_gemm_launches_per_step": all_launches / args.steps,
                "step_alg_tflop": round(f_step / 1e12, 4),
                "step_mfma_frac": round(f_step / step_s / 1e12 / peak, 4),  # per GPU: this rank's flops over this rank's step time
                "step_alg_bytes": alg_bytes(n_local, T, S, A),
                "step_hbm_frac_alg": round(alg_bytes(n_local, T, S, A) / step_s / 1e9 / HBM_PEAK_GBS, 5)}

    out = None
    if rank == 0:
        if shard_cand:
            par = f"candidate-shard x{world}" + (" (strong)" if args.strong else " (weak)")
        else:
            par = f"env-shard x{world} (one environment per GPU, no collective)" if world > 1 else "1 GPU"
        flight_txt = (f"{depth} independent plan steps in flight (policy pass / candidate pass / re-score of neighbouring steps on "
                      f"different streams)") if depth else "serial order"
        metric = "MPC plan-steps/sec (N=1024, H=16, hopper-medium-v2)" + (
            f"; independent plan steps, {depth} in flight" if depth else "; one step at a time (serial)")
        if args.config == "c4":
            metric = "MPC plan-steps/sec (N=16384, H=32, halfcheetah-medium-expert-v2, candidate-sharded)"
        out = {"metric": metric, "value": round(value, 2),
               "unit": "plan-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
               "scaling": "strong" if (shard_cand and args.strong) else "weak", "vs_baseline": None, "dtype": args.precision,
               "data": "synthetic",
               "config": {"workload": f"{args.env}-medium-v2 shapes (S={S},A={A}) {args.guidance} N={args.candidates}"
                                      + ("/GPU" if (shard_cand and not args.strong) else "") + f" H={H} "
                                      f"T={T} {args.precision} candidate pass + fp32 policy pass + fp32 re-score ("
                                      + (f"arg-max and multinomial draw certified under the calibrated bound delta, {sum(n_re) / max(len(n_re), 1):.1f} candidates "
                                          f"re-scored on average" if args.rescore == "bound" and not args.no_certify_sample
                                         else f"arg-max certified under the calibrated bound delta, {sum(n_re) / max(len(n_re), 1):.1f} candidates on average"
                                         if args.rescore == "bound"
                                         else f"top-{args.rescore_topk}") + f"); {flight_txt}; window resident in HBM",
                          "candidates_per_gpu": n_local, "global_candidates": n_global, "horizon": H, "traj_length": T,
                          "steps_in_flight": depth, "settle_steps": args.settle, "parallelism": par},
               # what `value` is and is not: the rate of INDEPENDENT plan steps with `steps_in_flight` of them in
               # flight; one step alone -- what a single environment's loop can use -- is serial_steps_per_s
               "serial_steps_per_s": round(1e3 / latency["p50"], 2),
               "latency_ms": latency, "roofline": roofline,
               "rescore": {"mode": args.rescore, "n_mean": round(sum(n_re) / max(len(n_re), 1), 2), "n_max": max(n_re) if n_re else None,
                           "delta": planner.last.get("delta"), "delta_grown": planner.delta_grown,
                           "min_margin_outside": planner.last.get("min_margin_outside")}}
        if shard_cand and args.strong:
            # what every rank repeats whatever the shard size: the policy pass and the re-score + select
            out["replicated_ms"] = round(latency["p50"] - candidate_only_ms(planner, mode, states, actions, rewards, h), 4)
        if world == 1 and not args.no_extras:
            try:
                out.update(extras(args, dims, cfg, hist, planner, S, A))
            except Exception as e:  # the side measurements never take the headline down with them
                out["extras_error"] = repr(e)[:300]
        if world == 1 and not args.no_cpu_baseline and args.config == "c2":
            base, ref, eps = cpu_baseline(dims, dict(horizon=H, action_samples=args.candidates), hist, 3.0)
            out["cpu_baseline"] = base
            if not args.no_extras and args.guidance == "rtg_guiding":
                try:
                    out["parity"] = parity(args, dims, cfg, hist, ref, eps)
                except Exception as e:
                    out["parity_error"] = repr(e)[:300]
    if world > 1 and not shard_cand and args.config == "c2" and not args.no_collective_legs:
        planner.handle.close()
        finish_with_collective_legs(args, out, rank, local_rank, world)
        return
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def finish_with_collective_legs(args, out, rank, local_rank, world):
    """Behind the headline measurement of an environment-sharded multi-GPU run: the candidate-sharded legs (RCCL all-gather
    on the data path) under a wall-clock watchdog, then rank 0's ONE JSON line.  A leg that hangs or fails costs its own
    entry, never the headline: on timeout rank 0 prints the line with {"error": "timeout"} and every rank exits with code 0."""
    store = None
    try:
        store = torch.distributed.distributed_c10d._get_default_store()
    except Exception:
        pass

    def peer_failure():
        """The message a failing rank left in the rendezvous store (a rank that raises before a collective would
        otherwise leave the others waiting in it until the watchdog's limit, and its message would be lost)."""
        try:
            if store is not None and store.check(["m3pc_leg_fail"]):
                return store.get("m3pc_leg_fail").decode()[:300]
        except Exception:
            pass
        return None

    def on_timeout(reason="timeout"):
        if rank == 0:
            for leg in ("c4", "c4_pipelined", "c2_candidates_strong"):
                out[leg] = {"error": reason}
            out["collective_legs_ok"] = False
            out["collective_timeout_s"] = args.collective_timeout
            print(json.dumps(out), flush=True)

    def legs():
        try:
            res = collective_legs(args, rank, local_rank, world)
        except Exception as e:
            msg = f"rank {rank}: {e!r}"[:300]
            try:
                if store is not None:
                    store.set("m3pc_leg_fail", msg)
            except Exception:
                pass
            res = {"c4": {"error": msg}, "collective_legs_ok": False}
            return res  # (no barrier: the other ranks leave through their watchdogs as soon as they see the message)
        torch.distributed.barrier()
        return res

    res = run_guarded(legs, args.collective_timeout, on_timeout, abort_reason=peer_failure)
    if rank == 0:
        out.update(res)
        out.setdefault("collective_legs_ok", all("error" not in v for v in res.values() if isinstance(v, dict)))
        print(json.dumps(out), flush=True)
    run_guarded(torch.distributed.destroy_process_group, 30.0, lambda *a: None)


def main_c5(args, rank, local_rank, world):
    """--config c5: BASELINE configs[4], zero-shot goal reaching.  64 environments x 1024 windows over 8 GPUs = 8192 windows per
    GPU and call; the windows are independent, so the ranks share nothing (environment sharding, no collective): weak scaling,
    value = windows planned per second over all ranks."""
    E = args.windows
    T, H, S, A = 8, 4, 11, 3   # zeroshot_omtm/config_hopper.yaml:6,78
    precs = (args.precision,) if args.no_extras else (("bf16", "fp32") if args.precision == "bf16" else ("fp32",))
    res = goal_leg(local_rank, E, T, H, S, A, steps=args.steps, warm=max(args.warmup, 1), precisions=precs, world=world,
                   fp32_steps=max(3, args.steps // 8))
    if rank == 0:
        r = res[args.precision]
        out = {"metric": "zero-shot goal-reaching windows/sec (zeroshot_omtm config_hopper, 64 envs x N=1024)",
               "value": round(world * r["windows_per_s"], 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": r["ms_per_call"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.precision, "data": "synthetic",
               "config": {"workload": res["what"], "windows_per_gpu": E, "traj_length": T, "horizon": H,
                          "parallelism": f"env-shard x{world} (no collective)" if world > 1 else "1 GPU"},
               "roofline": {"bound": "mfma", "achieved": round(res["alg_tflop_per_call"] / (r["ms_per_call"] * 1e-3), 2),
                            "peak": MFMA_PEAK_TFLOPS[args.precision], "unit": "TFLOP/s", "frac": r["mfma_frac"], "traffic": None,
                            "kernel": "whole call (two chained pruned forwards): F_alg / wall time per call"},
               "c5": res}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def c4_full_leg(local_rank, steps=6):
    """BASELINE configs[3] at its full size on ONE GPU: halfcheetah shapes, rtg_guiding, N=16384, H=32, T=64 (about 60 GB of
    workspace): the single-GPU figure a candidate-sharded N-rank run of the same step divides by.  Serial and pipelined."""
    ser = plan_leg(local_rank, "halfcheetah", "rtg_guiding", 16384, 32, 64, steps=steps, warm=2, settle=4, depth=0)
    pip = plan_leg(local_rank, "halfcheetah", "rtg_guiding", 16384, 32, 64, steps=steps, warm=2, settle=4, depth=3,
                   what="BASELINE configs[3] on ONE GPU: halfcheetah shapes rtg_guiding, all N=16384 candidates, H=32 T=64 bf16")
    pip["serial_ms_per_step"] = ser["ms_per_step"]
    return pip


def plan_leg(local_rank, env, guidance, N, H, T, steps=20, warm=4, settle=12, depth=3, group=None, what=""):
    """Another BASELINE plan-step shape on this GPU (or, with `group`, candidate-sharded over the ranks of this run): `steps`
    plan steps, `depth` in flight (0: serial -- a sharded step gathers on the current stream), with the F_alg-based MFMA fraction."""
    import types

    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner
    from m3pc_amd.dist import shard_range, world_info
    S, A = synth.ENV_DIMS[env]
    critic = guidance == "critic_lambda_guiding"
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=1.0 if critic else 0.01,
                                lmbda=0.6, plan_guidance=guidance)
    qsd, om, os_ = synth.make_critic(dims, 0) if critic else (None, None, None)
    gen = torch.Generator(device="cuda").manual_seed(1)
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_, precision="bf16",
                   device=local_rank, generator=gen, group=group, pipeline_depth=max(1, min(depth, capi.SLOTS - 1)))
    rank, world = world_info(group)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    s_, a_, r_, h, rtg = p.assemble_window(hist, rtg=3.0)
    mode = capi.MODE_CRITIC if critic else capi.MODE_RTG

    stamps = []  # host time at which each step's result was in hand (a leg that comes out slow: every step, or one stall?)

    def run(k):
        if depth == 0:
            for _ in range(k):
                p._guide(mode, s_, a_, r_, rtg, h, 0.6)
            return
        flight = deque()
        for _ in range(k):
            flight.append(p._issue(mode, s_, a_, r_, rtg, h, 0.6, pipelined=True, inputs_ready=True))
            if len(flight) > depth:
                flight.popleft().pair()
                stamps.append(time.perf_counter())
        while flight:
            flight.popleft().pair()
            stamps.append(time.perf_counter())

    with _NoGcPause():
        run(max(settle, p._cal_windows + capi.SLOTS))  # (every calibration pass of the weight load behind us before anything is timed)
        run(warm)
        if world > 1:
            torch.distributed.barrier(group)
        torch.cuda.synchronize()
        del stamps[:]
        t0 = time.perf_counter()
        run(steps)
        if world > 1:
            torch.distributed.barrier(group)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    gaps = sorted(1e3 * (b - a) for a, b in zip(stamps, stamps[1:]))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=group)
        dt = float(t.item())
    ms = 1e3 * dt / steps
    n_local = shard_range(N, rank, world)[1]
    f_rank = alg_flops(n_local, T, H, S, A, mode="critic" if critic else "rtg")
    out = {"what": what or f"{env} shapes (S={S},A={A}) {guidance} N={N} H={H} T={T} bf16",
           "ms_per_step": round(ms, 4), "plan_steps_per_s": round(1e3 / ms, 2), "steps": steps, "steps_in_flight": depth,
           "alg_tflop_per_step_per_gpu": round(f_rank / 1e12, 4), "mfma_frac": round(f_rank / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["bf16"], 4),
           "argmax": int(p.last["argmax"].item()), "n_rescored": p.last.get("n_rescored")}
    if gaps:  # (between consecutive results on the host: the median is the step, the maximum shows a stall)
        out["step_gap_ms"] = {"median": round(gaps[len(gaps) // 2], 4), "max": round(gaps[-1], 4)}
    if world > 1:
        out["candidates_per_gpu"] = n_local
        out["replicated_ms"] = round(_replicated_ms(p, mode, s_, a_, r_, rtg, h), 4)
    p.handle.close()
    return out


def _replicated_ms(planner, mode, states, actions, rewards, rtg, h):
    """What every rank of a candidate-sharded step repeats whatever the shard size (policy pass, re-score + select): one
    serial step minus the sharded part alone."""
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for e0, e1 in pairs:
        e0.record()
        planner._guide(mode, states, actions, rewards, rtg, h, 0.6)
        e1.record()
    torch.cuda.synchronize()
    lat = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
    return lat[len(lat) // 2] - candidate_only_ms(planner, mode, states, actions, rewards, h)


def rccl_ranks_seen(group=None):
    """The world size as observed INSIDE a collective of the data path's kind (all_gather_into_tensor on the compute stream):
    every rank contributes its rank id; the count of distinct ids that arrived."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([float(dist.get_rank(group))], device=dev)
    out = torch.full((world,), -1.0, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return int(torch.unique(out[out >= 0]).numel())


def collective_legs(args, rank, local_rank, world):
    """The candidate-sharded configurations of a multi-GPU run, through RCCL's all-gather on the data path (north_star:
    "candidate batches shard across the GPUs with an RCCL all-gather of per-shard scores"): BASELINE configs[3] (halfcheetah
    N=16384 H=32 T=64, strong scaling) and the headline shape with its 1024 candidates sharded (strong; `replicated_ms` is what
    caps it).  Serial steps: the gather sits on the step's path."""
    grp = torch.distributed.group.WORLD
    res = {"rccl_ranks_seen": rccl_ranks_seen(grp)}
    if args.dry_run:
        if os.environ.get("M3PC_BENCH_HANG_LEG"):
            time.sleep(3600)
        if os.environ.get("M3PC_BENCH_FAIL_LEG_RANK") == str(rank):  # (one rank raises before the collective: the others must not wait)
            raise RuntimeError("injected leg failure")
        if os.environ.get("M3PC_BENCH_FAIL_LEG_RANK"):
            time.sleep(3600)  # (stands for the collective the failed rank never enters)
        res["c4"] = {"dry_run": True}
        res["c4_pipelined"] = {"dry_run": True}
        res["c4_full"] = {"dry_run": True}
        res["c2_candidates_strong"] = {"dry_run": True}
        return res
    res["c4"] = plan_leg(local_rank, "halfcheetah", "rtg_guiding", 16384, 32, 64, steps=12, warm=3, settle=6, depth=0, group=grp,
                         what=f"BASELINE configs[3]: halfcheetah shapes rtg_guiding N=16384 H=32 T=64 bf16, candidates sharded over {world} "
                              f"ranks, one RCCL all-gather of scores + first actions per step, serial steps (strong scaling)")
    # the same sharded step with two independent plan steps in flight (the all-gather stays on the current stream, the policy
    # pass and the re-score + select of the neighbouring steps on the chain streams, as at one rank): the 1-GPU yardsticks are
    # pipelined, so this is the leg a scaling figure has to be read from
    res["c4_pipelined"] = plan_leg(local_rank, "halfcheetah", "rtg_guiding", 16384, 32, 64, steps=12, warm=3, settle=6, depth=3, group=grp,
                                   what=f"BASELINE configs[3] as `c4`, three independent plan steps in flight")
    # the denominator, same definition, same run: the WHOLE N=16384 step on ONE GPU (rank 0; the other ranks wait at the barrier)
    if rank == 0:
        try:
            res["c4_full"] = c4_full_leg(local_rank)
        except Exception as e:
            res["c4_full"] = {"error": repr(e)[:300]}
    torch.distributed.barrier(grp)
    if rank == 0 and "error" not in res["c4_full"]:
        for leg, key in (("c4", "serial_ms_per_step"), ("c4_pipelined", "ms_per_step")):
            if "error" not in res[leg]:
                res[leg]["scaling_eff"] = round(res["c4_full"][key] / (world * res[leg]["ms_per_step"]), 4)
                res[leg]["scaling_eff_what"] = f"c4_full.{key} (one GPU, all 16384 candidates) / ({world} x this leg's ms_per_step)"
    res["c2_candidates_strong"] = plan_leg(local_rank, "hopper", "rtg_guiding", 1024, 16, 32, steps=20, warm=4, settle=12, depth=0, group=grp,
                                           what=f"the headline shape with its 1024 candidates sharded over {world} ranks (strong), one "
                                                "RCCL all-gather per step, serial steps")
    return res


def run_guarded(fn, timeout_s, on_timeout, abort_reason=None):
    """fn() with a watchdog: when it has not returned after timeout_s seconds (a collective that never completes cannot be
    interrupted from Python) -- or as soon as abort_reason() returns a message (a peer rank's failure) -- on_timeout(reason)
    runs on the watchdog thread and the process ends with exit code 0: the headline measurement is done by then and must not
    be lost with the side legs (the line carries "collective_legs_ok": false for whoever reads it)."""
    import threading
    done = threading.Event()

    def watch():
        t_end = time.monotonic() + timeout_s
        reason = None
        while not done.wait(1.0):
            reason = abort_reason() if abort_reason is not None else None
            if reason is not None or time.monotonic() >= t_end:
                break
        if done.is_set():
            return
        try:
            on_timeout(reason or "timeout")
        finally:
            sys.stdout.flush()
            os._exit(0)

    threading.Thread(target=watch, daemon=True).start()
    try:
        return fn()
    finally:
        done.set()


def candidate_only_ms(planner, mode, states, actions, rewards, h):
    """Device time of the sharded part of a step alone (sample + candidate pass of this rank's shard)."""
    from m3pc_amd import dist as mdist
    cfg = planner.cfg
    N = int(cfg.action_samples)
    eps = planner._eps((N, 1, planner.T, 1, planner.A)).reshape(N, -1, planner.A)
    planner.handle.policy_pass(mode, states, actions, rewards, h, 3.0, slot=0)
    begin, count = mdist.shard_range(N, planner.rank, planner.world)
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        planner.handle.candidate_pass(mode, states, actions, rewards, eps, h, 0.6, float(cfg.discount), N, begin, count,
                                      precision=planner.precision, slot=0)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


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


def parity(args, dims, cfg, hist, ref, eps):
    """What the planner returns against the oracle's fp32 result on the bench workload (C2), for the bf16 headline mode and
    for fp32 end to end: the same eps (make_eps seed 1), the multinomial drawn from the same generator state."""
    from m3pc_amd import capi, synth
    from m3pc_amd.planner import HipPlanner
    out = {"reference": "oracle/mtm_oracle.py (fp32 PyTorch-CPU restatement, pinned to the reference by tests/golden), eps = make_eps(N, dims, 1)"}
    scale = float(ref["expect_return"].abs().max())
    for prec in ("bf16", "fp32"):
        gen = torch.Generator(device="cuda").manual_seed(1234)
        p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision=prec, generator=gen)
        s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
        p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6, eps=eps.cuda())  # (first step: calibrates delta)
        state = gen.get_state()
        sa, ev = p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6, eps=eps.cuda())
        gen.set_state(state)
        idx_ref = int(torch.multinomial(ref["p"].cuda(), 1, generator=gen).item())  # torch's own draw from the ORACLE's p, same variates
        er = p.last["expect_return"].cpu()
        out[prec] = {"argmax_match": int(p.last["argmax"].item()) == int(ref["argmax"]),
                     "eval_action_err": float((ev.cpu() - ref["eval_action"]).abs().max()),
                     "sample_idx_match": int(p.last["sample_idx"].item()) == idx_ref,
                     "expect_return_err_over_scale": float((er - ref["expect_return"]).abs().max()) / scale,
                     "p_err": float((p.last["p"].cpu() - ref["p"]).abs().max()),
                     "n_rescored": p.last.get("n_rescored")}
        p.handle.close()
    return out


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
        for _ in range(p8._cal_windows + capi.SLOTS if prec == "bf16" else 0):  # (the weight load's calibration passes, untimed)
            p8.action_sample(h8, plan=True, eval=True, rtg=3.0)
        ship[prec] = _time_calls(lambda: p8.action_sample(h8, plan=True, eval=True, rtg=3.0).cpu(), 40)
        if prec == "bf16":  # several environments in flight at the shipped config (rollout.evaluate_plan's pattern)
            hs8 = [dict(synth.make_history(d8, i), path_length=500) for i in range(8)]
            for _ in range(3):
                p8.action_sample_batch(hs8, eval=True, rtg=3.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                p8.action_sample_batch(hs8, eval=True, rtg=3.0)
            torch.cuda.synchronize()
            ship["bf16_pipelined_E8_ms_per_step"] = round(1e3 * (time.perf_counter() - t0) / 80, 4)
        if prec == "fp32":  # zero-shot goal reaching, one env per call (zeroshot_omtm/learner.py:151-261, config_hopper.yaml)
            ship["zeroshot_piid_B1"] = _time_calls(lambda: p8.action_piid_sample(h8, eval=True, rtg=2.5).cpu(), 40)
        p8.handle.close()
    out["latency_ms_shipped"] = {"config": "hopper rtg_guiding N=625 H=4 T=8 (closed loop, per call)", **ship}
    # batched multi-env planning: E windows x N candidates per call, incl. host window assembly and H2D copies
    if args.precision == "bf16" and cfg.plan_guidance == "rtg_guiding":
        bat = {}
        for E in (1, 4, 8):
            pb = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                            generator=torch.Generator(device="cuda").manual_seed(1))
            hs = []
            for i in range(E):
                hi = synth.make_history(dims, i)
                hi["path_length"] = 500
                hs.append(hi)
            for _ in range(max(8, -(-(pb._cal_windows + capi.SLOTS) // E) + 2)):  # (calibration passes + the bound's settling, untimed)
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
        # the environments stepping together: one policy pass at batch E, the windows' candidate passes back to back, one fp32
        # re-score pass over all windows' sets (action_sample_batch(lockstep=True))
        E = 8
        pl_ = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
                         generator=torch.Generator(device="cuda").manual_seed(1), max_batch=E, max_windows=E)
        hs = [dict(synth.make_history(dims, i), path_length=500) for i in range(E)]
        for _ in range(max(4, -(-pl_._cal_windows // E) + 2)):  # (the lock-step calls calibrate over the same number of windows, untimed)
            pl_.action_sample_batch(hs, eval=True, rtg=3.0, lockstep=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12):
            pl_.action_sample_batch(hs, eval=True, rtg=3.0, lockstep=True)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / 12
        bat["E8_lockstep"] = {"ms_per_call": round(1e3 * dt_, 4), "plan_steps_per_s": round(E / dt_, 2)}
        pl_.handle.close()
        out["batched"] = {"what": "action_sample_batch: E env windows x N=%d candidates per call, one pipelined plan step per window "
                                  "(incl. host window assembly and H2D copies; the call returns when all E are resolved)" % cfg.action_samples,
                          **bat}
    if args.precision == "bf16" and args.config == "c2" and cfg.plan_guidance == "rtg_guiding":
        # the other BASELINE configurations on this GPU, each with its F_alg-based MFMA fraction
        for name, fn in (("c5", lambda: goal_leg(0, 8192, steps=20, fp32_steps=4)),
                         ("c3", lambda: plan_leg(0, "walker2d", "critic_lambda_guiding", 4096, 16, 32, steps=20, settle=12,
                                                 what="BASELINE configs[2]: walker2d shapes critic_lambda_guiding N=4096 H=16 T=32 bf16")),
                         ("c4_shard", lambda: plan_leg(0, "halfcheetah", "rtg_guiding", 2048, 32, 64, steps=20, settle=12,
                                                       what="one rank's share of BASELINE configs[3] as a plan step of its own: halfcheetah "
                                                            "shapes rtg_guiding, 2048 of the 16384 candidates, H=32 T=64 bf16")),
                         ("c4_full", lambda: c4_full_leg(0))):
            try:
                out[name] = fn()
            except Exception as e:
                out[name] = {"error": repr(e)[:300]}
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
