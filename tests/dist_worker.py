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

CASES = {
    # name: env, guidance, mode, N per rank, H, T, temperature
    "c2": ("hopper", "rtg_guiding", capi.MODE_RTG, 512, 16, 32, 0.01),
    "c3": ("walker2d", "critic_lambda_guiding", capi.MODE_CRITIC, 384, 16, 32, 1.0),
    "c4": ("halfcheetah", "rtg_guiding", capi.MODE_RTG, 2048, 32, 64, 0.01),   # the 2048-candidate shard of BASELINE config 4
    "odd": ("hopper", "rtg_guiding", capi.MODE_RTG, 171, 16, 32, 0.01),         # N % world != 0 for world 2 (N = 513: +1 below)
}


def main():
    rank, world, port, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    env, guidance, mode, n_rank, H, T, temp = CASES[case]
    N = n_rank * world + (1 if case == "odd" else 0)
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    torch.cuda.set_device(0)
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
        p.handle.close()
        return out

    ref = plan(None, False)  # before the process group exists: world 1
    assert ref["world"] == 1
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    got = plan(dist.group.WORLD, True)
    assert got["world"] == world
    bad = [k for k in ("expect_return", "argmax", "sample_idx", "eval_action", "sample_action") if not torch.equal(ref[k], got[k])]
    if got["n_rescored"] != ref["n_rescored"]:
        bad.append("n_rescored")
    # a planner without an explicit generator must be refused when sharded (silent RNG divergence)
    try:
        HipPlanner(cfg, sd, st, qsd, om, os_, precision="bf16", group=dist.group.WORLD)
        bad.append("missing-generator not refused")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        print(f"rank {rank}/{world} case {case}: MISMATCH in {bad}", flush=True)
        sys.exit(3)
    print(f"rank {rank}/{world} case {case}: sharded == single ({N} candidates, {got['n_rescored']} re-scored)", flush=True)


if __name__ == "__main__":
    main()
