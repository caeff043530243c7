"""What the occasional 10-30 ms gap between two results of a pipelined leg is (bench.py's plan legs report step_gap_ms.max): per step,
the host time of issue and of result, new allocator segments (= hipMalloc calls of torch's caching allocator) and garbage collections.
python tools/stall_probe.py [c3|c4]"""
import gc
import os
import sys
import time
import types
from collections import deque

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
env, guidance, N, H, T = ("walker2d", "critic_lambda_guiding", 4096, 16, 32) if which == "c3" else ("halfcheetah", "rtg_guiding", 2048, 32, 64)
S, A = synth.ENV_DIMS[env]
critic = guidance == "critic_lambda_guiding"
dims = synth.Dims(S, A, T)
cfg = types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=1.0 if critic else 0.01, lmbda=0.6, plan_guidance=guidance)
qsd, om, os_ = synth.make_critic(dims, 0) if critic else (None, None, None)
gcs = []     # (start time, duration ms, generation) of every collection
_gc_t0 = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t0[0] = time.perf_counter()
    else:
        gcs.append((_gc_t0[0], 1e3 * (time.perf_counter() - _gc_t0[0]), info.get("generation")))


gc.callbacks.append(_gc_cb)
if "nogc" in sys.argv:
    gc.disable()
for rep in range(4):
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), qsd, om, os_, precision="bf16", device=0,
                   generator=torch.Generator(device="cuda").manual_seed(1), pipeline_depth=3)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    s_, a_, r_, h, rtg = p.assemble_window(hist, rtg=3.0)
    mode = capi.MODE_CRITIC if critic else capi.MODE_RTG
    rows = []
    flight = deque()

    def seg():
        return torch.cuda.memory_stats()["segment.all.allocated"]

    for i in range(60):
        t0 = time.perf_counter()
        s0 = seg()
        flight.append(p._issue(mode, s_, a_, r_, rtg, h, 0.6, pipelined=True, inputs_ready=True))
        t1 = time.perf_counter()
        if len(flight) > 3:
            tk = flight.popleft()
            tk.pair()
            t2 = time.perf_counter()
            info = tk.info or {}
            rows.append((i, 1e3 * (t1 - t0), 1e3 * (t2 - t1), seg() - s0, int(info.get("n_rescored", -1)), int(info.get("n_race", -1)),
                         bool(info.get("second_pass", info.get("n_rescored", 0) > info.get("n_first", 1 << 30)))))
    while flight:
        flight.popleft().pair()
    torch.cuda.synchronize()
    gaps = [(r[0], round(r[1] + r[2], 2)) for r in rows]
    big = [r for r in rows[20:] if r[1] + r[2] > 1.6 * sorted(x[1] + x[2] for x in rows[20:])[len(rows[20:]) // 2]]
    print(f"rep {rep}: median step {sorted(x[1] + x[2] for x in rows[20:])[len(rows[20:]) // 2]:.2f} ms; slow steps (index, issue ms, wait ms, new segments, n_rescored, n_race, second pass):",
          [(r[0], round(r[1], 2), round(r[2], 2), r[3], r[4], r[5], r[6]) for r in big],
          "gc: %d collections, longest %s" % (len(gcs), sorted(((round(d, 2), g) for _, d, g in gcs), reverse=True)[:3]), flush=True)
    del gcs[:]
    p.handle.close()
