"""Where a pipelined plan step spends its time on the device: event brackets around the policy pass (chain stream), the
candidate pass (current stream) and the re-score + select tail (chain stream) of every step, in steady state.
    python tools/pipeline_phases.py [steps] [depth] ["dict(planner kw)"]"""
import os
import sys
import types
from collections import deque

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
KW = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
               generator=torch.Generator(device="cuda").manual_seed(1), **KW)
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)
marks = []


def bracket(obj, name, tag):
    fn = getattr(obj, name)

    def wrapped(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        marks.append((tag, e0, e1))
        return out

    setattr(obj, name, wrapped)


bracket(p.handle, "policy_pass", "policy")
bracket(p.handle, "candidate_pass", "cand")
orig_tail = p._enqueue_tail


def tail(tk):
    with p._on(tk):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    orig_tail_inner(tk, e0, e1)


def orig_tail_inner(tk, e0, e1):
    # the tail waits for the candidate pass first; bracket what comes after the wait
    if tk.tchain is not None:
        tk.tchain.wait_event(tk.slot.ev_cand)
    with p._on(tk):
        e0.record()
    orig_tail(tk)
    with p._on(tk):
        e1.record()
    marks.append(("tail", e0, e1))


p._enqueue_tail = tail


def run(depth, k):
    fl = deque()
    for _ in range(k):
        if depth == 0:
            p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
            continue
        fl.append(p._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
        if len(fl) > depth:
            fl.popleft().pair()
    while fl:
        fl.popleft().pair()


run(DEPTH, 15)
torch.cuda.synchronize()
marks.clear()
g0 = torch.cuda.Event(enable_timing=True)
g0.record()
run(DEPTH, K)
torch.cuda.synchronize()
rows = {}
for tag, e0, e1 in marks:
    rows.setdefault(tag, []).append((g0.elapsed_time(e0), g0.elapsed_time(e1)))
for tag, v in rows.items():
    d = sorted(b - a_ for a_, b in v)
    print(f"{tag:7s} n={len(v):3d}  mean {sum(d) / len(d):.3f} ms  p50 {d[len(d) // 2]:.3f}  min {d[0]:.3f}  max {d[-1]:.3f}")
c = rows["cand"]
gaps = [c[i + 1][0] - c[i][1] for i in range(len(c) - 1)]
print("n_rescored", p.last.get("n_rescored"), "n_first", p.last.get("n_first"), "delta", p.last.get("delta"), "grown", p.delta_grown)
print(f"cand start-to-start {((c[-1][0] - c[0][0]) / (len(c) - 1)):.3f} ms; idle gap between candidate passes mean {sum(gaps) / len(gaps):.3f} max {max(gaps):.3f}")
for i in range(5, 9):
    print("step", i, " ".join(f"{t}:[{rows[t][i][0]:.2f},{rows[t][i][1]:.2f}]" for t in ("policy", "cand", "tail") if i < len(rows[t])))
