"""Phase clocks of fused-tail tiles AS THE PIPELINED STEP RUNS (lab build): workgroup 37 of every fused launch logs its stamps.
    M3PC_LIB=m3pc_amd/libm3pc_hip_lab.so python tools/tile_phases_insitu.py [steps] [depth]"""
import ctypes as C
import os
import sys
import types
from collections import deque

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3pc_amd import capi, synth  # noqa: E402
from m3pc_amd.planner import HipPlanner  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEPTH = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dims = synth.Dims(11, 3, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6,
                            plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16",
               generator=torch.Generator(device="cuda").manual_seed(1))
hist = synth.make_history(dims, 0)
hist["path_length"] = 500
s, a, r, h, rtg = p.assemble_window(hist, rtg=3.0)


def run(k):
    fl = deque()
    for _ in range(k):
        if DEPTH == 0:
            p._guide(capi.MODE_RTG, s, a, r, rtg, h, 0.6)
            continue
        fl.append(p._issue(capi.MODE_RTG, s, a, r, rtg, h, 0.6, pipelined=True, inputs_ready=True))
        if len(fl) > DEPTH:
            fl.popleft().pair()
    while fl:
        fl.popleft().pair()


run(30)
torch.cuda.synchronize()
lib = p.handle.lib
fn = lib.m3pc_debug_stamp_log
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
cap = 6 * K
assert fn(p.handle._h, cap, None, None) == 0
run(K)
torch.cuda.synchronize()
buf = np.zeros((cap, 4, 16), dtype=np.int64)
n = C.c_int(0)
assert fn(p.handle._h, 0, buf.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
names = ["prologue", "out-proj", "LN2", "FFN", "X store", "tail"]
kinds = ["L1 A", "L1 B", "L2 A", "L2 B", "dec A", "dec B"]
print("launches logged", n.value)
# launches come in the enqueue order of a step: stage by stage, half A then half B
for kind in range(6):
    rows = buf[kind::6][2:]  # (skip the first steps)
    d = np.diff(rows[:, 0, :7], axis=1)
    d = d[(d > 0).all(axis=1)]
    if len(d) == 0:
        continue
    extra = ""
    if kind < 2:  # layer-1 tail: the Q|K|V part's own stamps (clock at its first phase / behind its last)
        ok = (rows[:, 0, 8] > rows[:, 0, 5]) & (rows[:, 0, 9] > rows[:, 0, 8])
        if ok.any():
            extra = (f"  [fragments + X'' drain {int(np.median((rows[:, 0, 8] - rows[:, 0, 5])[ok]))}  48 phases "
                     f"{int(np.median((rows[:, 0, 9] - rows[:, 0, 8])[ok]))}]")
    print(f"{kinds[kind]:6s} n={len(d):3d}  " + "  ".join(f"{nm} {int(np.median(d[:, i]))}" for i, nm in enumerate(names))
          + f"  total {int(np.median(d.sum(axis=1)))}" + extra)
