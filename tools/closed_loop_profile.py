"""Host side of the closed loop (action_sample + .cpu() per call): median call time and a cProfile of where the host spends it.
    python tools/closed_loop_profile.py   (on the GPU box)"""
import os, sys, time, types, cProfile, pstats
import torch
sys.path.insert(0, os.getcwd())
from m3pc_amd import capi, synth
from m3pc_amd.planner import HipPlanner
S, A = synth.ENV_DIMS["hopper"]
dims = synth.Dims(S, A, 32)
cfg = types.SimpleNamespace(traj_length=32, action_samples=1024, horizon=16, discount=0.99, temperature=0.01, lmbda=0.6, plan_guidance="rtg_guiding")
p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16")
hist = synth.make_history(dims, 0); hist["path_length"] = 500
for _ in range(30): p.action_sample(hist, eval=True, rtg=3.0).cpu()
ts=[]
for _ in range(40):
    t0=time.perf_counter(); a=p.action_sample(hist, eval=True, rtg=3.0); t1=time.perf_counter(); a=a.cpu(); t2=time.perf_counter()
    ts.append((t1-t0, t2-t0))
ts.sort(key=lambda x:x[1])
print("host return %.3f ms, with .cpu() %.3f ms (median)" % (1e3*ts[20][0], 1e3*ts[20][1]))
pr=cProfile.Profile(); pr.enable()
for _ in range(40): p.action_sample(hist, eval=True, rtg=3.0).cpu()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
