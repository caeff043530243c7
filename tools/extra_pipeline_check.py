import sys, types, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); 
from m3pc_amd import synth
from m3pc_amd.planner import HipPlanner
KEYS = ("expect_return", "p", "argmax", "sample_idx", "eval_action", "sample_action")
def run(env, N, T, H, depth):
    S, A = synth.ENV_DIMS[env]
    dims = synth.Dims(S, A, T)
    cfg = types.SimpleNamespace(traj_length=T, action_samples=N, horizon=H, discount=0.99, temperature=0.01, lmbda=0.6, plan_guidance="rtg_guiding")
    wins = []
    for i in range(5):
        h = synth.make_history(dims, i % 3); h["path_length"] = [500, 40, 321, 998, 77][i]; wins.append(h)
    def mk(): return HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16", generator=torch.Generator(device="cuda").manual_seed(5), pipeline_depth=depth)
    ps = mk(); ser = []
    for w in wins:
        ps.action_sample(w, plan=True, eval=True, rtg=3.0); ser.append({k: ps.last[k].clone() for k in KEYS})
    ps.handle.close()
    pp = mk(); fl, got = [], []
    for w in wins:
        fl.append(pp.plan_async(w, eval=True, rtg=3.0))
        if len(fl) > depth:
            tk = fl.pop(0); tk.result(); got.append({k: tk.info[k].clone() for k in KEYS})
    while fl:
        tk = fl.pop(0); tk.result(); got.append({k: tk.info[k].clone() for k in KEYS})
    torch.cuda.synchronize(); pp.handle.close()
    bad = [(i, k) for i, (g, s) in enumerate(zip(got, ser)) for k in KEYS if not torch.equal(g[k], s[k])]
    print(env, N, T, H, "depth", depth, "OK" if not bad else ("MISMATCH", bad[:4]))
run("halfcheetah", 2048, 64, 32, 2)
run("hopper", 4096, 32, 16, 3)
run("walker2d", 1500, 32, 16, 2)
run("hopper", 625, 8, 4, 3)
