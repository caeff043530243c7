"""INTEGRATION.md is executable: every ```python block of it runs here, in order and in one namespace, against a learner
shaped like the reference's (tests/fake_learner.py) and files shaped like the reference's checkpoints, statistics cache and
way-point tables (the documented calls must be the real signatures)."""
import os
import pickle
import re
import sys
import types

import numpy as np
import pytest
import torch

from m3pc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GD = os.path.join(ROOT, "tests", "golden")


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def test_integration_md_has_runnable_blocks():
    assert len(_blocks()) >= 6


def test_every_python_block_of_integration_md_runs(tmp_path, monkeypatch):
    from fake_learner import ToyEnv, make_learner

    dims = synth.Dims(11, 3, 8)  # hopper shapes, the shipped T = 8 / H = 4 (finetune_omtm/config.yaml:5,77)
    cfg = types.SimpleNamespace(traj_length=8, action_samples=64, horizon=4, discount=0.99, temperature=0.01, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda", index_jump=4)
    learner = make_learner(dims, cfg)
    # the files the doc names, in the reference's formats, in the working directory
    monkeypatch.chdir(tmp_path)
    sd = {k: v.cpu() for k, v in learner.mtm.state_dict().items()}
    torch.save({"model": sd, "step": 140000}, "hopper-medium-v2_140000.pt")                      # train.py:1208-1216
    torch.save({"qf": {k: v.cpu() for k, v in learner.iql.qf.state_dict().items()}}, "iql_100000.pt")  # model.py:310-320
    st = synth.make_tokenizer_stats(dims, 0)
    mod = types.ModuleType("research.omtm.datasets.base")

    class DataStatistics:
        def __init__(self, mean, std, min, max):
            self.mean, self.std, self.min, self.max = mean, std, min, max

    DataStatistics.__module__, DataStatistics.__qualname__ = "research.omtm.datasets.base", "DataStatistics"
    mod.DataStatistics = DataStatistics
    for name in ("research", "research.omtm", "research.omtm.datasets"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["research.omtm.datasets.base"] = mod
    try:
        with open("d4rl_statistics_hopper-medium-v2.pkl", "wb") as f:                                # sequence_dataset.py:357-404
            pickle.dump({k: DataStatistics(v["mean"], v["std"], v["min"], v["max"]) for k, v in st.items()}, f)
    finally:
        for name in ("research.omtm.datasets.base", "research.omtm.datasets", "research.omtm", "research"):
            sys.modules.pop(name, None)
    np.savetxt("hopper-wiggle-f2.txt", np.load(os.path.join(GD, "g3_zeroshot.npz"))["waypoints_raw"])  # waypoint_gen/*.txt
    histories = []
    for i in range(4):
        h = synth.make_history(dims, i)
        h["path_length"] = 100 + 7 * i
        histories.append(h)
    ns = {"learner": learner, "sequence_history": histories[0], "histories": histories, "rtg": 3.0,
          "env_fns": [lambda i=i: ToyEnv(11, 3, i) for i in range(3)]}
    for i, block in enumerate(_blocks()):
        try:
            exec(compile(block, f"INTEGRATION.md[block {i}]", "exec"), ns)
        except Exception as e:
            raise AssertionError(f"INTEGRATION.md python block {i} failed: {e!r}\n{block[:400]}") from e
    torch.cuda.synchronize()
    # the stub's structures are the library's (a shorter m3pc_plan_args would let the library read past it)
    import ctypes
    from m3pc_amd import capi
    assert ctypes.sizeof(ns["PlanArgs"]) == ctypes.sizeof(capi.PlanArgs) and ctypes.sizeof(ns["Dims"]) == ctypes.sizeof(capi.Dims)
    assert ctypes.sizeof(ns["NamedTensor"]) == ctypes.sizeof(capi.NamedTensor)


def test_c_resolve_reference_matches_the_planner(tmp_path):
    """INTEGRATION.md section 2's ```c block -- the certified re-score's host protocol for a host that is not Python -- compiled
    with gcc against include/m3pc_hip.h and run on BASELINE config 2's golden step from a deliberately short first pass (one
    candidate by score, one racer): it must extend the lists as the certificates ask and end on the reference's arg-max and
    multinomial index (tests/golden/g2_c2.npz), as m3pc_amd/certificate.py:resolve does for HipPlanner."""
    import ctypes as C
    import subprocess

    from m3pc_amd import capi
    from m3pc_amd.planner import HipPlanner

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```c\n(.*?)```", text, flags=re.S)
    assert len(blocks) == 1 and "m3pc_resolve_reference" in blocks[0]
    src, so = tmp_path / "resolve.c", tmp_path / "libm3pc_resolve_ref.so"
    src.write_text(blocks[0])
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(so),
                           "-L", libdir, "-l:" + os.path.basename(capi.LIB_PATH), "-Wl,-rpath," + libdir])
    g = np.load(os.path.join(GD, "g2_c2.npz"))
    dims = synth.Dims(11, 3, 32)
    N, H, tau = 1024, 16, 0.01
    cfg = types.SimpleNamespace(traj_length=32, action_samples=N, horizon=H, discount=0.99, temperature=tau, lmbda=0.6,
                                plan_guidance="rtg_guiding", device="cuda")
    p = HipPlanner(cfg, synth.make_state_dict(dims, 0), synth.make_tokenizer_stats(dims, 0), None, precision="bf16")
    eps = synth.make_eps(N, dims, 1).cuda()
    q = torch.empty(N, dtype=torch.float32).exponential_(1, generator=torch.Generator().manual_seed(77)).cuda()  # the golden draw's variates
    p._eps, p._draw_expo = (lambda shape: eps), (lambda: q)
    hist = synth.make_history(dims, 0)
    hist["path_length"] = 500
    p.action_sample(hist, plan=True, eval=False, rtg=3.0)          # the Python protocol (calibrates delta on the way)
    py = dict(p.last)
    assert int(py["argmax"].item()) == int(g["argmax"]) and int(py["sample_idx"].item()) == int(g["sample_idx"].reshape(-1)[0])
    # the same step through the C ABI by hand, first pass of 1 + 1 entries, then the C routine
    hd = p.handle
    s_, a_, r_, h, rtg = p.assemble_window(hist, rtg=3.0)
    res = hd.plan_step(capi.MODE_RTG, s_, a_, r_, eps, H, rtg, 0.6, 0.99, N, precision=capi.PREC_BF16, slot=0)
    er_b, a0 = res["expect_return"], res["sample_actions"][:, 0]
    K, R, kmin, rfirst = 128, 32, 1, 1
    f32 = dict(dtype=torch.float32, device="cuda")
    lst = torch.empty(R + K + 1, dtype=torch.int32, device="cuda")
    lst_b, lst_f = torch.empty(R + K + 1, **f32), torch.zeros(R + K + 1, **f32)
    hd.topk_race_window(er_b, q, tau, K, kmin, R, lst=lst, list_scores=lst_b, want_stats=False)
    o = R - rfirst
    hd.rescore(capi.MODE_RTG, s_, a_, r_, eps, lst[o : R + kmin], H, rtg, 0.6, 0.99, N, slot=0, out=lst_f[o : R + kmin], want_actions=False)
    hs = capi.HostStats()
    seq = hs.next_seq()
    merged, mstats = torch.empty(N, **f32), torch.empty(8, **f32)
    delta0 = float(py["delta"])
    _, _, sel = hd.merge_race_select(er_b, q, tau, lst[o:], rfirst, kmin, lst_b[o:], lst_f[o:], a0, delta=delta0, merged=merged,
                                     stats=mstats, host_stats=hs.buf, seq=seq)
    pp, ev, am, si, sa = sel
    lib = C.CDLL(str(so))
    fn = lib.m3pc_resolve_reference
    fn.restype = C.c_int
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, C.c_longlong,
                   vp, vp, vp, vp, vp, vp]
    args = hd._args(capi.MODE_RTG, capi.PREC_FP32, H, N, 0, N, 0.6, 0.99, rtg, 0)
    n_done, r_done, delta, seq_c = ci(kmin), ci(rfirst), cf(delta0), cf(seq)
    ptr = lambda t: vp(t.data_ptr())
    rc = fn(hd._h, C.byref(args), ptr(s_), ptr(a_), ptr(r_), ptr(eps), ptr(er_b), ptr(q), tau, N, ptr(lst), ptr(lst_b), ptr(lst_f), R, K,
            C.byref(n_done), C.byref(r_done), C.byref(delta), ptr(merged), ptr(mstats), vp(hs.buf.data_ptr()), C.byref(seq_c),
            ptr(a0), a0.stride(0), ptr(pp), ptr(ev), ptr(am), ptr(si), ptr(sa), vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, (rc, hd.lib.m3pc_last_error())
    torch.cuda.synchronize()
    assert n_done.value > kmin, "the one-entry first pass cannot have certified this step"
    need, need_race = int(mstats[2]), int(mstats[5])
    assert need <= n_done.value and need_race <= r_done.value      # both certificates hold on return
    assert int(am.item()) == int(g["argmax"]) == int(py["argmax"].item())
    assert int(si.item()) == int(g["sample_idx"].reshape(-1)[0]) == int(py["sample_idx"].item())
    assert np.abs(sa.cpu().numpy().reshape(-1) - g["sample_action"].reshape(-1)).max() < 2e-5
    # the same merged scores on the entries both protocols re-scored, the same weights p to rounding
    top = py["topk"].long()
    both = top[: min(top.numel(), n_done.value)]
    # (fp32 re-scores of the same candidates through passes of different row counts: other tilings of the few-row kernels)
    assert float((merged[both] - py["expect_return"][both]).abs().max()) <= 5e-5 * float(py["expect_return"].abs().max())
    assert float((pp - py["p"]).abs().max()) < 1e-5
    p.handle.close()
