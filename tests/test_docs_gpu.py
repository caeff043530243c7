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
