"""An object shaped like the reference ``Learner`` as far as ``m3pc_amd.planner.attach`` reads it (learner.cfg, learner.mtm with
state_dict / config / parameters, learner.tokenizer_manager.tokenizers[k]._data_mean/_data_std/normalize/stats, learner.iql.qf),
built from the synthetic recipe -- plus a toy deterministic environment with gym's reset/step API.  Test infrastructure."""
import types

import numpy as np
import torch

from m3pc_amd import synth
from m3pc_amd.mtm import omtmConfig
from m3pc_amd.tokenizers import DataStatistics


class ParamModule(torch.nn.Module):
    """Parameters under their state_dict names (dots and all); state_dict(keep_vars=True) hands out the Parameters themselves,
    as torch.nn.Module does, so in-place updates are visible through their version counters."""

    def __init__(self, sd):
        super().__init__()
        self.names = list(sd.keys())
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(v.clone(), requires_grad=False) for v in sd.values()])

    def state_dict(self, *a, keep_vars=False, **k):
        return {n: (p if keep_vars else p.detach()) for n, p in zip(self.names, self.ps)}

    def param(self, name):
        return self.ps[self.names.index(name)]


def make_learner(dims, cfg, seed=0, device="cuda", with_critic=True):
    sd = synth.make_state_dict(dims, seed)
    mtm = ParamModule({k: v.to(device) for k, v in sd.items()})
    mtm.config = omtmConfig(n_embd=dims.n_embd, n_head=dims.n_head, n_enc_layer=dims.n_enc_layer, n_dec_layer=dims.n_dec_layer)
    st = synth.make_tokenizer_stats(dims, seed)
    toks = {k: types.SimpleNamespace(_data_mean=torch.tensor(st[k]["mean"]), _data_std=torch.tensor(st[k]["std"]),
                                     normalize=(k != "actions"),
                                     stats=DataStatistics(st[k]["mean"], st[k]["std"], st[k]["min"], st[k]["max"]))
            for k in synth.KEYS}
    iql = None
    if with_critic:
        qsd, om, os_ = synth.make_critic(dims, seed)
        qf = ParamModule({k: v.to(device) for k, v in qsd.items()})
        qf.obs_mean, qf.obs_std = om, os_
        iql = types.SimpleNamespace(qf=qf)
    return types.SimpleNamespace(cfg=cfg, mtm=mtm, tokenizer_manager=types.SimpleNamespace(tokenizers=toks), iql=iql)


class ToyEnv:
    """Deterministic linear system with gym's API: obs' = tanh(M obs + B action), reward = -|obs'|^2 / S, done after `length` steps."""

    def __init__(self, S, A, seed, length=1000):
        g = np.random.RandomState(seed)
        self.M = (g.randn(S, S) * 0.3).astype(np.float32)
        self.B = (g.randn(S, A) * 0.5).astype(np.float32)
        self.o0 = g.randn(S).astype(np.float32)
        self.length = length
        self.t = 0
        self.obs = None

    def reset(self):
        self.t = 0
        self.obs = self.o0.copy()
        return self.obs.copy()

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(-1)
        self.obs = np.tanh(self.M @ self.obs + self.B @ a).astype(np.float32)
        self.t += 1
        return self.obs.copy(), float(-np.mean(self.obs ** 2)), self.t >= self.length, {}
