"""Deterministic synthetic weights / statistics / histories for the MPC plan step.

No D4RL data and no pretrained checkpoint exist in the build or GPU containers
(SURVEY.md section 8c/8d), so every test, fixture and benchmark draws its MTM weights,
tokenizer statistics, critic weights and episode history from the recipes below.
The recipes only depend on (name, seed) and on CPU ``torch.Generator`` streams, so the
golden-fixture generator (which loads them into the *reference* model), the oracle,
the HIP path and ``bench.py`` all see bit-identical tensors.

State-dict names and shapes follow the reference model's ``state_dict()``
(research/omtm/models/mtm_model.py:348-437).
"""
from __future__ import annotations

import dataclasses
import math
import zlib
from typing import Dict, Tuple

import numpy as np
import torch

KEYS = ("states", "actions", "rewards", "returns")


@dataclasses.dataclass(frozen=True)
class Dims:
    """Static sizes of one planner instance."""

    state_dim: int
    action_dim: int
    traj_length: int
    n_embd: int = 512
    n_head: int = 4
    n_enc_layer: int = 2
    n_dec_layer: int = 1

    @property
    def feat(self) -> Dict[str, int]:
        return {"states": self.state_dim, "actions": self.action_dim, "rewards": 1, "returns": 1}

    @property
    def data_shapes(self) -> Dict[str, Tuple[int, int]]:
        return {k: (1, v) for k, v in self.feat.items()}


ENV_DIMS = {"hopper": (11, 3), "walker2d": (17, 6), "halfcheetah": (17, 6)}


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


def _randn(name: str, seed: int, *shape: int) -> torch.Tensor:
    return torch.randn(*shape, generator=_gen(name, seed), dtype=torch.float32)


def sincos_pos_embed(n_embd: int, length: int) -> torch.Tensor:
    """1-D sin/cos table halved, shape (1, T, 1, d) -- mtm_model.py:38-58, 435-437."""
    omega = np.arange(n_embd // 2, dtype=np.float32)
    omega /= n_embd / 2.0
    omega = 1.0 / 10000**omega
    pos = np.arange(length, dtype=np.float32).reshape(-1)
    out = np.einsum("m,d->md", pos, omega)
    emb = np.concatenate([np.sin(out), np.cos(out)], axis=1)
    return torch.from_numpy(emb).float()[None, :, None, :] / 2.0


def make_state_dict(dims: Dims, seed: int = 0) -> Dict[str, torch.Tensor]:
    """MTM weights with trained-like magnitudes (names as in SURVEY Appendix B)."""
    d = dims.n_embd
    sd: Dict[str, torch.Tensor] = {}

    def lin(name: str, out_f: int, in_f: int, gain: float = 1.0):
        sd[name + ".weight"] = _randn(name + ".weight", seed, out_f, in_f) * (gain / math.sqrt(in_f))
        sd[name + ".bias"] = _randn(name + ".bias", seed, out_f) * 0.05

    def ln(name: str):
        sd[name + ".weight"] = 1.0 + 0.1 * _randn(name + ".weight", seed, d)
        sd[name + ".bias"] = 0.05 * _randn(name + ".bias", seed, d)

    for k, f in dims.feat.items():
        lin(f"encoder_embed_dict.{k}", d, f)
        lin(f"decoder_embed_dict.{k}", d, d)
        sd[f"mask_token_dict.{k}"] = 0.3 * _randn(f"mask_token_dict.{k}", seed, 1, 1, d)
        sd[f"encoder_per_dim_encoding.{k}"] = 0.3 * _randn(f"encoder_per_dim_encoding.{k}", seed, 1, 1, 1, d)
        sd[f"decoder_per_dim_encoding.{k}"] = 0.3 * _randn(f"decoder_per_dim_encoding.{k}", seed, 1, 1, 1, d)

    def block(prefix: str):
        sd[prefix + ".self_attn.in_proj_weight"] = _randn(prefix + ".self_attn.in_proj_weight", seed, 3 * d, d) / math.sqrt(d)
        sd[prefix + ".self_attn.in_proj_bias"] = 0.05 * _randn(prefix + ".self_attn.in_proj_bias", seed, 3 * d)
        lin(prefix + ".self_attn.out_proj", d, d, 0.7)
        lin(prefix + ".linear1", 4 * d, d)
        lin(prefix + ".linear2", d, 4 * d, 0.7)
        ln(prefix + ".norm1")
        ln(prefix + ".norm2")

    for i in range(dims.n_enc_layer):
        block(f"encoder.layers.{i}")
    ln("encoder.norm")
    for i in range(dims.n_dec_layer):
        block(f"decoder.layers.{i}")
    ln("decoder.norm")

    for k, f in dims.feat.items():
        if k == "actions":
            lin("output_head_dict.actions.mu", f, d, 0.5)
            lin("output_head_dict.actions.log_std", f, d, 0.5)
        else:
            ln(f"output_head_dict.{k}.0")
            lin(f"output_head_dict.{k}.1", d, d)
            lin(f"output_head_dict.{k}.3", f, d)
    sd["pos_embed"] = sincos_pos_embed(d, dims.traj_length)
    return sd


def trained_like(sd: Dict[str, torch.Tensor], stats, seed: int = 0, linear_scale: float = 2.0, gain_lo: float = 0.5,
                 gain_hi: float = 2.0, returns_std_scale: float = 1.0):
    """A perturbation family of the recipe above that moves it towards what training does to a model (the
    certificate's statistics -- the bf16 deviation delta against the score spread sigma -- are properties of the weight
    distribution they were measured on): every Linear weight x ``linear_scale`` (larger pre-activations: sharper attention,
    more gelu curvature, a wider range in front of the x 1000 returns term of learner.py:305), every LayerNorm gain drawn
    from U(gain_lo, gain_hi) per feature, the returns tokenizer's std x ``returns_std_scale`` (the scale of the decoded
    return-to-go).  Returns (state_dict, stats) copies; the inputs are left alone."""
    out = {}
    for name, w in sd.items():
        if name == "pos_embed":
            out[name] = w
        elif w.dim() == 2 and (name.endswith(".weight") or name.endswith("in_proj_weight")):
            out[name] = w * float(linear_scale)
        elif w.dim() == 1 and name.endswith(".weight") and (".norm" in name or name.endswith(".0.weight")):
            u = torch.rand(w.shape, generator=_gen("trained_like." + name, seed), dtype=torch.float32)
            out[name] = (gain_lo + (gain_hi - gain_lo) * u).to(torch.float32)
        else:
            out[name] = w
    st = {k: {f: np.array(v, copy=True) for f, v in d.items()} for k, d in stats.items()}
    st["returns"]["std"] = (st["returns"]["std"] * np.float32(returns_std_scale)).astype(np.float32)
    return out, st


def make_tokenizer_stats(dims: Dims, seed: int = 0) -> Dict[str, Dict[str, np.ndarray]]:
    """Per-key mean/std/min/max as float32 arrays (SURVEY 8d recipe)."""
    stats = {}
    for k, f in dims.feat.items():
        mean = _randn(f"tok.{k}.mean", seed, f).numpy()
        std = (_randn(f"tok.{k}.std", seed, f).abs() + 0.5).numpy()
        stats[k] = {
            "mean": mean.astype(np.float32),
            "std": std.astype(np.float32),
            "min": np.full((f,), -5.0, np.float32),
            "max": np.full((f,), 5.0, np.float32),
        }
    return stats


def make_critic(dims: Dims, seed: int = 0, hidden: int = 256):
    """TwinQ weights (finetune_omtm/model.py:146-171) + observation mean/std."""
    sa = dims.state_dim + dims.action_dim
    sd = {}
    for q in ("q1", "q2"):
        for li, (o, i) in zip((0, 2, 4), ((hidden, sa), (hidden, hidden), (1, hidden))):
            name = f"{q}.net.{li}"
            sd[name + ".weight"] = _randn(name + ".weight", seed, o, i) / math.sqrt(i)
            sd[name + ".bias"] = 0.05 * _randn(name + ".bias", seed, o)
    obs_mean = _randn("critic.obs_mean", seed, dims.state_dim)
    obs_std = _randn("critic.obs_std", seed, dims.state_dim).abs() + 0.5
    return sd, obs_mean, obs_std


def make_history(dims: Dims, seed: int = 0, length: int = 1000) -> Dict[str, np.ndarray]:
    """An episode buffer shaped like ReplayBuffer.online_rollout's ``current_trajectory``
    (replay_buffer.py:188-203): float32 arrays of ``length`` rows."""
    S, A = dims.state_dim, dims.action_dim
    obs = _randn("hist.obs", seed, length, S).numpy()
    act = (torch.rand(length, A, generator=_gen("hist.act", seed)) * 2 - 1).numpy()
    rew = _randn("hist.rew", seed, length, 1).numpy()
    val = _randn("hist.val", seed, length, 1).numpy()
    return {
        "observations": obs.astype(np.float32),
        "actions": act.astype(np.float32),
        "rewards": rew.astype(np.float32),
        "values": val.astype(np.float32),
    }


def make_eps(n: int, dims: Dims, seed: int = 1) -> torch.Tensor:
    """Standard normals in the shape the reference draws them,
    ``dist.sample((N,))`` over loc of shape (1,T,1,A) (finetune_omtm/learner.py:285)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn((n, 1, dims.traj_length, 1, dims.action_dim), generator=g, dtype=torch.float32)
