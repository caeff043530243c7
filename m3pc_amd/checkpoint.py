"""Checkpoint ingestion: the files the reference writes, read into what ``HipPlanner`` takes.

* pre-trained model ``{env}_{step}.pt`` = ``{"model": state_dict, "optimizer": ..., "step": ..., "eval_max": ...}``
  (research/omtm/train.py:1208-1216; loaded by the reference at research/finetune_omtm/learner.py:33-35);
* IQL checkpoint ``iql_{step}.pt`` = ``{"qf": TwinQ.state_dict, "vf": ..., "actor": ..., ...}``
  (research/finetune_omtm/model.py:310-320); only ``qf`` is on the plan path (learner.py:253-256);
* tokenizer statistics: a mapping key -> object or dict with ``mean / std / min / max`` (the reference keeps them as
  ``DataStatistics`` dataclasses, research/omtm/datasets/base.py:32-48, produced by the dataset at run time).

Nothing here touches the GPU; the planner uploads the tensors.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping, Optional

import numpy as np
import torch

KEYS = ("states", "actions", "rewards", "returns")


def load_mtm_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """``torch.load(path)["model"]`` with the checks the reference leaves to ``load_state_dict``."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob["model"] if isinstance(blob, dict) and "model" in blob else blob
    if "encoder_embed_dict.states.weight" not in sd:
        raise ValueError(f"{path}: not an omtm state_dict (no encoder_embed_dict.states.weight)")
    return {k: v.detach().float().contiguous() for k, v in sd.items()}


def load_iql_qf(path: str) -> Dict[str, torch.Tensor]:
    """The twin-Q critic of an IQL checkpoint (``{"qf": ...}``) or a bare TwinQ state_dict."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    sd = blob["qf"] if isinstance(blob, dict) and "qf" in blob else blob
    if "q1.net.0.weight" not in sd:
        raise ValueError(f"{path}: no TwinQ weights (q1.net.0.weight)")
    return {k: v.detach().float().contiguous() for k, v in sd.items()}


def tokenizer_stats(stats: Mapping[str, Any]) -> Dict[str, Dict[str, np.ndarray]]:
    """Normalise per-key statistics (objects with .mean/.std/.min/.max or dicts) to the dict form ``HipPlanner`` takes.
    The values stay the RAW dataset statistics (what the reference pickles, sequence_dataset.py:357-395); the clamp
    ``std[std < 0.1] = 1`` of ``ContinuousTokenizer.create`` (continuous.py:58) is applied where the tokenizer is built
    (``HipPlanner`` / ``ContinuousTokenizer.create``), exactly once."""
    out = {}
    for k in KEYS:
        s = stats[k]
        get = (lambda n: s[n]) if isinstance(s, Mapping) else (lambda n: getattr(s, n))
        out[k] = {n: np.asarray(get(n), dtype=np.float32) for n in ("mean", "std", "min", "max")}
        if not np.all(out[k]["min"] <= out[k]["max"]):
            raise ValueError(f"statistics of '{k}': min > max")
    return out


def load_statistics_pickle(path: str) -> Dict[str, Dict[str, np.ndarray]]:
    """The statistics cache the reference's dataset writes (``/tmp/d4rl/d4rl_statistics_{env_name}[_d={discount}|_avg].pkl``,
    research/omtm/datasets/sequence_dataset.py:357-404): a pickled ``{key: DataStatistics}``.  Caches written by older runs
    carry the key ``values`` instead of ``returns``; they are accepted and renamed as ``trajectory_statistics`` does
    (sequence_dataset.py:372-377).  The pickle refers to the
    reference's ``research.omtm.datasets.base.DataStatistics`` class, which is not importable here: objects of that one
    class are rebuilt as plain namespaces, anything else is refused (no arbitrary code runs)."""
    import pickle
    import types

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if name == "DataStatistics":
                return _Stats
            if module.startswith("numpy") and name in ("ndarray", "dtype", "_reconstruct", "scalar", "_frombuffer"):
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"statistics pickle refers to {module}.{name}; only DataStatistics of numpy arrays is accepted")

    class _Stats(types.SimpleNamespace):
        def __init__(self, mean=None, std=None, min=None, max=None):
            super().__init__(mean=mean, std=std, min=min, max=max)

    with open(path, "rb") as f:
        obj = _Unpickler(f).load()
    if isinstance(obj, dict) and "values" in obj and "returns" not in obj:
        obj["returns"] = obj.pop("values")
    if not isinstance(obj, dict) or not all(k in obj for k in KEYS):
        raise ValueError(f"{path}: not a {{key: DataStatistics}} pickle with keys {KEYS}")
    return tokenizer_stats(obj)


def model_dims(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, int]:
    """Architecture read off the tensor shapes (n_head is not recoverable from shapes: pass it explicitly)."""
    d = state_dict["encoder_embed_dict.states.weight"].shape[0]
    n_enc = 1 + max(int(k.split(".")[2]) for k in state_dict if k.startswith("encoder.layers."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in state_dict if k.startswith("decoder.layers."))
    return {"n_embd": int(d), "n_enc_layer": n_enc, "n_dec_layer": n_dec}


def planner_from_checkpoints(cfg, mtm_path: str, stats: Mapping[str, Any], iql_path: Optional[str] = None, obs_mean=None,
                             obs_std=None, n_head: int = 4, **planner_kw):
    """Build a ``HipPlanner`` from the reference's checkpoint files (what ``Learner.__init__`` does at
    learner.py:32-62 for the plan path)."""
    from .planner import HipPlanner

    sd = load_mtm_state_dict(mtm_path)
    qf = load_iql_qf(iql_path) if iql_path else None
    if qf is not None and (obs_mean is None or obs_std is None):
        raise ValueError("the critic needs obs_mean / obs_std (finetune_omtm/model.py:146-171 normalises its input)")
    return HipPlanner(cfg, sd, tokenizer_stats(stats), qf, obs_mean, obs_std, n_head=n_head, **model_dims(sd), **planner_kw)
