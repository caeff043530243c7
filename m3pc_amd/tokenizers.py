"""Host-side mirror of the reference tokenizer API (research/omtm/tokenizers/base.py:64-99,
continuous.py:25-94, research/omtm/datasets/base.py:31-48).  The arithmetic runs in the HIP library
(m3pc_tokenize / m3pc_detokenize, or folded into the embedding / head kernels inside a plan step);
these classes only carry the statistics and keep the reference's call signatures.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import capi

KEYS = capi.KEYS


@dataclass
class DataStatistics:
    """datasets/base.py:31-48."""

    mean: np.ndarray
    std: np.ndarray
    min: np.ndarray
    max: np.ndarray

    def __post_init__(self):
        self.mean = np.array(self.mean, dtype=np.float32)
        self.std = np.array(self.std, dtype=np.float32)
        self.min = np.array(self.min, dtype=np.float32)
        self.max = np.array(self.max, dtype=np.float32)
        assert self.mean.shape == self.std.shape == self.min.shape == self.max.shape
        assert np.all(self.min <= self.max)


class ContinuousTokenizer:
    """continuous.py:25-94.  ``encode`` (B,L,D) -> (B,L,1,D) fp32; ``decode`` the inverse;
    a SquashedNormal passes through ``decode`` untouched, as in the reference."""

    def __init__(self, data_mean, data_std, stats: DataStatistics, normalize: bool = True):
        self._data_mean = torch.tensor(np.asarray(data_mean), dtype=torch.float32)
        self._data_std = torch.tensor(np.asarray(data_std), dtype=torch.float32)
        self.stats = stats
        self.normalize = normalize
        self._handle: Optional[capi.Handle] = None
        self._key: Optional[int] = None

    @classmethod
    def create(cls, key: str, train_dataset, normalize: bool = True) -> "ContinuousTokenizer":
        """continuous.py:50-62: std < 0.1 -> 1; actions are never normalised."""
        stats = train_dataset.trajectory_statistics()[key]
        data_mean = stats.mean
        data_std = stats.std
        data_std[data_std < 0.1] = 1
        if key == "actions":
            return cls(data_mean, data_std, stats, normalize=False)
        return cls(data_mean, data_std, stats, normalize=normalize)

    @classmethod
    def from_statistics(cls, key: str, stats) -> "ContinuousTokenizer":
        """The tokenizer ``create`` builds, from RAW dataset statistics given as a dict or an object with
        mean / std / min / max (what the reference caches in /tmp/d4rl/d4rl_statistics_*.pkl,
        sequence_dataset.py:357-395): std < 0.1 -> 1 (continuous.py:58), actions never normalised (59-61)."""
        get = (lambda n: stats[n]) if isinstance(stats, dict) else (lambda n: getattr(stats, n))
        mean = np.array(get("mean"), dtype=np.float32)
        std = np.array(get("std"), dtype=np.float32)
        std[std < 0.1] = 1
        ds = DataStatistics(mean, std, np.asarray(get("min"), dtype=np.float32), np.asarray(get("max"), dtype=np.float32))
        return cls(mean, std, ds, normalize=(key != "actions"))

    @property
    def discrete(self) -> bool:
        return False

    def bind(self, handle: capi.Handle, key: int):
        self._handle, self._key = handle, key
        handle.set_tokenizer(key, self._data_mean, self._data_std, self.normalize)

    def _need(self):
        if self._handle is None:
            raise capi.M3pcError("tokenizer is not bound to a HIP handle (TokenizerManager.bind)")
        return self._handle

    def encode(self, trajectory: torch.Tensor) -> torch.Tensor:
        assert trajectory.dim() == 3
        return self._need().tokenize(self._key, trajectory).unsqueeze(2)

    def decode(self, trajectory):
        if isinstance(trajectory, SquashedNormal):
            return trajectory
        assert trajectory.dim() == 4 and trajectory.size(2) == 1
        return self._need().detokenize(self._key, trajectory.squeeze(2))


class TokenizerManager:
    """tokenizers/base.py:64-99: dict-wise encode / decode."""

    def __init__(self, tokenizers: Dict[str, ContinuousTokenizer]):
        self.tokenizers = dict(tokenizers)

    def bind(self, handle: capi.Handle):
        for k, name in enumerate(KEYS):
            self.tokenizers[name].bind(handle, k)
        return self

    def encode(self, trajectories: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        out = {}
        for key, value in trajectories.items():
            if key in self.tokenizers:
                out[key] = self.tokenizers[key].encode(value)
                assert out[key].dim() == 4
        return out

    def decode(self, tokenized: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {key: self.tokenizers[key].decode(value) for key, value in tokenized.items()}


class SquashedNormal:
    """tanh(Normal(loc, std)) -- the attributes and methods of the reference distribution that the plan
    path reads (mtm_model.py:254-291): ``loc``, ``std``, ``mean``, ``sample``.  ``sample`` takes an
    optional ``eps`` so that callers can make the noise explicit."""

    def __init__(self, loc: torch.Tensor, std: torch.Tensor):
        self.loc = loc
        self.std = std

    @property
    def mean(self) -> torch.Tensor:
        return torch.tanh(self.loc)

    def sample(self, sample_shape=(), eps: Optional[torch.Tensor] = None) -> torch.Tensor:
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        if eps is None:
            eps = torch.randn(shape, device=self.loc.device, dtype=self.loc.dtype)
        return torch.tanh(eps * self.std + self.loc)
