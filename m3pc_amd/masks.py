"""Deterministic test-time masks with the reference's signatures and return types
(research/finetune_omtm/masks.py:7-61, research/zeroshot_omtm/masks.py:30-108):
``create_*_mask(traj_length, device, idx) -> {key: float64 tensor (T,)}`` in the key order
states, actions, rewards, returns.  The HIP library consumes them as 0/1 byte rows.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

KEYS = ("states", "actions", "rewards", "returns")


def _pack(device, states, actions, rewards, returns) -> Dict[str, torch.Tensor]:
    rows = (states, actions, rewards, returns)
    return {k: torch.from_numpy(v).to(device) for k, v in zip(KEYS, rows)}


def _ones_upto(T: int, n: int) -> np.ndarray:
    m = np.zeros(T)
    if n > 0:
        m[:n] = 1
    return m


def create_rcbc_mask(traj_length: int, device, idx: int):
    """Return-conditioned BC: states[:idx+1], actions[:idx], every return, no reward (masks.py:7-27)."""
    T = traj_length
    return _pack(device, _ones_upto(T, idx + 1), _ones_upto(T, idx), np.zeros(T), np.ones(T))


def create_fd_mask(traj_length: int, device, idx: int):
    """Forward dynamics: states[:idx+1] and every action (masks.py:30-44)."""
    T = traj_length
    return _pack(device, _ones_upto(T, idx + 1), np.ones(T), np.zeros(T), np.zeros(T))


def create_ret_mask(traj_length: int, device, idx: int):
    """Return prediction: states[:idx+1], actions[:idx+1] (masks.py:47-61)."""
    T = traj_length
    return _pack(device, _ones_upto(T, idx + 1), _ones_upto(T, idx + 1), np.zeros(T), np.zeros(T))


def create_fid_mask(traj_length: int, device, idx: int):
    """Full inverse dynamics: every state, actions[:idx] (zeroshot masks.py:30-47)."""
    T = traj_length
    return _pack(device, np.ones(T), _ones_upto(T, idx), np.zeros(T), np.zeros(T))


def create_gid_mask(traj_length: int, device, idx: int):
    """Goal inverse dynamics: as fid, but states idx+1..T-2 hidden when idx > 0 (zeroshot masks.py:50-69)."""
    T = traj_length
    s = np.ones(T)
    if idx > 0:
        s[idx + 1 : -1] = 0
    return _pack(device, s, _ones_upto(T, idx), np.zeros(T), np.zeros(T))


def create_pi_mask(traj_length: int, device, idx: int):
    """Path inference mask; identical rows to gid (zeroshot masks.py:72-91)."""
    return create_gid_mask(traj_length, device, idx)


def mask_rows(masks) -> list:
    """{key: (T,) tensor/array} -> four python lists of 0/1 in ABI key order (host side; masks are
    built on the host by every caller in the reference, so this costs no device sync for CPU masks)."""
    out = []
    for k in KEYS:
        m = masks[k]
        if torch.is_tensor(m):
            m = m.detach().cpu().numpy()
        m = np.asarray(m)
        if m.ndim == 2:  # (T, P) with P == 1
            assert m.shape[1] == 1, "one token per timestep and modality"
            m = m[:, 0]
        out.append([int(v != 0) for v in m])
    return out
