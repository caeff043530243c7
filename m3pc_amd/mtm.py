"""Host-side mirror of the reference model interface (research/omtm/models/mtm_model.py:200-221,
324-437, 593-607): ``omtmConfig.create(data_shapes, traj_length, discrete_map) -> omtm`` and
``omtm.forward(trajectories, masks)``.  The transformer itself lives in libm3pc_hip.so.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import torch

from . import capi
from .masks import mask_rows
from .tokenizers import SquashedNormal

KEYS = capi.KEYS


@dataclasses.dataclass
class omtmConfig:
    """Field-for-field the reference dataclass (mtm_model.py:200-221); only the architecture fields
    matter at test time."""

    n_embd: int = 128
    n_head: int = 2
    n_enc_layer: int = 1
    n_dec_layer: int = 1
    dropout: float = 0
    embd_pdrop: float = 0
    resid_pdrop: float = 0
    attn_pdrop: float = 0
    norm: str = "l2"
    loss: str = "total"
    reduce_use_sum: bool = False
    loss_keys: Optional[List[str]] = None
    latent_dim: Optional[int] = None
    use_masked_loss: bool = False
    init_temperature: float = 0.1
    target_entropy: float = -3
    use_entropy: bool = True

    def create(self, data_shape, traj_length, discrete_map, **handle_kw):
        return omtm(data_shape, traj_length, discrete_map, self, **handle_kw)


class omtm:
    """Inference-only masked trajectory model on one MI355X.

    ``data_shapes`` = {key: (tokens_per_timestep == 1, feature_dim)} (mtm_model.py:327-331).
    ``handle_kw`` sizes the device workspace: max_candidates, max_batch, critic_hidden, device.
    """

    def __init__(self, data_shapes: Dict[str, Tuple[int, ...]], traj_length: int, discrete_map: Dict[str, bool],
                 config: omtmConfig, max_candidates: int = 1024, max_batch: int = 1, critic_hidden: int = 256,
                 device: int = 0, precision: int = capi.PREC_FP32):
        if config.latent_dim is not None:
            raise capi.M3pcError("latent_dim is not used by any m3pc config and is not supported")
        if any(discrete_map.get(k, False) for k in KEYS):
            raise capi.M3pcError("discrete heads are not used by any m3pc config and are not supported")
        for k in KEYS:
            if k not in data_shapes or data_shapes[k][0] != 1:
                raise capi.M3pcError(f"data_shapes[{k!r}] must be (1, feature_dim)")
        if data_shapes["rewards"][1] != 1 or data_shapes["returns"][1] != 1:
            raise capi.M3pcError("rewards / returns are scalar per timestep")
        self.data_shapes = dict(data_shapes)
        self.config = config
        self.n_embd = config.n_embd
        self.max_len = traj_length
        self.precision = precision
        self.handle = capi.Handle(data_shapes["states"][1], data_shapes["actions"][1], traj_length, config.n_embd,
                                  config.n_head, config.n_enc_layer, config.n_dec_layer, max_candidates=max_candidates,
                                  max_batch=max_batch, critic_hidden=critic_hidden, device=device)
        self._sd: Dict[str, torch.Tensor] = {}

    # nn.Module-shaped conveniences used by the reference's callers (learner.py:32-36, finetune.py:298)
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        self.handle.load_weights(state_dict)
        self._sd = {k: v.detach().clone() for k, v in state_dict.items()}
        return self

    def state_dict(self):
        return dict(self._sd)

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def forward(self, trajectories: Dict[str, torch.Tensor], masks) -> Dict[str, object]:
        """mtm_model.py:593-607.  trajectories[k]: (B,T,1,D_k) tokenised; masks[k]: (T,) or (T,1).
        Returns {k: (B,T,1,D_k)} in the input key order, ``SquashedNormal`` for "actions"."""
        keys = list(trajectories.keys())
        toks = []
        for k in KEYS:
            t = trajectories.get(k)
            if t is not None:
                assert t.dim() == 4 and t.shape[2] == 1, f"{k}: expected (B,T,1,D)"
                t = t[:, :, 0]
            toks.append(t)
        out = self.handle.forward(toks, mask_rows(masks), want=keys, precision=self.precision)
        res = {}
        for k in keys:
            if k == "actions":
                mu, sd = out[k]
                res[k] = SquashedNormal(mu.unsqueeze(2), sd.unsqueeze(2))
            else:
                res[k] = out[k].unsqueeze(2)
        return res

    __call__ = forward
