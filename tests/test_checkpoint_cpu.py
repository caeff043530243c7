"""Checkpoint ingestion (m3pc_amd/checkpoint.py) on files shaped like the reference's (train.py:1208-1216,
finetune_omtm/model.py:310-320).  CPU only: no library call is made."""
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from m3pc_amd import checkpoint, synth  # noqa: E402


def _files(tmp_path):
    dims = synth.Dims(11, 3, 8, n_embd=64, n_head=2)
    sd = synth.make_state_dict(dims, 0)
    qsd, om, os_ = synth.make_critic(dims, 0)
    mtm = tmp_path / "hopper-medium-v2_100.pt"
    iql = tmp_path / "iql_100.pt"
    torch.save({"model": sd, "optimizer": {"state": {}, "param_groups": [{"lr": 1e-4}]}, "step": 100, "eval_max": {"a": 1.0}}, mtm)
    torch.save({"qf": qsd, "q_optimizer": {}, "vf": {}, "v_optimizer": {}, "actor": {}, "total_it": 7}, iql)
    return dims, sd, qsd, str(mtm), str(iql)


def test_mtm_and_iql_checkpoints_round_trip(tmp_path):
    dims, sd, qsd, mtm, iql = _files(tmp_path)
    got = checkpoint.load_mtm_state_dict(mtm)
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k].float())
    q = checkpoint.load_iql_qf(iql)
    assert set(q) == set(qsd) and all(torch.equal(q[k], qsd[k].float()) for k in qsd)
    assert checkpoint.model_dims(got) == {"n_embd": 64, "n_enc_layer": 2, "n_dec_layer": 1}


def test_wrong_files_are_rejected(tmp_path):
    _, _, _, mtm, iql = _files(tmp_path)
    with pytest.raises(ValueError):
        checkpoint.load_mtm_state_dict(iql)
    with pytest.raises(ValueError):
        checkpoint.load_iql_qf(mtm)


def test_statistics_accept_dataclass_like_objects_and_dicts():
    dims = synth.Dims(11, 3, 8)
    st = synth.make_tokenizer_stats(dims, 0)
    objs = {k: types.SimpleNamespace(**v) for k, v in st.items()}  # the reference's DataStatistics has these attributes
    a, b = checkpoint.tokenizer_stats(st), checkpoint.tokenizer_stats(objs)
    for k in synth.KEYS:
        for n in ("mean", "std", "min", "max"):
            assert a[k][n].dtype == np.float32 and np.array_equal(a[k][n], b[k][n])
    bad = {k: dict(v) for k, v in st.items()}
    bad["states"]["min"] = bad["states"]["max"] + 1.0
    with pytest.raises(ValueError):
        checkpoint.tokenizer_stats(bad)
