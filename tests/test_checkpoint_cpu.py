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


def test_small_std_is_clamped_like_the_reference_create():
    """continuous.py:58: a dimension whose dataset std is below 0.1 is tokenized with std 1."""
    from m3pc_amd.tokenizers import ContinuousTokenizer
    from oracle import mtm_oracle as O

    dims = synth.Dims(11, 3, 8)
    st = synth.make_tokenizer_stats(dims, 0)
    st["states"]["std"] = st["states"]["std"].copy()
    st["states"]["std"][[1, 7]] = [0.05, 0.0999]
    st["rewards"]["std"] = np.array([0.01], dtype=np.float32)
    raw = {k: {n: np.array(v[n]) for n in v} for k, v in st.items()}
    toks = {k: ContinuousTokenizer.from_statistics(k, checkpoint.tokenizer_stats(raw)[k]) for k in synth.KEYS}
    assert toks["states"]._data_std[1] == 1 and toks["states"]._data_std[7] == 1 and toks["states"]._data_std[0] == st["states"]["std"][0]
    assert float(toks["rewards"]._data_std[0]) == 1.0 and not toks["actions"].normalize and toks["returns"].normalize
    assert raw["states"]["std"][1] == np.float32(0.05)  # the caller's arrays are left alone
    # the oracle's tokenizer with the clamped std is what the reference computes: (x - mean) / 1
    x = torch.randn(2, 8, 11)
    ost = O.Stats(toks["states"]._data_mean.numpy(), toks["states"]._data_std.numpy(), st["states"]["min"], st["states"]["max"], True)
    enc = O.tok_encode(x, ost)
    assert torch.allclose(enc[..., 0, 1], x[..., 1] - float(st["states"]["mean"][1]))


def test_statistics_pickle_reader(tmp_path):
    """sequence_dataset.py:357-404 caches {key: DataStatistics} with pickle; the reader rebuilds it without importing
    the reference and refuses anything else."""
    import pickle
    import sys as _sys
    import types as _types

    dims = synth.Dims(11, 3, 8)
    st = synth.make_tokenizer_stats(dims, 0)
    # a stand-in module with the reference's class path, only to WRITE a pickle shaped like the reference's
    mod = _types.ModuleType("research.omtm.datasets.base")

    class DataStatistics:
        def __init__(self, mean, std, min, max):
            self.mean, self.std, self.min, self.max = mean, std, min, max

    DataStatistics.__module__ = "research.omtm.datasets.base"
    DataStatistics.__qualname__ = "DataStatistics"
    mod.DataStatistics = DataStatistics
    for name in ("research", "research.omtm", "research.omtm.datasets"):
        _sys.modules.setdefault(name, _types.ModuleType(name))
    _sys.modules["research.omtm.datasets.base"] = mod
    try:
        path = tmp_path / "d4rl_statistics_hopper-medium-v2.pkl"
        with open(path, "wb") as f:
            pickle.dump({k: DataStatistics(v["mean"], v["std"], v["min"], v["max"]) for k, v in st.items()}, f)
        # a cache written by an older run of the reference: "values" instead of "returns" (sequence_dataset.py:372-377)
        old_path = tmp_path / "d4rl_statistics_hopper-medium-v2_d=1.0.pkl"
        with open(old_path, "wb") as f:
            pickle.dump({("values" if k == "returns" else k): DataStatistics(v["mean"], v["std"], v["min"], v["max"])
                         for k, v in st.items()}, f)
    finally:
        for name in ("research.omtm.datasets.base", "research.omtm.datasets", "research.omtm", "research"):
            _sys.modules.pop(name, None)
    got = checkpoint.load_statistics_pickle(str(path))
    for k in synth.KEYS:
        for n in ("mean", "std", "min", "max"):
            assert np.array_equal(got[k][n], st[k][n].astype(np.float32))
    got_old = checkpoint.load_statistics_pickle(str(old_path))
    assert set(got_old) == set(synth.KEYS) and np.array_equal(got_old["returns"]["mean"], got["returns"]["mean"])
    evil = tmp_path / "evil.pkl"
    with open(evil, "wb") as f:
        pickle.dump({k: os.system for k in synth.KEYS}, f)
    with pytest.raises(Exception):
        checkpoint.load_statistics_pickle(str(evil))
