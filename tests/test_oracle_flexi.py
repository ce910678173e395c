"""Pins oracle/flexi_oracle.py against golden vectors minted from the REAL reference GShellFlexiCubes
(oracle/make_golden_flexi.py): faces bit-exact, floats 1e-6, gradients 1e-5."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import flexi_oracle as fo
from oracle.make_golden_flexi import make_inputs

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "flexi_*.npz")))


def run_case(g, extract):
    res = int(g["res"])
    x, s, nu, w = make_inputs(res, str(g["sdf_kind"]), str(g["msdf_kind"]), str(g["weights_kind"]), int(g["seed"]))
    verts, cubes = fo.construct_voxel_grid(res)
    np.testing.assert_array_equal(cubes.numpy(), g["cubes"])                 # grid layout == the reference's construct_voxel_grid
    return x, s, nu, w, cubes, res


TRAIN_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "flexitrain_*.npz")))


@pytest.mark.parametrize("path", FILES + TRAIN_FILES, ids=[os.path.basename(f)[:-4] for f in FILES + TRAIN_FILES])
def test_flexi_oracle_matches_reference(path):
    """flexi_*: training=False; flexitrain_*: training=True (gshell_flexicubes.py:523-551, centre-vertex fans), both minted from the real reference"""
    g = np.load(path)
    assert len(TRAIN_FILES) >= 4
    x, s, nu, w, cubes, res = run_case(g, fo.extract)
    X, S, NU, Wt = (torch.tensor(a, requires_grad=True) for a in (x, s[:, None], nu, w))
    out = fo.extract(X, S, NU, cubes, res, Wt[:, :12], Wt[:, 12:20], Wt[:, 20], training=os.path.basename(path).startswith("flexitrain_"))
    if bool(g["empty"]):
        assert len(out) == 3 and out[0].shape == (0, 3) and out[1].shape == (0, 3) and out[2].shape == (0,)
        return
    v, f, L, ex = out
    np.testing.assert_array_equal(f.numpy(), g["faces_open"])
    np.testing.assert_array_equal(ex["faces_watertight"].numpy(), g["faces_watertight"])
    assert ex["n_verts_watertight"] == int(g["n_verts_watertight"])
    for name, t in (("vertices_open", v), ("L_dev", L), ("vertices_watertight", ex["vertices_watertight"]), ("msdf", ex["msdf"]),
                    ("msdf_watertight", ex["msdf_watertight"]), ("msdf_boundary", ex["msdf_boundary"])):
        np.testing.assert_allclose(t.detach().numpy(), g[name], rtol=1e-5, atol=1e-6, err_msg=name)
        assert tuple(t.shape) == g[name].shape, name
    loss = (v * torch.tensor(g["w_v"])).sum() + (ex["msdf"] * torch.tensor(g["w_m"])).sum() + (L * torch.tensor(g["w_l"])).sum() \
        + (ex["msdf_watertight"] * 0.3).sum()
    loss.backward()
    for name, t in (("g_x", X), ("g_s", S), ("g_nu", NU), ("g_w", Wt)):
        got = t.grad.numpy() if t.grad is not None else np.zeros_like(g[name])
        np.testing.assert_allclose(got, g[name], rtol=2e-4, atol=1e-5 * max(1.0, np.abs(g[name]).max()), err_msg=name)
