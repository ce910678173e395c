"""Pins oracle/mtets_oracle.py to the golden vectors minted from the real reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fields, mtets_oracle
from tests.helpers import golden_inputs

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mtets_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    verts, tets, sdf, msdf = golden_inputs(g)
    pos = torch.tensor(verts, requires_grad=True)
    s = torch.tensor(sdf, requires_grad=True)
    m = torch.tensor(msdf, requires_grad=True)
    out = mtets_oracle.extract(pos, s, m, torch.tensor(tets))
    # topology: bit exact
    assert out["n_verts_watertight"] == int(g["n_verts_watertight"])
    np.testing.assert_array_equal(out["faces_watertight"].numpy(), g["faces_watertight"])
    np.testing.assert_array_equal(out["faces_aug"].numpy(), g["faces_aug"])
    # float outputs: same IEEE ops in the same order -> identical on CPU
    for k in ("verts_aug", "vertices_watertight", "msdf", "msdf_watertight", "msdf_boundary"):
        np.testing.assert_array_equal(out[k].detach().numpy(), g[k], err_msg=k)
    for k in ("v_tng_aug", "v_tng_watertight"):
        np.testing.assert_allclose(out[k].detach().numpy(), g[k], rtol=0, atol=2e-5, err_msg=k)
    wv, wm, ww = fields.loss_weights(out["verts_aug"].shape[0], out["n_verts_watertight"], int(g["seed"]))
    loss = (out["verts_aug"] * torch.tensor(wv)).sum() + (out["msdf"] * torch.tensor(wm)).sum() \
        + (out["vertices_watertight"] * torch.tensor(ww)).sum()
    if loss.requires_grad:
        loss.backward()
        for name, t in (("grad_pos", pos), ("grad_sdf", s), ("grad_msdf", m)):
            ref = g[name]
            got = t.grad.numpy() if t.grad is not None else np.zeros_like(ref)
            scale = max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * scale, err_msg=name)


NOWT_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mtetsnowt_*.npz")))


@pytest.mark.parametrize("path", NOWT_FILES, ids=[os.path.basename(p)[10:-4] for p in NOWT_FILES])
def test_oracle_matches_reference_golden_without_the_watertight_template(path):
    """output_watertight_template=False (reference gshell_tets.py:256-263: the mSDF pre-filter of the valid tets; :436-441: three entries in `extra`)"""
    g = np.load(path)
    assert len(NOWT_FILES) >= 5
    verts, tets, sdf, msdf = golden_inputs(g)
    pos = torch.tensor(verts, requires_grad=True)
    s = torch.tensor(sdf, requires_grad=True)
    m = torch.tensor(msdf, requires_grad=True)
    out = mtets_oracle.extract(pos, s, m, torch.tensor(tets), output_watertight_template=False)
    assert not any(k in out for k in ("n_verts_watertight", "vertices_watertight", "faces_watertight", "v_tng_watertight"))
    np.testing.assert_array_equal(out["faces_aug"].numpy().reshape(-1, 3), g["faces_aug"].reshape(-1, 3))
    for k in ("verts_aug", "msdf", "msdf_watertight", "msdf_boundary"):
        np.testing.assert_array_equal(out[k].detach().numpy().reshape(g[k].shape), g[k], err_msg=k)
    np.testing.assert_allclose(out["v_tng_aug"].detach().numpy().reshape(g["v_tng_aug"].shape), g["v_tng_aug"], rtol=0, atol=2e-5)
    wv, wm, _ = fields.loss_weights(out["verts_aug"].shape[0], out["msdf_watertight"].shape[0], int(g["seed"]))
    loss = (out["verts_aug"] * torch.tensor(wv)).sum() + (out["msdf"] * torch.tensor(wm)).sum()
    if loss.requires_grad:
        loss.backward()
    for name, t in (("grad_pos", pos), ("grad_sdf", s), ("grad_msdf", m)):
        ref = g[name]
        got = t.grad.numpy() if t.grad is not None else np.zeros_like(ref)
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * scale, err_msg=name)


AUG_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "auggrid_*.npz")))


@pytest.mark.parametrize("path", AUG_FILES, ids=[os.path.basename(p)[8:-4] for p in AUG_FILES])
def test_auggrid_oracle_matches_reference_golden(path):
    from tests.helpers import auggrid_inputs
    g = np.load(path)
    pos, tets, sdf, vdisc, coeff, mgrid, occ = auggrid_inputs(g)
    out = mtets_oracle.extract_from_auggrid(torch.tensor(pos), torch.tensor(sdf), torch.tensor(tets), torch.tensor(vdisc),
                                            torch.tensor(coeff), torch.tensor(mgrid), torch.tensor(occ))
    np.testing.assert_array_equal(out["faces_aug"].numpy(), g["faces_aug"])
    np.testing.assert_array_equal(out["valid_tet_gidx"].numpy(), g["valid_tet_gidx"])
    for k in ("verts_aug", "vertices_watertight", "msdf", "msdf_watertight"):
        np.testing.assert_array_equal(out[k].numpy(), g[k], err_msg=k)
    np.testing.assert_allclose(out["v_tng_aug"].numpy(), g["v_tng_aug"], rtol=0, atol=2e-5)
