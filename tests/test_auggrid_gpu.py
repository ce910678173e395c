"""Generative-decode extraction (marching_from_auggrid): HIP path vs goldens minted from the
real reference (tests/golden/auggrid_*.npz) and vs the CPU oracle at a larger size; plus the
geometry-level entry point getMesh_from_augmented_grid_withocc.  Run: pytest -m gpu."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fields, mtets_oracle
from tests.helpers import assert_tangents_match, auggrid_inputs

pytestmark = pytest.mark.gpu
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "auggrid_*.npz")))
EDGE_A, EDGE_B = [0, 0, 0, 1, 1, 2], [1, 2, 3, 2, 3, 3]


def _sorted_edges(t):
    a, b = t[:, EDGE_A], t[:, EDGE_B]
    return torch.stack([torch.minimum(a, b), torch.maximum(a, b)], -1)


def _run_hip(pos, tets, sdf, vdisc, coeff, mgrid, occ):
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    dev = torch.device("cuda")
    t = torch.tensor(tets, dtype=torch.long, device=dev)
    ext = GShell_Tets()
    out = ext.marching_from_auggrid(torch.tensor(pos, device=dev), torch.tensor(sdf, device=dev), t, _sorted_edges(t),
                                    torch.tensor(coeff, device=dev), torch.tensor(vdisc, device=dev).float(),
                                    torch.tensor(mgrid, device=dev), torch.tensor(occ, device=dev))
    assert len(out) == 9 and out[2] is None and out[3] is None
    names = ("verts_aug", "faces_aug", None, None, "v_tng_aug", "vertices_watertight", "valid_tet_gidx", "msdf", "msdf_watertight")
    return {n: v.cpu().numpy() for n, v in zip(names, out) if n}


def _compare(out, ref, F, faces_wt=None):
    np.testing.assert_array_equal(out["faces_aug"], np.asarray(ref["faces_aug"]))            # index work: bit exact
    np.testing.assert_array_equal(out["valid_tet_gidx"], np.asarray(ref["valid_tet_gidx"]))
    for k in ("verts_aug", "vertices_watertight", "msdf", "msdf_watertight"):                 # same IEEE ops, no contraction
        np.testing.assert_array_equal(out[k], np.asarray(ref[k]), err_msg=k)
    # tangents use float atomics (as the reference's scatter_add_ does): 1e-4 + the vertex's own conditioning bound, no quota
    assert_tangents_match(out["v_tng_aug"], np.asarray(ref["v_tng_aug"]), np.asarray(ref["vertices_watertight"]),
                          np.asarray(ref["faces_watertight"] if faces_wt is None else faces_wt), F)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[8:-4] for p in FILES])
def test_auggrid_hip_matches_reference_golden(path):
    g = np.load(path)
    ins = auggrid_inputs(g)
    pos, tets, sdf, vdisc, coeff, mgrid, occ = ins
    # the golden holds the reference's 9-tuple, which has no watertight faces: take them from the oracle (bit-exact topology)
    o = mtets_oracle.extract_from_auggrid(torch.tensor(pos), torch.tensor(sdf), torch.tensor(tets), torch.tensor(vdisc), torch.tensor(coeff),
                                          torch.tensor(mgrid), torch.tensor(occ))
    _compare(_run_hip(*ins), g, tets.shape[0], faces_wt=o["faces_watertight"].numpy())


def test_auggrid_hip_matches_oracle_bcc40():
    from gshell_amd import grid
    n, seed = 40, 11
    verts, tets = grid.bcc_grid(n)
    verts = verts.numpy().astype(np.float32)
    vdisc = fields.discretize_verts(verts)
    pos = (verts + fields.make_deform(verts, 1.0 / n, seed)).astype(np.float32)
    sdf = np.sign(fields.make_sdf(verts, "skirt", seed, 50)).astype(np.float32)
    coeff, mgrid, occ = fields.make_aug_grids(int(vdisc.max()) + 1, seed, "sign")
    ref = mtets_oracle.extract_from_auggrid(torch.tensor(pos), torch.tensor(sdf), tets, torch.tensor(vdisc), torch.tensor(coeff),
                                            torch.tensor(mgrid), torch.tensor(occ))
    assert ref["faces_aug"].shape[0] > 20000
    _compare(_run_hip(pos, tets.numpy(), sdf, vdisc, coeff, mgrid, occ), {k: v.numpy() for k, v in ref.items()}, tets.shape[0])


def test_auggrid_rejects_foreign_edge_table_and_small_grids():
    from gshell_amd import grid
    from gshell_amd._lib import GShellHipError
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    dev = torch.device("cuda")
    verts, tets = grid.kuhn_grid(4)
    vdisc = torch.tensor(fields.discretize_verts(verts.numpy()), device=dev).float()
    G = int(vdisc.max()) + 1
    t = tets.to(dev)
    ones, occ = torch.ones(G, G, G, device=dev), torch.zeros(2 * G, 2 * G, 2 * G, device=dev)
    sdf = torch.tensor(np.sign(fields.make_sdf(verts.numpy(), "sphere", 0, 0)).astype(np.float32), device=dev)
    ext = GShell_Tets()
    with pytest.raises(GShellHipError):
        ext.marching_from_auggrid(verts.to(dev), sdf, t, _sorted_edges(t).flip(1), ones, vdisc, ones, occ)
    with pytest.raises(IndexError):
        ext.marching_from_auggrid(verts.to(dev), sdf, t, _sorted_edges(t), ones[:-2, :-2, :-2].contiguous(), vdisc,
                                  ones[:-2, :-2, :-2].contiguous(), occ)


def test_getmesh_from_augmented_grid():
    """Geometry-level decode of one generated grid, following eval_gmeshdiffusion_generated_samples.py:160-180."""
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets_geometry import GShellTetsGeometry
    from gshell_amd.train import default_flags
    FLAGS = default_flags()
    FLAGS.use_sdf_mlp = False
    n = 16
    verts, tets = grid.bcc_grid(n)
    geo = GShellTetsGeometry(n, 1.0, FLAGS, tet_grid=(verts.numpy(), tets.numpy()), extract_from_generative=True)
    vd = geo.verts_discretized.long()
    G = int(vd.max()) + 1
    gen = torch.Generator(device="cuda").manual_seed(3)
    # a "generated" cubic grid: channel 0 = sdf sign at vertex cells and mSDF sign at edge-midpoint cells, 1:4 = deformation
    centre = (G - 1) / 2.0
    ax = torch.arange(G, device="cuda").float()
    r = ((ax[:, None, None] - centre) ** 2 + (ax[None, :, None] - centre) ** 2 + (ax[None, None, :] - centre) ** 2).sqrt()
    cubic = torch.zeros(4, G, G, G, device="cuda")
    cubic[0] = torch.sign(r - 0.3 * G)
    cubic[1:4] = torch.rand(3, G, G, G, device="cuda", generator=gen) * 0.6 - 0.3
    sdf_sign = cubic[0, vd[:, 0], vd[:, 1], vd[:, 2]]
    geo.deform.data[:] = cubic[1:4, vd[:, 0], vd[:, 1], vd[:, 2]].transpose(0, 1).clamp(-1, 1)
    sdf_coeff = torch.full((G, G, G), 0.5, device="cuda")
    msdf_sign = torch.sign(0.15 * G + centre - ax)[None, :, None].expand(G, G, G).contiguous()      # open top
    occgrid = torch.rand(2 * G, 2 * G, 2 * G, device="cuda", generator=gen) * 2 - 1
    out = geo.getMesh_from_augmented_grid_withocc(None, torch.sign(sdf_sign), sdf_coeff, msdf_sign, occgrid)
    m = out['imesh']
    assert m.t_pos_idx.shape[0] > 1000 and m.v_tng.shape == m.v_pos.shape and out['v_msdf'].shape[0] == m.v_pos.shape[0]
    assert torch.isfinite(m.v_pos).all() and torch.isfinite(m.v_nrm).all() and torch.isfinite(m.v_tng).all()
    # tangent frame is orthonormal to the smooth normal wherever a vertex is referenced
    used = torch.zeros(m.v_pos.shape[0], dtype=torch.bool, device="cuda")
    used[m.t_pos_idx.reshape(-1)] = True
    assert ((m.v_tng * m.v_nrm).sum(-1)[used].abs() < 1e-3).all()
    # and it equals the oracle on the same inputs
    ref = mtets_oracle.extract_from_auggrid((geo.verts + geo.max_displacement * geo.deform).detach().cpu(), torch.sign(sdf_sign).cpu(),
                                            geo.indices.cpu(), vd.cpu(), sdf_coeff.cpu(), msdf_sign.cpu(), occgrid.cpu(),
                                            with_tangents=False)
    np.testing.assert_array_equal(m.t_pos_idx.cpu().numpy(), ref["faces_aug"].numpy())
    np.testing.assert_array_equal(m.v_pos.cpu().numpy(), ref["verts_aug"].numpy())
