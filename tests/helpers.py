import numpy as np

from gshell_amd import grid
from oracle import fields


def golden_inputs(g):
    """Rebuild (verts, tets, sdf, msdf) of a golden fixture (literal arrays if stored,
    otherwise regenerated from its recipe -- see oracle/make_golden_mtets.py)."""
    if "in_verts" in g.files:
        return g["in_verts"], g["in_tets"].astype(np.int64), g["in_sdf"], g["in_msdf"]
    kind, n, seed = str(g["grid_kind"]), int(g["grid_n"]), int(g["seed"])
    verts, tets = (grid.bcc_grid(n) if kind == "bcc" else grid.kuhn_grid(n))
    verts = verts.numpy()
    if bool(g["deform"]):
        verts = verts + fields.make_deform(verts, 1.0 / n, seed)
    sdf = fields.make_sdf(verts, str(g["sdf_kind"]), seed, int(g["sdf_zeros"]))
    msdf = fields.make_msdf(verts, str(g["msdf_kind"]), seed, int(g["msdf_zeros"]))
    return verts.astype(np.float32), tets.numpy(), sdf, msdf


def auggrid_inputs(g):
    """Rebuild the inputs of an auggrid golden fixture from its recipe
    (see oracle/make_golden_auggrid.py): pos, tets, sdf sign, cells, coeff, msdf grid, occgrid."""
    kind, n, seed = str(g["grid_kind"]), int(g["grid_n"]), int(g["seed"])
    verts, tets = (grid.bcc_grid(n) if kind == "bcc" else grid.kuhn_grid(n))
    verts = verts.numpy().astype(np.float32)
    vdisc = fields.discretize_verts(verts)
    pos = verts + fields.make_deform(verts, 1.0 / n, seed) if bool(g["deform"]) else verts
    sdf = np.sign(fields.make_sdf(verts, str(g["sdf_kind"]), seed, int(g["sdf_zeros"]))).astype(np.float32)
    G = int(vdisc.max()) + 1
    coeff, mgrid, occ = fields.make_aug_grids(G, seed, str(g["msdf_kind"]))
    return pos.astype(np.float32), tets.numpy(), sdf, vdisc, coeff, mgrid, occ
