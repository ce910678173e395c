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
