"""Wavefront OBJ at the edge of the path: `write_obj(folder, mesh)` writes what the reference's
exporter writes (render/obj.py:143-196: `mesh.obj` with v / vt (v flipped) / vn records and
1-based `f a/ta/na` corners) and `load_obj` reads positions + triangles back.  Materials are
an MLP field here (no texture atlas), so `save_material` only emits a stub .mtl."""
import os

import numpy as np
import torch

from . import mesh as meshlib


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def write_obj(folder, mesh, save_material=True, name="mesh"):
    os.makedirs(folder, exist_ok=True)
    obj_file = os.path.join(folder, name + ".obj")
    v_pos, v_nrm, v_tex = _np(mesh.v_pos), _np(mesh.v_nrm), _np(mesh.v_tex)
    t_pos, t_nrm, t_tex = _np(mesh.t_pos_idx), _np(mesh.t_nrm_idx), _np(mesh.t_tex_idx)
    # numpy scalars are formatted exactly as the reference formats them ('{}'.format(np.float32) = shortest float32 repr;
    # `1.0 - v` follows numpy's own promotion), so the file is byte-identical to the reference writer's for the same mesh
    lines = [f"mtllib {name}.mtl", "g default"]
    lines += ["v {} {} {} ".format(v[0], v[1], v[2]) for v in v_pos]
    if v_tex is not None:
        assert len(t_pos) == len(t_tex)
        lines += ["vt {} {} ".format(v[0], 1.0 - v[1]) for v in v_tex]
    if v_nrm is not None:
        assert len(t_pos) == len(t_nrm)
        lines += ["vn {} {} {}".format(v[0], v[1], v[2]) for v in v_nrm]
    lines += ["s 1 ", "g pMesh1", "usemtl defaultMat"]
    for i in range(len(t_pos)):
        lines.append("f " + "".join(" %s/%s/%s" % (str(t_pos[i][j] + 1), "" if v_tex is None else str(t_tex[i][j] + 1),
                                                   "" if v_nrm is None else str(t_nrm[i][j] + 1)) for j in range(3)))
    with open(obj_file, "w") as f:
        f.write("\n".join(lines) + "\n")
    if save_material:
        with open(os.path.join(folder, name + ".mtl"), "w") as f:
            f.write("newmtl defaultMat\nbsdf   pbr\nKd 0.5 0.5 0.5\nKs 0.0 0.25 0.0\n")
    return obj_file


def load_obj(filename, device="cuda"):
    """Positions and triangle indices (fans for polygons) of an OBJ file -> Mesh."""
    verts, faces = [], []
    with open(filename) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                idx = [int(c.split("/")[0]) - 1 for c in tok[1:]]
                faces += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
    v = torch.tensor(np.asarray(verts, np.float32).reshape(-1, 3), device=device)
    t = torch.tensor(np.asarray(faces, np.int64).reshape(-1, 3), device=device)
    return meshlib.Mesh(v, t)
