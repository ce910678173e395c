"""Mints tests/golden/obj_reference_writer.{npz,obj}: the text the REAL reference exporter (render/obj.py:143-196 `write_obj`)
writes for a small mesh with normals and texture coordinates.  Build container only (needs /root/reference).

    python -m oracle.make_golden_obj
The function is lifted out of the reference source with `ast` (render/obj.py imports nvdiffrast-dependent modules at the top;
write_obj itself needs only `os`)."""
import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402


def reference_write_obj():
    path = os.path.join(refload.REF_ROOT, "render", "obj.py")
    tree = ast.parse(open(path).read(), path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "write_obj"][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"os": os, "material": types.SimpleNamespace(save_mtl=lambda *a, **k: None), "print": lambda *a, **k: None}
    exec(compile(mod, path, "exec"), ns)
    return ns["write_obj"]


def sample_mesh():
    g = torch.Generator().manual_seed(7)
    v = torch.randn(9, 3, generator=g) * torch.tensor([1.0, 1e-3, 1e3])
    v[0] = torch.tensor([0.1, -0.5, 2.0])
    n = torch.nn.functional.normalize(torch.randn(7, 3, generator=g), dim=1)
    uv = torch.rand(8, 2, generator=g)
    t = torch.randint(0, 9, (13, 3), generator=g)
    tn = torch.randint(0, 7, (13, 3), generator=g)
    tt = torch.randint(0, 8, (13, 3), generator=g)
    return v, n, uv, t, tn, tt


if __name__ == "__main__":
    v, n, uv, t, tn, tt = sample_mesh()
    mesh = types.SimpleNamespace(v_pos=v, v_nrm=n, v_tex=uv, t_pos_idx=t, t_nrm_idx=tn, t_tex_idx=tt, material=None)
    with tempfile.TemporaryDirectory() as d:
        reference_write_obj()(d, mesh, save_material=False)
        text = open(os.path.join(d, "mesh.obj")).read()
    out = os.path.join(ROOT, "tests", "golden")
    np.savez(os.path.join(out, "obj_reference_writer.npz"), v=v.numpy(), n=n.numpy(), uv=uv.numpy(), t=t.numpy(), tn=tn.numpy(), tt=tt.numpy())
    open(os.path.join(out, "obj_reference_writer.obj"), "w").write(text)
    print(text[:400])
