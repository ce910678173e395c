"""OBJ writer/reader round trip (format of the reference's render/obj.py:143-196)."""
import torch

from gshell_amd.render import obj
from gshell_amd.render.mesh import Mesh


def test_write_then_load_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    v = torch.randn(11, 3, generator=g)
    t = torch.randint(0, 11, (17, 3), generator=g)
    n = torch.nn.functional.normalize(torch.randn(11, 3, generator=g), dim=1)
    path = obj.write_obj(str(tmp_path), Mesh(v, t, v_nrm=n, t_nrm_idx=t))
    text = open(path).read().splitlines()
    assert text[0] == "mtllib mesh.mtl" and sum(l.startswith("v ") for l in text) == 11
    assert sum(l.startswith("vn ") for l in text) == 11 and sum(l.startswith("f ") for l in text) == 17
    first_face = [l for l in text if l.startswith("f ")][0].split()
    a = int(t[0, 0]) + 1
    assert first_face[1] == f"{a}//{a}"                       # 1-based, empty texcoord slot
    m = obj.load_obj(path, device="cpu")
    assert torch.equal(m.t_pos_idx, t) and torch.allclose(m.v_pos, v, rtol=0, atol=1e-6)


def test_writer_is_byte_identical_to_the_reference_exporter(tmp_path):
    """Golden minted by running the REAL render/obj.py:143-196 write_obj (oracle/make_golden_obj.py): same records, same
    order, same number formatting, same face corner syntax -- byte for byte."""
    import os
    import numpy as np
    g = os.path.join(os.path.dirname(__file__), "golden")
    a = np.load(os.path.join(g, "obj_reference_writer.npz"))
    m = Mesh(torch.tensor(a["v"]), torch.tensor(a["t"]), v_nrm=torch.tensor(a["n"]), t_nrm_idx=torch.tensor(a["tn"]), v_tex=torch.tensor(a["uv"]),
             t_tex_idx=torch.tensor(a["tt"]))
    path = obj.write_obj(str(tmp_path), m, save_material=False)
    assert open(path).read() == open(os.path.join(g, "obj_reference_writer.obj")).read()
    from oracle import refload
    if refload.reference_available():           # build container: also against the live reference function
        from oracle.make_golden_obj import reference_write_obj
        d = tmp_path / "ref"
        d.mkdir()
        reference_write_obj()(str(d), m, save_material=False)
        assert open(d / "mesh.obj").read() == open(path).read()
