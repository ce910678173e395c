"""G-MarchingTets: HIP path (through the C ABI) vs golden vectors from the real reference
and vs the CPU oracle on fresh seeded inputs.  Run on the MI355X box: pytest -m gpu."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import fields, mtets_oracle
from tests.helpers import assert_tangents_match, golden_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda"
FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mtets_*.npz")))


def _run_hip(verts, tets, sdf, msdf, seed, sdf_2d=False):
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    dev = torch.device("cuda")
    pos = torch.tensor(verts, device=dev, requires_grad=True)
    s = torch.tensor(sdf, device=dev)
    if sdf_2d:
        s = s[:, None]
    s.requires_grad_(True)
    m = torch.tensor(msdf, device=dev, requires_grad=True)
    t = torch.tensor(tets, dtype=torch.long, device=dev)
    ext = GShell_Tets()
    v_aug, f_aug, uvs, uv_idx, tng, extra = ext(pos, s, m, t)
    assert uvs is None and uv_idx is None
    wv, wm, ww = fields.loss_weights(v_aug.shape[0], extra["n_verts_watertight"], seed)
    loss = (v_aug * torch.tensor(wv, device=dev)).sum() + (extra["msdf"] * torch.tensor(wm, device=dev)).sum() \
        + (extra["vertices_watertight"] * torch.tensor(ww, device=dev)).sum()
    loss.backward()
    out = dict(verts_aug=v_aug, faces_aug=f_aug, v_tng_aug=tng, vertices_watertight=extra["vertices_watertight"],
               faces_watertight=extra["faces_watertight"], v_tng_watertight=extra["v_tng_watertight"], msdf=extra["msdf"],
               msdf_watertight=extra["msdf_watertight"], msdf_boundary=extra["msdf_boundary"], faces_i32=extra["faces_i32"])
    out = {k: v.detach().cpu().numpy() for k, v in out.items()}
    out["n_verts_watertight"] = extra["n_verts_watertight"]
    out["grad_pos"] = pos.grad.cpu().numpy()
    out["grad_sdf"] = s.grad.reshape(-1).cpu().numpy()
    out["grad_msdf"] = m.grad.cpu().numpy()
    return out


def _compare(out, ref, F, grads=True):
    assert out["n_verts_watertight"] == int(ref["n_verts_watertight"])
    # integer / index work: bit exact
    np.testing.assert_array_equal(out["faces_watertight"], np.asarray(ref["faces_watertight"]))
    np.testing.assert_array_equal(out["faces_aug"], np.asarray(ref["faces_aug"]))
    np.testing.assert_array_equal(out["faces_i32"], np.asarray(ref["faces_aug"]))
    assert out["faces_aug"].dtype == np.int64
    # forward floats: same IEEE ops, same order, no FMA contraction -> exact
    for k in ("verts_aug", "vertices_watertight", "msdf", "msdf_watertight", "msdf_boundary"):
        np.testing.assert_array_equal(out[k], np.asarray(ref[k]), err_msg=k)
    # tangents use float atomics (as the reference's scatter_add_ does): 1e-4 + the vertex's own conditioning bound, no quota
    assert_tangents_match(out["v_tng_aug"], np.asarray(ref["v_tng_aug"]), np.asarray(ref["vertices_watertight"]), np.asarray(ref["faces_watertight"]), F)
    np.testing.assert_array_equal(out["v_tng_watertight"], out["v_tng_aug"][:out["n_verts_watertight"]])
    if grads:
        for k in ("grad_pos", "grad_sdf", "grad_msdf"):
            r = np.asarray(ref[k])
            scale = max(1.0, float(np.abs(r).max()))
            # north_star tolerance: gradients within 1e-4 relative (fp32, atomics reorder sums)
            np.testing.assert_allclose(out[k], r, rtol=1e-4, atol=1e-4 * scale, err_msg=k)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[6:-4] for p in FILES])
def test_hip_matches_reference_golden(path):
    g = np.load(path)
    verts, tets, sdf, msdf = golden_inputs(g)
    out = _run_hip(verts, tets, sdf, msdf, int(g["seed"]))
    _compare(out, g, tets.shape[0])


NOWT_FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "mtetsnowt_*.npz")))


@pytest.mark.parametrize("path", NOWT_FILES, ids=[os.path.basename(p)[10:-4] for p in NOWT_FILES])
def test_hip_matches_reference_golden_without_the_watertight_template(path):
    """GShell_Tets()(..., output_watertight_template=False) against the real reference's output (gshell_tets.py:256-263, :436-441): topology bit-exact (vertex
    numbering of the mSDF-filtered tet set), forward floats bit-exact, gradients to 1e-4, three mSDF entries in `extra`; one case filters every tet away."""
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    g = np.load(path)
    verts, tets, sdf, msdf = golden_inputs(g)
    dev = torch.device("cuda")
    pos = torch.tensor(verts, device=dev, requires_grad=True)
    s = torch.tensor(sdf, device=dev, requires_grad=True)
    m = torch.tensor(msdf, device=dev, requires_grad=True)
    v_aug, f_aug, uvs, uv_idx, tng, extra = GShell_Tets()(pos, s, m, torch.tensor(tets, dtype=torch.long, device=dev), output_watertight_template=False)
    assert uvs is None and uv_idx is None and f_aug.dtype == torch.int64
    assert {"msdf", "msdf_watertight", "msdf_boundary"} <= set(extra) and not any("watertight" in k and k != "msdf_watertight" for k in extra)
    np.testing.assert_array_equal(f_aug.cpu().numpy().reshape(-1, 3), g["faces_aug"].reshape(-1, 3))
    for k, t in (("verts_aug", v_aug), ("msdf", extra["msdf"]), ("msdf_watertight", extra["msdf_watertight"]), ("msdf_boundary", extra["msdf_boundary"])):
        np.testing.assert_array_equal(t.detach().cpu().numpy().reshape(g[k].shape), g[k], err_msg=k)
    V = int(g["msdf_watertight"].shape[0])
    if V:
        # tangents: float atomics on both sides; the watertight faces that condition them are those of the filtered tet set (oracle, pinned to this golden on the CPU)
        o = mtets_oracle.extract(torch.tensor(verts), torch.tensor(sdf), torch.tensor(msdf), torch.tensor(tets), output_watertight_template=False)
        assert_tangents_match(tng.detach().cpu().numpy(), g["v_tng_aug"], o["_vertices_watertight"].numpy(), o["_faces_watertight"].numpy(), tets.shape[0])      # the uv atlas is sized by the WHOLE grid (ref :301, :309)
    wv, wm, _ = fields.loss_weights(v_aug.shape[0], V, int(g["seed"]))
    loss = (v_aug * torch.tensor(wv, device=dev)).sum() + (extra["msdf"] * torch.tensor(wm, device=dev)).sum()
    if loss.requires_grad:
        loss.backward()
    for name, t in (("grad_pos", pos), ("grad_sdf", s), ("grad_msdf", m)):
        r = g[name]
        got = t.grad.reshape(r.shape).cpu().numpy() if t.grad is not None else np.zeros_like(r)
        np.testing.assert_allclose(got, r, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(r).max())), err_msg=name)


def _oracle(verts, tets, sdf, msdf, seed):
    pos = torch.tensor(verts, requires_grad=True)
    s = torch.tensor(sdf, requires_grad=True)
    m = torch.tensor(msdf, requires_grad=True)
    o = mtets_oracle.extract(pos, s, m, torch.tensor(tets))
    wv, wm, ww = fields.loss_weights(o["verts_aug"].shape[0], o["n_verts_watertight"], seed)
    loss = (o["verts_aug"] * torch.tensor(wv)).sum() + (o["msdf"] * torch.tensor(wm)).sum() \
        + (o["vertices_watertight"] * torch.tensor(ww)).sum()
    loss.backward()
    ref = {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in o.items()}
    ref["grad_pos"], ref["grad_sdf"], ref["grad_msdf"] = pos.grad.numpy(), s.grad.numpy(), m.grad.numpy()
    return ref


@pytest.mark.parametrize("gspec,sk,mk,seed,zeros,sdf2d", [
    (("bcc", 20), "sphere_noise", "rand", 11, 0, False),
    (("bcc", 26), "two_spheres", "wavy", 12, 200, True),
    (("kuhn", 24), "skirt", "halfspace", 13, 100, False),
    (("bcc", 9), "plane", "negative", 14, 0, False),
    (("bcc", 40), "skirt", "wavy", 15, 0, True),
])
def test_hip_matches_oracle_seeded(gspec, sk, mk, seed, zeros, sdf2d):
    from gshell_amd import grid
    verts, tets = (grid.bcc_grid(gspec[1]) if gspec[0] == "bcc" else grid.kuhn_grid(gspec[1]))
    verts = verts.numpy()
    verts = verts + fields.make_deform(verts, 1.0 / gspec[1], seed)
    sdf = fields.make_sdf(verts, sk, seed, zeros)
    msdf = fields.make_msdf(verts, mk, seed, zeros)
    out = _run_hip(verts, tets.numpy(), sdf, msdf, seed, sdf_2d=sdf2d)
    ref = _oracle(verts, tets.numpy(), sdf, msdf, seed)
    _compare(out, ref, tets.shape[0])


def test_empty_surface_is_legal():
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    verts, tets = grid.bcc_grid(4, device="cuda")
    sdf = torch.full((verts.shape[0],), -1.0, device="cuda", requires_grad=True)
    msdf = torch.ones_like(sdf)
    v, f, _, _, tng, extra = GShell_Tets()(verts, sdf, msdf, tets)
    assert v.shape == (0, 3) and f.shape == (0, 3) and tng.shape == (0, 3) and extra["n_verts_watertight"] == 0
    (v.sum() + extra["msdf"].sum()).backward()
    assert float(sdf.grad.abs().sum()) == 0.0


def test_full_size_res256_properties():
    """BASELINE config 3 grid (N=2.28 M, F=13.4 M): size-independent invariants."""
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    dev = torch.device("cuda")
    verts, tets = grid.bcc_grid(104, device=dev)
    r = torch.sqrt(verts[:, 0] ** 2 + verts[:, 2] ** 2)
    sdf = torch.minimum(0.26 - 0.18 * verts[:, 1] - r, 0.36 - verts[:, 1].abs())
    msdf = 0.12 - verts[:, 1] + 0.05 * torch.sin(8.0 * verts[:, 0])
    ext = GShell_Tets()
    v, f, _, _, tng, extra = ext(verts, sdf, msdf, tets)
    V, fw = extra["n_verts_watertight"], extra["faces_watertight"]
    assert f.shape[0] > 0 and int(f.max()) < v.shape[0] and int(f.min()) >= 0
    # closed surface strictly inside the box: every watertight edge is shared by exactly two faces
    e = torch.cat([fw[:, [0, 1]], fw[:, [1, 2]], fw[:, [2, 0]]], 0)
    key = torch.minimum(e[:, 0], e[:, 1]) * V + torch.maximum(e[:, 0], e[:, 1])
    _, cnt = torch.unique(key, return_counts=True)
    assert bool((cnt == 2).all())
    # Euler characteristic of a sphere-like closed surface
    assert V - cnt.shape[0] + fw.shape[0] == 2
    # vertices lie on the zero level set of the (linear-per-edge) field: |sdf| small at verts
    rr = torch.sqrt(extra["vertices_watertight"][:, 0] ** 2 + extra["vertices_watertight"][:, 2] ** 2)
    yy = extra["vertices_watertight"][:, 1]
    s_at = torch.minimum(0.26 - 0.18 * yy - rr, 0.36 - yy.abs())
    assert float(s_at.abs().max()) < 5e-3
    # idempotence: same inputs -> identical outputs
    v2, f2, _, _, _, _ = ext(verts, sdf, msdf, tets)
    assert torch.equal(f, f2) and torch.equal(v, v2)
    # the open mesh keeps only the msdf>0 side
    assert float(extra["msdf"][f.reshape(-1)].min()) > -1e-6


def test_presigned_extraction_equals_plain_extraction():
    """Fused geometry front end: the SDF-network kernel writes the sign bits of its result into the extractor's
    occupancy array and the extraction skips its own sign pass -- same mesh, bit for bit; a stale tag must not be trusted."""
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    from gshell_amd.geometry.mlp import MLP, forward_row_sparse_backward
    torch.manual_seed(0)
    verts, tets = grid.bcc_grid(12)
    verts, tets = (verts * 1.4).to(DEV), tets.to(DEV)
    net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).to(DEV)
    with torch.no_grad():      # a surface inside the grid: shift the output bias to the median
        lin = [m for m in net.net if isinstance(m, torch.nn.Linear)]
        lin[-1].bias -= net(verts).median()
    msdf = torch.rand(verts.shape[0], device=DEV) - 0.1
    ext = GShell_Tets(compute_tangents=False)
    topo = ext.topology(tets, verts.shape[0])
    sdf_tagged = forward_row_sparse_backward(net, verts, sign_sink=topo)
    assert getattr(sdf_tagged, "_gs_presigned", None) is not None
    v1, f1, _, _, _, e1 = ext(verts, sdf_tagged, msdf, tets)
    sdf_plain = sdf_tagged.detach().clone()
    v2, f2, _, _, _, e2 = ext(verts, sdf_plain, msdf, tets)
    assert f1.shape[0] > 100
    assert torch.equal(f1, f2) and torch.equal(v1, v2) and torch.equal(e1["msdf"], e2["msdf"])
    # the plain call rewrote the bits: the old tag is stale now and must be ignored (other field, same topology)
    other = -sdf_plain
    ext(verts, other, msdf, tets)
    v3, f3, _, _, _, _ = ext(verts, sdf_tagged, msdf, tets)
    assert torch.equal(f3, f1) and torch.equal(v3, v1)


@pytest.mark.parametrize("gspec,sk,mk,seed", [(("bcc", 12), "sphere", "wavy", 0), (("bcc", 20), "sphere_noise", "rand", 11), (("kuhn", 16), "skirt", "halfspace", 13),
                                              (("bcc", 26), "skirt", "wavy", 15)])
def test_tangent_gradient_matches_autograd_of_the_oracle(gspec, sk, mk, seed):
    """E5: d loss / d (pos, sdf, msdf) through v_tng_aug -- compute_tangents + auto_normals + the boundary interpolation with the mSDF
    weights (gshell_tets.py:9-78, :318-319, :375-380) -- gs_mtets_tangents_bwd vs autograd through oracle/mtets_oracle._tangents (pinned to
    goldens minted from the real reference).  The unit tangent of a vertex whose face terms nearly cancel has an unbounded derivative, so
    the float32 answer itself is only defined up to the float32-vs-float64 difference of the ORACLE: the bar is 1e-4 or 4 x that floor."""
    from gshell_amd import grid
    from gshell_amd.geometry.gshell_tets import GShell_Tets
    verts, tets = (grid.bcc_grid(gspec[1]) if gspec[0] == "bcc" else grid.kuhn_grid(gspec[1]))
    vn = verts.numpy()
    vn = (vn + fields.make_deform(vn, 1.0 / gspec[1], seed)).astype(np.float32)
    sdf, msdf = fields.make_sdf(vn, sk, seed).astype(np.float32), fields.make_msdf(vn, mk, seed).astype(np.float32)

    def oracle(dt):
        pos = torch.tensor(vn, dtype=dt, requires_grad=True)
        s = torch.tensor(sdf, dtype=dt, requires_grad=True)
        m = torch.tensor(msdf, dtype=dt, requires_grad=True)
        o = mtets_oracle.extract(pos, s, m, tets)
        w = torch.tensor(np.random.default_rng(seed).normal(size=tuple(o["v_tng_aug"].shape)), dtype=dt)
        (o["v_tng_aug"] * w).sum().backward()
        return o, w, (pos.grad, s.grad, m.grad)
    o32, w, g32 = oracle(torch.float32)
    pos = torch.tensor(vn, device=DEV, requires_grad=True)
    s = torch.tensor(sdf, device=DEV, requires_grad=True)
    m = torch.tensor(msdf, device=DEV, requires_grad=True)
    v, f, _, _, tng, extra = GShell_Tets()(pos, s, m, tets.to(DEV))
    assert tng.requires_grad and extra["v_tng_watertight"].requires_grad and tng.shape == o32["v_tng_aug"].shape
    (tng * w.to(DEV)).sum().backward()
    if o32["v_tng_aug"].shape[0] == 0:
        return
    # float64 run of the same oracle on the same float32 inputs: what float32 round-off alone does to this gradient
    try:
        _, _, g64 = oracle(torch.float64)
    except Exception:
        g64 = None
    for name, a, b32, k in (("pos", pos.grad, g32[0], 0), ("sdf", s.grad, g32[1], 1), ("msdf", m.grad, g32[2], 2)):
        a = a.cpu()
        assert torch.isfinite(a).all() and float(b32.abs().max()) > 0, name
        rel = float((a - b32).norm() / b32.norm())
        floor = float((b32.double() - g64[k]).norm() / g64[k].norm()) if g64 is not None else 0.0
        print(f"  tangent gradient {name}: HIP vs oracle(f32) rel L2 {rel:.2e}; oracle f32 vs f64 {floor:.2e}")
        assert rel <= max(1e-4, 4.0 * floor), (name, rel, floor)
    # and the training path is untouched: tangents unused -> the extraction's own gradients only
    pos.grad = None
    v2, _, _, _, tng2, _ = GShell_Tets()(pos, s, m, tets.to(DEV))
    v2.sum().backward()
    assert torch.isfinite(pos.grad).all()
