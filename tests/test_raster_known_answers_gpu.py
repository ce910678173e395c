"""the analytic known answers of the rasterise / interpolate / antialias stage THROUGH THE HIP ENTRY POINTS (the same
cases pin the oracle itself in tests/test_raster_oracle.py).  These do not depend on any restatement of nvdiffrast: exact coverage of an
axis-aligned square, watertightness of shared edges (no gap, no double hit, both windings), nearest-depth / lower-id / depth-range rules,
perspective-correct barycentrics (interpolating clip-space position returns the pixel's own NDC), antialias coverage = the crossing
fraction of a silhouette edge between two pixel centres, no blending across interior edges, near-plane clipping = explicit clipping."""
import numpy as np
import pytest
import torch

from oracle import scenes

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ndc(pts, z=0.0, w=1.0):
    return torch.tensor([[x * w, y * w, z * w, w] for x, y in pts], dtype=torch.float32, device=DEV)[None]


def _ids(pos, tri, H, W):
    from gshell_amd.render import rast as dr
    rast, db = dr.rasterize(None, pos, tri, (H, W))
    return rast, db, (rast[..., 3].long() - 1)


def _tri(rows):
    return torch.tensor(rows, dtype=torch.int32, device=DEV)


def test_axis_aligned_square_coverage():
    pos = _ndc([(-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)])
    _, _, ids = _ids(pos, _tri([[0, 1, 2], [0, 2, 3]]), 8, 8)
    expect = torch.zeros(8, 8, dtype=torch.bool, device=DEV)
    expect[2:6, 2:6] = True                      # pixel centres -0.375 .. 0.375
    assert torch.equal(ids[0] >= 0, expect)
    assert set(ids[0][expect].unique().tolist()) == {0, 1}       # the diagonal's centres belong to exactly one of the two


def test_shared_edges_watertight_fan_both_windings():
    rng = np.random.default_rng(3)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 9))
    pos = _ndc([(0.03, -0.02)] + [(0.8 * np.cos(a), 0.8 * np.sin(a)) for a in ang])
    fan = [[0, 1 + i, 1 + (i + 1) % 9] for i in range(9)]
    Hh = Ww = 64
    _, _, all_ids = _ids(pos, _tri(fan), Hh, Ww)
    count = torch.zeros(Hh, Ww, dtype=torch.int64, device=DEV)
    for t in range(9):
        count += (_ids(pos, _tri(fan[t:t + 1]), Hh, Ww)[2][0] >= 0).long()
    assert int(count.max()) == 1, "a pixel centre is covered by two fan triangles"
    assert torch.equal(count == 1, all_ids[0] >= 0)
    _, _, flipped = _ids(pos, _tri([f[::-1] for f in fan]), Hh, Ww)
    assert torch.equal(flipped >= 0, all_ids >= 0)


def test_depth_order_range_and_ties():
    shape = [(-1, -1), (1, -1), (0, 1)]
    pos = torch.cat([_ndc(shape, z=0.5), _ndc(shape, z=-0.5), _ndc(shape, z=1.5)], dim=1)       # far, near, beyond the far plane
    tri = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert set(_ids(pos, _tri(tri), 16, 16)[2].unique().tolist()) == {-1, 1}                      # nearest wins
    assert set(_ids(pos, _tri([tri[0], tri[2]]), 16, 16)[2].unique().tolist()) == {-1, 0}         # z/w > 1 is clipped
    assert set(_ids(pos, _tri([tri[1], [3, 4, 5]]), 16, 16)[2].unique().tolist()) == {-1, 0}      # equal depth: lower id
    rast = _ids(pos, _tri(tri), 16, 16)[0]
    assert torch.allclose(rast[..., 2][rast[..., 3] > 0], torch.tensor(-0.5, device=DEV))


def test_perspective_correct_barycentrics_return_the_pixel_ndc():
    from gshell_amd.render import rast as dr, renderutils as ru
    verts, tri = scenes.grid_sheet(4, seed=1)
    mvp, _ = scenes.orbit_views(2)
    pos = ru.xfm_points(torch.tensor(verts, device=DEV)[None], torch.tensor(mvp, device=DEV))
    Hh = Ww = 48
    tri_d = torch.tensor(tri, device=DEV)
    rast, db, ids = _ids(pos, tri_d, Hh, Ww)
    m = ids >= 0
    assert float(m.float().mean()) > 0.05
    out = dr.interpolate(pos, rast, tri_d)[0]
    X = ((torch.arange(Ww, device=DEV) + 0.5) * 2 / Ww - 1)[None, None, :].expand(2, Hh, Ww)
    Y = ((torch.arange(Hh, device=DEV) + 0.5) * 2 / Hh - 1)[None, :, None].expand(2, Hh, Ww)
    assert torch.allclose((out[..., 0] / out[..., 3])[m], X[m], atol=2e-4)
    assert torch.allclose((out[..., 1] / out[..., 3])[m], Y[m], atol=2e-4)
    assert torch.allclose((out[..., 2] / out[..., 3])[m], rast[..., 2][m], atol=1e-5)
    # rast_db = d(u)/dX: finite differences inside one triangle
    u = rast[..., 0]
    same = (ids[:, :, 1:] == ids[:, :, :-1]) & m[:, :, 1:]
    fd = (u[:, :, 1:] - u[:, :, :-1])[same]
    an = 0.5 * (db[..., 0][:, :, 1:] + db[..., 0][:, :, :-1])[same]
    assert torch.allclose(fd, an, atol=2e-3)
    # barycentrics of a constant attribute: exactly the coverage mask
    one = dr.interpolate(torch.ones(1, verts.shape[0], 1, device=DEV), rast, tri_d)[0]
    assert torch.allclose(one[..., 0], m.float(), atol=1e-6)


@pytest.mark.parametrize("axis", ["vertical", "horizontal"])
@pytest.mark.parametrize("e_pix", [3.2, 3.5, 3.9, 4.3])
def test_antialias_coverage_is_the_crossing_fraction(axis, e_pix):
    """A half-plane x <= e (or y <= e): exact area coverage of pixel column c is clamp(e - c, 0, 1); the op blends the two pixels adjacent
    to the edge linearly with the crossing fraction between their centres, and conserves total coverage along the edge."""
    from gshell_amd.render import rast as dr
    Hh = Ww = 8
    e = e_pix / Ww * 2 - 1
    pts = [(-3, -3), (e, -3), (e, 3), (-3, 3)]
    if axis == "horizontal":
        pts = [(y, x) for x, y in pts]
    pos = _ndc(pts)
    tri = _tri([[0, 1, 2], [0, 2, 3]])
    rast, _, ids = _ids(pos, tri, Hh, Ww)
    color = (ids >= 0).float()[..., None].contiguous()
    out = dr.antialias(color, rast, pos, tri)
    line = out[0, 4, :, 0] if axis == "vertical" else out[0, :, 4, 0]
    base = color[0, 4, :, 0] if axis == "vertical" else color[0, :, 4, 0]
    last_in = int(np.floor(e_pix - 0.5))
    dc = e_pix - (last_in + 0.5)
    expect = base.clone()
    if dc < 0.5:
        expect[last_in] = 1 - (0.5 - dc)
    else:
        expect[last_in + 1] = dc - 0.5
    assert torch.allclose(line, expect, atol=1e-5), (e_pix, line.tolist(), expect.tolist())
    assert abs(float(out.sum()) - Hh * e_pix) < 1e-3
    # d(coverage) / d(edge position): moving the edge by dx pixels changes the total by H * dx.  (Edge exactly THROUGH a pixel centre:
    # the crossing distance sits on the clamp's corner, where either one-sided derivative is legitimate -- the kernel takes 0.)
    if dc == 0.0:
        return
    p = pos.clone().requires_grad_(True)
    dr.antialias(color, rast, p, tri).sum().backward()
    k = 0 if axis == "vertical" else 1
    g_edge = float(p.grad[0, 1, k] + p.grad[0, 2, k])                       # the two vertices on the edge
    assert abs(g_edge - Hh * Ww / 2) < 1e-2 * Hh * Ww / 2, g_edge         # d total / d e_ndc = H * (W / 2)


def test_antialias_does_not_blend_interior_edges():
    from gshell_amd.render import rast as dr, renderutils as ru
    verts, tri = scenes.grid_sheet(6, seed=2)
    mvp, _ = scenes.orbit_views(1)
    pos = ru.xfm_points(torch.tensor(verts, device=DEV)[None], torch.tensor(mvp, device=DEV))
    tri_d = torch.tensor(tri, device=DEV)
    rast, _, ids = _ids(pos, tri_d, 40, 40)
    alpha = dr.aa_analyze(rast, pos, tri_d, dr.AATopology(tri_d, verts.shape[0]))
    both_r = (ids[:, :, :-1] >= 0) & (ids[:, :, 1:] >= 0)
    both_d = (ids[:, :-1] >= 0) & (ids[:, 1:] >= 0)
    assert (alpha[:, :, :-1, 0][both_r] == 0).all() and (alpha[:, :-1, :, 1][both_d] == 0).all()
    assert int((alpha != 0).sum()) > 10


def test_near_plane_clipping_equals_explicit_clipping():
    n, f = 0.1, 100.0
    proj = np.array([[1.5, 0, 0, 0], [0, 1.5, 0, 0], [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)], [0, 0, -1, 0]], dtype=np.float32)

    def clip(v):
        return torch.tensor((np.concatenate([v, np.ones((len(v), 1), np.float32)], 1) @ proj.T)[None].astype(np.float32), device=DEV)
    y = -0.5
    quad = np.array([[-1, y, -4.0], [1, y, -4.0], [1, y, 3.0], [-1, y, 3.0]], dtype=np.float32)        # z = +3 is behind the eye
    cut = np.array([[-1, y, -4.0], [1, y, -4.0], [1, y, -0.05], [-1, y, -0.05]], dtype=np.float32)     # cut in front of the eye, before the near plane
    tri = _tri([[0, 1, 2], [0, 2, 3]])
    cov = _ids(clip(quad), tri, 48, 64)[2] >= 0
    cov_cut = _ids(clip(cut), tri, 48, 64)[2] >= 0
    assert int(cov_cut.sum()) > 200
    assert int((cov != cov_cut).sum()) <= 2
    assert int(cov[0, :24].sum()) == 0 or int(cov[0, 24:].sum()) == 0
