"""Known-answer tests that pin the rasterise / interpolate / antialias ORACLE itself (CPU only).
The reference has no tests or golden vectors for these third-party ops, so the
oracle is pinned by analytic cases: exact coverage of axis-aligned shapes, watertightness of shared
edges (no gaps, no double hits), barycentric identities, and the antialias crossing fraction."""
import numpy as np
import torch

from oracle import raster_oracle as ro
from oracle import scenes


def _ndc_tri(pts, z=0.0, w=1.0):
    return np.array([[x * w, y * w, z * w, w] for x, y in pts], dtype=np.float32)[None]


def test_axis_aligned_square_coverage():
    # square covering NDC [-0.5,0.5]^2 on a 8x8 image = pixels 2..5 (centres at -0.375..0.375)
    pos = _ndc_tri([(-0.5, -0.5), (0.5, -0.5), (0.5, 0.5), (-0.5, 0.5)])
    tri = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    ids = ro.rasterize_ids(pos, tri, 8, 8)[0]
    cov = ids >= 0
    expect = np.zeros((8, 8), bool)
    expect[2:6, 2:6] = True
    assert (cov == expect).all()
    # diagonal x == y: centres on the shared edge are owned by exactly one triangle
    assert set(np.unique(ids[cov])) == {0, 1}


def test_shared_edges_watertight_random_fan():
    rng = np.random.default_rng(3)
    # triangle fan around a centre: every pixel inside the polygon is hit by exactly one triangle
    ang = np.sort(rng.uniform(0, 2 * np.pi, 9))
    ring = [(0.8 * np.cos(a), 0.8 * np.sin(a)) for a in ang]
    pos = _ndc_tri([(0.03, -0.02)] + ring)
    tri = np.array([[0, 1 + i, 1 + (i + 1) % 9] for i in range(9)], dtype=np.int32)
    H = W = 64
    ids_all = ro.rasterize_ids(pos, tri, H, W)[0]
    count = np.zeros((H, W), int)
    for t in range(9):
        count += ro.rasterize_ids(pos, tri[t:t + 1], H, W)[0] >= 0
    assert count.max() == 1, "a pixel centre is covered by two fan triangles (double hit)"
    assert ((count == 1) == (ids_all >= 0)).all()
    # both windings rasterise identically
    ids_flip = ro.rasterize_ids(pos, tri[:, ::-1].copy(), H, W)[0]
    assert ((ids_flip >= 0) == (ids_all >= 0)).all()


def test_depth_order_and_clip():
    near = _ndc_tri([(-1, -1), (1, -1), (0, 1)], z=-0.5)[0]
    far = _ndc_tri([(-1, -1), (1, -1), (0, 1)], z=0.5)[0]
    behind = _ndc_tri([(-1, -1), (1, -1), (0, 1)], z=1.5)[0]
    pos = np.concatenate([far, near, behind])[None]
    tri = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], dtype=np.int32)
    ids = ro.rasterize_ids(pos, tri, 16, 16)[0]
    assert set(np.unique(ids)) == {-1, 1}
    ids = ro.rasterize_ids(pos, tri[[0, 2]], 16, 16)[0]
    assert set(np.unique(ids)) == {-1, 0}        # z/w > 1 is clipped
    # equal depth: lower id wins
    pos2 = np.concatenate([near, near])[None]
    ids = ro.rasterize_ids(pos2, np.array([[3, 4, 5], [0, 1, 2]], dtype=np.int32), 16, 16)[0]
    assert set(np.unique(ids)) == {-1, 0}


def test_barycentrics_reproduce_vertices_and_perspective():
    verts, tri = scenes.grid_sheet(4, seed=1)
    mvp, _ = scenes.orbit_views(2)
    pos = ro.xfm_points(torch.tensor(verts)[None], torch.tensor(mvp))
    H = W = 48
    ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri, H, W))
    tri_t = torch.tensor(tri).long()
    rast, db = ro.rast_from_ids(pos, tri_t, ids)
    assert (ids >= 0).float().mean() > 0.05
    # interpolating clip-space w-normalised position must give the pixel's NDC back (perspective-correct)
    out = ro.interpolate(pos, rast, tri_t)
    m = ids >= 0
    X = (torch.arange(W) + 0.5) * 2 / W - 1
    Y = (torch.arange(H) + 0.5) * 2 / H - 1
    ndc_x = (out[..., 0] / out[..., 3])[m]
    ndc_y = (out[..., 1] / out[..., 3])[m]
    assert torch.allclose(ndc_x, X[None, None, :].expand(2, H, W)[m], atol=2e-4)
    assert torch.allclose(ndc_y, Y[None, :, None].expand(2, H, W)[m], atol=2e-4)
    # z/w channel equals interpolated z over interpolated w
    assert torch.allclose((out[..., 2] / out[..., 3])[m], rast[..., 2][m], atol=1e-5)
    # pixel derivatives: finite differences of u along X inside one triangle
    u = rast[..., 0]
    same = (ids[:, :, 1:] == ids[:, :, :-1]) & m[:, :, 1:]
    fd = (u[:, :, 1:] - u[:, :, :-1])[same]
    an = 0.5 * (db[..., 0][:, :, 1:] + db[..., 0][:, :, :-1])[same]
    assert torch.allclose(fd, an, atol=2e-3)


def test_antialias_crossing_fraction_vertical_edge():
    # a half-plane x <= e (one big quad), background colour 0, object colour 1, 8x8 image.
    H = W = 8
    for e_pix in (3.2, 3.5, 3.9, 4.3):          # edge position in pixel units
        e = e_pix / W * 2 - 1
        pos = torch.tensor(_ndc_tri([(-3, -3), (e, -3), (e, 3), (-3, 3)]))
        tri = torch.tensor([[0, 1, 2], [0, 2, 3]])
        ids = torch.tensor(ro.rasterize_ids(pos.numpy(), tri.numpy().astype(np.int32), H, W))
        rast, _ = ro.rast_from_ids(pos, tri, ids)
        color = (ids >= 0).float()[..., None]
        out = ro.antialias(color, rast, pos, tri)
        # exact area coverage of pixel column c by the half-plane is clamp(e_pix - c, 0, 1); the op blends the
        # two pixels adjacent to the edge linearly with the crossing fraction between their centres
        row = out[0, 4, :, 0]
        last_in = int(np.floor(e_pix - 0.5))               # last column whose centre is inside
        dc = e_pix - (last_in + 0.5)                        # crossing distance from that centre
        expect = color[0, 4, :, 0].clone()
        if dc < 0.5:
            expect[last_in] = 1 - (0.5 - dc)
        else:
            expect[last_in + 1] = dc - 0.5
        assert torch.allclose(row, expect, atol=1e-5), (e_pix, row, expect)
        assert torch.allclose(out.sum(), torch.tensor(H * e_pix), atol=1e-3)  # coverage is conserved along the edge


def test_antialias_ignores_interior_edges_and_has_gradients():
    verts, tri = scenes.grid_sheet(6, seed=2)
    mvp, _ = scenes.orbit_views(1)
    pos = ro.xfm_points(torch.tensor(verts)[None], torch.tensor(mvp)).requires_grad_(True)
    tri_t = torch.tensor(tri).long()
    H = W = 40
    ids = torch.tensor(ro.rasterize_ids(pos.detach().numpy(), tri, H, W))
    rast, _ = ro.rast_from_ids(pos.detach(), tri_t, ids)
    color = torch.rand(1, H, W, 3, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    opp = torch.tensor(ro.tri_adjacency(tri))
    alpha = ro.aa_alpha(rast, pos, tri_t, opp)
    # interior pixel pairs (both covered, front-facing smooth sheet) must not blend
    both = (ids[:, :, :-1] >= 0) & (ids[:, :, 1:] >= 0)
    assert (alpha[:, :, :-1, 0][both] == 0).all()
    assert (alpha != 0).sum() > 10
    out = ro.aa_apply(color, alpha)
    out.square().sum().backward()
    assert pos.grad.abs().sum() > 0 and torch.isfinite(pos.grad).all() and torch.isfinite(color.grad).all()


def test_near_plane_clipping_known_answer():
    """A ground quad that runs from in front of the camera to BEHIND it: the triangles with a vertex behind the eye must cover
    exactly what the explicitly clipped geometry (cut at a plane in front of the eye, all w > 0) covers."""
    import numpy as np
    from oracle import raster_oracle as ro
    n, f = 0.1, 100.0
    proj = np.array([[1.5, 0, 0, 0], [0, 1.5, 0, 0], [0, 0, -(f + n) / (f - n), -2 * f * n / (f - n)], [0, 0, -1, 0]], dtype=np.float32)

    def clip(v):
        return (np.concatenate([v, np.ones((len(v), 1), np.float32)], 1) @ proj.T)[None].astype(np.float32)
    y = -0.5
    quad = np.array([[-1, y, -4.0], [1, y, -4.0], [1, y, 3.0], [-1, y, 3.0]], dtype=np.float32)        # z = +3 is behind the eye
    tri = np.array([[0, 1, 2], [0, 2, 3]])
    ids = ro.rasterize_ids(clip(quad), tri, 48, 64)
    zc = -0.05                                                                                        # cut in front of the eye, before the near plane
    cut = np.array([[-1, y, -4.0], [1, y, -4.0], [1, y, zc], [-1, y, zc]], dtype=np.float32)
    ids_cut = ro.rasterize_ids(clip(cut), tri, 48, 64)
    cov, cov_cut = ids >= 0, ids_cut >= 0
    assert cov_cut.sum() > 200
    # the near plane (z_eye = -0.1) hides everything between the cut and the eye in both renderings -> identical coverage
    assert (cov != cov_cut).sum() <= 2, int((cov != cov_cut).sum())
    assert cov[:24].sum() == 0 or cov[24:].sum() == 0            # the floor occupies one half of the image only


# ---- the C restatement (oracle/raster_c.c) and the sorted adjacency are pinned to the python loops above ------------------------------------

def _pin_scenes():
    """(name, clip-space positions [B,V,4], triangles) -- small enough for the python loop, chosen to reach every branch of it:
    micro triangles, screen-filling triangles, depth ties, degenerate / zero-area triangles, vertices behind the eye (near-clip
    path), vertices far outside the 2^24 fixed-point range, NaN positions, an empty mesh."""
    out = []
    for kind, seed, first, radius in (("soup", 1, 0, 3.0), ("sheet", 2, 3, 3.0), ("sheet", 5, 1, 0.45), ("soup", 7, 2, 0.7)):
        verts, tri = scenes.random_soup(300, seed) if kind == "soup" else scenes.grid_sheet(12, seed)
        mvp, _ = scenes.orbit_views(2, radius=radius, first=first)
        out.append((f"{kind}-r{radius}", ro.xfm_points(torch.tensor(verts)[None], torch.tensor(mvp)).numpy(), tri))
    # two screen-filling quads at the SAME depth (ties -> lower id), one degenerate triangle, one zero-area triangle
    v = np.array([[-4, -4, -0.6], [4, -4, -0.6], [4, 4, -0.6], [-4, 4, -0.6], [0, 0, 0], [0.1, 0.1, 0]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 2], [4, 4, 5], [4, 5, 4], [1, 0, 2]], np.int32)
    mvp, _ = scenes.orbit_views(2, first=4)
    out.append(("ties", ro.xfm_points(torch.tensor(v)[None], torch.tensor(mvp)).numpy(), tri))
    pos = out[1][1].copy()
    pos[0, 5, 0] = 3e9
    pos[0, 17, 3] = 1e-9
    pos[1, 40, 1] = np.nan
    pos[1, 41, 3] = -0.3
    out.append(("overflow-nan", pos, out[1][2]))
    out.append(("empty", np.zeros((1, 3, 4), np.float32), np.zeros((0, 3), np.int32)))
    return out


def test_c_restatement_equals_the_python_rasteriser_bit_for_bit():
    fired = dict(near_clip=0, covered=0)
    for name, pos, tri in _pin_scenes():
        for (H, W) in ((40, 56), (64, 64)):
            a = ro.rasterize_ids(pos, tri, H, W)
            b = ro.rasterize_ids_c(pos, tri, H, W)
            np.testing.assert_array_equal(a, b, err_msg=name)
            fired["covered"] += int((a >= 0).sum())
        if tri.shape[0]:
            front = pos[:, tri.reshape(-1), 3].reshape(pos.shape[0], -1, 3) > 1e-6
            fired["near_clip"] += int((front.any(-1) & ~front.all(-1)).sum())
    assert fired["near_clip"] > 20 and fired["covered"] > 10000, fired


def test_sorted_adjacency_equals_the_dictionary_version():
    for kind, seed in (("soup", 1), ("sheet", 2)):
        verts, tri = scenes.random_soup(200, seed) if kind == "soup" else scenes.grid_sheet(9, seed)
        np.testing.assert_array_equal(ro.tri_adjacency_sorted(tri), ro.tri_adjacency(tri))
    # non-manifold edge (three triangles on one edge) and a duplicated triangle: neither is "exactly two"
    tri = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4], [5, 6, 7], [5, 6, 7], [7, 8, 5]], np.int64)
    np.testing.assert_array_equal(ro.tri_adjacency_sorted(tri), ro.tri_adjacency(tri))
    assert ro.tri_adjacency_sorted(np.zeros((0, 3), np.int64)).shape == (0, 3)
