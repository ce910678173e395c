"""The shadow-ray CHECKER at config size (oracle/anyhit_grid.h, anyhit_c.c, and the same header inside oracle/_ref/ref_envshade.so):
   numpy loop (shade_oracle.any_hit_bruteforce)  ==  C loop over every triangle  ==  C loop over the grid-filtered candidates,
bit for bit, on the scenes of the committed goldens, on degenerate / grazing / axis-parallel / far rays, and on surface-start rays
(origin = point + 1e-3 normal, the rays the shader traces: reference render/render.py:131) against a thin shell of ~5 10^4 micro triangles.
The reference's own any-hit is OptiX hardware (kernel.cu:101-117): the predicate itself is PARITY UNPINNED, see oracle/anyhit_grid.h."""
import os

import numpy as np
import pytest

from oracle import refnative as rn
from oracle import scenes
from oracle import shade_oracle as so

G = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def _rays(n, seed, extent=1.0):
    rng = np.random.default_rng(seed)
    org = rng.uniform(-extent, extent, (n, 3)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return org, d


def _special_rays(verts, tris, seed):
    """rays chosen to stress a candidate filter: axis-parallel, in-plane (grazing), through vertices and edge midpoints, starting far
    outside the mesh, zero / NaN / inf / denormal directions, unnormalised directions."""
    rng = np.random.default_rng(seed)
    v = verts[tris[rng.integers(0, len(tris), 400)]]                    # [400, 3, 3]
    c = v.mean(1)
    org, d = [], []
    for ax in range(3):                                                  # axis-parallel through triangle centroids, both ways
        e = np.zeros(3, f32)
        e[ax] = 1
        org += [c - 3 * e, c + 3 * e]
        d += [np.tile(e, (len(c), 1)), np.tile(-e, (len(c), 1))]
    e1 = v[:, 1] - v[:, 0]
    org += [v[:, 0] - e1, c - 2 * e1, v[:, 2] - 0.5 * e1]                # rays IN the plane of a triangle (det ~ 0)
    d += [e1, e1, e1]
    far = rng.normal(size=c.shape).astype(f32)
    far = far / np.linalg.norm(far, axis=1, keepdims=True) * f32(50.0)
    org += [far, far]
    d += [v[:, 0] - far, (v[:, 0] + v[:, 1]) * f32(0.5) - far]           # from far away through a vertex / an edge midpoint (unnormalised)
    o_rand = rng.uniform(-1, 1, c.shape).astype(f32)
    org += [o_rand, o_rand, o_rand, o_rand, c]
    bad = np.zeros_like(c)
    nan = np.full_like(c, np.nan)
    inf = np.tile(np.array([np.inf, 0, 0], f32), (len(c), 1))
    tiny = (c - o_rand) * f32(1e-30)
    d += [bad, nan, inf, tiny, np.tile(np.array([0, 0, 1e-3], f32), (len(c), 1))]
    return np.concatenate(org).astype(f32), np.concatenate(d).astype(f32)


def _shell(n_lat=110, seed=0):
    """a bumpy closed-ish shell of 2 n_lat * 2 n_lat micro triangles (a garment-like thin surface), + its vertex normals"""
    rng = np.random.default_rng(seed)
    th = np.linspace(0.15, np.pi - 0.15, n_lat + 1)
    ph = np.linspace(0, 1.9 * np.pi, 2 * n_lat + 1)                      # open along a seam
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = 0.7 + 0.08 * np.sin(7 * P + rng.uniform(0, 3)) * np.cos(5 * T) + 0.03 * np.sin(23 * T + 11 * P)
    verts = np.stack([r * np.sin(T) * np.cos(P), 1.2 * r * np.cos(T), r * np.sin(T) * np.sin(P)], -1).reshape(-1, 3).astype(f32)
    idx = np.arange(verts.shape[0]).reshape(T.shape)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tris = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    fn = np.cross(verts[tris[:, 1]] - verts[tris[:, 0]], verts[tris[:, 2]] - verts[tris[:, 0]])
    vn = np.zeros_like(verts)
    for k in range(3):
        np.add.at(vn, tris[:, k], fn)
    vn /= np.maximum(np.linalg.norm(vn, axis=1, keepdims=True), 1e-20)
    return verts, tris, vn.astype(f32)


@pytest.mark.parametrize("kind,ntri", [("sheet", 0), ("soup", 3000), ("soup", 5), ("soup", 1), ("soup", 0)])
def test_c_loop_equals_numpy_loop_and_grid_equals_every_triangle(kind, ntri):
    verts, tris = scenes.grid_sheet(30, 2) if kind == "sheet" else scenes.random_soup(ntri, 4)
    org, d = _rays(6000, 1)
    if len(tris):
        o2, d2 = _special_rays(verts, tris, 2)
        org, d = np.concatenate([org, o2]), np.concatenate([d, d2])
    ref = so.any_hit_bruteforce(org, d, verts, tris.astype(np.int64))
    brute = so.any_hit_c(org, d, verts, tris, grid=False)
    st = {}
    grid = so.any_hit_c(org, d, verts, tris, grid=True, stats=st)
    np.testing.assert_array_equal(brute, ref)
    np.testing.assert_array_equal(grid, ref)
    if ntri >= 3000 or kind == "sheet":
        assert 0.02 < ref.mean() < 0.98
        assert st["tests"] < 0.2 * len(org) * len(tris)                 # the filter filters


def test_grid_equals_every_triangle_on_surface_start_rays_against_a_thin_shell():
    """the config-size situation in small: ~4.8 10^4 triangles a few 1e-2 wide, rays leaving the surface 1e-3 above it into the
    hemisphere of the normal (most graze the neighbouring triangles), + rays into the surface (immediate hits)."""
    verts, tris, vn = _shell()
    assert 45000 < len(tris) < 60000
    rng = np.random.default_rng(5)
    pick = rng.integers(0, len(tris), 30000)
    w = rng.dirichlet((1, 1, 1), len(pick)).astype(f32)
    p = (verts[tris[pick]] * w[..., None]).sum(1)
    n = (vn[tris[pick]] * w[..., None]).sum(1)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n *= np.sign((n * p).sum(1, keepdims=True))                           # outward: most rays leave the shell and MISS, as in the bench
    org = (p + n * f32(0.001)).astype(f32)
    d = rng.normal(size=p.shape).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = (d * n).sum(1) < 0
    d[flip & (np.arange(len(d)) % 8 != 0)] *= -1                         # 1 in 16 rays goes INTO the surface
    st_b, st_g = {}, {}
    brute = so.any_hit_c(org, d, verts, tris, grid=False, stats=st_b)
    grid = so.any_hit_c(org, d, verts, tris, grid=True, stats=st_g)
    assert 0.05 < brute.mean() < 0.9, brute.mean()
    np.testing.assert_array_equal(grid, brute)
    assert st_g["tests"] < 0.01 * st_b["tests"]
    o2, d2 = _special_rays(verts, tris, 6)
    np.testing.assert_array_equal(so.any_hit_c(o2, d2, verts, tris, grid=True), so.any_hit_c(o2, d2, verts, tris, grid=False))


def test_grid_equals_every_triangle_on_the_mesh_of_a_real_extraction():
    """the geometry of the config-size GPU tests, made on the CPU: the OPEN surface G-MarchingTets extracts from the suite's "skirt" field with the
    "wavy" mSDF cut (oracle/mtets_oracle on BCC 26: ~10^4 triangles incl. the slivers and the cut boundary a marching scheme produces, which
    the lat-long shell above does not have), rays from 1e-3 above and below the surface, cosine-distributed + grazing + into the surface."""
    import torch
    from gshell_amd import grid as ggrid
    from oracle import fields, mtets_oracle
    gv, tets = ggrid.bcc_grid(26)
    vn_ = gv.numpy()
    ex = mtets_oracle.extract(torch.tensor(vn_), torch.tensor(fields.make_sdf(vn_, "skirt", 3)), torch.tensor(fields.make_msdf(vn_, "wavy", 3)), tets, with_tangents=False)
    verts = (ex["verts_aug"].detach().numpy() * 2.0).astype(f32)
    tris = ex["faces_aug"].numpy().astype(np.int32)
    assert 8000 < len(tris) < 13000
    rng = np.random.default_rng(9)
    pick = rng.integers(0, len(tris), 40000)
    t = verts[tris[pick]]
    w = rng.dirichlet((1, 1, 1), len(pick)).astype(f32)
    p = (t * w[..., None]).sum(1)
    n = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    ok = np.linalg.norm(n, axis=1) > 0                                    # (exact-zero slivers have no normal: start those rays along x)
    n[~ok] = [1.0, 0.0, 0.0]
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n *= np.where(rng.random(len(n)) < 0.5, 1.0, -1.0)[:, None].astype(f32)  # an open surface is seen from both sides
    org = (p + n * f32(0.001)).astype(f32)
    d = rng.normal(size=p.shape).astype(f32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    flip = (d * n).sum(1) < 0
    d[flip & (np.arange(len(d)) % 16 != 0)] *= -1                        # 1 in 32 goes into the surface
    graze = np.arange(len(d)) % 5 == 0                                    # every fifth ray nearly tangent to its own triangle
    d[graze] = d[graze] - n[graze] * (d[graze] * n[graze]).sum(1, keepdims=True) * f32(0.98)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    st_b, st_g = {}, {}
    brute = so.any_hit_c(org, d.astype(f32), verts, tris, grid=False, stats=st_b)
    grid = so.any_hit_c(org, d.astype(f32), verts, tris, grid=True, stats=st_g)
    assert 0.02 < brute.mean() < 0.9, brute.mean()
    np.testing.assert_array_equal(grid, brute)
    assert st_g["tests"] < 0.05 * st_b["tests"]
    o2, d2 = _special_rays(verts, tris, 8)
    np.testing.assert_array_equal(so.any_hit_c(o2, d2, verts, tris, grid=True), so.any_hit_c(o2, d2, verts, tris, grid=False))


@pytest.mark.parametrize("name", ["ref_envshade_pbr_n4.npz", "ref_envshade_pbr_n8_64x64.npz", "ref_envshade_pbr_n4_occluder.npz", "ref_envshade_white_n8.npz"])
def test_checker_reproduces_the_visibility_column_of_the_goldens(name):
    """every sample record of the golden (direction + the any-hit outcome kernel.cu's shadow_test saw in the host build, minted with the
    every-triangle loop) is re-answered by the C loop and by the grid: identical"""
    g = dict(np.load(os.path.join(G, name)))
    S2 = g["samples"].shape[1]
    pix = np.flatnonzero(g["mask"].reshape(-1) > 0)
    org = np.repeat(g["ro"].reshape(-1, 3)[pix], S2, axis=0)
    d = g["samples"][..., :3].reshape(-1, 3)
    vis = g["samples"][..., 5].reshape(-1) > 0
    for grid in (False, True):
        hit = so.any_hit_c(org, d, g["verts"], g["tris"], grid=grid)
        np.testing.assert_array_equal(~hit, vis)
    if "occluder" in name:
        assert (~vis).mean() > 0.2


@pytest.mark.skipif(not rn.available("ref_envshade"), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_kernel_gives_the_same_goldens_with_either_any_hit_evaluation():
    """kernel.cu compiled for the host, its optixTrace answered by every triangle vs by the grid: forward, gradients and sample records
    bit-identical (the golden re-mint of tests/test_oracle_ref_cpu.py runs in the default = grid mode)."""
    from oracle import make_golden_ref as mg
    rn.set_threads(1)
    try:
        for case in (mg.envshade_case("pbr", 4, occluder=True), mg.envshade_case("pbr", 8, frame=(1, 64, 64), probe=(64, 128))):
            outs = []
            for grid in (False, True):
                rn.set_anyhit_mode(grid)
                outs.append(mg.run_envshade(case))
            for k in outs[0]:
                np.testing.assert_array_equal(outs[0][k], outs[1][k], err_msg=k)
    finally:
        rn.set_anyhit_mode(True)
        rn.set_threads(0)
