"""Synthetic cameras / meshes for the render-stage tests and the benchmark (TEST INFRASTRUCTURE).

Cameras follow the reference's dataset conventions: util.perspective(fovy, aspect, n, f)
(render/util.py:242-248, note the flipped y row) and an eye on a sphere looking at the origin
(dataset/dataset_deepfashion.py:59-138 builds mvp = proj @ mv, campos = inverse(mv)[:3,3])."""
import numpy as np


def perspective(fovy=np.deg2rad(60.0), aspect=1.0, n=0.1, f=1000.0):
    y = np.tan(fovy / 2)
    return np.array([[1 / (y * aspect), 0, 0, 0], [0, 1 / -y, 0, 0], [0, 0, -(f + n) / (f - n), -(2 * f * n) / (f - n)], [0, 0, -1, 0]],
                    dtype=np.float32)


def lookat(eye, at=(0, 0, 0), up=(0, 1, 0)):
    eye, at, up = (np.asarray(v, dtype=np.float64) for v in (eye, at, up))
    w = eye - at
    w /= np.linalg.norm(w)
    u = np.cross(up, w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    mv = np.eye(4)
    mv[0, :3], mv[1, :3], mv[2, :3] = u, v, w
    mv[:3, 3] = -mv[:3, :3] @ eye
    return mv.astype(np.float32)


def orbit_views(n, radius=3.0, fovy_deg=60.0, seed=0, first=0):
    """n (mvp [4,4], campos [3]) pairs on a sphere of `radius` (azimuth x elevation spiral)."""
    proj = perspective(np.deg2rad(fovy_deg))
    mvps, cams = [], []
    for k in range(first, first + n):
        az = 2 * np.pi * ((k * 0.61803398875) % 1.0)
        el = np.deg2rad(-25.0 + 50.0 * ((k * 0.41421356237 + 0.25) % 1.0))
        eye = radius * np.array([np.cos(el) * np.sin(az), np.sin(el), np.cos(el) * np.cos(az)])
        mv = lookat(eye)
        mvps.append(proj @ mv)
        cams.append(eye.astype(np.float32))
    return np.stack(mvps).astype(np.float32), np.stack(cams).astype(np.float32)


def random_soup(n_tri, seed, extent=0.8):
    """Random small triangles (shared vertices in strips so that some edges are manifold)."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-extent, extent, size=(n_tri, 1, 3))
    verts = (centers + rng.normal(0, 0.08, size=(n_tri, 3, 3))).reshape(-1, 3).astype(np.float32)
    tri = np.arange(3 * n_tri, dtype=np.int32).reshape(-1, 3)
    return verts, tri


def grid_sheet(n, seed=0, amp=0.15, size=1.2):
    """A wavy n x n quad sheet (2 n^2 triangles, shared vertices): manifold interior + open boundary."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(-0.5, 0.5, n + 1), np.linspace(-0.5, 0.5, n + 1), indexing="ij")
    z = amp * np.sin(5 * u + rng.uniform(0, 3)) * np.cos(4 * v + rng.uniform(0, 3))
    verts = np.stack([size * u, size * v, z], -1).reshape(-1, 3).astype(np.float32)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    tri = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, d], -1).reshape(-1, 3)]).astype(np.int32)
    return verts, tri


def sheet_gbuffer(B, H, W, seed, n_sheet=12):
    """g-buffer of a wavy sheet seen by `B` orbit cameras, rendered with the CPU oracle rasteriser (independent of the HIP
    rasteriser): -> verts, tri, mask [B,H,W], gb_pos, gb_nrm [B,H,W,3], view [B,1,1,3], kd, ks [B,H,W,3] (torch, float32)."""
    import torch
    from oracle import pixel_oracle as po
    from oracle import raster_oracle as ro
    verts, tri = grid_sheet(n_sheet, seed)
    mvp, cam = orbit_views(B, first=seed)
    pos_clip = ro.xfm_points(torch.tensor(verts)[None], torch.tensor(mvp))
    tri_l = torch.tensor(tri).long()
    ids = torch.tensor(ro.rasterize_ids(pos_clip.numpy(), tri, H, W))
    rast, _ = ro.rast_from_ids(pos_clip, tri_l, ids)
    gb_pos = ro.interpolate(torch.tensor(verts)[None], rast, tri_l)
    nrm_v = po.auto_normals(torch.tensor(verts), tri_l)
    gb_nrm = ro.interpolate(nrm_v[None], rast, tri_l)
    gen = torch.Generator().manual_seed(seed)
    kd = torch.rand(B, H, W, 3, generator=gen)
    ks = torch.rand(B, H, W, 3, generator=gen) * torch.tensor([0.3, 1.0, 1.0])
    mask = (ids >= 0).float()
    view = torch.tensor(cam)[:, None, None, :]
    return verts, tri, mask, gb_pos, gb_nrm, view, kd, ks
