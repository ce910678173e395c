"""Synthetic tetrahedral grids.

The reference loads Quartet-generated grids from data/tets/{res}_tets.npz
(reference geometry/gshell_tets_geometry.py:59-67, data/tets/generate_tets.py:47);
those files are not redistributable here, so the build synthesises grids with the
same on-disk format (keys `vertices` float32 [N,3], `indices` int64 [F,4]).

  bcc_grid(M)  : body-centred-cubic lattice, M cells per axis, in [-0.5, 0.5]^3.
                 N = (M+1)^3 + M^3, F = 12 M^2 (M-1)   
                 "res64/128/256" <-> M = 26/52/104.
  kuhn_grid(n) : 6 tets per cube (Kuhn / Freudenthal subdivision), n cubes per axis.

Generators are written with torch ops so the res-256 grid (13.4 M tets) can be
built directly in HBM (`device='cuda'`) in milliseconds.
"""
import itertools
import numpy as np
import torch

RES_TO_BCC_CELLS = {64: 26, 128: 52, 256: 104}


def _orient_positive(verts, tets):
    p = verts[tets.reshape(-1)].reshape(-1, 4, 3).double()
    vol = (torch.linalg.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]) * (p[:, 3] - p[:, 0])).sum(-1)
    flip = vol < 0
    t2 = torch.where(flip, tets[:, 3], tets[:, 2])
    t3 = torch.where(flip, tets[:, 2], tets[:, 3])
    return torch.stack([tets[:, 0], tets[:, 1], t2, t3], -1)


def bcc_grid(M: int, device="cpu"):
    """Returns (verts float32 [N,3], tets int64 [F,4]) torch tensors on `device`."""
    M = int(M)
    assert M >= 2
    P = M + 1
    dev = torch.device(device)
    ax = torch.linspace(-0.5, 0.5, P, dtype=torch.float64, device=dev)
    kk, jj, ii = torch.meshgrid(ax, ax, ax, indexing="ij")
    corners = torch.stack([ii, jj, kk], -1).reshape(-1, 3)
    cx = (ax[:-1] + ax[1:]) * 0.5
    kk, jj, ii = torch.meshgrid(cx, cx, cx, indexing="ij")
    centres = torch.stack([ii, jj, kk], -1).reshape(-1, 3)
    verts = torch.cat([corners, centres], 0).float()
    n_corner = P ** 3

    def cid(i, j, k):
        return (k * P + j) * P + i

    def mid(i, j, k):
        return n_corner + (k * M + j) * M + i

    rng_f = torch.arange(1, M, device=dev)   # interior face index along the axis
    rng_c = torch.arange(0, M, device=dev)   # cell index on the other two axes
    a, b, c = torch.meshgrid(rng_c, rng_c, rng_f, indexing="ij")
    a, b, c = a.reshape(-1), b.reshape(-1), c.reshape(-1)
    tets = []
    for axis in range(3):
        if axis == 0:
            c_lo, c_hi = mid(c - 1, b, a), mid(c, b, a)
            q = [cid(c, b, a), cid(c, b + 1, a), cid(c, b + 1, a + 1), cid(c, b, a + 1)]
        elif axis == 1:
            c_lo, c_hi = mid(b, c - 1, a), mid(b, c, a)
            q = [cid(b, c, a), cid(b, c, a + 1), cid(b + 1, c, a + 1), cid(b + 1, c, a)]
        else:
            c_lo, c_hi = mid(b, a, c - 1), mid(b, a, c)
            q = [cid(b, a, c), cid(b + 1, a, c), cid(b + 1, a + 1, c), cid(b, a + 1, c)]
        # 4 tets around the centre-centre axis, one per face edge (q[e], q[e+1])
        per_face = torch.stack([torch.stack([c_lo, c_hi, q[e], q[(e + 1) % 4]], -1) for e in range(4)], 1)
        tets.append(per_face.reshape(-1, 4))
    tets = torch.cat(tets, 0).long()
    return verts, _orient_positive(verts, tets)


def kuhn_grid(n: int, device="cpu"):
    n = int(n)
    P = n + 1
    dev = torch.device(device)
    ax = torch.linspace(-0.5, 0.5, P, dtype=torch.float64, device=dev)
    kk, jj, ii = torch.meshgrid(ax, ax, ax, indexing="ij")
    verts = torch.stack([ii, jj, kk], -1).reshape(-1, 3).float()
    r = torch.arange(n, device=dev)
    k, j, i = torch.meshgrid(r, r, r, indexing="ij")
    base = torch.stack([i.reshape(-1), j.reshape(-1), k.reshape(-1)], -1)
    tets = []
    for perm in itertools.permutations(range(3)):
        p = [base]
        for ax_ in perm:
            nxt = p[-1].clone()
            nxt[:, ax_] += 1
            p.append(nxt)
        tets.append(torch.stack([(q[:, 2] * P + q[:, 1]) * P + q[:, 0] for q in p], -1))
    tets = torch.stack(tets, 1).reshape(-1, 4).long()
    return verts, _orient_positive(verts, tets)


def save_npz(path, verts, tets):
    """Same layout as the reference's data/tets/generate_tets.py:47."""
    np.savez_compressed(path, vertices=verts.cpu().numpy().astype(np.float32),
                        indices=tets.cpu().numpy().astype(np.int64))


def grid_for_res(res: int, device="cpu"):
    """Tet grid standing in for data/tets/{res}_tets.npz."""
    M = RES_TO_BCC_CELLS.get(int(res), max(2, int(round(res * 26 / 64))))
    return bcc_grid(M, device=device)
