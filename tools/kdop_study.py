"""Would a k-DOP bound pay in the shadow-ray tree (VERDICT r4 item 5 ii)?  CPU-only study, numpy: the product's tree SHAPE -- a complete 8-ary implicit
heap over Hilbert-sorted triangles, one triangle per leaf (csrc/bvh.hip) -- rebuilt here with float64 boxes over the mesh of a real extraction
(oracle/mtets_oracle on a BCC grid, the test suite's "skirt" field), traversed WITHOUT early exit by rays that start 1e-3 above their own surface
with cosine-distributed directions (what the shader's sampler produces for the 99.5 % of rays that miss), counting the internal nodes a ray visits
(a visit = 8 child tests) and the triangles it reaches, for three child bounds: the AABB (3 slabs), AABB + the 4 cube diagonals (14-DOP, 7 slabs),
AABB + diagonals + the 6 face diagonals (26-DOP, 13 slabs).  The slab test is the traversal kernel's whole inner cost, so a bound with s slabs
has to cut the visits below 3 / s of the AABB's to break even.
usage: python tools/kdop_study.py [cells=52] [rays=20000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gshell_amd import grid
from oracle import fields, mtets_oracle


def hilbert_key(p, bits=10):
    """3-D Hilbert index of integer points p [n,3] < 2^bits (Skilling's transpose algorithm, vectorised)"""
    x = p.astype(np.uint32).copy()
    m = np.uint32(1 << (bits - 1))
    q = m
    while q > 1:
        pm = np.uint32(q - 1)
        for i in range(3):
            hi = (x[:, i] & q) != 0
            x[hi, 0] ^= pm
            t = (x[:, 0] ^ x[:, i]) & pm
            t[hi] = 0
            x[:, 0] ^= t
            x[:, i] ^= t
        q >>= 1
    for i in range(1, 3):
        x[:, i] ^= x[:, i - 1]
    t = np.zeros(len(x), np.uint32)
    q = m
    while q > 1:
        sel = (x[:, 2] & q) != 0
        t[sel] ^= np.uint32(q - 1)
        q >>= 1
    for i in range(3):
        x[:, i] ^= t
    key = np.zeros(len(x), np.uint64)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << np.uint64(1)) | ((x[:, i] >> np.uint32(b)) & 1).astype(np.uint64)
    return key


def main():
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 52
    n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    verts, tets = grid.bcc_grid(cells)
    vn = verts.numpy()
    ex = mtets_oracle.extract(torch.tensor(vn), torch.tensor(fields.make_sdf(vn, "skirt", 3)), torch.tensor(fields.make_msdf(vn, "wavy", 3)), tets, with_tangents=False)
    V, F = ex["verts_aug"].detach().numpy().astype(np.float64), ex["faces_aug"].numpy()
    tri = V[F]                                                     # [T,3,3]
    T = len(tri)
    cen = tri.mean(1)
    lo, hi = cen.min(0), cen.max(0)
    order = np.argsort(hilbert_key(np.minimum(((cen - lo) / (hi - lo).max() * 1023.0), 1023.0).astype(np.uint32)), kind="stable")
    tri = tri[order]
    depth = int(np.ceil(np.log(T) / np.log(8.0)))
    L = 8 ** depth
    s3 = 1.0 / np.sqrt(3.0)
    s2 = 1.0 / np.sqrt(2.0)
    axes_aabb = np.eye(3)
    axes_diag = np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [-1, 1, 1]], float) * s3
    axes_face = np.array([[1, 1, 0], [1, -1, 0], [1, 0, 1], [1, 0, -1], [0, 1, 1], [0, 1, -1]], float) * s2
    sets = {"AABB (3 slabs)": axes_aabb, "14-DOP (7 slabs)": np.vstack([axes_aabb, axes_diag]), "26-DOP (13 slabs)": np.vstack([axes_aabb, axes_diag, axes_face])}
    A = sets["26-DOP (13 slabs)"]
    proj = tri @ A.T                                               # [T,3,13]
    leaf_lo = np.full((L, 13), np.inf)
    leaf_hi = np.full((L, 13), -np.inf)
    leaf_lo[:T], leaf_hi[:T] = proj.min(1), proj.max(1)
    levels = [(leaf_lo, leaf_hi)]                                  # levels[0] = leaves ... levels[depth] = root
    for _ in range(depth):
        a, b = levels[-1]
        levels.append((a.reshape(-1, 8, 13).min(1), b.reshape(-1, 8, 13).max(1)))
    levels = levels[::-1]                                          # levels[l]: 8^l nodes
    # rays: 1e-3 above the surface point of a random triangle, cosine-distributed around its normal (both orientations of an open surface occur)
    rng = np.random.default_rng(0)
    pick = rng.integers(0, T, n_rays)
    t = tri[pick]
    w = rng.dirichlet(np.ones(3), n_rays)
    p = (t * w[:, :, None]).sum(1)
    nrm = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm *= np.where(rng.random(n_rays) < 0.5, 1.0, -1.0)[:, None]
    org = p + 1e-3 * nrm
    u1, u2 = rng.random(n_rays), rng.random(n_rays)
    r, ph = np.sqrt(u1), 2 * np.pi * u2
    tx = np.cross(nrm, np.where(np.abs(nrm[:, :1]) < 0.9, [[1.0, 0, 0]], [[0, 1.0, 0]]))
    tx /= np.linalg.norm(tx, axis=1, keepdims=True)
    ty = np.cross(nrm, tx)
    d = tx * (r * np.cos(ph))[:, None] + ty * (r * np.sin(ph))[:, None] + nrm * np.sqrt(1 - u1)[:, None]
    o_p, d_p = org @ A.T, d @ A.T                                  # projections on all 13 axes
    print(f"mesh: {T} triangles (BCC {cells}), tree depth {depth} ({L} leaf slots), {n_rays} rays")
    base = None
    for name, ax in sets.items():
        k = len(ax)
        ray = np.arange(n_rays)
        node = np.zeros(n_rays, np.int64)
        visits = np.zeros(n_rays)
        reached = np.zeros(n_rays)
        for lvl in range(depth):
            visits += np.bincount(ray, minlength=n_rays)           # every (ray, node) pair alive here is one visit = 8 child tests
            ch = (node[:, None] * 8 + np.arange(8)[None, :]).reshape(-1)
            rr = np.repeat(ray, 8)
            lo_c, hi_c = levels[lvl + 1][0][ch, :k], levels[lvl + 1][1][ch, :k]
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / d_p[rr, :k]
                t0, t1 = (lo_c - o_p[rr, :k]) * inv, (hi_c - o_p[rr, :k]) * inv
            tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
            par = ~np.isfinite(inv)                                 # ray parallel to a slab: inside or outside for good
            inside = (o_p[rr, :k] >= lo_c) & (o_p[rr, :k] <= hi_c)
            tn = np.where(par, np.where(inside, -np.inf, np.inf), tn)
            tf = np.where(par, np.where(inside, np.inf, -np.inf), tf)
            hit = (np.maximum(tn.max(1), 0.0) <= tf.min(1)) & (lo_c[:, 0] <= hi_c[:, 0])      # (padding slots of the complete tree are empty: lo > hi)
            ray, node = rr[hit], ch[hit]
        reached += np.bincount(ray, minlength=n_rays)
        v, tr_ = visits.mean(), reached.mean()
        if base is None:
            base = v
        print(f"{name:18s}: {v:6.2f} node visits / ray ({v / base:5.2f} x), {tr_:5.2f} triangles reached; slab tests per ray {v * 8 * k:7.0f} ({v * 8 * k / (base * 8 * 3):4.2f} x the AABB's); "
              f"break-even at {3.0 / k:4.2f} x the visits")

    # ---- the variant VERDICT r4 names: ONE extra slab per child along the child's OWN mean normal (area-weighted; for a leaf: the triangle's plane),
    # at every level ("all") or only for the children of the leaf parents ("last").  Cost per child: the 3 AABB slabs + two dot products and a
    # slab = ~1.7 x an AABB test, and the node record grows by the axis (3 floats) and the interval (2) per child.
    nrm_t = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])               # area-weighted normals, Hilbert order
    # orient consistently enough for a mean: flip to the dominant component's sign
    ax_lvls, lohi_lvls = [], []
    acc = np.zeros((L, 3))
    acc[:T] = nrm_t
    mem = np.arange(L)
    for lvl in range(depth, -1, -1):
        groups = 8 ** lvl
        per = L // groups
        a = acc.reshape(groups, per, 3)
        # sum of normals with signs aligned to the group's first non-zero normal
        ref = a[np.arange(groups), np.argmax(np.linalg.norm(a, axis=2) > 0, axis=1)]
        sgn = np.sign((a * ref[:, None, :]).sum(2))
        axis = (a * sgn[:, :, None]).sum(1)
        nn = np.linalg.norm(axis, axis=1, keepdims=True)
        axis = np.where(nn > 0, axis / np.maximum(nn, 1e-300), [[1.0, 0, 0]])
        pv = np.full((L, 3), np.nan)
        pv[:T] = (tri * axis[np.arange(L) // per][:T, None, :]).sum(2)
        pv = pv.reshape(groups, per * 3)
        with np.errstate(all="ignore"):
            import warnings
            warnings.simplefilter("ignore")
            lo_n, hi_n = np.nanmin(pv, axis=1), np.nanmax(pv, axis=1)
        ax_lvls.append(axis)
        lohi_lvls.append((lo_n, hi_n))
    ax_lvls, lohi_lvls = ax_lvls[::-1], lohi_lvls[::-1]
    for mode in ("all", "last"):
        ray, node = np.arange(n_rays), np.zeros(n_rays, np.int64)
        visits, extra = np.zeros(n_rays), 0.0
        for lvl in range(depth):
            visits += np.bincount(ray, minlength=n_rays)
            ch = (node[:, None] * 8 + np.arange(8)[None, :]).reshape(-1)
            rr = np.repeat(ray, 8)
            lo_c, hi_c = levels[lvl + 1][0][ch, :3], levels[lvl + 1][1][ch, :3]
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / d[rr]
                t0, t1 = (lo_c - org[rr]) * inv, (hi_c - org[rr]) * inv
            tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
            use = mode == "all" or lvl == depth - 1
            if use:
                a = ax_lvls[lvl + 1][ch]
                oa, da = (org[rr] * a).sum(1), (d[rr] * a).sum(1)
                lo_n, hi_n = lohi_lvls[lvl + 1][0][ch], lohi_lvls[lvl + 1][1][ch]
                with np.errstate(divide="ignore", invalid="ignore"):
                    ia = 1.0 / da
                    s0, s1 = (lo_n - 1e-9 - oa) * ia, (hi_n + 1e-9 - oa) * ia
                sn, sf = np.minimum(s0, s1), np.maximum(s0, s1)
                par = ~np.isfinite(ia)
                ins = (oa >= lo_n - 1e-9) & (oa <= hi_n + 1e-9)
                sn = np.where(par, np.where(ins, -np.inf, np.inf), sn)
                sf = np.where(par, np.where(ins, np.inf, -np.inf), sf)
                tn, tf = np.maximum(tn, sn), np.minimum(tf, sf)
                extra += np.bincount(ray, minlength=n_rays).mean() * 8
            hit = (np.maximum(tn, 0.0) <= tf) & (lo_c[:, 0] <= hi_c[:, 0])
            ray, node = rr[hit], ch[hit]
        v, tr_ = visits.mean(), np.bincount(ray, minlength=n_rays).mean()
        cost = v * 8 * 3 + extra * 2.0          # an own-axis slab ~ 2 fixed-axis slabs (two dot products + the interval)
        print(f"own-normal slab, {mode:4s}: {v:6.2f} node visits / ray ({v / base:5.2f} x), {tr_:5.2f} triangles reached; slab-equivalents per ray {cost:7.0f} "
              f"({cost / (base * 8 * 3):4.2f} x the AABB's)")


if __name__ == "__main__":
    main()
