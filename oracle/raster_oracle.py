"""CPU oracle for the rasterise / interpolate / antialias stage (TEST INFRASTRUCTURE -- checker only).

PARITY UNPINNED: the reference delegates these ops to nvdiffrast (third-party, installed unpinned from
git per reference README.md:38, NOT in /root/reference; call sites render/render.py:26,358,377-379).
No reference test or golden vector touches them.  This file restates the
published semantics of nvdiffrast's rasterize / interpolate / antialias ops and is, by definition,
the specification the HIP kernels in gshell_amd/csrc/{raster,antialias}.hip are checked against:

  * pixel (x,y) centre <-> NDC ((x+.5)/W*2-1, (y+.5)/H*2-1); row 0 = NDC y -1
  * rast = (u, v, z/w, tri_id+1), perspective-correct barycentrics of vertices 0 and 1
  * coverage: exact fixed-point edge functions (8 sub-pixel bits) with a top-left ownership rule
  * visibility: smallest z/w, ties to the lower triangle id; samples outside -1 <= z/w <= 1 clipped
  * interpolate: out = u a0 + v a1 + (1-u-v) a2
  * antialias: silhouette-edge blending of adjacent pixel pairs (see `antialias`)

Integer decisions (ids) are made with numpy float32 scalar arithmetic in exactly the operation
order of the kernels (no FMA), so they are bit-comparable; differentiable quantities are torch
expressions whose autograd is the gradient oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch

SUBPIX = 256
f32 = np.float32


def xfm_points(points, matrix):
    """reference twin: render/renderutils/ops.py:528-531 (use_python branch of xfm_points)."""
    out = torch.matmul(torch.nn.functional.pad(points, pad=(0, 1), mode='constant', value=1.0), torch.transpose(matrix, 1, 2))
    return out


def xfm_points_kernel_order(points, matrix):
    """`xfm_points` with the sums in the CUDA kernel's order (renderutils/c_src/mesh.cu:43-46: x m0 + y m1 + z m2 + m3, left to right), each
    product and sum rounded to the tensor's dtype (no contraction) -- what the end-to-end chains use, so that the clip-space coordinates the
    oracle rasterises are bit for bit the ones the HIP path (compiled -ffp-contract=off) rasterises: the last bit of a clip coordinate decides
    a handful of the 5 10^5 coverage tests of a 512 x 512 frame.  (nvcc contracts the same expression into fused multiply-adds, so the
    reference's own last bits are a third variant; xfm_points above is the reference's python branch.)  points [1 or B, V, 3], matrix [B, 4, 4]."""
    x, y, z = points[..., 0:1], points[..., 1:2], points[..., 2:3]              # [Bp, V, 1]
    m = matrix[:, None, :, :]                                                   # [B, 1, 4(out), 4(in)]
    return ((x * m[..., 0] + y * m[..., 1]) + z * m[..., 2]) + m[..., 3]


def _project_fix(p, H, W):
    x, y, w = p[..., 0], p[..., 1], p[..., 3]
    with np.errstate(all="ignore"):
        xn, yn = x / w, y / w
        sx = (xn * f32(0.5) + f32(0.5)) * f32(W)
        sy = (yn * f32(0.5) + f32(0.5)) * f32(H)
        fx = np.floor(sx * f32(SUBPIX) + f32(0.5))
        fy = np.floor(sy * f32(SUBPIX) + f32(0.5))
    ok = (w > f32(1e-6)) & (np.abs(fx) < f32(16777216.0)) & (np.abs(fy) < f32(16777216.0))
    fx = np.where(ok, fx, 0).astype(np.int64)
    fy = np.where(ok, fy, 0).astype(np.int64)
    return fx, fy, ok


def _pix_ndc(p, n):
    return (p.astype(f32) + f32(0.5)) * (f32(2.0) / f32(n)) - f32(1.0)


def _bary(p0, p1, p2, fx, fy):
    """2-D homogeneous barycentrics at NDC sample (fx, fy); float32, kernel operation order."""
    p0x, p0y = p0[0] - fx * p0[3], p0[1] - fy * p0[3]
    p1x, p1y = p1[0] - fx * p1[3], p1[1] - fy * p1[3]
    p2x, p2y = p2[0] - fx * p2[3], p2[1] - fy * p2[3]
    a0 = p1x * p2y - p1y * p2x
    a1 = p2x * p0y - p2y * p0x
    a2 = p0x * p1y - p0y * p1x
    s = a0 + a1 + a2
    with np.errstate(all="ignore"):
        iw = f32(1.0) / s
        b0, b1 = a0 * iw, a1 * iw
        z = p0[2] * a0 + p1[2] * a1 + p2[2] * a2
        w = p0[3] * a0 + p1[3] * a1 + p2[3] * a2
        zw = z / w
    return b0, b1, zw


def _depth_key(zw):
    u = zw.astype(f32).view(np.uint32).astype(np.uint64)
    neg = (u & np.uint64(0x80000000)) != 0
    return np.where(neg, (~u) & np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))


def _near_clipped(zb, p0, p1, p2, t, H, W):
    """A triangle with a vertex at w <= eps (behind the eye): nvdiffrast clips it against the view volume; equivalently every
    pixel whose ray hits the triangle's plane polygon (homogeneous barycentrics >= 0) IN FRONT of the eye (w > 0) and inside
    the depth range is covered.  Every pixel is tested here (the kernel bounds the search with the clipped polygon's box)."""
    if not ((p0[2] + p0[3] >= 0) or (p1[2] + p1[3] >= 0) or (p2[2] + p2[3] >= 0)):
        return
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    fx, fy = _pix_ndc(px, W), _pix_ndc(py, H)
    p0x, p0y = p0[0] - fx * p0[3], p0[1] - fy * p0[3]
    p1x, p1y = p1[0] - fx * p1[3], p1[1] - fy * p1[3]
    p2x, p2y = p2[0] - fx * p2[3], p2[1] - fy * p2[3]
    a0 = p1x * p2y - p1y * p2x
    a1 = p2x * p0y - p2y * p0x
    a2 = p0x * p1y - p0y * p1x
    s = a0 + a1 + a2
    with np.errstate(all="ignore"):
        iw = f32(1.0) / s
        b0, b1 = a0 * iw, a1 * iw
        b2 = f32(1.0) - b0 - b1
        zw = (p0[2] * a0 + p1[2] * a1 + p2[2] * a2) / (p0[3] * a0 + p1[3] * a1 + p2[3] * a2)
        w = p0[3] * b0 + p1[3] * b1 + p2[3] * b2
        good = (s != 0) & (b0 >= 0) & (b1 >= 0) & (b2 >= 0) & (w > 0) & (zw >= f32(-1.0)) & (zw <= f32(1.0))
    key = (_depth_key(zw) << np.uint64(32)) | np.uint64(t)
    zb[good] = np.minimum(zb[good], key[good])


def rasterize_ids(pos, tri, H, W):
    """pos [B,V,4] float32 clip space, tri [T,3] int -> ids [B,H,W] int64 (triangle id, -1 = empty)."""
    pos = np.ascontiguousarray(pos, dtype=f32)
    tri = np.asarray(tri, dtype=np.int64)
    B = pos.shape[0]
    zbuf = np.full((B, H, W), np.iinfo(np.uint64).max, dtype=np.uint64)
    for b in range(B):
        ix, iy, ok = _project_fix(pos[b], H, W)
        for t in range(tri.shape[0]):
            i = tri[t]
            front = pos[b, i, 3] > f32(1e-6)
            if front.any() and not front.all():
                _near_clipped(zbuf[b], pos[b, i[0]], pos[b, i[1]], pos[b, i[2]], t, H, W)
                continue
            if not ok[i].all():
                continue
            x, y = ix[i], iy[i]
            area2 = (x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0])
            if area2 == 0:
                continue
            sgn = 1 if area2 > 0 else -1
            x0 = max(0, (x.min() - 128 + 255) >> 8)
            x1 = min(W - 1, (x.max() - 128) >> 8)
            y0 = max(0, (y.min() - 128 + 255) >> 8)
            y1 = min(H - 1, (y.max() - 128) >> 8)
            if x0 > x1 or y0 > y1:
                continue
            py, px = np.meshgrid(np.arange(y0, y1 + 1), np.arange(x0, x1 + 1), indexing="ij")
            cx, cy = px * SUBPIX + 128, py * SUBPIX + 128
            inside = np.ones_like(px, dtype=bool)
            for e in range(3):
                a, c = (e + 1) % 3, (e + 2) % 3
                dx, dy = (x[c] - x[a]) * sgn, (y[c] - y[a]) * sgn
                v = -dy * cx + dx * cy + (dy * x[a] - dx * y[a])
                own = (dy > 0) or (dy == 0 and dx > 0)
                inside &= (v > 0) | ((v == 0) & own)
            if not inside.any():
                continue
            p0, p1, p2 = pos[b, i[0]], pos[b, i[1]], pos[b, i[2]]
            _, _, zw = _bary(p0, p1, p2, _pix_ndc(px, W), _pix_ndc(py, H))
            with np.errstate(invalid="ignore"):
                good = inside & (zw >= f32(-1.0)) & (zw <= f32(1.0))
            key = (_depth_key(zw) << np.uint64(32)) | np.uint64(t)
            sub = zbuf[b, y0:y1 + 1, x0:x1 + 1]
            sub[good] = np.minimum(sub[good], key[good])
    ids = np.where(zbuf == np.iinfo(np.uint64).max, -1, (zbuf & np.uint64(0xFFFFFFFF)).astype(np.int64))
    return ids


_C_LIB = None


def _c_lib():
    """oracle/_c/raster_c.so: the C restatement of `rasterize_ids` (oracle/raster_c.c, built by oracle/Makefile)."""
    global _C_LIB
    if _C_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_c", "raster_c.so")
        if not os.path.isfile(path):          # a checkout without build(): the checker builds itself (gcc is part of the image on both boxes)
            import subprocess
            subprocess.run(["make", "-C", os.path.dirname(os.path.abspath(__file__)), "_c/raster_c.so"], check=False, capture_output=True)
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        L.ro_rasterize_ids.restype = ctypes.c_int
        L.ro_rasterize_ids.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        _C_LIB = L
    return _C_LIB


def rasterize_ids_c(pos, tri, H, W):
    """`rasterize_ids` evaluated by oracle/raster_c.c -- the same statements in C (float32, no contraction), pinned to the python
    loop above bit for bit by tests/test_raster_oracle.py; this is what the config-size parity tests use (10^5+ triangles at 512^2)."""
    pos = np.ascontiguousarray(pos, dtype=f32)
    tri = np.ascontiguousarray(tri, dtype=np.int32).reshape(-1, 3)
    B, V = pos.shape[0], pos.shape[1]
    assert tri.size == 0 or (tri.min() >= 0 and tri.max() < V)
    ids = np.empty((B, H, W), np.int64)
    zbuf = np.empty((B, H, W), np.uint64)
    fix = np.empty((max(V, 1), 3), np.int64)
    rc = _c_lib().ro_rasterize_ids(pos.ctypes.data, B, V, tri.ctypes.data, tri.shape[0], H, W, ids.ctypes.data, zbuf.ctypes.data, fix.ctypes.data)
    assert rc == 0
    return ids


def rast_from_ids(pos, tri, ids):
    """Differentiable (torch) rast / rast_db given the integer winners.
    pos [B,V,4] float tensor, tri [T,3] long, ids [B,H,W] long (-1 empty) -> rast [B,H,W,4], rast_db [B,H,W,4]."""
    B, H, W = ids.shape
    dt = pos.dtype
    valid = ids >= 0
    t = ids.clamp(min=0)
    bidx = torch.arange(B)[:, None, None].expand(B, H, W)
    P = [pos[bidx, tri[t, k]] for k in range(3)]                       # each [B,H,W,4]
    X = (torch.arange(W, dtype=dt) + 0.5) * (2.0 / W) - 1.0
    Y = (torch.arange(H, dtype=dt) + 0.5) * (2.0 / H) - 1.0
    fx = X[None, None, :].expand(B, H, W)
    fy = Y[None, :, None].expand(B, H, W)
    px = [p[..., 0] - fx * p[..., 3] for p in P]
    py = [p[..., 1] - fy * p[..., 3] for p in P]
    a0 = px[1] * py[2] - py[1] * px[2]
    a1 = px[2] * py[0] - py[2] * px[0]
    a2 = px[0] * py[1] - py[0] * px[1]
    s = a0 + a1 + a2
    s = torch.where(valid, s, torch.ones_like(s))
    b0, b1 = a0 / s, a1 / s
    z = P[0][..., 2] * a0 + P[1][..., 2] * a1 + P[2][..., 2] * a2
    w = P[0][..., 3] * a0 + P[1][..., 3] * a1 + P[2][..., 3] * a2
    zw = z / torch.where(valid, w, torch.ones_like(w))
    w_ = [p[..., 3] for p in P]
    da0x, da0y = py[1] * w_[2] - w_[1] * py[2], w_[1] * px[2] - px[1] * w_[2]
    da1x, da1y = py[2] * w_[0] - w_[2] * py[0], w_[2] * px[0] - px[2] * w_[0]
    da2x, da2y = py[0] * w_[1] - w_[0] * py[1], w_[0] * px[1] - px[0] * w_[1]
    dsx, dsy = da0x + da1x + da2x, da0y + da1y + da2y
    db = torch.stack([(da0x - b0 * dsx) / s * (2.0 / W), (da0y - b0 * dsy) / s * (2.0 / H),
                      (da1x - b1 * dsx) / s * (2.0 / W), (da1y - b1 * dsy) / s * (2.0 / H)], -1)
    # forward values are clamped like the kernel; the clamp is transparent to gradients (kernel bwd ignores it)
    b0c = b0 + (b0.clamp(0, 1) - b0).detach()
    b1c = b1 + (b1.clamp(0, 1) - b1).detach()
    rast = torch.stack([b0c, b1c, zw.clamp(-1, 1).detach(), (ids + 1).to(dt)], -1)
    m = valid[..., None].to(dt)
    return rast * m, (db * m).detach()


def interpolate(attr, rast, tri, rast_db=None):
    """attr [1|B,V,A], rast [B,H,W,4], tri [T,3] long -> out [B,H,W,A] (, out_da [B,H,W,2A])."""
    B, H, W, _ = rast.shape
    ids = rast[..., 3].long() - 1
    valid = (ids >= 0) & (ids < tri.shape[0])
    t = ids.clamp(min=0, max=max(tri.shape[0] - 1, 0))
    if tri.shape[0] == 0:
        out = torch.zeros(B, H, W, attr.shape[-1], dtype=attr.dtype)
        return (out, torch.zeros(B, H, W, 2 * attr.shape[-1], dtype=attr.dtype)) if rast_db is not None else out
    bidx = torch.arange(B)[:, None, None].expand(B, H, W) if attr.shape[0] > 1 else torch.zeros(B, H, W, dtype=torch.long)
    a = [attr[bidx, tri[t, k]] for k in range(3)]
    b0, b1 = rast[..., 0:1], rast[..., 1:2]
    b2 = 1.0 - b0 - b1
    m = valid[..., None].to(attr.dtype)
    out = (b0 * a[0] + b1 * a[1] + b2 * a[2]) * m
    if rast_db is None:
        return out
    e0, e1 = a[0] - a[2], a[1] - a[2]
    dX = rast_db[..., 0:1] * e0 + rast_db[..., 2:3] * e1
    dY = rast_db[..., 1:2] * e0 + rast_db[..., 3:4] * e1
    da = torch.stack([dX, dY], -1).reshape(B, H, W, -1) * m
    return out, da


def tri_adjacency(tri):
    """opp[t,e] = vertex of the other triangle sharing edge e (opposite vertex e), -1 unless the edge is
    shared by exactly two triangles."""
    tri = np.asarray(tri, dtype=np.int64)
    T = tri.shape[0]
    opp = np.full((T, 3), -1, dtype=np.int64)
    groups = {}
    for t in range(T):
        for e in range(3):
            a, b = tri[t, (e + 1) % 3], tri[t, (e + 2) % 3]
            groups.setdefault((min(a, b), max(a, b)), []).append((t, e))
    for members in groups.values():
        if len(members) == 2:
            (t0, e0), (t1, e1) = members
            opp[t0, e0] = tri[t1, e1]
            opp[t1, e1] = tri[t0, e0]
    return opp


def tri_adjacency_sorted(tri):
    """`tri_adjacency` without the python dictionary: half-edges sorted by their (min, max) vertex pair; an edge is manifold when its
    key occurs exactly twice.  Pinned to `tri_adjacency` by tests/test_raster_oracle.py."""
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    T = tri.shape[0]
    opp = np.full((T, 3), -1, dtype=np.int64)
    if T == 0:
        return opp
    a, b = tri[:, [1, 2, 0]].reshape(-1), tri[:, [2, 0, 1]].reshape(-1)          # edge e = (v[e+1], v[e+2]), opposite vertex e
    key = np.minimum(a, b) * (tri.max() + 1) + np.maximum(a, b)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    first = np.r_[True, ks[1:] != ks[:-1]]
    start = np.flatnonzero(first)
    count = np.diff(np.r_[start, ks.size])
    two = start[count == 2]
    h0, h1 = order[two], order[two + 1]
    flat_tri = tri.reshape(-1)
    opp.reshape(-1)[h0] = flat_tri[h1]
    opp.reshape(-1)[h1] = flat_tri[h0]
    return opp


def _same_sign(a, b):
    return torch.signbit(a) == torch.signbit(b)


def aa_alpha(rast, pos, tri, opp):
    """Blend factors alpha [B,H,W,2] for the (right, down) neighbour pair of every pixel.
    Differentiable w.r.t. pos through the crossing distance.  See module docstring."""
    B, H, W, _ = rast.shape
    dt = pos.dtype
    ids = rast[..., 3].long() - 1
    zw = rast[..., 2]
    alphas = []
    for d in (0, 1):
        if d == 0:
            sl0, sl1 = (slice(None), slice(None), slice(0, W - 1)), (slice(None), slice(None), slice(1, W))
        else:
            sl0, sl1 = (slice(None), slice(0, H - 1), slice(None)), (slice(None), slice(1, H), slice(None))
        t0, t1 = ids[sl0], ids[sl1]
        z0, z1 = zw[sl0], zw[sl1]
        shape = t0.shape
        py, px = torch.meshgrid(torch.arange(shape[1]), torch.arange(shape[2]), indexing="ij")
        px, py = px[None].expand(shape), py[None].expand(shape)
        differ = t0 != t1
        t = torch.where(t0 >= 0, t0, t1)
        both = (t0 >= 0) & (t1 >= 0)
        t = torch.where(both, torch.where(z0 < z1, t0, t1), t)
        use1 = differ & (t == t1) & ~((t == t0))
        ds = torch.where(use1, -torch.ones((), dtype=dt), torch.ones((), dtype=dt))
        qx = px + (use1.long() * (1 - d))
        qy = py + (use1.long() * d)
        tt = t.clamp(min=0)
        bidx = torch.arange(B)[:, None, None].expand(shape)
        vi = tri[tt]                                                  # [...,3]
        oi = opp[tt]
        hx, hy = 0.5 * W, 0.5 * H
        fx = qx.to(dt) + 0.5 - hx
        fy = qy.to(dt) + 0.5 - hy

        def proj(idx):
            p = pos[bidx, idx]
            iw = 1.0 / p[..., 3]
            return p[..., 0] * iw * hx - fx, p[..., 1] * iw * hy - fy
        P = [proj(vi[..., k]) for k in range(3)]
        O = []
        for k in range(3):
            ox, oy = proj(oi[..., k].clamp(min=0))
            has = oi[..., k] >= 0
            O.append((torch.where(has, ox, P[k][0]), torch.where(has, oy, P[k][1])))
        Pxl, Pyl = [p[0] for p in P], [p[1] for p in P]
        bb = (Pxl[1] - Pxl[0]) * (Pyl[2] - Pyl[0]) - (Pxl[2] - Pxl[0]) * (Pyl[1] - Pyl[0])
        sil = []
        for k in range(3):
            a, b = (k + 1) % 3, (k + 2) % 3
            wing = (Pxl[a] - O[k][0]) * (Pyl[b] - O[k][1]) - (Pxl[b] - O[k][0]) * (Pyl[a] - O[k][1])
            sil.append(_same_sign(wing, bb))
        any_sil = sil[0] | sil[1] | sil[2]
        if d == 1:
            Pxl, Pyl = Pyl, Pxl
        best = torch.full(shape, -1, dtype=torch.long)
        best_dc = torch.zeros(shape, dtype=dt)
        dcs, steep = [], []
        for k in range(3):
            a, b = (k + 1) % 3, (k + 2) % 3
            cross = ~_same_sign(Pyl[a], Pyl[b])
            num = ds * (Pxl[a] * Pyl[b] - Pxl[b] * Pyl[a])
            den = Pyl[b] - Pyl[a]
            dc = num / torch.where(cross, den, torch.ones_like(den))
            take = cross & ((best < 0) | (dc > best_dc))
            best = torch.where(take, torch.full_like(best, k), best)
            best_dc = torch.where(take, dc, best_dc)
            dcs.append(dc)
            steep.append((Pyl[b] - Pyl[a]).abs() >= (Pxl[b] - Pxl[a]).abs())
        dc_sel = torch.zeros(shape, dtype=dt)
        ok = torch.zeros(shape, dtype=torch.bool)
        for k in range(3):
            sel = best == k
            dc_sel = torch.where(sel, dcs[k], dc_sel)
            ok = ok | (sel & sil[k] & steep[k])
        eps = 0.0625
        ok = ok & differ & (t >= 0) & any_sil & (dc_sel > -eps) & (dc_sel < 1.0 + eps)
        dcc = dc_sel.clamp(0.0, 1.0)
        alpha = torch.where(ok, ds * (0.5 - dcc), torch.zeros_like(dcc))
        full = torch.zeros(B, H, W, dtype=dt)
        full[sl0] = alpha
        alphas.append(full)
    return torch.stack(alphas, -1)


def aa_apply(color, alpha):
    """out = color + sum_pairs alpha (c1 - c0) added to (alpha > 0 ? p0 : p1)."""
    out = color.clone()
    B, H, W, C = color.shape
    ar, ad = alpha[..., 0], alpha[..., 1]
    # right pairs
    diff = color[:, :, 1:] - color[:, :, :-1]
    a = ar[:, :, :-1, None]
    out[:, :, :-1] = out[:, :, :-1] + torch.where(a > 0, a * diff, torch.zeros_like(diff))
    out[:, :, 1:] = out[:, :, 1:] + torch.where(a < 0, a * diff, torch.zeros_like(diff))
    diff = color[:, 1:] - color[:, :-1]
    a = ad[:, :-1, :, None]
    out[:, :-1] = out[:, :-1] + torch.where(a > 0, a * diff, torch.zeros_like(diff))
    out[:, 1:] = out[:, 1:] + torch.where(a < 0, a * diff, torch.zeros_like(diff))
    return out


def antialias(color, rast, pos, tri, opp=None):
    if tri.shape[0] == 0:
        return color.clone()
    if opp is None:
        opp = torch.as_tensor(tri_adjacency(tri.numpy()))
    return aa_apply(color, aa_alpha(rast, pos, tri, opp))
