"""CPU oracle for Monte-Carlo environment shading, shadow rays and the bilateral denoiser
(TEST INFRASTRUCTURE -- checker only).

Restates the reference's OptiX raygen program render/optixutils/c_src/envsampling/kernel.cu:30-541
(PCG stream :30-45, cosine / GGX-VNDF sampling :57-300, light CDF sampling :140-193, MIS :403-460,
launch loop :463-541), its BSDF c_src/bsdf.h:21-275 and helpers c_src/math_utils.h:80-163, with a
brute-force Moeller-Trumbore any-hit test standing in for `optixTrace` (kernel.cu:101-117), and the
bilateral filter c_src/denoising.cu:14-130.

Pins:  * BSDF terms are checked against the reference's own python twins
         (render/renderutils/bsdf.py: bsdf_lambert, bsdf_pbr_specular) in tests/test_oracle_shade.py,
       * the bilateral filter against the reference's python BilateralDenoiser
         (render/optixutils/tests/filter_test.py:31-74) via tests/golden/shade_bilateral.npz,
       * analytic cases: constant white probe + Lambert -> diffuse ~= 1, fully occluded -> 1 - shadow_scale.
The sampling / shadow-ray code path itself has NO reference test or golden vector (OptiX is not runnable):
PARITY UNPINNED for ray hits and sample placement; this restatement is the specification.

Sampling decisions are made in numpy float32 (discrete, not differentiated -- exactly like the reference, whose
backward pass treats sample directions, pdfs and visibility as constants); the BSDF / light evaluation is torch,
so autograd w.r.t. gb_pos, gb_normal, kd, ks and the probe is the gradient oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
import torch

f32 = np.float32
PI = f32(3.14159265358979323846)
U32 = np.uint32


# ---- PCG ------------------------------------------------------------------------------------------
def _pcg_out(s):
    s = s.astype(np.uint32)
    word = ((s >> ((s >> U32(28)) + U32(4))) ^ s) * U32(277803737)
    return (word >> U32(22)) ^ word


def _lcg_next(s):
    return s * U32(747796405) + U32(2891336453)


def _u01(s):
    return (_pcg_out(s) & U32(0xFFFFFF)).astype(f32) / f32(0x1000000)


# ---- float32 vector helpers (numpy, [P,3]) ----------------------------------------------------------
def _dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def _normalize(v):
    l = np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2])
    with np.errstate(all="ignore"):
        out = v / l[..., None]
    return np.where((l > 0)[..., None], out, f32(0)).astype(f32)


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1).astype(f32)


def _onb(n):
    sign = np.copysign(f32(1.0), n[..., 2]).astype(f32)
    with np.errstate(all="ignore"):
        a = f32(-1.0) / (sign + n[..., 2])
    b = n[..., 0] * n[..., 1] * a
    b1 = np.stack([f32(1.0) + sign * n[..., 0] * n[..., 0] * a, sign * b, -sign * n[..., 0]], -1).astype(f32)
    b2 = np.stack([b, sign + n[..., 1] * n[..., 1] * a, -n[..., 1]], -1).astype(f32)
    return b1, b2


def _tolocal(a, u, v, w):
    return np.stack([_dot(a, u), _dot(a, v), _dot(a, w)], -1).astype(f32)


def _toworld(a, u, v, w):
    return (u * a[..., 0:1] + v * a[..., 1:2] + w * a[..., 2:3]).astype(f32)


def _luminance(c):
    return c[..., 0] * f32(0.2126) + c[..., 1] * f32(0.7152) + c[..., 2] * f32(0.0722)


# ---- sampling pdfs ---------------------------------------------------------------------------------
def _ndf_ggx(alpha, c):
    a2 = alpha * alpha
    d = (c * a2 - c) * c + f32(1.0)
    with np.errstate(all="ignore"):
        return (a2 / (d * d * PI)).astype(f32)


def _g1_ggx(a2, c):
    with np.errstate(all="ignore"):
        c2 = c * c
        t2 = np.maximum(f32(1.0) - c2, f32(0)) / c2
        r = f32(2.0) / (f32(1.0) + np.sqrt(f32(1.0) + a2 * t2))
    return np.where(c <= 0, f32(0), r).astype(f32)


def _ggx_pdf(N, wo, wi, alpha):
    W = _normalize(N)
    U, V = _onb(W)
    wo_l, wi_l = _tolocal(wo, U, V, W), _tolocal(wi, U, V, W)
    ok = (wo_l[..., 2] > 0) & (wi_l[..., 2] > 0)
    m = _normalize(wi_l + wo_l)
    woH = _dot(m, wo_l)
    with np.errstate(all="ignore"):
        pdf = _g1_ggx(alpha * alpha, wo_l[..., 2]) * _ndf_ggx(alpha, m[..., 2]) * np.maximum(f32(0), _dot(wo_l, m)) / wo_l[..., 2]
        pdf = pdf / (f32(4.0) * woH)
    return np.where(ok, pdf, f32(0)).astype(f32)


def _bsdf_pdf(pD, pS, N, wo, wi, alpha):
    NdotL, NdotV = _dot(N, wi), _dot(N, wo)
    pdf = np.zeros_like(pD)
    pdf = pdf + np.where((pD > 0) & (pD > f32(1e-6)), np.maximum(NdotL, f32(0)) / PI * pD, f32(0))
    b = f32(1.0) - pD
    pdf = pdf + np.where((pS > 0) & (b > f32(1e-6)), _ggx_pdf(N, wo, wi, alpha) * b, f32(0))
    return np.where(np.minimum(NdotV, NdotL) < f32(1e-6), f32(1.0), pdf).astype(f32)


def _cosine_sample(N, u, v):
    N = _normalize(N)
    dx, dy = _onb(N)
    phi = f32(2.0) * PI * u
    ct, st = np.sqrt(v), np.sqrt(f32(1.0) - v)
    x, y = np.cos(phi) * st, np.sin(phi) * st
    pdf = np.maximum(f32(0.000001), ct / PI)
    vec = dx * x[..., None] + dy * y[..., None] + N * ct[..., None]
    return _normalize(vec.astype(f32)), pdf.astype(f32)


def _sample_vndf(alpha, wo, ux, uy):
    Vh = _normalize(np.stack([alpha * wo[..., 0], alpha * wo[..., 1], wo[..., 2]], -1).astype(f32))
    z = np.zeros_like(Vh)
    z[..., 2] = 1
    T1 = np.where((Vh[..., 2] < f32(0.9999))[..., None], _normalize(_cross(z, Vh)), np.array([1, 0, 0], f32))
    T2 = _cross(Vh, T1)
    r, phi = np.sqrt(ux), (f32(2.0) * PI) * uy
    t1, t2 = r * np.cos(phi), r * np.sin(phi)
    s = f32(0.5) * (f32(1.0) + Vh[..., 2])
    t2 = (f32(1.0) - s) * np.sqrt(f32(1.0) - t1 * t1) + s * t2
    Nh = T1 * t1[..., None] + T2 * t2[..., None] + Vh * np.sqrt(np.maximum(f32(0), f32(1.0) - t1 * t1 - t2 * t2))[..., None]
    h = _normalize(np.stack([alpha * Nh[..., 0], alpha * Nh[..., 1], np.maximum(f32(0), Nh[..., 2])], -1).astype(f32))
    with np.errstate(all="ignore"):
        pdf = _g1_ggx(alpha * alpha, wo[..., 2]) * _ndf_ggx(alpha, h[..., 2]) * np.maximum(f32(0), _dot(wo, h)) / wo[..., 2]
    return h, pdf.astype(f32)


def _ggx_sample(N, wo, u, v, alpha):
    W = _normalize(N)
    U, V = _onb(W)
    wo_l = _normalize(_tolocal(wo, U, V, W))
    ok = wo_l[..., 2] > 0
    h, pdf = _sample_vndf(alpha, wo_l, u, v)
    woH = _dot(wo_l, h)
    wi_l = h * (woH * f32(2.0))[..., None] - wo_l
    with np.errstate(all="ignore"):
        pdf = pdf / (f32(4.0) * woH)
    wi = _normalize(_toworld(wi_l.astype(f32), U, V, W))
    return np.where(ok[..., None], wi, f32(0)).astype(f32), np.where(ok, pdf, f32(0)).astype(f32)


def _bsdf_sample(pD, pS, N, wo, sx, sy, sz, alpha):
    diffuse = sz < pD
    wi_d, pdf_d = _cosine_sample(N, sx, sy)
    pdf_d = pdf_d * pD
    b = f32(1.0) - pD
    pdf_d = pdf_d + np.where((pS > 0) & (b > f32(1e-6)), _ggx_pdf(N, wo, wi_d, alpha) * b, f32(0))
    tiny = pD < f32(0.0001)
    wi_d = np.where(tiny[..., None], N, wi_d)
    pdf_d = np.where(tiny, f32(1.0), pdf_d)
    wi_s, pdf_s = _ggx_sample(N, wo, sx, sy, alpha)
    pdf_s = pdf_s * b
    pdf_s = pdf_s + np.where((pD > 0) & (pD > f32(1e-6)), np.maximum(_dot(N, wi_s), f32(0)) / PI * pD, f32(0))
    return np.where(diffuse[..., None], wi_d, wi_s).astype(f32), np.where(diffuse, pdf_d, pdf_s).astype(f32)


# ---- light probe -----------------------------------------------------------------------------------
def _dir_to_tc(d):
    u = np.arctan2(d[..., 0], -d[..., 2]).astype(f32) / (f32(2.0) * PI) + f32(0.5)
    v = np.arccos(np.clip(d[..., 1], f32(-1.0), f32(1.0))).astype(f32) / PI
    return u.astype(f32), v.astype(f32)


def _tc_to_dir(u, v):
    phi, th = (u * f32(2.0) - f32(1.0)) * PI, v * PI
    return np.stack([np.sin(th) * np.sin(phi), np.cos(th), -np.sin(th) * np.cos(phi)], -1).astype(f32)


def _sample_cdf(cdf, x):
    """cdf [P,size] (per-sample rows), x [P] -> idx, pdf, resampled x  (kernel.cu:140-169)."""
    size = cdf.shape[-1]
    x = np.minimum(x, f32(0.99999994))
    lo = np.zeros(x.shape, np.int64)
    hi = np.full(x.shape, size - 1, np.int64)
    iters = int(math.ceil(math.log2(float(size - 1)))) + 1
    ar = np.arange(x.shape[0])
    for _ in range(iters):
        mid = (lo + hi) // 2
        c = cdf[ar, mid]
        lo = np.where(x >= c, mid, lo)
        hi = np.where(x < c, mid, hi)
    idx = hi
    d0, d1 = cdf[ar, idx], cdf[ar, np.maximum(idx - 1, 0)]
    pdf = np.where(idx == 0, cdf[ar, 0], d0 - d1).astype(f32)
    sample = np.where(idx == 0, x, x - d1).astype(f32)
    with np.errstate(all="ignore"):
        return idx, pdf, np.minimum(sample / pdf, f32(0.99999994)).astype(f32)


def _light_pdf(pdf_img, d):
    Hl, Wl = pdf_img.shape
    u, v = _dir_to_tc(d)
    x = np.clip((u * f32(Wl)).astype(np.int64), 0, Wl - 1)
    y = np.clip((v * f32(Hl)).astype(np.int64), 0, Hl - 1)
    w = f32(Hl) * f32(Wl) / (f32(2.0) * PI * PI * np.maximum(np.sin(v * PI), f32(0.0001)))
    return (pdf_img[y, x] * w).astype(f32)


def _light_sample(pdf_img, rows, cols, u, v):
    Hl, Wl = pdf_img.shape
    y, _, ry = _sample_cdf(np.broadcast_to(rows[None], (u.shape[0], Hl)), v)
    x, _, rx = _sample_cdf(cols[y], u)
    d = _tc_to_dir((x.astype(f32) + rx) / f32(Wl), (y.astype(f32) + ry) / f32(Hl))
    return d, _light_pdf(pdf_img, d)


# ---- shadow rays -----------------------------------------------------------------------------------
def any_hit_bruteforce(org, dirs, verts, tris, chunk=256):
    """org, dirs [n,3] float32; verts [V,3]; tris [T,3] -> bool [n] (t in (0, 1e16))."""
    n = org.shape[0]
    hit = np.zeros(n, bool)
    if tris.shape[0] == 0 or n == 0:
        return hit
    v0 = verts[tris[:, 0]].astype(f32)
    e1 = (verts[tris[:, 1]] - v0).astype(f32)
    e2 = (verts[tris[:, 2]] - v0).astype(f32)
    for s in range(0, n, chunk):
        o, d = org[s:s + chunk, None, :].astype(f32), dirs[s:s + chunk, None, :].astype(f32)
        with np.errstate(all="ignore"):
            p = _cross(np.broadcast_to(d, (d.shape[0], e2.shape[0], 3)), e2[None])
            det = _dot(e1[None], p)
            inv = f32(1.0) / det
            tv = o - v0[None]
            u = _dot(tv, p) * inv
            q = _cross(tv, np.broadcast_to(e1[None], tv.shape))
            v = _dot(np.broadcast_to(d, q.shape), q) * inv
            t = _dot(np.broadcast_to(e2[None], q.shape), q) * inv
            ok = (np.abs(det) > f32(1e-20)) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 0) & (t < f32(1e16))
        nz = (dirs[s:s + chunk] != 0).any(-1) & np.isfinite(dirs[s:s + chunk]).all(-1)
        hit[s:s + chunk] = ok.any(-1) & nz
    return hit


_AH_LIB = None


def _ah_lib():
    global _AH_LIB
    if _AH_LIB is None:
        import ctypes
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_c", "anyhit_c.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is not built: run `make -C oracle`")
        _AH_LIB = ctypes.CDLL(path)
    return _AH_LIB


def any_hit_c(org, dirs, verts, tris, grid=True, stats=None):
    """`any_hit_bruteforce` evaluated by oracle/anyhit_c.c: the same float32 predicate (no contraction) over every triangle
    (`grid=False`: the definition, pinned to the numpy loop above by tests/test_oracle_anyhit_cpu.py) or over the candidates of a
    conservative uniform grid (`grid=True`: what the config-size tests use -- 10^6+ rays against 10^5+ triangles in seconds; asserted
    identical to the definition by the same test file).  `stats`: a dict that receives the number of predicate evaluations."""
    import ctypes
    org = np.ascontiguousarray(org, dtype=f32).reshape(-1, 3)
    dirs = np.ascontiguousarray(dirs, dtype=f32).reshape(-1, 3)
    verts = np.ascontiguousarray(verts, dtype=f32).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    assert org.shape == dirs.shape
    assert tris.size == 0 or (tris.min() >= 0 and tris.max() < verts.shape[0])
    out = np.zeros(org.shape[0], np.uint8)
    st = np.zeros(1, np.int64)
    rc = _ah_lib().ah_any_hit(ctypes.c_void_p(org.ctypes.data), ctypes.c_void_p(dirs.ctypes.data), ctypes.c_longlong(org.shape[0]),
                              ctypes.c_void_p(verts.ctypes.data), ctypes.c_void_p(tris.ctypes.data), ctypes.c_longlong(tris.shape[0]),
                              ctypes.c_int(1 if grid else 0), ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(st.ctypes.data))
    assert rc == 0, rc
    if stats is not None:
        stats["tests"] = int(st[0])
    return out.astype(bool)


ANY_HIT = any_hit_bruteforce        # what env_shade below traces its shadow rays with; the config-size chains set any_hit_c


# ---- BSDF evaluation (torch, differentiable) ----------------------------------------------------------
SPECULAR_EPSILON = 1e-4


def t_dot(a, b):
    return (a * b).sum(-1, keepdim=True)


def t_safe_normalize(v):
    l = torch.sqrt((v * v).sum(-1, keepdim=True))
    return torch.where(l > 0, v / torch.where(l > 0, l, torch.ones_like(l)), torch.zeros_like(v))


def lambert(nrm, wi):
    return torch.clamp(t_dot(nrm, wi) / math.pi, min=0.0)


def _fresnel(f0, f90, c):
    cc = torch.clamp(c, SPECULAR_EPSILON, 1.0 - SPECULAR_EPSILON)
    scale = (1.0 - cc) ** 5
    return f0 * (1.0 - scale) + f90 * scale


def _ndf(a2, c):
    cc = torch.clamp(c, SPECULAR_EPSILON, 1.0 - SPECULAR_EPSILON)
    d = (cc * a2 - cc) * cc + 1.0
    return a2 / (d * d * math.pi)


def _lambda(a2, c):
    cc = torch.clamp(c, SPECULAR_EPSILON, 1.0 - SPECULAR_EPSILON)
    c2 = cc * cc
    t2 = (1.0 - c2) / c2
    return 0.5 * (torch.sqrt(1.0 + a2 * t2) - 1.0)


def pbr_specular(col, nrm, wo, wi, alpha, min_roughness=0.08):
    al = torch.clamp(alpha, min_roughness * min_roughness, 1.0)
    a2 = al * al
    h = t_safe_normalize(wo + wi)
    woN, wiN, woH, nH = t_dot(wo, nrm), t_dot(wi, nrm), t_dot(wo, h), t_dot(nrm, h)
    D = _ndf(a2, nH)
    G = 1.0 / (1.0 + _lambda(a2, woN) + _lambda(a2, wiN))
    F = _fresnel(col, torch.ones_like(col), woH)
    front = (woN > SPECULAR_EPSILON) & (wiN > SPECULAR_EPSILON)
    w = F * D * G * 0.25 / torch.where(front, woN, torch.ones_like(woN))
    return torch.where(front, w, torch.zeros_like(w))


DECISION_PIN = None      # {'mode': 'record' | 'replay', 'calls': [...]}: see env_shade
CHECKPOINT = False       # activation checkpointing of the per-sample graphs of env_shade / the per-row graphs of bilateral (memory, not arithmetic)


def env_shade(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris):
    """All image tensors torch [B,H,W,C] float32 (view_pos [B,1,1,3]); light [Hl,Wl,3]; pdf [Hl,Wl]; rows [Hl]; cols [Hl,Wl];
    perms [P, n*n] int; verts / tris = occluder mesh (numpy).  -> diff, spec [B,H,W,3] (differentiable)."""
    B, H, W, _ = gb_pos.shape
    S = n * n
    m = (mask.reshape(-1) > 0)
    pix = torch.nonzero(m).reshape(-1)
    P = pix.numel()
    dt = gb_pos.dtype
    diff = torch.zeros(B * H * W, 3, dtype=dt)
    spec = torch.zeros(B * H * W, 3, dtype=dt)
    if P == 0:
        return diff.reshape(B, H, W, 3), spec.reshape(B, H, W, 3)

    def sel(t):
        return t.reshape(-1, 3)[pix]
    pos, nrm, kd, ks, ro_s = sel(gb_pos), sel(gb_normal), sel(gb_kd), sel(gb_ks), sel(ro)
    view = view_pos.expand(B, H, W, 3).reshape(-1, 3)[pix]
    # numpy copies for the (non-differentiated) sampling decisions
    npos, nnrm, nkd, nks, nview = (x.detach().numpy().astype(f32) for x in (pos, nrm, kd, ks, view))
    ro_n = ro_s.detach().numpy().astype(f32)
    if DECISION_PIN is not None:
        # float64 arbiter runs (oracle/make_golden_chain.py): sample placement, lobe choice and shadow rays are decided from the float32
        # run's inputs, so the two runs evaluate the SAME samples and differ by arithmetic only
        if DECISION_PIN["mode"] == "record":
            DECISION_PIN["calls"].append((npos, nnrm, nkd, nks, nview, ro_n))
        else:
            npos, nnrm, nkd, nks, nview, ro_n = DECISION_PIN["calls"].pop(0)
    npdf, nrows, ncols = pdf.numpy().astype(f32), rows.numpy().astype(f32), cols.numpy().astype(f32)
    alpha_n = nks[:, 1] * nks[:, 1]
    wo_n = _normalize(nview - npos)
    metallic = nks[:, 2]
    spec_color = f32(0.04) * (f32(1.0) - metallic)[:, None] + nkd * metallic[:, None]
    diffuse_w = (f32(1.0) - metallic) * _luminance(nkd)
    Wn = _normalize(nnrm)
    Un, Vn = _onb(Wn)
    wo_l = _normalize(_tolocal(wo_n, Un, Vn, Wn))
    cosNO = wo_l[:, 2]
    cc = np.clip(cosNO, f32(SPECULAR_EPSILON), f32(1.0 - SPECULAR_EPSILON))
    scale = np.power(f32(1.0) - cc, f32(5.0))
    fres = spec_color * (f32(1.0) - scale)[:, None] + scale[:, None]
    specular_w = np.where(cosNO > 0, _luminance(fres.astype(f32)), f32(0)).astype(f32)
    with np.errstate(all="ignore"):
        pD = np.where((diffuse_w + specular_w) > 0, diffuse_w / (diffuse_w + specular_w), f32(1.0)).astype(f32)
    pS = (f32(1.0) - pD).astype(f32)

    with np.errstate(over="ignore"):
        pidx = pix.numpy().astype(np.uint32)
        rng = _pcg_out(np.full(P, seed & 0xFFFFFFFF, np.uint32)) ^ _pcg_out(pidx)
        Pn = perms.shape[0]
        light_idx = (_pcg_out(rng) % U32(Pn)).astype(np.int64)
        rng = _lcg_next(rng)
        bsdf_idx = (_pcg_out(rng) % U32(Pn)).astype(np.int64)
        rng = _lcg_next(rng)
    perms = np.asarray(perms)
    strata, frac = f32(1.0) / f32(n), f32(1.0) / f32(n * n)
    alpha_t = (ks[:, 1:2] * ks[:, 1:2])
    wo_t = t_safe_normalize(view - pos)
    Hl, Wl = npdf.shape
    acc_d, acc_s = torch.zeros(P, 3, dtype=dt), torch.zeros(P, 3, dtype=dt)

    def shade(light_, nrm_, kd_, ks_, wo_, alpha_, ly, lx, wi, k):
        """the differentiable part of one batch of samples (one sample slot of every covered pixel)"""
        light_col = light_[ly, lx]
        d_ = lambert(nrm_, wi).expand(-1, 3)
        if bsdf in (1, 2):
            s_ = torch.zeros_like(d_)
        else:
            spec_col = (0.04 * (1.0 - ks_[:, 2:3]) + kd_ * ks_[:, 2:3]) * (1.0 - ks_[:, 0:1])
            s_ = pbr_specular(spec_col, nrm_, wo_, wi, alpha_)
        return d_ * light_col * k, s_ * light_col * k

    def process(dirs, pdf_sum):
        u, v = _dir_to_tc(dirs)
        lx = torch.as_tensor(np.clip((u * f32(Wl)).astype(np.int64), 0, Wl - 1))
        ly = torch.as_tensor(np.clip((v * f32(Hl)).astype(np.int64), 0, Hl - 1))
        mis = torch.as_tensor((f32(1.0) / np.maximum(pdf_sum, f32(0.0001))).astype(f32))[:, None]
        occl = ANY_HIT(ro_n, dirs, verts, tris)
        Vis = torch.as_tensor(((~occl).astype(f32) * f32(shadow_scale) + (f32(1.0) - f32(shadow_scale))).astype(f32))[:, None]
        k = Vis * mis * float(frac)
        args = (light, nrm, kd, ks, wo_t, alpha_t, ly, lx, torch.as_tensor(dirs), k)
        if CHECKPOINT:      # config-size chains (oracle/make_golden_chain.py): same arithmetic, the graph of ONE sample batch alive at a time
            from torch.utils.checkpoint import checkpoint
            return checkpoint(shade, *args, use_reentrant=False)
        return shade(*args)

    ar = np.arange(P)
    for i in range(S):
        with np.errstate(over="ignore"):
            r = []
            for _ in range(5):
                r.append(_u01(rng))
                rng = _lcg_next(rng)
        pl = perms[light_idx, i].astype(np.int64)
        sx = ((pl % n).astype(f32) + r[0]) * strata
        sy = ((pl // n).astype(f32) + r[1]) * strata
        dirs, pdf_l = _light_sample(npdf, nrows, ncols, sx, sy)
        pdf_b = _bsdf_pdf(pD, pS, nnrm, wo_n, dirs, alpha_n)
        d, s = process(dirs, pdf_l + pdf_b)
        acc_d, acc_s = acc_d + d, acc_s + s
        pb = perms[bsdf_idx, i].astype(np.int64)
        sx = ((pb % n).astype(f32) + r[2]) * strata
        sy = ((pb // n).astype(f32) + r[3]) * strata
        dirs, pdf_b = _bsdf_sample(pD, pS, nnrm, wo_n, sx, sy, r[4], alpha_n)
        pdf_l = _light_pdf(npdf, dirs)
        d, s = process(dirs, pdf_l + pdf_b)
        acc_d, acc_s = acc_d + d, acc_s + s
    diff = diff.index_add(0, pix, acc_d)
    spec = spec.index_add(0, pix, acc_s)
    return diff.reshape(B, H, W, 3), spec.reshape(B, H, W, 3)


# ---- bilateral denoiser ------------------------------------------------------------------------------
def bilateral(col, nrm, zdz, sigma):
    """-> [B,H,W,4] = (sum_w col, max(sum_w, 1e-4)); weights are constants w.r.t. autograd (denoising.cu:14-72).
    Autograd of this expression w.r.t. col equals the reference's backward kernel (denoising.cu:74-130)."""
    B, H, W, _ = col.shape
    eps = 0.0001
    rad = 2 * math.ceil(sigma * 2.5) + 1
    var = sigma * sigma
    acc = torch.zeros_like(col)
    acc_w = torch.zeros(B, H, W, 1)
    for fy in range(-rad, rad + 1):
        if CHECKPOINT:
            from torch.utils.checkpoint import checkpoint
            acc, acc_w = checkpoint(_bilateral_row, col, nrm, zdz, acc, acc_w, fy, rad, var, use_reentrant=False)
        else:
            acc, acc_w = _bilateral_row(col, nrm, zdz, acc, acc_w, fy, rad, var)
    return torch.cat([acc, torch.clamp(acc_w, min=eps)], -1)


def _bilateral_row(col, nrm, zdz, acc, acc_w, fy, rad, var):
    """the 2 rad + 1 taps of filter row fy added to the running sums, in the kernel's tap order"""
    B, H, W, _ = col.shape
    eps = 0.0001
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    for fx in range(-rad, rad + 1):
        yy, xx = ys + fy, xs + fx
        valid = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W))[None, ..., None]
        yc, xc = yy.clamp(0, H - 1), xx.clamp(0, W - 1)
        t_col, t_nrm, t_zdz = col[:, yc, xc], nrm[:, yc, xc], zdz[:, yc, xc]
        dist_sqr = float(fx * fx + fy * fy)
        dist = math.sqrt(dist_sqr)
        with torch.no_grad():
            w_xy = math.exp(-dist_sqr / (2.0 * var))
            w_n = torch.clamp(t_dot(t_nrm, nrm), eps, 1.0) ** 128.0
            w_d = torch.exp(-(torch.abs(t_zdz[..., 0:1] - zdz[..., 0:1]) / torch.clamp(zdz[..., 1:2] * dist, min=eps)))
            w = torch.where(valid, w_xy * w_n * w_d, torch.zeros(()))
        acc = acc + t_col * w
        acc_w = acc_w + w
    return acc, acc_w
