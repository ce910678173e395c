"""ctypes binding of oracle/_ref/*.so -- the REFERENCE'S OWN native kernels compiled for the host cores
(TEST INFRASTRUCTURE -- checker only; nothing under gshell_amd/ may import this).

oracle/Makefile builds the libraries from the sources where they lie under /root/reference (kernel.cu, bsdf.h,
denoising.cu, loss.cu, normal.cu, mesh.cu -- nothing is copied) through the execution shim in oracle/ref_stub.
They are used (i) to mint the golden vectors tests/golden/ref_*.npz (oracle/make_golden_ref.py), (ii) to pin the python
restatements oracle/shade_oracle.py / pixel_oracle.py, (iii) as bench.py's `cpu_baseline` of kind "reference".
`available()` is False where the libraries were not built (no /root/reference at build time and none shipped).

All functions take / return numpy arrays (float32 / int32, C-contiguous); image tensors are [B,H,W,C] like the
reference's torch tensors.
"""
import ctypes
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_LIBS = {}
f32 = np.float32


def _path(name):
    return os.path.join(_DIR, name + ".so")


def available(name="ref_envshade"):
    return os.path.isfile(_path(name))


def _lib(name):
    if name not in _LIBS:
        if not available(name):
            raise RuntimeError(f"{_path(name)} is not built: run `make -C oracle` on a box that has /root/reference")
        _LIBS[name] = ctypes.CDLL(_path(name))
    return _LIBS[name]


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _out(shape):
    a = np.empty(shape, np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n):
    """host threads of the env-shade pixel loop; 1 = the launch order z, y, x (fixed atomicAdd order into light_grad)."""
    _lib("ref_envshade").ref_set_threads(ctypes.c_int(int(n)))


# ---- render/optixutils: env shading (kernel.cu) -------------------------------------------------------------------------
def _shade_args(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris):
    keep = []

    def F(a):
        a, p = _f(a)
        keep.append(a)
        return p

    def I(a):
        a, p = _i(a)
        keep.append(a)
        return p
    B, H, W = np.shape(mask)
    vB, vH, vW = np.shape(view_pos)[:3]
    Hl, Wl = np.shape(pdf)
    perms = np.asarray(perms)
    assert perms.shape[1] == n * n
    verts = np.asarray(verts, np.float32).reshape(-1, 3)
    tris = np.asarray(tris, np.int32).reshape(-1, 3)
    args = [F(mask), F(ro), F(gb_pos), F(gb_normal), F(view_pos), F(gb_kd), F(gb_ks), F(light), F(pdf), F(rows), F(cols), I(perms),
            ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(vB), ctypes.c_int(vH), ctypes.c_int(vW),
            ctypes.c_int(Hl), ctypes.c_int(Wl), ctypes.c_int(perms.shape[0]), ctypes.c_int(int(bsdf)), ctypes.c_int(int(n)),
            ctypes.c_uint(int(seed) & 0xFFFFFFFF), ctypes.c_float(float(shadow_scale)), F(verts), I(tris), ctypes.c_longlong(tris.shape[0])]
    return args, keep, (B, H, W, Hl, Wl)


def env_shade_fwd(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris):
    """`_plugin.env_shade_fwd` (optixutils/ops.py:88, torch_bindings.cpp:123-189) -> diff, spec [B,H,W,3]."""
    args, keep, (B, H, W, _, _) = _shade_args(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris)
    diff, pd = _out((B, H, W, 3))
    spec, ps = _out((B, H, W, 3))
    rc = _lib("ref_envshade").ref_env_shade_fwd(*args, pd, ps)
    assert rc == 0
    return diff, spec


def env_shade_bwd(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris,
                  diff_grad, spec_grad):
    """`_plugin.env_shade_bwd` (optixutils/ops.py:103, torch_bindings.cpp:191-266)
    -> gb_pos_grad, gb_normal_grad, gb_kd_grad, gb_ks_grad [B,H,W,3], light_grad [Hl,Wl,3]."""
    args, keep, (B, H, W, Hl, Wl) = _shade_args(mask, ro, gb_pos, gb_normal, view_pos, gb_kd, gb_ks, light, pdf, rows, cols, perms, bsdf, n, seed, shadow_scale, verts, tris)
    dg, pdg = _f(diff_grad)
    sg, psg = _f(spec_grad)
    outs = [_out((B, H, W, 3)) for _ in range(4)] + [_out((Hl, Wl, 3))]
    rc = _lib("ref_envshade").ref_env_shade_bwd(*args, pdg, psg, *[p for _, p in outs])
    assert rc == 0
    return tuple(a for a, _ in outs)


def set_anyhit_mode(grid):
    """any-hit evaluation of the following env-shade launches: False = every triangle (the definition), True = the grid-filtered
    candidates of the same predicate (oracle/anyhit_grid.h; the default)."""
    _lib("ref_envshade").ref_set_anyhit_mode(ctypes.c_int(1 if grid else 0))


def env_shade_trace_pixels(pix, n):
    """Per-sample records [len(pix), 2 n^2, 6] of the pixels `pix` (linear indices (z H + y) W + x) of the LAST launch."""
    pix = np.ascontiguousarray(pix, dtype=np.int64)
    out, p = _out((pix.shape[0], 2 * n * n, 6))
    _lib("ref_envshade").ref_env_shade_trace_pixels(pix.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(pix.shape[0]), p)
    return out


def env_shade_trace_pixel(x, y, z, n):
    """Per-sample record [2 n^2, 6] = (dir xyz, pdf_light, pdf_bsdf, visible) of one pixel of the LAST launch."""
    out, p = _out((2 * n * n, 6))
    _lib("ref_envshade").ref_env_shade_trace_pixel(ctypes.c_int(x), ctypes.c_int(y), ctypes.c_int(z), p)
    return out


# ---- render/optixutils: bilateral denoiser (denoising.cu) -----------------------------------------------------------------
def bilateral_fwd(col, nrm, zdz, sigma):
    """`_plugin.bilateral_denoiser_fwd` (torch_bindings.cpp:268-290) -> [B,H,W,4] = (sum w col, max(sum w, 1e-4))."""
    col, pc = _f(col)
    nrm, pn = _f(nrm)
    zdz, pz = _f(zdz)
    B, H, W, _ = col.shape
    out, po = _out((B, H, W, 4))
    _lib("ref_denoise").ref_bilateral_fwd(pc, pn, pz, ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_float(sigma), po)
    return out


def bilateral_bwd(col, nrm, zdz, sigma, out_grad):
    """`_plugin.bilateral_denoiser_bwd` (torch_bindings.cpp:292-314) -> col_grad [B,H,W,3]."""
    col, pc = _f(col)
    nrm, pn = _f(nrm)
    zdz, pz = _f(zdz)
    og, pg = _f(out_grad)
    B, H, W, _ = col.shape
    out, po = _out((B, H, W, 3))
    _lib("ref_denoise").ref_bilateral_bwd(pc, pn, pz, pg, ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(og.shape[-1]),
                                          ctypes.c_float(sigma), po)
    return out


# ---- render/renderutils: image loss (loss.cu), shading normal (normal.cu), xfm_points (mesh.cu) ---------------------------
_LOSS = {"l1": 0, "mse": 1, "relmse": 2, "smape": 3}          # loss.h:22-28
_TONE = {"none": 0, "log_srgb": 1}                            # loss.h:16-20


def image_loss_fwd(img, target, loss, tonemapper):
    """`ru.image_loss(..., use_python=False)` (renderutils/ops.py:466-501): the per-warp partial sums of the CUDA kernel summed
    and divided by B*H*W as in ops.py:497.  Returns (scalar, partial sums)."""
    img, pi = _f(img)
    tgt, pt = _f(target)
    B, H, W, _ = img.shape
    dims = (ctypes.c_int * 3)()
    _lib("ref_renderutils").ref_image_loss_out_dims(ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), dims)
    out, po = _out((dims[0], dims[1], dims[2], 1))
    _lib("ref_renderutils").ref_image_loss_fwd(pi, pt, ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(_LOSS[loss]),
                                               ctypes.c_int(_TONE[tonemapper]), po)
    return np.float32(out.sum(dtype=np.float32) / np.float32(B * H * W)), out


def image_loss_bwd(img, target, loss, tonemapper, dscalar=1.0):
    """Gradient of the scalar above w.r.t. img and target: torch's backward of `sum(out) / (B*H*W)` hands every partial sum
    dscalar / (B*H*W); the kernel does the rest (loss.cu:137-209)."""
    img, pi = _f(img)
    tgt, pt = _f(target)
    B, H, W, _ = img.shape
    dims = (ctypes.c_int * 3)()
    _lib("ref_renderutils").ref_image_loss_out_dims(ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W), dims)
    dout = np.full((dims[0], dims[1], dims[2], 1), np.float32(dscalar) / np.float32(B * H * W), np.float32)
    gi, pgi = _out(img.shape)
    gt, pgt = _out(img.shape)
    _lib("ref_renderutils").ref_image_loss_bwd(pi, pt, dout.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(B), ctypes.c_int(H), ctypes.c_int(W),
                                               ctypes.c_int(_LOSS[loss]), ctypes.c_int(_TONE[tonemapper]), pgi, pgt)
    return gi, gt


def _bcast_dims(a):
    return (ctypes.c_int * 4)(*a.shape)


def prepare_shading_normal_fwd(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, two_sided_shading=True, opengl=True):
    """`_plugin.prepare_shading_normal_fwd` (renderutils torch_bindings.cpp:161-201); inputs [b,h,w,3] broadcastable."""
    ts = [np.ascontiguousarray(t, np.float32) for t in (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)]
    B, H, W = (max(t.shape[k] for t in ts) for k in range(3))
    out, po = _out((B, H, W, 3))
    args = []
    for t in ts:
        args += [t.ctypes.data_as(ctypes.c_void_p), _bcast_dims(t)]
    _lib("ref_renderutils").ref_shading_normal_fwd(*args, ctypes.c_int(int(two_sided_shading)), ctypes.c_int(int(opengl)), po)
    return out


def prepare_shading_normal_bwd(pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm, grad, two_sided_shading=True, opengl=True):
    """`_plugin.prepare_shading_normal_bwd` (torch_bindings.cpp:203-232): six FULL-resolution gradients [B,H,W,3]; the python
    wrapper sums broadcast dimensions afterwards (renderutils/ops.py:171-196) -- `reduce_like` does that here."""
    ts = [np.ascontiguousarray(t, np.float32) for t in (pos, view_pos, perturbed_nrm, smooth_nrm, smooth_tng, geom_nrm)]
    B, H, W = (max(t.shape[k] for t in ts) for k in range(3))
    g, pg = _f(grad)
    outs = [_out((B, H, W, 3)) for _ in range(6)]
    args = []
    for t in ts:
        args += [t.ctypes.data_as(ctypes.c_void_p), _bcast_dims(t)]
    _lib("ref_renderutils").ref_shading_normal_bwd(*args, pg, ctypes.c_int(int(two_sided_shading)), ctypes.c_int(int(opengl)), *[p for _, p in outs])
    return tuple(reduce_like(a, t) for (a, _), t in zip(outs, ts))


def reduce_like(full, like):
    for k in range(like.ndim):
        if like.shape[k] == 1 and full.shape[k] != 1:
            full = full.sum(axis=k, keepdims=True, dtype=np.float32)
    return full


def xfm_points_fwd(points, matrix):
    """`_plugin.xfm_fwd(points, matrix, isPoints=True)` (torch_bindings.cpp:970-1002): [1|B,V,3] x [B,4,4] -> [B,V,4]."""
    pts, pp = _f(points)
    mtx, pm = _f(matrix)
    B, V = mtx.shape[0], pts.shape[1]
    out, po = _out((B, V, 4))
    _lib("ref_renderutils").ref_xfm_points_fwd(pp, ctypes.c_int(pts.shape[0]), ctypes.c_int(V), pm, ctypes.c_int(B), po)
    return out


def xfm_points_bwd(points, matrix, grad):
    """`_plugin.xfm_bwd` (torch_bindings.cpp:1004-1032) -> full-resolution points_grad [B,V,3] (summed over B by the python
    wrapper when points has batch 1)."""
    pts, pp = _f(points)
    mtx, pm = _f(matrix)
    g, pg = _f(grad)
    B, V = mtx.shape[0], pts.shape[1]
    out, po = _out((B, V, 3))
    _lib("ref_renderutils").ref_xfm_points_bwd(pp, ctypes.c_int(pts.shape[0]), ctypes.c_int(V), pm, ctypes.c_int(B), pg, po)
    return out
