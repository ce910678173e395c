"""Rasterise / interpolate / antialias / texture ops with the call surface the reference uses from
`nvdiffrast.torch` (imported as `dr` in render/render.py:13), backed by the HIP kernels in
gshell_amd/csrc/{raster,antialias,texture}.hip through the C ABI (include/gshell_hip.h).

Covered call sites of the reference (everything the G-Shell training path touches):
    dr.RasterizeGLContext() / dr.RasterizeCudaContext()          train_gshelltet_*.py:625
    dr.DepthPeeler(ctx, pos, tri, res).rasterize_next_layer()    render/render.py:377-379 (first layer only)
    dr.rasterize(ctx, pos, tri, res)                             render/render.py:458
    dr.interpolate(attr, rast, tri, rast_db=, diff_attrs=)       render/render.py:25-26
    dr.antialias(color, rast, pos, tri)                          render/render.py:358
    dr.texture(tex, uv, filter_mode='linear', boundary_mode='clamp')   render/render.py:59, :110

There is no CPU / eager fallback: tensors must live in HBM and the HIP library must load.
"""
import torch

from .. import _lib
from .._lib import c_int, c_int64, c_void_p, check, ptr, stream


def _scratch(nbytes, device):
    return torch.empty((max(int(nbytes), 8) + 7) // 8, dtype=torch.int64, device=device)


class RasterizeContext:
    """Stateless stand-in for dr.RasterizeGLContext / dr.RasterizeCudaContext (no GL, no interop)."""

    def __init__(self, output_db=True, mode=None, device=None):
        self.output_db = output_db


RasterizeGLContext = RasterizeContext
RasterizeCudaContext = RasterizeContext


def _check_mesh(pos, tri):
    if pos.dim() != 3 or pos.shape[-1] != 4:
        raise _lib.GShellHipError(f"pos must be [B,V,4] clip-space (instanced mode), got {tuple(pos.shape)}")
    if tri.dim() != 2 or tri.shape[-1] != 3 or tri.dtype != torch.int32:
        raise _lib.GShellHipError(f"tri must be int32 [T,3], got {tri.dtype} {tuple(tri.shape)}")


class _RasterizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W, want_visible):
        _check_mesh(pos, tri)
        L = _lib.lib()
        pos_c = pos.detach().contiguous().float()
        tri_c = tri.contiguous()
        B, V, T = pos_c.shape[0], pos_c.shape[1], tri_c.shape[0]
        dev = pos_c.device
        with torch.cuda.device(dev):
            rast = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            rast_db = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            vis = torch.zeros((T,), dtype=torch.uint8, device=dev) if want_visible else None
            scratch = _scratch(L.gs_rasterize_scratch_bytes(c_int64(B), c_int64(T), c_int64(H), c_int64(W)), dev)
            check(L.gs_rasterize_fwd(ptr(pos_c, torch.float32, "pos"), c_int64(B), c_int64(V), ptr(tri_c, torch.int32, "tri"), c_int64(T),
                                     c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(rast_db), ptr(vis), stream()), "gs_rasterize_fwd")
        ctx.save_for_backward(pos_c, tri_c, rast)
        ctx.dims = (B, V, T, H, W)
        ctx.mark_non_differentiable(rast_db)
        if want_visible:
            ctx.mark_non_differentiable(vis)
            return rast, rast_db, vis
        return rast, rast_db

    @staticmethod
    def backward(ctx, g_rast, *_):
        pos_c, tri_c, rast = ctx.saved_tensors
        B, V, T, H, W = ctx.dims
        g_pos = torch.zeros_like(pos_c)
        if g_rast is not None and T > 0:
            g = g_rast.contiguous().float()
            with torch.cuda.device(pos_c.device):
                check(_lib.lib().gs_rasterize_bwd(ptr(pos_c), c_int64(B), c_int64(V), ptr(tri_c), c_int64(T), c_int64(H), c_int64(W), ptr(rast),
                                                  ptr(g), ptr(g_pos), stream()), "gs_rasterize_bwd")
        return g_pos, None, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True, return_visible=False):
    """-> (rast [B,H,W,4], rast_db [B,H,W,4]) [, tri_visible [T] uint8]."""
    if ranges is not None:
        raise NotImplementedError("range mode is not used by the reference and is not implemented")
    H, W = int(resolution[0]), int(resolution[1])
    return _RasterizeFn.apply(pos, tri, H, W, bool(return_visible))


class DepthPeeler:
    """First (nearest) layer only: the reference asserts num_layers == 1 (render/render.py:378)."""

    def __init__(self, glctx, pos, tri, resolution):
        self._args = (glctx, pos, tri, resolution)
        self._layer = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def rasterize_next_layer(self):
        if self._layer > 0:
            raise NotImplementedError("depth peeling beyond the first layer is not exercised by the reference (render.py:378)")
        self._layer += 1
        return rasterize(*self._args)


class _InterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db):
        L = _lib.lib()
        if attr.dim() == 2:
            attr = attr[None]
        attr_c = attr.detach().contiguous().float()
        rast_c = rast.detach().contiguous().float()
        tri_c = tri.contiguous()
        if tri_c.dtype != torch.int32:
            raise _lib.GShellHipError("tri must be int32")
        Ba, V, A = attr_c.shape
        B, H, W, _ = rast_c.shape
        T = tri_c.shape[0]
        dev = rast_c.device
        with torch.cuda.device(dev):
            out = torch.empty((B, H, W, A), dtype=torch.float32, device=dev)
            db_c = rast_db.detach().contiguous().float() if rast_db is not None else None
            out_da = torch.empty((B, H, W, 2 * A), dtype=torch.float32, device=dev) if rast_db is not None else None
            check(L.gs_interpolate_fwd(ptr(attr_c, torch.float32, "attr"), c_int64(Ba), c_int64(V), c_int64(A), ptr(rast_c), ptr(db_c),
                                       ptr(tri_c), c_int64(T), c_int64(B), c_int64(H), c_int64(W), ptr(out), ptr(out_da), stream()),
                  "gs_interpolate_fwd")
        ctx.save_for_backward(attr_c, rast_c, tri_c)
        ctx.attr_shape = attr.shape
        ctx.need = (attr.requires_grad, rast.requires_grad)
        if out_da is None:
            return out
        ctx.mark_non_differentiable(out_da)
        return out, out_da

    @staticmethod
    def backward(ctx, g_out, *_):
        attr_c, rast_c, tri_c = ctx.saved_tensors
        Ba, V, A = attr_c.shape
        B, H, W, _ = rast_c.shape
        T = tri_c.shape[0]
        need_attr, need_rast = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_attr = torch.zeros_like(attr_c) if need_attr else None
        g_rast = torch.empty_like(rast_c) if need_rast else None
        g = g_out.contiguous().float()
        with torch.cuda.device(rast_c.device):
            check(_lib.lib().gs_interpolate_bwd(ptr(attr_c), c_int64(Ba), c_int64(V), c_int64(A), ptr(rast_c), ptr(tri_c), c_int64(T), c_int64(B),
                                                c_int64(H), c_int64(W), ptr(g), ptr(g_attr), ptr(g_rast), stream()), "gs_interpolate_bwd")
        if g_attr is not None:
            g_attr = g_attr.reshape(ctx.attr_shape)
        return g_attr, g_rast, None, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """-> (out [B,H,W,A], out_da or None).  diff_attrs: None, 'all' (out_da [B,H,W,2A], what the reference passes, render.py:275) or a list of attribute
    indices (out_da [B,H,W,2 len(list)], pairs in the order given -- nvdiffrast's convention; computed as the 'all' pass and sliced)."""
    if diff_attrs is None or rast_db is None:
        return _InterpolateFn.apply(attr, rast, tri, None), None
    out, out_da = _InterpolateFn.apply(attr, rast, tri, rast_db)
    if isinstance(diff_attrs, str):
        if diff_attrs != 'all':
            raise ValueError(f"diff_attrs must be None, 'all' or a list of attribute indices, got {diff_attrs!r}")
        return out, out_da
    A = out.shape[-1]
    idx = [int(i) for i in diff_attrs]
    if any(i < 0 or i >= A for i in idx):
        raise ValueError(f"diff_attrs index out of range for {A} attributes: {idx}")
    cols = torch.tensor([c for i in idx for c in (2 * i, 2 * i + 1)], dtype=torch.long, device=out_da.device)
    return out, out_da.index_select(-1, cols)


class _InterpolateGroupsFn(torch.autograd.Function):
    """interpolate() of several per-vertex attribute tensors in one launch, one contiguous output each (gs_interpolate_groups_*)."""

    @staticmethod
    def forward(ctx, rast, tri, *attrs):
        import ctypes
        L = _lib.lib()
        rast_c = rast.detach().contiguous().float()
        tri_c = tri.detach().contiguous()
        if tri_c.dtype != torch.int32:
            tri_c = tri_c.int()
        a_c = [a.detach().contiguous().float() for a in attrs]
        V = a_c[0].shape[0]
        for a in a_c:
            if a.dim() != 2 or a.shape[0] != V:
                raise _lib.GShellHipError("interpolate_groups: attributes must be [V, C] over the same vertices")
        B, H, W, _ = rast_c.shape
        n = len(a_c)
        outs = [torch.empty((B, H, W, a.shape[1]), dtype=torch.float32, device=rast_c.device) for a in a_c]
        ch = (ctypes.c_int32 * n)(*[int(a.shape[1]) for a in a_c])
        ap = (ctypes.c_void_p * n)(*[ptr(a, torch.float32, "attr").value for a in a_c])
        op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
        ctx.set_materialize_grads(False)          # an output nobody differentiates arrives as None in backward, not as a zero tensor
        with torch.cuda.device(rast_c.device):
            check(L.gs_interpolate_groups_fwd(c_int(n), ch, ap, ptr(rast_c), ptr(tri_c, torch.int32, "tri"), c_int64(tri_c.shape[0]), c_int64(B), c_int64(H),
                                              c_int64(W), op, stream()), "gs_interpolate_groups_fwd")
        ctx.save_for_backward(rast_c, tri_c, *a_c)
        ctx.shapes = [a.shape for a in attrs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        import ctypes
        rast_c, tri_c, *a_c = ctx.saved_tensors
        B, H, W, _ = rast_c.shape
        n = len(a_c)
        need_rast = ctx.needs_input_grad[0]
        g_c = [None if g is None else g.contiguous().float() for g in g_outs]
        g_attr = [torch.zeros_like(a) if (ctx.needs_input_grad[2 + k] and g_c[k] is not None) else None for k, a in enumerate(a_c)]
        g_rast = torch.empty_like(rast_c) if need_rast else None
        ch = (ctypes.c_int32 * n)(*[int(a.shape[1]) for a in a_c])
        ap = (ctypes.c_void_p * n)(*[a.data_ptr() for a in a_c])
        gp = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in g_c])
        gap = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in g_attr])
        with torch.cuda.device(rast_c.device):
            check(_lib.lib().gs_interpolate_groups_bwd(c_int(n), ch, ap, ptr(rast_c), ptr(tri_c), c_int64(tri_c.shape[0]), c_int64(B), c_int64(H), c_int64(W),
                                                       gp, gap, ptr(g_rast), stream()), "gs_interpolate_groups_bwd")
        return (g_rast, None) + tuple(None if g is None else g.reshape(s) for g, s in zip(g_attr, ctx.shapes))


def interpolate_groups(attrs, rast, tri):
    """[interpolate(a[None], rast, tri)[0] for a in attrs] in ONE launch each way: attrs = per-vertex tensors [V, C_k] over the same
    vertices -> list of [B,H,W,C_k].  Values and gradients are bit-identical to interpolating torch.cat(attrs, -1) and slicing."""
    if not all(a.is_cuda for a in attrs):
        raise _lib.GShellHipError("interpolate_groups: attributes must live in HBM; the HIP path has no CPU fallback")
    return list(_InterpolateGroupsFn.apply(rast, tri, *attrs))


class _FaceNormalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_pos, tri, rast):
        v = v_pos.detach().contiguous().float()
        r = rast.detach().contiguous().float()
        B, H, W, _ = r.shape
        out = torch.empty((B, H, W, 3), dtype=torch.float32, device=r.device)
        with torch.cuda.device(r.device):
            check(_lib.lib().gs_face_normal_fwd(ptr(v, torch.float32, "v_pos"), c_int64(v.shape[0]), ptr(tri, torch.int32, "tri"), c_int64(tri.shape[0]),
                                                ptr(r), c_int64(B), c_int64(H), c_int64(W), ptr(out), stream()), "gs_face_normal_fwd")
        ctx.save_for_backward(v, tri, r)
        return out

    @staticmethod
    def backward(ctx, g_out):
        v, tri, r = ctx.saved_tensors
        B, H, W, _ = r.shape
        g = g_out.contiguous().float()
        g_v = torch.zeros_like(v)
        with torch.cuda.device(r.device):
            check(_lib.lib().gs_face_normal_bwd(ptr(v), c_int64(v.shape[0]), ptr(tri), c_int64(tri.shape[0]), ptr(r), c_int64(B), c_int64(H), c_int64(W),
                                                ptr(g), ptr(g_v), stream()), "gs_face_normal_bwd")
        return g_v, None, None


def face_normals(v_pos, tri, rast):
    """Per-pixel geometric normal of the covering triangle, [B,H,W,3] (what the reference obtains by interpolating
    a per-face attribute with index [[i,i,i]], render/render.py:243-248)."""
    return _FaceNormalFn.apply(v_pos, tri, rast)


# ---- antialias -------------------------------------------------------------------------------------

class AATopology:
    """Per-mesh triangle adjacency (opposite vertex across each edge); replaces nvdiffrast's
    per-call topology hash (antialias_construct_topology_hash)."""

    def __init__(self, tri, num_verts):
        tri = tri.contiguous()
        T = tri.shape[0]
        self.opp = torch.empty((T, 3), dtype=torch.int32, device=tri.device)
        if T > 0:
            L = _lib.lib()
            with torch.cuda.device(tri.device):
                scratch = _scratch(L.gs_tri_adjacency_scratch_bytes(c_int64(T)), tri.device)
                check(L.gs_tri_adjacency(ptr(tri, torch.int32, "tri"), c_int64(T), c_int64(int(num_verts)), ptr(scratch), ptr(self.opp), stream()),
                      "gs_tri_adjacency")


def aa_analyze(rast, pos, tri, topo):
    """alpha [B,H,W,2] (no grad; gradients w.r.t. pos flow through `_AntialiasFn`)."""
    L = _lib.lib()
    pos_c, rast_c = pos.detach().contiguous().float(), rast.detach().contiguous().float()
    B, H, W, _ = rast_c.shape
    alpha = torch.empty((B, H, W, 2), dtype=torch.float32, device=rast_c.device)
    with torch.cuda.device(rast_c.device):
        check(L.gs_aa_analyze(ptr(pos_c), c_int64(pos_c.shape[0]), c_int64(pos_c.shape[1]), ptr(tri), c_int64(tri.shape[0]), ptr(topo.opp),
                              ptr(rast_c), c_int64(H), c_int64(W), ptr(alpha), stream()), "gs_aa_analyze")
    return alpha


class _AntialiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, pos, rast, tri, topo, alpha):
        L = _lib.lib()
        color_c = color.detach().contiguous().float()
        B, H, W, C = color_c.shape
        if alpha is None:
            alpha = aa_analyze(rast, pos, tri, topo)
        out = torch.empty_like(color_c)
        with torch.cuda.device(color_c.device):
            check(L.gs_aa_apply_fwd(ptr(color_c, torch.float32, "color"), ptr(alpha), c_int64(B), c_int64(H), c_int64(W), c_int64(C), ptr(out),
                                    stream()), "gs_aa_apply_fwd")
        ctx.save_for_backward(color_c, alpha, pos.detach().contiguous().float(), rast.detach().contiguous().float(), tri, topo.opp)
        return out

    @staticmethod
    def backward(ctx, g_out):
        color_c, alpha, pos_c, rast_c, tri, opp = ctx.saved_tensors
        L = _lib.lib()
        B, H, W, C = color_c.shape
        need_color, need_pos = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g = g_out.contiguous().float()
        g_color = torch.empty_like(color_c) if need_color else None
        g_alpha = torch.empty_like(alpha) if need_pos else None
        g_pos = None
        with torch.cuda.device(color_c.device):
            check(L.gs_aa_apply_bwd(ptr(color_c), ptr(alpha), c_int64(B), c_int64(H), c_int64(W), c_int64(C), ptr(g), ptr(g_color), ptr(g_alpha),
                                    stream()), "gs_aa_apply_bwd")
            if need_pos:
                g_pos = torch.zeros_like(pos_c)
                check(L.gs_aa_analyze_bwd(ptr(pos_c), c_int64(pos_c.shape[0]), c_int64(pos_c.shape[1]), ptr(tri), c_int64(tri.shape[0]), ptr(opp),
                                          ptr(rast_c), c_int64(H), c_int64(W), ptr(alpha), ptr(g_alpha), ptr(g_pos), stream()),
                      "gs_aa_analyze_bwd")
        return g_color, g_pos, None, None, None, None


class _AntialiasInplaceFn(torch.autograd.Function):
    """antialias of a frame the caller OWNS, in place (gs_aa_apply_*_inplace): only the silhouette pixels are read and written instead of two
    passes over the whole [B,H,W,C] frame each way.  Bit-identical to _AntialiasFn.  The incoming gradient is updated in place as well WHEN it
    the caller vouches that it is this node's alone (`grad_exclusive`).  An incoming gradient can be shared: AddBackward hands ONE tensor to
    both operands, a tensor hook may keep what it is shown.  render_mesh can vouch: the frame's direct consumers are `frame_sums` (its
    backward allocates what it returns) and VIEWS of the frame (split / slice backward allocate a fresh full frame), and several consumers
    are summed into the engine's own buffer.  Every other caller gets a copy of the gradient first."""

    @staticmethod
    def forward(ctx, color, pos, rast, tri, topo, grad_exclusive=False):
        L = _lib.lib()
        if not (color.is_cuda and color.dtype == torch.float32 and color.is_contiguous()):
            raise _lib.GShellHipError("in-place antialias needs a contiguous fp32 frame in HBM")
        ctx.grad_exclusive = bool(grad_exclusive)
        B, H, W, C = color.shape
        alpha = aa_analyze(rast, pos, tri, topo)
        saved = torch.empty_like(color)          # only the entries of modified pixels are ever touched
        with torch.cuda.device(color.device):
            check(L.gs_aa_apply_fwd_inplace(ptr(color), ptr(alpha), c_int64(B), c_int64(H), c_int64(W), c_int64(C), ptr(saved), stream()),
                  "gs_aa_apply_fwd_inplace")
        ctx.mark_dirty(color)
        ctx.save_for_backward(color, saved, alpha, pos.detach().contiguous().float(), rast.detach().contiguous().float(), tri, topo.opp)
        return color

    @staticmethod
    def backward(ctx, g_out):
        out, saved, alpha, pos_c, rast_c, tri, opp = ctx.saved_tensors
        L = _lib.lib()
        B, H, W, C = out.shape
        need_pos = ctx.needs_input_grad[1]
        g = g_out if (g_out.is_contiguous() and g_out.dtype == torch.float32) else g_out.contiguous().float()
        if g is g_out and not ctx.grad_exclusive:
            g = g_out.clone()
            INPLACE_GRAD_COPIES[0] += 1
        g_scratch = torch.empty_like(g)
        g_alpha = torch.empty_like(alpha) if need_pos else None
        g_pos = None
        with torch.cuda.device(out.device):
            check(L.gs_aa_apply_bwd_inplace(ptr(out), ptr(saved), ptr(alpha), c_int64(B), c_int64(H), c_int64(W), c_int64(C), ptr(g), ptr(g_scratch),
                                            ptr(g_alpha), stream()), "gs_aa_apply_bwd_inplace")
            if need_pos:
                g_pos = torch.zeros_like(pos_c)
                check(L.gs_aa_analyze_bwd(ptr(pos_c), c_int64(pos_c.shape[0]), c_int64(pos_c.shape[1]), ptr(tri), c_int64(tri.shape[0]), ptr(opp),
                                          ptr(rast_c), c_int64(H), c_int64(W), ptr(alpha), ptr(g_alpha), ptr(g_pos), stream()),
                      "gs_aa_analyze_bwd")
        return g, g_pos, None, None, None, None


INPLACE_GRAD_COPIES = [0]      # how often the in-place antialias backward had to copy its incoming gradient (0 on the training path)


_aa_cache = {"key": None, "refs": None, "topo": None, "alpha": None}


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """dr.antialias drop-in.  The adjacency table and the silhouette analysis are cached across the per-buffer calls of
    ONE render: the entry is keyed on the identity of the rast / pos / tri tensor OBJECTS and holds strong references to
    them, so the caching allocator cannot hand their addresses to the next iteration's tensors while the entry is alive
    (data_ptr / _version alone can collide across iterations: kernel-written tensors keep _version 0)."""
    if pos_gradient_boost != 1.0:
        raise NotImplementedError("pos_gradient_boost is not used by the reference")
    if tri.shape[0] == 0:
        return color
    key = (id(rast), rast._version, id(pos), pos._version, id(tri), tri._version, tuple(rast.shape), tuple(pos.shape), tuple(tri.shape))
    refs = _aa_cache["refs"]
    if _aa_cache["key"] != key or refs is None or refs[0] is not rast or refs[1] is not pos or refs[2] is not tri:
        topo = topology_hash if isinstance(topology_hash, AATopology) else AATopology(tri, pos.shape[1])
        _aa_cache.update(key=key, refs=(rast, pos, tri), topo=topo, alpha=aa_analyze(rast, pos, tri, topo))
    return _AntialiasFn.apply(color, pos, rast, tri, _aa_cache["topo"], _aa_cache["alpha"])


def antialias_cache_clear():
    """Drop the cached analysis (and the references that keep its rast / pos / tri alive)."""
    _aa_cache.update(key=None, refs=None, topo=None, alpha=None)


def antialias_stacked(colors, rast, pos, tri, topo=None, inplace=False, grad_exclusive=False):
    """Antialias several [B,H,W,Ci] buffers with ONE analysis and ONE apply launch (channels stacked).  `inplace` (a single buffer that the
    caller owns and hands over): the frame is updated in place, only its silhouette pixels are touched.  `grad_exclusive`: the caller
    guarantees that the gradient arriving for the result is referenced by nobody else, so the backward may update it in place too."""
    if tri.shape[0] == 0:
        return list(colors)
    if topo is None:
        topo = AATopology(tri, pos.shape[1])
    if len(colors) == 1:          # (torch.cat / torch.split of ONE tensor still copy it, forward and backward: 2 x 190 MB at 4 x 512^2 x 45)
        c = colors[0]
        if inplace and c.is_cuda and c.dtype == torch.float32 and c.is_contiguous() and not c.is_leaf:
            return [_AntialiasInplaceFn.apply(c, pos, rast, tri, topo, grad_exclusive)]
        return [_AntialiasFn.apply(c, pos, rast, tri, topo, None)]
    sizes = [c.shape[-1] for c in colors]
    out = _AntialiasFn.apply(torch.cat(colors, dim=-1), pos, rast, tri, topo, None)
    return list(torch.split(out, sizes, dim=-1))


# ---- texture (bilinear, clamp) ----------------------------------------------------------------------

class _TextureLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv):
        tex_c, uv_c = tex.detach().contiguous().float(), uv.detach().contiguous().float()
        B, H, W, C = tex_c.shape
        if uv_c.shape[0] != B or uv_c.shape[-1] != 2:
            raise _lib.GShellHipError(f"texture: uv {tuple(uv_c.shape)} does not match tex {tuple(tex_c.shape)}")
        n = uv_c[0].numel() // 2
        out = torch.empty(tuple(uv_c.shape[:-1]) + (C,), dtype=torch.float32, device=tex_c.device)
        with torch.cuda.device(tex_c.device):
            check(_lib.lib().gs_texture_linear_fwd(ptr(tex_c, torch.float32, "tex"), c_int64(B), c_int64(H), c_int64(W), c_int64(C),
                                                   ptr(uv_c, torch.float32, "uv"), c_int64(n), ptr(out), stream()), "gs_texture_linear_fwd")
        ctx.save_for_backward(uv_c)
        ctx.dims = (B, H, W, C, n)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (uv_c,) = ctx.saved_tensors
        B, H, W, C, n = ctx.dims
        g = g_out.contiguous().float()
        g_tex = torch.zeros((B, H, W, C), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(_lib.lib().gs_texture_linear_bwd(c_int64(B), c_int64(H), c_int64(W), c_int64(C), ptr(uv_c), c_int64(n), ptr(g), ptr(g_tex),
                                                   stream()), "gs_texture_linear_bwd")
        return g_tex, None


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap', max_mip_level=None):
    """Only the form the G-Shell path uses: bilinear, clamp, no mips (render/render.py:59, :110)."""
    if filter_mode not in ('linear', 'auto') or boundary_mode != 'clamp' or uv_da is not None or mip is not None:
        raise NotImplementedError("only filter_mode='linear', boundary_mode='clamp' without mipmaps is used on the G-Shell path")
    return _TextureLinearFn.apply(tex, uv)
