"""Mesh container + auto_normals with the reference's surface (render/mesh.py:20-100, :212-237)."""
import torch

from .. import _lib
from . import util
from .._lib import c_int64, check, ptr, stream


class Mesh:
    """Same attribute names as the reference container (render/mesh.py:20-60); only what the
    G-Shell render path touches."""

    def __init__(self, v_pos=None, t_pos_idx=None, v_nrm=None, t_nrm_idx=None, v_tex=None, t_tex_idx=None, v_tng=None, t_tng_idx=None,
                 material=None, base=None):
        self.v_pos, self.t_pos_idx = v_pos, t_pos_idx
        self.v_nrm, self.t_nrm_idx = v_nrm, t_nrm_idx
        self.v_tex, self.t_tex_idx = v_tex, t_tex_idx
        self.v_tng, self.t_tng_idx = v_tng, t_tng_idx
        self.material = material
        self.t_pos_idx_i32 = None
        if base is not None:
            self.copy_none(base)

    def copy_none(self, other):
        for k in ("v_pos", "t_pos_idx", "v_nrm", "t_nrm_idx", "v_tex", "t_tex_idx", "v_tng", "t_tng_idx", "material", "t_pos_idx_i32"):
            if getattr(self, k) is None:
                setattr(self, k, getattr(other, k))

    def clone(self):
        out = Mesh(base=self)
        for k in ("v_pos", "t_pos_idx", "v_nrm", "t_nrm_idx", "v_tex", "t_tex_idx", "v_tng", "t_tng_idx"):
            v = getattr(out, k)
            if v is not None:
                setattr(out, k, v.clone().detach())
        return out

    def faces_i32(self):
        if self.t_pos_idx_i32 is None:
            self.t_pos_idx_i32 = self.t_pos_idx.int().contiguous()
        return self.t_pos_idx_i32


class _AutoNormalsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v_pos, tri_i32):
        v = v_pos.detach().contiguous().float()
        V, T = v.shape[0], tri_i32.shape[0]
        acc = torch.empty_like(v)
        nrm = torch.empty_like(v)
        with torch.cuda.device(v.device):
            check(_lib.lib().gs_auto_normals_fwd(ptr(v, torch.float32, "v_pos"), c_int64(V), ptr(tri_i32, torch.int32, "tri"), c_int64(T), ptr(acc),
                                                 ptr(nrm), stream()), "gs_auto_normals_fwd")
        ctx.save_for_backward(v, tri_i32, acc)
        return nrm

    @staticmethod
    def backward(ctx, g_nrm):
        v, tri, acc = ctx.saved_tensors
        g = g_nrm.contiguous().float()
        g_acc = torch.empty_like(v)
        g_pos = torch.zeros_like(v)
        with torch.cuda.device(v.device):
            check(_lib.lib().gs_auto_normals_bwd(ptr(v), c_int64(v.shape[0]), ptr(tri), c_int64(tri.shape[0]), ptr(acc), ptr(g), ptr(g_acc),
                                                 ptr(g_pos), stream()), "gs_auto_normals_bwd")
        return g_pos, None


def auto_normals(imesh):
    """Area-weighted smooth vertex normals (ref render/mesh.py:212-237)."""
    v_nrm = _AutoNormalsFn.apply(imesh.v_pos, imesh.faces_i32())
    return Mesh(v_nrm=v_nrm, t_nrm_idx=imesh.t_pos_idx, base=imesh)


def compute_tangents(imesh, v_tng=None):
    """Tangent frame from precomputed per-vertex tangents: normalise, make orthogonal to the
    smooth normal, normalise again (reference render/mesh.py:243-247).  The uv-derived branch
    (:249-287) needs a texture atlas, which no call site on the hot path provides."""
    if v_tng is None:
        raise NotImplementedError("compute_tangents without v_tng needs uv coordinates (reference mesh.py:249-287); "
                                  "the G-Shell paths always pass v_tng or run with use_uv=False")
    v_tng = util.safe_normalize(v_tng)
    v_tng = util.safe_normalize(v_tng - util.dot(v_tng, imesh.v_nrm) * imesh.v_nrm)
    return Mesh(v_tng=v_tng, t_tng_idx=imesh.t_nrm_idx, base=imesh)
