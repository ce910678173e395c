"""`render.render_mesh` / `render_layer` / `shade` with the reference's signatures and buffer dictionary
(reference render/render.py:31-191, :199-317, :325-444), built on the HIP ops of this package.

What differs from the reference's structure (results are the same):
  * one stacked `interpolate` launch for position + normal + mSDF instead of three; the per-face geometric
    normal is a gather by triangle id (interpolating a per-face constant is the identity);
  * the hash-grid texture is evaluated only where a triangle covers the pixel (the reference evaluates the
    background too and then multiplies it by alpha = 0);
  * all output buffers are composited and antialiased by ONE analysis + ONE apply launch
    (`rast.antialias_stacked`) instead of ~12 independent dr.antialias calls;
  * `visible_triangles` comes from a flag array written by the rasteriser's resolve pass.
"""
import torch

from . import light
from . import optixutils as ou
from . import rast as dr
from . import renderutils as ru
from . import util

rnd_seed = 0

# Test hook: the reference draws three noise tensors from the global torch RNG (render/render.py:55, :68, :265).
# Parity tests inject the same tensors on both sides through this table {name: tensor}; None = draw fresh noise.
noise_override = None


class NoiseStream:
    """Counter-based source of the render noise (jitter / texture / tangent tensors) and of the eikonal surface samples for
    view-sharded training (SURVEY.md 8e ii, iv): every draw is a function of (seed, iteration, name, GLOBAL view index),
    so rank r of an N-GPU job sees exactly the rows r, r + N, ... of the tensor the single-GPU job draws.  The reference
    is single-GPU and draws from the global torch RNG (render/render.py:55, :68, :265); that stays the default when
    FLAGS carries no `noise_stream`."""
    _IDS = {'jitter': 1, 'texture': 2, 'tangent': 3, 'eikonal': 4}

    def __init__(self, seed=0, rank=0, world=1):
        self.seed, self.rank, self.world, self.it = int(seed), int(rank), int(world), 0
        self.global_batch = None          # views of the global batch (None: local views x world)
        self._gens = {}
        self._calls = {}                  # draws of each name within the current iteration

    def set_iteration(self, it, global_batch=None):
        self.it = int(it)
        self.global_batch = None if global_batch is None else int(global_batch)
        self._calls = {}

    def generator(self, name, device):
        """Seeded by (seed, iteration, name, how often this name has been drawn in this iteration): the second render of an
        iteration (use_img_2nd_layer, every view of validate()) gets FRESH noise, like the reference's draws from the global RNG
        (render.py:55, :68, :265), and all ranks -- which issue the same call sequence -- still agree."""
        g = self._gens.get(str(device))
        if g is None:
            g = self._gens[str(device)] = torch.Generator(device=device)
        k = self._calls.get(name, 0)
        self._calls[name] = k + 1
        g.manual_seed(((self.seed * 1000003 + self.it) * 4099 + k) * 16 + self._IDS[name])
        return g

    def normal(self, name, std, size, device):
        """[B_local, ...] = this rank's rows (views rank, rank + world, ...) of the [B_global, ...] global-batch draw."""
        B = self.global_batch if self.global_batch is not None else size[0] * self.world
        if self.world > 1 and -(-B // self.world) != size[0] and B // self.world != size[0]:
            # a render outside the training step's batch (validate(): one view at a time after step(global_batch=G)): every rank
            # renders `size[0]` views of its own, i.e. a global batch of size[0] * world
            B = size[0] * self.world
        full = torch.randn((B,) + tuple(size[1:]), device=device, generator=self.generator(name, device))
        if self.world > 1:
            full = full[self.rank::self.world]
        if full.shape[0] != size[0]:
            raise ValueError(f"NoiseStream: {size[0]} local views do not match views {self.rank}::{self.world} of a global batch of {B}")
        return (full * std).contiguous() if std != 1.0 else full.contiguous()

    def state_dict(self):
        return {'seed': self.seed, 'it': self.it}

    def load_state_dict(self, sd):
        self.seed, self.it = int(sd['seed']), int(sd['it'])


def _noise(name, draw, FLAGS=None, std=1.0, size=None, device=None):
    if noise_override is not None and name in noise_override:
        return noise_override[name]
    ns = getattr(FLAGS, 'noise_stream', None) if FLAGS is not None else None
    if ns is not None:
        return ns.normal(name, std, size, device)
    return draw()


_consts = {}


def _const_011(dev):
    """[1,1,1,3] mask (0,1,1), created once per device (a fresh torch.tensor(...) per call is a blocking H2D copy)."""
    key = ("011", str(dev))
    if key not in _consts:
        _consts[key] = torch.tensor([0, 1, 1], dtype=torch.float32, device=dev)[None, None, None, :]
    return _consts[key]


def _view_map(FLAGS):
    """Global view index of local view b = b * world + rank under the round-robin view shard (train.ViewShard)."""
    sh = getattr(FLAGS, 'view_shard', None)
    return {'view_offset': sh.rank, 'view_stride': sh.world} if sh is not None and sh.world > 1 else {}


def interpolate(attr, rast, attr_idx, rast_db=None):
    return dr.interpolate(attr.contiguous(), rast, attr_idx, rast_db=rast_db, diff_attrs=None if rast_db is None else 'all')


def _sample_texture(tex, pos, mask):
    """MLPTexture3D.sample with the coverage mask when the object supports it (duck-typed materials do not)."""
    try:
        return tex.sample(pos, mask=mask)
    except TypeError:
        return tex.sample(pos)


class _ShadeAssembleFn(torch.autograd.Function):
    """The tail of shade() + the background composite of render_mesh as ONE kernel pair (gs_shade_assemble_fwd / bwd):
    -> [B,H,W,44 | 45] = shaded z_grad normal geometric_normal kd ks kd_grad ks_grad normal_grad diffuse_light specular_light
    [msdf_image], every buffer with its alpha channel, composited over the background (shaded) or zero."""

    @staticmethod
    def forward(ctx, rast, tex, texj, n_in, n_jit, mask_tap, n_shade, n_geo, depth, dif, spc, msdf_img, background):
        from .. import _lib
        from .._lib import c_int, c_int64, check, ptr, stream
        B, H, W = rast.shape[0], rast.shape[1], rast.shape[2]
        f = lambda t: None if t is None else t.detach().contiguous().float()
        t = [f(x) for x in (rast, tex, texj, n_in, n_jit, mask_tap, n_shade, n_geo, depth, dif, spc, msdf_img, background)]
        cw = int(t[9].shape[-1])
        C = 45 if msdf_img is not None else 44
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=rast.device)
        with torch.cuda.device(rast.device):
            check(_lib.lib().gs_shade_assemble_fwd(*[ptr(x) for x in t[:11]], c_int(cw), ptr(t[11]), ptr(t[12]), c_int(int(t[12].shape[0])), c_int64(B), c_int64(H),
                                                   c_int64(W), ptr(out), stream()), "gs_shade_assemble_fwd")
        ctx.t, ctx.cw, ctx.dims = t, cw, (B, H, W)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from .. import _lib
        from .._lib import c_int, c_int64, check, ptr, stream
        t, cw, (B, H, W) = ctx.t, ctx.cw, ctx.dims
        dev = g_out.device
        g = g_out.contiguous().float()
        e = lambda c: torch.empty((B, H, W, c), dtype=torch.float32, device=dev)
        g_tex, g_texj, g_nin, g_njit, g_nsh, g_ngeo, g_dif, g_spc = e(6), e(6), e(3), e(3), e(3), e(3), e(cw), e(cw)
        g_msdf = e(1) if t[11] is not None else None
        with torch.cuda.device(dev):
            check(_lib.lib().gs_shade_assemble_bwd(*[ptr(x) for x in t[:11]], c_int(cw), ptr(t[11]), c_int64(B), c_int64(H), c_int64(W), ptr(g), ptr(g_tex),
                                                   ptr(g_texj), ptr(g_nin), ptr(g_njit), ptr(g_nsh), ptr(g_ngeo), ptr(g_dif), ptr(g_spc), ptr(g_msdf), stream()),
                  "gs_shade_assemble_bwd")
        return None, g_tex, g_texj, g_nin, g_njit, None, g_nsh, g_ngeo, None, g_dif, g_spc, g_msdf, None


class PendingFrame:
    """What shade() hands to render_mesh on the fused path: the per-pixel inputs of gs_shade_assemble (the buffer dictionary
    of the reference is materialised by the composite, as channel slices of one tensor)."""
    KEYS = ['shaded', 'z_grad', 'normal', 'geometric_normal', 'kd', 'ks', 'kd_grad', 'ks_grad', 'normal_grad', 'diffuse_light', 'specular_light']

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.msdf_image = None


# ==============================================================================================
#  pixel shader
# ==============================================================================================
def shade(FLAGS, rast, gb_depth, gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_texc, gb_texc_deriv, view_pos, lgt, material, optix_ctx,
          mesh, bsdf, denoiser, shadow_scale, use_uv=True, finetune_normal=True, xfm_lgt=None, shade_data=False, _defer=False):
    dev = gb_pos.device
    B, H, W = gb_depth.shape[0], gb_depth.shape[1], gb_depth.shape[2]
    offset = _noise('jitter', lambda: torch.normal(mean=0, std=0.005, size=(B, H, W, 2), device=dev), FLAGS, 0.005, (B, H, W, 2), dev)
    jitter = (util.pixel_grid(W, H, device=dev)[None, ...] + offset).contiguous()

    mask = (rast[..., -1:] > 0).float()
    mask_tap = dr.texture(mask.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
    grad_weight = mask * mask_tap

    # ---- texture lookups ------------------------------------------------------------------------
    perturbed_nrm = None
    if 'kd_ks' in material:
        noise = _noise('texture', lambda: torch.normal(mean=0, std=0.01, size=gb_pos.shape, device=dev), FLAGS, 0.01, tuple(gb_pos.shape), dev)
        tex_obj = material['kd_ks']
        if hasattr(tex_obj, 'sample_many'):      # both lookups in one pass over the concatenated rows (same values)
            all_tex_jitter, all_tex = tex_obj.sample_many([gb_pos + noise, gb_pos], mask)
        else:
            all_tex_jitter = _sample_texture(tex_obj, gb_pos + noise, mask)
            all_tex = _sample_texture(tex_obj, gb_pos, mask)
        assert all_tex.shape[-1] == 6, "Combined kd_ks must be 6 channels"
        kd, ks = all_tex[..., 0:3], all_tex[..., 3:6]
        if not (_defer and getattr(FLAGS, "fused_assemble", True)):
            kd_grad = torch.abs(all_tex_jitter[..., 0:3] - kd)
            ks_grad = torch.abs(all_tex_jitter[..., 3:6] - ks) * _const_011(dev)      # omit the o-component (reference :74)
        else:
            kd_grad = ks_grad = None
    else:
        raise NotImplementedError("only the combined 'kd_ks' material of the G-Shell scripts is supported (uv-textured materials need "
                                  "mip-mapped texture sampling, which the G-Shell path never enters)")

    alpha = kd[..., 3:4] if kd.shape[-1] == 4 else None       # None: all ones, materialised only where the unfused tail needs it
    kd = kd[..., 0:3]

    # ---- normal regulariser + shading normal ---------------------------------------------------------
    if (not finetune_normal) or ('no_perturbed_nrm' in material and material['no_perturbed_nrm']):
        perturbed_nrm = None
    nrm_jitter = dr.texture(gb_normal.contiguous(), jitter, filter_mode='linear', boundary_mode='clamp')
    gb_normal_interp = gb_normal
    fused = (_defer and getattr(FLAGS, "fused_assemble", True) and (material['bsdf'] if bsdf is None else bsdf) == 'pbr'
             and kd.shape[-1] == 3 and (denoiser is None or (FLAGS.denoiser_demodulate and hasattr(denoiser, "filter_raw"))))
    nrm_grad = None if fused else torch.abs(nrm_jitter - gb_normal) * grad_weight

    gb_normal = ru.prepare_shading_normal(gb_pos, view_pos, perturbed_nrm, gb_normal, gb_tangent, gb_geometric_normal, two_sided_shading=True,
                                          opengl=True)

    # ---- BSDF ------------------------------------------------------------------------------------------
    assert 'bsdf' in material or bsdf is not None, "Material must specify a BSDF type"
    bsdf = material['bsdf'] if bsdf is None else bsdf
    diffuse_accum = specular_accum = None
    if bsdf in ('pbr', 'diffuse', 'white'):
        kd = torch.ones_like(kd) if bsdf == 'white' else kd
        assert isinstance(lgt, light.EnvironmentLight) and optix_ctx is not None
        global rnd_seed
        # reference (render.py:131-133): ro = gb_pos + gb_normal * 0.001, kd / ks as channel slices of the texture sample.  Here the
        # kernel forms ro itself (ro=None) and, when kd / ks are the plain halves of the MLP texture's output, reads that tensor in
        # place (kd_ks=...): no slice copies forward, ONE gradient tensor instead of two slice backward passes
        packed = fused and bsdf == 'pbr'
        extra = dict(kd_ks=all_tex) if packed else {}
        diffuse_accum, specular_accum = ou.optix_env_shade(optix_ctx, rast[..., -1], None, gb_pos, gb_normal, view_pos, kd, ks, lgt.base, lgt._pdf,
                                                           lgt.rows[:, 0], lgt.cols, BSDF=bsdf, n_samples_x=FLAGS.n_samples,
                                                           rnd_seed=None if FLAGS.decorrelated else rnd_seed, shadow_scale=shadow_scale,
                                                           **extra, **_view_map(FLAGS))
        rnd_seed += 1
        if ou.last_covered_pixels is not None:
            # the texture field's backward sizes the bins of its table gradient by the rows that carry gradient: the covered pixels of this
            # frame (known on the host since the shader's pixel list) x the two coordinate sets sampled above
            from . import mlptexture as _mt
            _mt.ACTIVE_ROWS_HINT[0] = 2 * int(ou.last_covered_pixels)
        if fused:
            # everything below (demodulated filtering weights, kd * (1 - metalness), the buffer dictionary with its alpha
            # channels) happens inside gs_shade_assemble, called by render_mesh once the background is known
            if denoiser is not None:
                with torch.no_grad():
                    nrm_unit = util.safe_normalize(gb_normal)       # the filter's guides carry no gradient (optixutils.py backward)
                # only covered pixels are read by gs_shade_assemble (alpha = 0 elsewhere): the others are taps, not centres
                if hasattr(denoiser, "filter_raw_pair"):      # same guides, same weights: one pass for both images
                    diffuse_accum, specular_accum = denoiser.filter_raw_pair(diffuse_accum, specular_accum, nrm_unit, gb_depth, mask)
                else:
                    diffuse_accum = denoiser.filter_raw(diffuse_accum, nrm_unit, gb_depth, mask)        # [B,H,W,4] = (sum w c, sum w)
                    specular_accum = denoiser.filter_raw(specular_accum, nrm_unit, gb_depth, mask)
            return PendingFrame(tex=all_tex, texj=all_tex_jitter, n_in=gb_normal_interp, n_jit=nrm_jitter, mask_tap=grad_weight, n_shade=gb_normal,
                                n_geo=gb_geometric_normal, depth=gb_depth, dif=diffuse_accum, spc=specular_accum)
        if denoiser is not None and FLAGS.denoiser_demodulate:
            diffuse_accum = denoiser.forward(torch.cat((diffuse_accum, gb_normal, gb_depth), dim=-1))
            specular_accum = denoiser.forward(torch.cat((specular_accum, gb_normal, gb_depth), dim=-1))
        if bsdf in ('white', 'diffuse'):
            shaded_col = diffuse_accum * kd
        else:
            kd = kd * (1.0 - ks[..., 2:3])      # kd * (1 - metalness)
            shaded_col = diffuse_accum * kd + specular_accum
        if denoiser is not None and not FLAGS.denoiser_demodulate:
            shaded_col = denoiser.forward(torch.cat((shaded_col, gb_normal, gb_depth), dim=-1))
    elif bsdf == 'normal':
        shaded_col = (gb_normal + 1.0) * 0.5
    elif bsdf == 'tangent':
        shaded_col = (gb_tangent + 1.0) * 0.5
    elif bsdf == 'kd':
        shaded_col = kd
    elif bsdf == 'ks':
        shaded_col = ks
    else:
        assert False, "Invalid BSDF '%s'" % bsdf

    if alpha is None:
        alpha = torch.ones_like(kd[..., 0:1])
    if kd_grad is None:          # the fused path was requested but this configuration is not covered by the kernel
        kd_grad = torch.abs(all_tex_jitter[..., 0:3] - all_tex[..., 0:3])
        ks_grad = torch.abs(all_tex_jitter[..., 3:6] - ks) * _const_011(dev)
    if nrm_grad is None:
        nrm_grad = torch.abs(nrm_jitter - gb_normal_interp) * grad_weight
    buffers = {
        'shaded': torch.cat((shaded_col, alpha), dim=-1),
        'z_grad': torch.cat((gb_depth, torch.zeros_like(alpha), alpha), dim=-1),
        'normal': torch.cat((gb_normal, alpha), dim=-1),
        'geometric_normal': torch.cat((gb_geometric_normal, alpha), dim=-1),
        'kd': torch.cat((kd, alpha), dim=-1),
        'ks': torch.cat((ks, alpha), dim=-1),
        'kd_grad': torch.cat((kd_grad, alpha), dim=-1),
        'ks_grad': torch.cat((ks_grad, alpha), dim=-1),
        'normal_grad': torch.cat((nrm_grad, alpha), dim=-1),
    }
    if diffuse_accum is not None:
        buffers['diffuse_light'] = torch.cat((diffuse_accum, alpha), dim=-1)
    if specular_accum is not None:
        buffers['specular_light'] = torch.cat((specular_accum, alpha), dim=-1)
    return buffers


# ==============================================================================================
#  one depth layer
# ==============================================================================================
def render_layer(FLAGS, v_pos_clip, rast, rast_deriv, mesh, view_pos, lgt, resolution, spp, msaa, optix_ctx, bsdf, denoiser, shadow_scale,
                 use_uv=True, finetune_normal=True, extra_dict=None, xfm_lgt=None, shade_data=False, _defer=False):
    full_res = [resolution[0] * spp, resolution[1] * spp]
    if spp > 1 and msaa:
        rast_out_s = util.scale_img_nhwc(rast, resolution, mag='nearest', min='nearest')
        rast_out_deriv_s = util.scale_img_nhwc(rast_deriv, resolution, mag='nearest', min='nearest') * spp
    else:
        rast_out_s, rast_out_deriv_s = rast, rast_deriv
    if use_uv:
        raise NotImplementedError("use_uv=True (uv-unwrapped second pass) is not on the G-Shell training path")
    tri = mesh.faces_i32()
    assert mesh.v_nrm is not None

    # position + smooth normal (+ mSDF) in one interpolation pass (reference render.py:240, :263, :306: one call each)
    msdf = None
    if extra_dict is not None and extra_dict.get('msdf', None) is not None:
        msdf = extra_dict['msdf']
        assert msdf.dim() == 1 or (msdf.dim() == 2 and msdf.size(1) == 1)
    stack = [mesh.v_pos, mesh.v_nrm] + ([msdf.reshape(-1, 1)] if msdf is not None else [])
    gb = dr.interpolate_groups(stack, rast_out_s, tri)          # one launch, one contiguous tensor per attribute
    gb_pos, gb_normal = gb[0], gb[1]

    # geometric (face) normal of the covering triangle (fused gather + normalise; a torch gather here would
    # back-propagate through index_put with ~1e6 duplicate indices: 260 ms at 4 x 512^2)
    gb_geometric_normal = dr.face_normals(mesh.v_pos, tri, rast_out_s)

    with torch.no_grad():
        noise = _noise('tangent', lambda: torch.randn_like(gb_normal), FLAGS, 1.0, tuple(gb_normal.shape), gb_normal.device)
        noise = noise / noise.norm(dim=-1, keepdim=True)
    # only used to add isotropic noise (no uv maps).  With perturbed_nrm = None the shading normal is tng * 0 + btng * 0 + nrm * 1: the
    # tangent's gradient is exactly zero, so the cross product is taken of the detached normal (saves its backward and an add of zeros)
    gb_tangent = torch.cross(noise, gb_normal.detach(), dim=-1)
    gb_texc, gb_texc_deriv = None, None

    with torch.no_grad():
        eps = 0.00001
        clip_pos, clip_pos_deriv = interpolate(v_pos_clip, rast_out_s, tri, rast_db=rast_out_deriv_s)
        # clip_pos_deriv layout: (d/dX, d/dY) interleaved per attribute -> the reference reads [2:3], [3:4]
        gb_depth = ru.depth_zgrad(clip_pos, clip_pos_deriv, eps)          # (z0, |z1 - z0|), reference :276-279 in one launch

    buffers = shade(FLAGS, rast_out_s, gb_depth, gb_pos, gb_geometric_normal, gb_normal, gb_tangent, gb_texc, gb_texc_deriv, view_pos, lgt,
                    mesh.material, optix_ctx, mesh, bsdf, denoiser, shadow_scale, use_uv=use_uv, finetune_normal=finetune_normal, xfm_lgt=xfm_lgt,
                    shade_data=shade_data, _defer=_defer and spp == 1)
    if isinstance(buffers, PendingFrame):
        buffers.msdf_image = gb[2] if msdf is not None else None
        return buffers
    if msdf is not None:
        buffers['msdf_image'] = gb[2]
    if spp > 1 and msaa:
        for key in buffers.keys():
            buffers[key] = util.scale_img_nhwc(buffers[key], full_res, mag='nearest', min='nearest')
    return buffers


class _LazyVisible:
    """`visible_triangles` of the reference (render.py:380-383: sorted unique triangle ids of the frame) costs a device ->
    host round trip for its SIZE; the training loop only needs the per-triangle flags, so the index list is materialised on
    first access."""

    def __init__(self, flags):
        self.flags = flags

    def resolve(self):
        return torch.nonzero(self.flags).reshape(-1)          # sorted ids == rast[...,-1].long().unique() - 1


class FrameBuffers(dict):
    """The reference's dict of output buffers (render.py:436-444) + `.stacked` = (tensor [B,H,W,sum C], names, channel counts):
    all buffers are channel slices of that one antialiased tensor, which lets the loss read the frame in a single pass;
    `.visible_flags` [T] uint8 = the rasteriser's per-triangle visibility flags (what `visible_triangles` is the nonzero of)."""
    stacked = None
    visible_flags = None

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, _LazyVisible):
            v = v.resolve()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default


_layout_cache = {}


def _composite_layout(sizes, dev):
    """Per-channel constants of the stacked composite: index of the alpha channel of each channel's buffer, alpha mask."""
    key = (sizes, str(dev))
    lay = _layout_cache.get(key)
    if lay is None:
        alpha_of, is_alpha, o = [], [], 0
        for c in sizes:
            alpha_of += [o + c - 1] * c
            is_alpha += [False] * (c - 1) + [True]
            o += c
        n = len(alpha_of)
        pick = torch.zeros(n, n, dtype=torch.float32)
        pick[torch.tensor(alpha_of), torch.arange(n)] = 1.0                              # column c selects the alpha channel of c's buffer
        lay = {'pick_alpha': pick.to(dev), 'is_alpha': torch.tensor(is_alpha, dtype=torch.bool, device=dev),
               'one': torch.ones((), dtype=torch.float32, device=dev)}
        _layout_cache[key] = lay
    return lay


# ==============================================================================================
#  render a mesh (single layer)
# ==============================================================================================
def render_mesh(FLAGS, ctx, mesh, mtx_in, view_pos, lgt, resolution, spp=1, num_layers=1, msaa=False, background=None, optix_ctx=None, bsdf=None,
                denoiser=None, shadow_scale=1.0, use_uv=True, finetune_normal=True, extra_dict=None, xfm_lgt=None, shade_data=False):
    dev = mesh.v_pos.device

    def prepare_input_vector(x):
        x = torch.tensor(x, dtype=torch.float32, device=dev) if not torch.is_tensor(x) else x
        return x[:, None, None, :] if len(x.shape) == 2 else x

    assert num_layers == 1, "the reference asserts a single layer (render/render.py:378)"
    full_res = [resolution[0] * spp, resolution[1] * spp]
    mtx_in = torch.tensor(mtx_in, dtype=torch.float32, device=dev) if not torch.is_tensor(mtx_in) else mtx_in
    view_pos = prepare_input_vector(view_pos)
    tri = mesh.faces_i32()

    v_pos_clip = ru.xfm_points(mesh.v_pos[None, ...], mtx_in)
    rast, db, vis = dr.rasterize(ctx, v_pos_clip, tri, full_res, return_visible=True)
    # the shader's covered-pixel list: requested now, its count read ~0.5 ms of queued work later (optixutils.CoveredPixels).  The request belongs to
    # THIS frame: an unconsumed one of an earlier frame (a render that never shaded) is dropped here, and ours below once the layer is rendered --
    # the caching allocator hands the next frame's `rast` the same address, so a stale request must never outlive its frame
    ou.PENDING_PIXELS = ou.CoveredPixels(rast) if (spp == 1 and optix_ctx is not None and rast.is_cuda and getattr(FLAGS, "async_pixel_list", True)) else None

    buffers = render_layer(FLAGS, v_pos_clip, rast, db, mesh, view_pos, lgt, resolution, spp, msaa, optix_ctx, bsdf, denoiser, shadow_scale,
                           use_uv=use_uv, finetune_normal=finetune_normal, extra_dict=extra_dict, xfm_lgt=xfm_lgt, shade_data=shade_data, _defer=True)
    ou.PENDING_PIXELS = None         # (the shader has run inside render_layer: consumed, or never needed)

    if background is not None:
        if spp > 1:
            background = util.scale_img_nhwc(background, full_res, mag='nearest', min='nearest')
        background = torch.cat((background, torch.zeros_like(background[..., 0:1])), dim=-1)
    else:
        background = torch.zeros(1, full_res[0], full_res[1], 4, dtype=torch.float32, device=dev)

    # composite every buffer over its background, then antialias all of them in one stacked launch.  The reference loops
    # over the ~12 buffers (render.py:417-433: alpha, cat, lerp, antialias each); here the buffers are stacked once along
    # the channel axis and composited with a handful of whole-stack ops: out = lerp(bg, [rgb.., 1], cover * alpha_of_group).
    if isinstance(buffers, PendingFrame):
        pf = buffers
        keys = list(PendingFrame.KEYS) + (['msdf_image'] if pf.msdf_image is not None else [])
        sizes = [4] * 11 + ([1] if pf.msdf_image is not None else [])
        comp = _ShadeAssembleFn.apply(rast, pf.tex, pf.texj, pf.n_in, pf.n_jit, pf.mask_tap, pf.n_shade, pf.n_geo, pf.depth, pf.dif, pf.spc,
                                      pf.msdf_image, background[..., 0:3])
        # comp is this function's own: updated in place.  Its gradient is exclusive too: `aa` leaves this function only as `out_buffers.stacked`
        # (consumed by regularizer.frame_sums, whose backward allocates what it returns) and as the split views below, whose backward
        # (cat of the parts' gradients) allocates a fresh frame; several consumers are summed into the engine's own buffer
        aa = dr.antialias_stacked([comp], rast, v_pos_clip, tri, inplace=True, grad_exclusive=True)[0]
        out_list = list(torch.split(aa, sizes, dim=-1))
        buffers = None
    else:
        keys = [k for k, b in buffers.items() if b is not None]
    if buffers is None:
        pass
    elif keys:
        sizes = [buffers[k].shape[-1] for k in keys]
        cover = (rast[..., -1:] > 0).float()
        layout = _composite_layout(tuple(sizes), dev)
        stacked = torch.cat([buffers[k] for k in keys], dim=-1)                       # [B,H,W,sum C]
        a = cover * torch.matmul(stacked, layout['pick_alpha'])                        # alpha of each channel's own buffer (0/1 matrix: exact,
                                                                                       # and its backward is a dense product, not 47 M atomics)
        fg = torch.where(layout['is_alpha'], layout['one'], stacked)                   # (rgb.., 1)
        bg = torch.zeros((background.shape[0],) + tuple(stacked.shape[1:]), dtype=torch.float32, device=dev)
        if 'shaded' in keys:
            o = sum(sizes[:keys.index('shaded')])
            bg[..., o:o + background.shape[-1]] = background
        comp = torch.lerp(bg.expand_as(fg), fg, a)
        aa = dr.antialias_stacked([comp], rast, v_pos_clip, tri)[0]
        out_list = list(torch.split(aa, sizes, dim=-1))
    else:
        aa, sizes, out_list = None, [], []

    out_buffers = FrameBuffers({'visible_triangles': _LazyVisible(vis)})
    out_buffers.visible_flags = vis
    if aa is not None and spp == 1:
        out_buffers.stacked = (aa, keys, sizes)              # every buffer below is a channel slice of this tensor
    for key, accum in zip(keys, out_list):
        out_buffers[key] = util.avg_pool_nhwc(accum, spp) if spp > 1 else accum
    return out_buffers
