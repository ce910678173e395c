"""CPU oracle of the WHOLE render path: composition of the stage oracles in the order of the reference's
render/render.py (render_mesh :325-444 -> render_layer :199-317 -> shade :31-191), used
  * by tests/test_render_gpu.py as the end-to-end parity check of gshell_amd.render.render.render_mesh, and
  * by bench.py's `cpu_baseline` leg (timed_sample) as the CPU port timed next to the GPU number.
TEST INFRASTRUCTURE -- checker only; never imported by gshell_amd/."""
import os
import time

import numpy as np
import torch

from oracle import hashgrid_oracle as ho
from oracle import mtets_oracle
from oracle import pixel_oracle as po
from oracle import raster_oracle as ro
from oracle import shade_oracle as so


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


class TextureOracle:
    """MLPTexture3D.sample (render/mlptexture.py:87-98) on top of the hash-grid oracle; weights are given."""

    def __init__(self, aabb, cfg, params, weights, mn, mx):
        self.aabb, self.cfg, self.params, self.weights, self.mn, self.mx = aabb, cfg, params, weights, mn, mx

    def sample_covered(self, texc, covered):
        """`sample` on the covered pixels only, zeros elsewhere.  The reference evaluates the field on every pixel and the composite then discards
        the background ones (render.py:352-359: lerp(bg, fg, mask * alpha) with mask = 0 there), so values and gradients of everything the
        frame shows are the same; what it saves is 6/7 of the hash-grid oracle's memory at the config-size frames (oracle/make_golden_chain.py)."""
        rows = torch.nonzero(covered.reshape(-1)).reshape(-1)
        flat = texc.reshape(-1, 3)
        val = self.sample(flat[rows])
        out = torch.zeros(flat.shape[0], val.shape[-1], dtype=val.dtype).index_put((rows,), val)
        return out.reshape(*texc.shape[:-1], -1)

    def sample(self, texc):
        x = (texc.reshape(-1, 3) - self.aabb[0][None]) / (self.aabb[1][None] - self.aabb[0][None])
        h = ho.encode(torch.clamp(x, 0, 1), self.params, *self.cfg)
        for i, w in enumerate(self.weights):
            h = h @ w.t()
            if i + 1 < len(self.weights):
                h = torch.relu(h)
        out = torch.sigmoid(h) * (self.mx - self.mn)[None] + self.mn[None]
        return out.reshape(*texc.shape[:-1], -1)


class ConstantTextureOracle:
    """BASELINE.json configs[0] "constant kd": an object whose .sample(pos) returns a constant [..., 6] (duck-types
    render/mlptexture.py:87)."""

    def __init__(self, value):
        self.value = value            # [6] kd rgb, ks (o, roughness, metalness); may require grad

    def sample(self, texc):
        return self.value.expand(*texc.shape[:-1], 6) + 0.0 * texc[..., :1]


def render_mesh(v_pos, faces, v_nrm, msdf, mvp, campos, light, background, noise, texture, n_samples, seed, shadow_scale, perms, bsdf='pbr',
                denoise_sigma=None, resolution=(32, 32), xfm=None, covered_texture=False):
    """All tensors torch CPU float32.  faces [T,3] long.  noise = {'jitter','texture','tangent'} as drawn by the product.
    Returns the dict of composited + antialiased buffers (same keys as the reference).
    xfm: the point transform -- default raster_oracle.xfm_points (the reference's python branch, a matmul); the config-size chains pass
    raster_oracle.xfm_points_kernel_order (the CUDA kernel's sum order without contraction = the HIP path's clip coordinates bit for bit)."""
    xfm = xfm or ro.xfm_points
    H, W = resolution
    B = mvp.shape[0]
    tri_np = faces.numpy().astype(np.int32)
    v_pos_clip = xfm(v_pos[None], mvp)
    # discrete decisions (coverage, sample placement) are always made from float32 values, whatever dtype the floats run in
    ids = torch.tensor(ro.rasterize_ids_c(xfm(v_pos.detach().float()[None], mvp.float()).numpy(), tri_np, H, W))     # oracle/raster_c.c
    rast, rast_db = ro.rast_from_ids(v_pos_clip, faces, ids)
    visible = torch.unique(ids[ids >= 0])
    gb_pos = ro.interpolate(v_pos[None], rast, faces)
    gb_normal = ro.interpolate(v_nrm[None], rast, faces)
    v0, v1, v2 = v_pos[faces[:, 0]], v_pos[faces[:, 1]], v_pos[faces[:, 2]]
    face_normals = safe_normalize(torch.cross(v1 - v0, v2 - v0, dim=-1))
    covered = (ids >= 0)[..., None]
    gb_geo = torch.where(covered, face_normals[ids.clamp(min=0)], torch.zeros(())) if faces.shape[0] else torch.zeros_like(gb_pos)
    tn = noise['tangent'] / noise['tangent'].norm(dim=-1, keepdim=True)
    gb_tangent = torch.cross(tn, gb_normal, dim=-1)
    with torch.no_grad():
        eps = 0.00001
        clip_pos, clip_da = ro.interpolate(v_pos_clip.detach(), rast.detach(), faces, rast_db)
        z0 = torch.clamp(clip_pos[..., 2:3], min=eps) / torch.clamp(clip_pos[..., 3:4], min=eps)
        z1 = torch.clamp(clip_pos[..., 2:3] + clip_da[..., 2:3].abs(), min=eps) / torch.clamp(clip_pos[..., 3:4] + clip_da[..., 3:4].abs(), min=eps)
        gb_depth = torch.cat((z0, (z1 - z0).abs()), -1)
    view_pos = campos[:, None, None, :]
    # ---- shade
    jitter = po.pixel_grid(W, H)[None] + noise['jitter']
    mask = (rast[..., -1:] > 0).to(rast.dtype)
    grad_weight = mask * po.texture_linear_clamp(mask, jitter)
    if covered_texture and hasattr(texture, 'sample_covered'):
        all_jit = texture.sample_covered(gb_pos + noise['texture'], covered)
        all_tex = texture.sample_covered(gb_pos, covered)
    else:
        all_jit = texture.sample(gb_pos + noise['texture'])
        all_tex = texture.sample(gb_pos)
    kd, ks = all_tex[..., 0:3], all_tex[..., 3:6]
    kd_grad = (all_jit[..., 0:3] - kd).abs()
    ks_grad = (all_jit[..., 3:6] - ks).abs() * torch.tensor([0.0, 1.0, 1.0])
    alpha = torch.ones_like(kd[..., 0:1])
    nrm_grad = (po.texture_linear_clamp(gb_normal, jitter) - gb_normal).abs() * grad_weight
    gb_nrm_s = po.prepare_shading_normal(gb_pos, view_pos, None, gb_normal, gb_tangent, gb_geo, True, True)
    ro_ = gb_pos + gb_nrm_s * 0.001
    pdf, rows, cols = po.update_pdf(light.detach())
    diff, spec = so.env_shade(rast[..., -1], ro_, gb_pos, gb_nrm_s, view_pos, kd, ks, light, pdf, rows[:, 0], cols, perms, ['pbr', 'diffuse', 'white'].index(bsdf),
                              n_samples, seed, shadow_scale, v_pos.detach().numpy(), faces.numpy())
    if denoise_sigma is not None:
        nn_ = safe_normalize(gb_nrm_s)
        d4 = so.bilateral(diff, nn_, gb_depth, denoise_sigma)
        s4 = so.bilateral(spec, nn_, gb_depth, denoise_sigma)
        diff, spec = d4[..., 0:3] / d4[..., 3:4], s4[..., 0:3] / s4[..., 3:4]
    kd_m = kd * (1.0 - ks[..., 2:3])
    shaded = diff * kd_m + spec
    buffers = {
        'shaded': torch.cat((shaded, alpha), -1), 'z_grad': torch.cat((gb_depth, torch.zeros_like(alpha), alpha), -1),
        'normal': torch.cat((gb_nrm_s, alpha), -1), 'geometric_normal': torch.cat((gb_geo, alpha), -1), 'kd': torch.cat((kd_m, alpha), -1),
        'ks': torch.cat((ks, alpha), -1), 'kd_grad': torch.cat((kd_grad, alpha), -1), 'ks_grad': torch.cat((ks_grad, alpha), -1),
        'normal_grad': torch.cat((nrm_grad, alpha), -1), 'diffuse_light': torch.cat((diff, alpha), -1), 'specular_light': torch.cat((spec, alpha), -1),
    }
    if msdf is not None:
        buffers['msdf_image'] = ro.interpolate(msdf.reshape(1, -1, 1), rast, faces)
    bg4 = torch.cat((background, torch.zeros_like(background[..., 0:1])), -1)
    opp = torch.as_tensor(ro.tri_adjacency_sorted(tri_np)) if faces.shape[0] else None
    aa_alpha = ro.aa_alpha(rast.detach(), v_pos_clip, faces, opp) if faces.shape[0] else None
    out = {'visible_triangles': visible}
    for key, buf in buffers.items():
        a = mask * buf[..., -1:]
        fg = torch.cat((buf[..., :-1], torch.ones_like(buf[..., -1:])), -1)
        bg = bg4 if key == 'shaded' else torch.zeros_like(fg)
        comp = torch.lerp(bg.expand_as(fg), fg, a)
        out[key] = ro.aa_apply(comp, aa_alpha) if aa_alpha is not None else comp
    return out


def timed_sample(cells=16, res=(64, 64), n_samples=4, seed=0, constant_kd=False):
    """CPU baseline for bench.py: ONE forward + backward pass of the oracle pipeline (extraction -> normals -> raster ->
    interpolate -> hash-grid texture -> MC shading with brute-force shadow rays -> bilateral -> composite/antialias ->
    image loss) on a bounded sample of the workload, single process, torch/numpy on the host cores."""
    from gshell_amd import grid
    from oracle import fields, scenes
    torch.manual_seed(seed)
    t_start = time.perf_counter()
    verts, tets = grid.bcc_grid(cells)
    vn = verts.numpy()
    pos = torch.tensor(vn, requires_grad=True)
    sdf = torch.tensor(fields.make_sdf(vn, "skirt", 3), requires_grad=True)
    msdf = torch.tensor(fields.make_msdf(vn, "wavy", 3), requires_grad=True)
    ex = mtets_oracle.extract(pos, sdf, msdf, tets, with_tangents=False)
    v, f = ex["verts_aug"] * 2.0, ex["faces_aug"]
    nrm = po.auto_normals(v, f)
    H, W = res
    mvp, cam = scenes.orbit_views(1)
    gen = torch.Generator().manual_seed(seed)
    noise = {'jitter': torch.randn(1, H, W, 2, generator=gen) * 0.005, 'texture': torch.randn(1, H, W, 3, generator=gen) * 0.01,
             'tangent': torch.randn(1, H, W, 3, generator=gen)}
    cfg = (16, 2, 19, 16, float(np.exp(np.log(4096 / 16) / 15)))
    _, total = ho.level_meta(*cfg)
    params = ((torch.rand(total, generator=gen) * 2 - 1) * 1e-4).requires_grad_(True)
    weights = [(torch.randn(32, 32, generator=gen) * 0.3).requires_grad_(True), (torch.randn(32, 32, generator=gen) * 0.3).requires_grad_(True),
               (torch.randn(6, 32, generator=gen) * 0.3).requires_grad_(True)]
    tex = TextureOracle((torch.tensor([-1.0, -1, -1]), torch.tensor([1.0, 1, 1])), cfg, params, weights, torch.tensor([0, 0, 0, 0, 0.001, 0]),
                        torch.tensor([1, 1, 1, 0, 1.0, 1]))
    if constant_kd:
        tex = ConstantTextureOracle(torch.tensor([0.7, 0.55, 0.4, 0.0, 0.5, 0.1], requires_grad=True))
    light = torch.full((32, 64, 3), 0.5, requires_grad=True)
    perms = torch.argsort(torch.rand(256, n_samples * n_samples, generator=gen), dim=-1).int().numpy()
    bg = torch.rand(1, H, W, 3, generator=gen)
    out = render_mesh(v, f, nrm, ex["msdf"], torch.tensor(mvp), torch.tensor(cam), light, bg, noise, tex, n_samples, 1, 1.0, perms, denoise_sigma=1.0,
                      resolution=res)
    target = torch.rand(1, H, W, 3, generator=gen)
    loss = po.image_loss(out['shaded'][..., 0:3], target, 'l1', 'log_srgb') + out['msdf_image'].abs().mean()
    loss.backward()
    dt = time.perf_counter() - t_start
    mpix = H * W / dt / 1e6
    return {"value": round(mpix, 8), "unit": "Mpixels/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle pipeline fwd+bwd, 1 view {H}x{W}, BCC {cells} cells ({tets.shape[0]} tets, {f.shape[0]} faces), "
                      f"{'constant kd/ks' if constant_kd else 'hash-grid texture'}, n_samples={n_samples} "
                      f"({2 * n_samples ** 2} brute-force shadow rays/px), bilateral sigma 1, {dt:.1f} s wall"}
