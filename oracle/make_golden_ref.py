"""Mint golden vectors from the REFERENCE'S OWN native kernels compiled for the host (oracle/_ref, build container only).

    make -C oracle && python -m oracle.make_golden_ref

  tests/golden/ref_envshade_{pbr,diffuse,white}_n{1,4,8}.npz,  ref_envshade_pbr_n8_64x64.npz,  ref_envshade_pbr_n4_occluder.npz
                                                                kernel.cu raygen, forward + the five gradients of backward = 1,
                                                                + the per-sample record (direction, pdf_light, pdf_bsdf, visible)
                                                                of every covered pixel (ref_env_shade_trace_pixel)
  tests/golden/ref_image_loss.npz                               loss.cu: {l1,mse,smape,relmse} x {none,log_srgb}, value + both gradients
  tests/golden/ref_shading_normal.npz                           normal.cu: forward + six gradients, two_sided x opengl
  tests/golden/ref_xfm_points.npz                               mesh.cu: forward + gradient
  tests/golden/ref_bilateral.npz                                denoising.cu: sigma 0.4 / 1 / 2, forward + gradient

One host thread (refnative.set_threads(1)) makes the atomicAdd order into light_grad the launch order z, y, x.
tests/test_oracle_ref_cpu.py re-mints and compares bit for bit wherever oracle/_ref is built."""
import os

import numpy as np
import torch

from oracle import pixel_oracle as po
from oracle import refnative as rn
from oracle import scenes

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
BSDFS = ("pbr", "diffuse", "white")
ENVSHADE_CASES = [(b, n) for b in BSDFS for n in (1, 4, 8)]
PERM_ROWS = 64


def envshade_case(bsdf, n, frame=None, probe=None, occluder=False):
    """Inputs of one env-shade golden (deterministic).  -> dict of numpy arrays + scalars.
    occluder: the shadow-ray mesh also holds a plate hovering above ONE HALF of the sheet (on the side its normals face), so that a
    large share of that half's shadow rays HIT -- the sheet alone occludes almost nothing."""
    big = n == 8
    B, H, W = frame or ((1, 20, 20) if big else (2, 20, 20))
    probe = probe or ((32, 64) if big else (16, 32))
    seed_scene = 3 + BSDFS.index(bsdf)
    verts, tri, mask, gb_pos, gb_nrm, view, kd, ks = scenes.sheet_gbuffer(B, H, W, seed_scene)
    gen = torch.Generator().manual_seed(9 + n)
    light = torch.rand(probe[0], probe[1], 3, generator=gen) * 2 + 0.05
    pdf, rows, cols = po.update_pdf(light)
    perms = torch.argsort(torch.rand(PERM_ROWS, n * n, generator=gen), dim=-1).to(torch.uint8)
    wd, ws = torch.rand(B, H, W, 3, generator=gen), torch.rand(B, H, W, 3, generator=gen)
    shadow = 0.6 if (bsdf, n) == ("pbr", 4) else 1.0
    ro = gb_pos + gb_nrm * 0.001
    if occluder:
        # plates at distance 0.18 on BOTH sides of the sheet (the cameras see either face), over the half x < 0, 40 x 40 quads each so
        # that the BVH has real work; rays of the other half reach them only at grazing angles
        m = 40
        u, v = np.meshgrid(np.linspace(-0.75, 0.0, m + 1), np.linspace(-0.75, 0.75, m + 1), indexing="ij")
        idx = np.arange((m + 1) * (m + 1)).reshape(m + 1, m + 1)
        a, b, c, e = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        quad_tri = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([a, c, e], -1).reshape(-1, 3)]).astype(np.int32)
        for z in (0.33, -0.33):
            plate = np.stack([u, v, np.full_like(u, z)], -1).reshape(-1, 3).astype(np.float32)
            tri = np.concatenate([tri, quad_tri + len(verts)])
            verts = np.concatenate([verts, plate])
    d = dict(mask=mask, ro=ro, gb_pos=gb_pos, gb_normal=gb_nrm, view_pos=view, gb_kd=kd, gb_ks=ks, light=light, pdf=pdf, rows=rows[:, 0].contiguous(),
             cols=cols, perms=perms, verts=torch.tensor(verts), tris=torch.tensor(tri), diff_grad=wd, spec_grad=ws)
    d = {k: v.numpy() for k, v in d.items()}
    d.update(bsdf=np.int32(BSDFS.index(bsdf)), n=np.int32(n), seed=np.uint32(1234 + 17 * n), shadow_scale=np.float32(shadow))
    return d


def run_envshade(d):
    a = [d[k] for k in ("mask", "ro", "gb_pos", "gb_normal", "view_pos", "gb_kd", "gb_ks", "light", "pdf", "rows", "cols")]
    tail = (d["perms"].astype(np.int32), int(d["bsdf"]), int(d["n"]), int(d["seed"]), float(d["shadow_scale"]), d["verts"], d["tris"])
    diff, spec = rn.env_shade_fwd(*a, *tail)
    g = rn.env_shade_bwd(*a, *tail, d["diff_grad"], d["spec_grad"])
    n = int(d["n"])
    B, H, W = d["mask"].shape
    pix = np.flatnonzero(d["mask"].reshape(-1) > 0)
    samples = np.stack([rn.env_shade_trace_pixel(int(p % W), int((p // W) % H), int(p // (W * H)), n) for p in pix]) if len(pix) else np.zeros((0, 2 * n * n, 6), np.float32)
    out = dict(diff=diff, spec=spec, samples=samples)
    out.update({f"g_{k}": v for k, v in zip(("gb_pos", "gb_normal", "gb_kd", "gb_ks", "light"), g)})
    return out


def image_loss_case():
    g = torch.Generator().manual_seed(11)
    B, H, W = 2, 13, 21                                  # not a multiple of the 8x8 block / 8x4 warp tile: partial warps
    img = torch.rand(B, H, W, 3, generator=g) * 3 - 0.3    # negatives exercise the clamp and the kernel's unclamped backward
    tgt = torch.rand(B, H, W, 3, generator=g) * 3 - 0.3
    img[0, 0, 0] = 70000.0                               # above the 65535 clamp
    tgt[0, 1, 1] = 0.0
    img[1, 2, 3] = 0.001                                 # below the sRGB knee after log
    tgt[1, 2, 3] = 0.002
    return img.numpy(), tgt.numpy()


def run_image_loss(img, tgt):
    out = {}
    for loss in ("l1", "mse", "smape", "relmse"):
        for tm in ("none", "log_srgb"):
            v, part = rn.image_loss_fwd(img, tgt, loss, tm)
            gi, gt = rn.image_loss_bwd(img, tgt, loss, tm)
            out[f"{loss}_{tm}_value"] = np.float32(v)
            out[f"{loss}_{tm}_partial"] = part
            out[f"{loss}_{tm}_g_img"] = gi
            out[f"{loss}_{tm}_g_target"] = gt
    return out


def shading_normal_case():
    g = torch.Generator().manual_seed(12)
    B, H, W = 2, 9, 11
    t = {k: torch.randn(B, H, W, 3, generator=g) for k in ("pos", "perturbed_nrm", "smooth_nrm", "smooth_tng", "geom_nrm")}
    t["view_pos"] = torch.randn(B, 1, 1, 3, generator=g) * 3
    t["smooth_nrm"][0, 0, 0] = 0.0                       # safeNormalize(0) = 0 (vec3f.h:90), unlike F.normalize's eps
    t["perturbed_nrm"][0, 1, 1, 2] = -0.5                 # max(z, 0) branch
    t["grad"] = torch.randn(B, H, W, 3, generator=g)
    return {k: v.numpy() for k, v in t.items()}


def run_shading_normal(t):
    out = {}
    names = ("pos", "view_pos", "perturbed_nrm", "smooth_nrm", "smooth_tng", "geom_nrm")
    for two_sided in (True, False):
        for opengl in (True, False):
            tag = f"ts{int(two_sided)}_gl{int(opengl)}"
            ins = [t[k] for k in names]
            out[f"{tag}_out"] = rn.prepare_shading_normal_fwd(*ins, two_sided, opengl)
            for k, gk in zip(names, rn.prepare_shading_normal_bwd(*ins, t["grad"], two_sided, opengl)):
                out[f"{tag}_g_{k}"] = gk
    # the training path: perturbed_nrm = (0,0,1) broadcast (renderutils/ops.py:219-220)
    ins = [t["pos"], t["view_pos"], np.array([0, 0, 1], np.float32)[None, None, None], t["smooth_nrm"], t["smooth_tng"], t["geom_nrm"]]
    out["flat_out"] = rn.prepare_shading_normal_fwd(*ins, True, True)
    for k, gk in zip(names, rn.prepare_shading_normal_bwd(*ins, t["grad"], True, True)):
        out[f"flat_g_{k}"] = gk
    return out


def xfm_case():
    g = torch.Generator().manual_seed(13)
    return dict(points=torch.randn(1, 77, 3, generator=g).numpy(), matrix=torch.randn(3, 4, 4, generator=g).numpy(),
                grad=torch.randn(3, 77, 4, generator=g).numpy())


def run_xfm(t):
    return dict(out=rn.xfm_points_fwd(t["points"], t["matrix"]), g_points_full=rn.xfm_points_bwd(t["points"], t["matrix"], t["grad"]))


def bilateral_case():
    g = torch.Generator().manual_seed(14)
    B, H, W = 2, 19, 23
    col = torch.rand(B, H, W, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(B, H, W, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    nrm[0, 3:6, 4:9] = 0.0                               # background pixels: zero normal
    zdz = torch.cat([torch.rand(B, H, W, 1, generator=g) * 0.2 + 2.0, torch.rand(B, H, W, 1, generator=g) * 0.02], -1)
    zdz[1, 5, 5, 1] = 0.0                                # max(dz * dist, eps) branch
    return dict(col=col.numpy(), nrm=nrm.numpy(), zdz=zdz.numpy(), out_grad=torch.randn(B, H, W, 4, generator=g).numpy())


def run_bilateral(t):
    out = {}
    for sigma in (0.4, 1.0, 2.0):
        out[f"out_{sigma}"] = rn.bilateral_fwd(t["col"], t["nrm"], t["zdz"], sigma)
        out[f"g_col_{sigma}"] = rn.bilateral_bwd(t["col"], t["nrm"], t["zdz"], sigma, t["out_grad"])
    return out


def all_goldens():
    """{file name: dict of arrays} -- everything this script writes (also used by the re-mint test)."""
    files = {}
    rn.set_threads(1)
    for bsdf, n in ENVSHADE_CASES:
        d = envshade_case(bsdf, n)
        files[f"ref_envshade_{bsdf}_n{n}.npz"] = {**d, **run_envshade(d)}
    d = envshade_case("pbr", 8, frame=(1, 64, 64), probe=(64, 128))      # the benchmarked sample count on a larger frame and a finer probe
    files["ref_envshade_pbr_n8_64x64.npz"] = {**d, **run_envshade(d)}
    d = envshade_case("pbr", 4, frame=(2, 32, 32), occluder=True)          # half of the covered pixels under an occluder: shadow rays that HIT
    files["ref_envshade_pbr_n4_occluder.npz"] = {**d, **run_envshade(d)}
    img, tgt = image_loss_case()
    files["ref_image_loss.npz"] = dict(img=img, target=tgt, **run_image_loss(img, tgt))
    t = shading_normal_case()
    files["ref_shading_normal.npz"] = {**t, **run_shading_normal(t)}
    t = xfm_case()
    files["ref_xfm_points.npz"] = {**t, **run_xfm(t)}
    t = bilateral_case()
    files["ref_bilateral.npz"] = {**t, **run_bilateral(t)}
    return files


def main():
    for name, arrays in all_goldens().items():
        np.savez_compressed(os.path.join(OUT, name), **arrays)
        print(f"wrote {name}: {os.path.getsize(os.path.join(OUT, name)) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
