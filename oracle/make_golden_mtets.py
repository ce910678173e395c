"""Mint golden fixtures for G-MarchingTets by running the REAL reference
(/root/reference/geometry/gshell_tets.py:245-443) on CPU.

Run in the build container only:   python -B oracle/make_golden_mtets.py
Writes tests/golden/mtets_*.npz (committed). TEST INFRASTRUCTURE.
"""
import os
import sys
import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refload, fields          # noqa: E402
from gshell_amd import grid                 # noqa: E402

CASES = [
    # name, grid, sdf kind, msdf kind, seed, sdf zeros, msdf zeros, deform
    ("bcc8_sphere_rand",        ("bcc", 8),   "sphere",       "rand",      0, 0, 0, True),
    ("bcc8_noise_wavy_zeros",   ("bcc", 8),   "sphere_noise", "wavy",      1, 40, 40, True),
    ("bcc8_sphere_positive",    ("bcc", 8),   "sphere",       "positive",  2, 0, 0, False),
    ("bcc8_sphere_negative",    ("bcc", 8),   "sphere",       "negative",  3, 0, 0, False),
    ("kuhn8_two_halfspace",     ("kuhn", 8),  "two_spheres",  "halfspace", 4, 0, 0, True),
    ("kuhn6_plane_rand_zeros",  ("kuhn", 6),  "plane",        "rand",      5, 25, 25, False),
    ("bcc12_skirt_wavy",        ("bcc", 12),  "skirt",        "wavy",      6, 0, 0, True),
    ("bcc26_skirt_wavy",        ("bcc", 26),  "skirt",        "wavy",      7, 0, 0, True),
]


# output_watertight_template=False (ref :260-263, :436-441): the mSDF pre-filter of the valid tets -> tests/golden/mtetsnowt_*.npz
NOWT_CASES = [
    ("bcc8_sphere_rand",        ("bcc", 8),   "sphere",       "rand",      0, 0, 0, True),
    ("bcc8_noise_wavy_zeros",   ("bcc", 8),   "sphere_noise", "wavy",      1, 40, 40, True),
    ("bcc8_sphere_negative",    ("bcc", 8),   "sphere",       "negative",  3, 0, 0, False),
    ("kuhn8_two_halfspace",     ("kuhn", 8),  "two_spheres",  "halfspace", 4, 0, 0, True),
    ("bcc12_skirt_wavy",        ("bcc", 12),  "skirt",        "wavy",      6, 0, 0, True),
]


def make_inputs(gspec, sdf_kind, msdf_kind, seed, zs, zm, deform):
    kind, n = gspec
    verts, tets = (grid.bcc_grid(n) if kind == "bcc" else grid.kuhn_grid(n))
    verts = verts.numpy()
    if deform:
        verts = verts + fields.make_deform(verts, 1.0 / n, seed)
    sdf = fields.make_sdf(verts, sdf_kind, seed, zs)
    msdf = fields.make_msdf(verts, msdf_kind, seed, zm)
    return verts.astype(np.float32), tets.numpy(), sdf, msdf


def run_reference(ref, verts, tets, sdf, msdf, seed):
    pos = torch.tensor(verts, requires_grad=True)
    s = torch.tensor(sdf, requires_grad=True)
    m = torch.tensor(msdf, requires_grad=True)
    t = torch.tensor(tets, dtype=torch.long)
    with refload.CudaToCpu():
        ext = ref.GShell_Tets()
        v_aug, f_aug, _u, _ui, tng_aug, extra = ext(pos, s, m, t)
    wv, wm, ww = fields.loss_weights(v_aug.shape[0], extra["vertices_watertight"].shape[0], seed)
    loss = (v_aug * torch.tensor(wv)).sum() + (extra["msdf"] * torch.tensor(wm)).sum() \
        + (extra["vertices_watertight"] * torch.tensor(ww)).sum()
    loss.backward()
    out = dict(
        verts_aug=v_aug.detach().numpy(), faces_aug=f_aug.numpy().astype(np.int32),
        v_tng_aug=tng_aug.detach().numpy(),
        n_verts_watertight=np.int64(extra["n_verts_watertight"]),
        vertices_watertight=extra["vertices_watertight"].detach().numpy(),
        faces_watertight=extra["faces_watertight"].numpy().astype(np.int32),
        v_tng_watertight=extra["v_tng_watertight"].detach().numpy(),
        msdf=extra["msdf"].detach().numpy(),
        msdf_watertight=extra["msdf_watertight"].detach().numpy(),
        msdf_boundary=extra["msdf_boundary"].detach().numpy(),
        grad_pos=pos.grad.numpy(), grad_sdf=s.grad.numpy(), grad_msdf=m.grad.numpy(),
    )
    return out


def run_reference_nowt(ref, verts, tets, sdf, msdf, seed):
    pos = torch.tensor(verts, requires_grad=True)
    s = torch.tensor(sdf, requires_grad=True)
    m = torch.tensor(msdf, requires_grad=True)
    t = torch.tensor(tets, dtype=torch.long)
    with refload.CudaToCpu():
        ext = ref.GShell_Tets()
        v_aug, f_aug, _u, _ui, tng_aug, extra = ext(pos, s, m, t, output_watertight_template=False)
    assert sorted(extra.keys()) == ["msdf", "msdf_boundary", "msdf_watertight"], sorted(extra.keys())
    wv, wm, _ = fields.loss_weights(v_aug.shape[0], extra["msdf_watertight"].shape[0], seed)
    loss = (v_aug * torch.tensor(wv)).sum() + (extra["msdf"] * torch.tensor(wm)).sum()
    if loss.requires_grad:
        loss.backward()
    z = lambda t_, like: (t_.grad.numpy() if t_.grad is not None else np.zeros_like(like))
    return dict(verts_aug=v_aug.detach().numpy(), faces_aug=f_aug.numpy().astype(np.int32), v_tng_aug=tng_aug.detach().numpy(),
                msdf=extra["msdf"].detach().numpy(), msdf_watertight=extra["msdf_watertight"].detach().numpy(), msdf_boundary=extra["msdf_boundary"].detach().numpy(),
                grad_pos=z(pos, verts), grad_sdf=z(s, sdf), grad_msdf=z(m, msdf))


def main():
    assert refload.reference_available(), "needs /root/reference"
    ref = refload.load_gshell_tets()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for name, gspec, sk, mk, seed, zs, zm, deform in CASES:
        verts, tets, sdf, msdf = make_inputs(gspec, sk, mk, seed, zs, zm, deform)
        out = run_reference(ref, verts, tets, sdf, msdf, seed)
        meta = dict(grid_kind=gspec[0], grid_n=np.int64(gspec[1]), sdf_kind=sk, msdf_kind=mk,
                    seed=np.int64(seed), sdf_zeros=np.int64(zs), msdf_zeros=np.int64(zm),
                    deform=np.bool_(deform))
        if gspec[1] <= 12:   # small: keep the literal inputs too
            meta.update(in_verts=verts, in_tets=tets.astype(np.int32), in_sdf=sdf, in_msdf=msdf)
        np.savez_compressed(os.path.join(outdir, f"mtets_{name}.npz"), **meta, **out)
        print(f"{name}: N={verts.shape[0]} F={tets.shape[0]} V={int(out['n_verts_watertight'])} "
              f"V_aug={out['verts_aug'].shape[0]} T={out['faces_aug'].shape[0]}")
    for name, gspec, sk, mk, seed, zs, zm, deform in NOWT_CASES:
        verts, tets, sdf, msdf = make_inputs(gspec, sk, mk, seed, zs, zm, deform)
        out = run_reference_nowt(ref, verts, tets, sdf, msdf, seed)
        meta = dict(grid_kind=gspec[0], grid_n=np.int64(gspec[1]), sdf_kind=sk, msdf_kind=mk, seed=np.int64(seed), sdf_zeros=np.int64(zs), msdf_zeros=np.int64(zm),
                    deform=np.bool_(deform), in_verts=verts, in_tets=tets.astype(np.int32), in_sdf=sdf, in_msdf=msdf)
        np.savez_compressed(os.path.join(outdir, f"mtetsnowt_{name}.npz"), **meta, **out)
        print(f"nowt {name}: N={verts.shape[0]} F={tets.shape[0]} V={out['msdf_watertight'].shape[0]} V_aug={out['verts_aug'].shape[0]} T={out['faces_aug'].shape[0]}")


if __name__ == "__main__":
    main()
