"""Where does the G-FlexiCubes extraction's gradient w.r.t. sdf / positions differ from the oracle's?  (GPU box)
    python tools/flexi_grad_diag.py [res]
Same inputs on both sides (the chain fixture's state, the float32 oracle's SDF values), one loss term at a time."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import chain_recipe as cr, flexi_oracle as fo, mlp_oracle as mlp  # noqa: E402


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    sc = cr.inputs('flexi32' if res == 32 else 'flexi80')
    from gshell_amd.geometry.gshell_flexicubes import GShellFlexiCubes
    v_def = sc['verts'] + sc['max_displacement'] * sc['deform']
    sdf = mlp.forward_chunked(sc['sdf_net'], v_def).reshape(-1)
    w = sc['cube_w']
    gen = torch.Generator().manual_seed(3)
    ext = GShellFlexiCubes()
    terms = {}
    for dt in (torch.float32, torch.float64):
        x, s, nu, ww = (t.to(dt).clone().requires_grad_(True) for t in (v_def, sdf, sc['msdf'], w))
        fv, ff, L, ex = fo.extract(x, s[:, None], nu, sc['indices'], res, ww[:, :12], ww[:, 12:20], ww[:, 20])
        if dt == torch.float32:
            wv = torch.randn(fv.shape, generator=gen)
            wm = torch.randn(ex['msdf'].shape, generator=gen)
        for name, loss in (("L_dev", L.mean() * 0.25), ("verts", (fv * wv.to(dt)).sum()), ("msdf", (ex['msdf'].reshape(-1) * wm.to(dt).reshape(-1)).sum())):
            gx, gs, gn, gw = torch.autograd.grad(loss, (x, s, nu, ww), retain_graph=True, allow_unused=True)
            terms[(name, dt)] = (gx, gs, gn, gw)
    x, s, nu, ww = (t.cuda().clone().requires_grad_(True) for t in (v_def, sdf, sc['msdf'], w))
    out = ext(x, s[:, None], nu, sc['indices'].cuda(), res, ww[:, :12], ww[:, 12:20], ww[:, 20])
    fv, ff, L, ex = out
    assert torch.equal(ff.cpu(), terms and fo.extract(v_def, sdf[:, None], sc['msdf'], sc['indices'], res, w[:, :12], w[:, 12:20], w[:, 20])[1])
    for name, loss in (("L_dev", L.mean() * 0.25), ("verts", (fv * wv.cuda()).sum()), ("msdf", (ex['msdf'].reshape(-1) * wm.cuda().reshape(-1)).sum())):
        g = torch.autograd.grad(loss, (x, s, nu, ww), retain_graph=True, allow_unused=True)
        for label, a, b32, b64 in zip(("x", "s", "nu", "w"), g, terms[(name, torch.float32)], terms[(name, torch.float64)]):
            if a is None or b64 is None or float(b64.norm()) == 0:
                continue
            a = a.cpu().double()
            e_h, e_32 = float((a - b64).norm() / b64.norm()), float((b32.double() - b64).norm() / b64.norm())
            print(f"loss {name:6s} d/d{label:2s}: HIP vs float64 {e_h:.2e}; float32 oracle vs float64 {e_32:.2e}")
            if e_h > 4 * max(e_32, 1e-5):
                err = (a - b64).reshape(a.shape[0], -1).norm(dim=-1)
                top = torch.topk(err, 8)
                print("    worst rows:", [(int(i), f"{float(e):.2e}", f"g64 {float(b64.reshape(a.shape[0], -1)[i].norm()):.2e}", f"hip {float(a.reshape(a.shape[0], -1)[i].norm()):.2e}") for e, i in zip(top.values, top.indices)])
                print(f"    share of the squared error in the 8 worst rows: {float(top.values.square().sum() / err.square().sum()):.2f}; rows with error > 1e-3 |g|max: {int((err > 1e-3 * float(b64.abs().max())).sum())}")


if __name__ == "__main__":
    main()
