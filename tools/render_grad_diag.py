"""Where does the end-to-end v_pos gradient of render_mesh differ from the CPU pipeline oracle?  (GPU box)
    python tools/render_grad_diag.py
Back-propagates one output buffer at a time through both sides and prints, per buffer: relative L2 error of the v_pos gradient,
the share of the squared error carried by the 10 worst vertices, and the error after removing them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_render_gpu import render_both  # noqa: E402


def main():
    for denoise, levels in ((False, 16), (True, 16), (False, 6), (True, 6)):
        out, ref, leaves, (B, H, W) = render_both(denoise, levels)
        vd, v_ref = leaves["v_pos"]
        gen = torch.Generator().manual_seed(9)
        for key in ("shaded", "msdf_image", "kd_grad", "normal", "occlusion" if "occlusion" in ref else "shaded"):
            w = torch.rand(ref[key].shape, generator=gen)
            ga, = torch.autograd.grad((out[key] * w.to(out[key].device)).sum(), vd, retain_graph=True)
            gb, = torch.autograd.grad((ref[key] * w).sum(), v_ref, retain_graph=True)
            ga = ga.cpu()
            err = (ga - gb).square().sum(-1)
            tot = float(err.sum())
            top = torch.topk(err, 10)
            rel = (tot ** 0.5) / float(gb.norm())
            rest = ((tot - float(top.values.sum())) ** 0.5) / float(gb.norm())
            print(f"denoise={denoise} levels={levels} buffer {key:12s}: v_pos grad rel L2 {rel:.2e}; 10 worst vertices carry {float(top.values.sum()) / max(tot, 1e-30):.2f} of the squared error; "
                  f"without them {rest:.2e}; worst vertex |err| {float(top.values[0]) ** 0.5:.3e} of |g|max {float(gb.norm(dim=-1).max()):.3e}")


if __name__ == "__main__":
    main()
