"""Would 16-bit D planes (the reverse chain's per-layer deltas, which k_h2_bwd writes and the weight-gradient kernel reads back) meet the gradient bars?
CPU emulation in float64: the exact deltas of the reference network's backward pass are rounded to bf16 / fp16 (fp16 with one
power-of-two scale per row, what the kernels' row scaling could supply) before the weight-gradient products dW_l = D_l^T A_l, everything else exact.
Prints the relative L2 error of every dW_l for the row counts of tests/test_mlp_grad_gpu.py and of the training iteration.
    python tools/d_plane_precision.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import chain_recipe as cr, mlp_oracle as mlp  # noqa: E402


def deltas(state, x, gy):
    """activations A_l (inputs of every Linear) and deltas D_l (d loss / d pre-activation of every Linear), float64"""
    emb = mlp.embed(x)
    ids = mlp.layer_keys(state)
    A, Z = [], []
    h = emb
    for j, lid in enumerate(ids):
        if j - 1 in (3,) and j + 1 < len(ids):
            h = torch.cat([h, emb], -1)
        A.append(h)
        z = torch.nn.functional.linear(h, state[f'net.{lid}.weight'], state[f'net.{lid}.bias'])
        Z.append(z)
        h = mlp.softplus100(z) if j + 1 < len(ids) else z
    D = [None] * len(ids)
    g = gy
    for j in reversed(range(len(ids))):
        D[j] = g                                              # d / d z_j
        if j > 0:
            gh = g @ state[f'net.{ids[j]}.weight']           # d / d input of layer j
            gh = gh[:, :256]                                  # (the encoding columns of the skip layer carry no parameter gradient upstream)
            g = gh * torch.sigmoid(100.0 * Z[j - 1])          # softplus'(z) = sigmoid(beta z)
    return A, D


def round_rows(D, kind):
    if kind == "bf16":
        return D.float().bfloat16().double()
    # fp16 with one power-of-two scale per row (max |d| of the row -> [0.5, 1))
    m = D.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    s = torch.pow(2.0, -torch.floor(torch.log2(m)) - 1)
    return (D * s).float().half().double() / s


def main():
    state = {k: v.double() for k, v in cr.load_net().items()}
    g = torch.Generator().manual_seed(0)
    for n in (16, 300, 1007, 5000, 78000):
        x = (torch.rand(n, 3, generator=g, dtype=torch.float64) - 0.5) * 1.4
        # upstream gradients spanning six decades, as in the test and in training (means over 10^6 pixels)
        gy = (torch.randn(n, 1, generator=g, dtype=torch.float64) * torch.pow(10.0, torch.rand(n, 1, generator=g, dtype=torch.float64) * 6 - 8))
        A, D = deltas(state, x, gy)
        line = []
        for kind in ("bf16", "fp16-row-scaled"):
            worst = 0.0
            for a, d in zip(A, D):
                exact = d.t() @ a
                approx = round_rows(d, kind).t() @ a
                worst = max(worst, float((approx - exact).norm() / exact.norm()))
            line.append(f"{kind}: worst layer {worst:.1e}")
        print(f"rows {n:6d}: " + "; ".join(line) + "   (bar of tests/test_mlp_grad_gpu.py: 1e-4; eikonal bar 5e-6 class)")


if __name__ == "__main__":
    main()
