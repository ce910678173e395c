"""Known-answer tests pinning the hash-grid ORACLE (no reference implementation or vectors exist for it)."""
import numpy as np
import torch

from oracle import hashgrid_oracle as ho

CFG = (16, 2, 19, 16, float(np.exp(np.log(4096 / 16) / 15)))


def test_level_layout_matches_instant_ngp_table_sizes():
    metas, total = ho.level_meta(*CFG)
    assert metas[0][1] == 16 and abs(metas[0][0] - 15.0) < 1e-5            # base level: scale 15, res 16
    assert metas[-1][1] == 4096                                            # finest level reaches the desired resolution
    sizes = [m[3] for m in metas]
    assert sizes[0] == 4096 and all(s <= 2 ** 19 for s in sizes) and sizes[-1] == 2 ** 19
    assert total == sum(sizes) * 2


def test_dense_level_is_trilinear_interpolation_of_its_grid():
    # single dense level: params laid out x-fastest; encoding at a grid vertex returns that vertex' features
    cfg = (1, 2, 19, 4, 1.0)
    metas, total = ho.level_meta(*cfg)
    scale, res, off, size = metas[0]                                       # scale 3, res 4
    params = torch.arange(total, dtype=torch.float32)
    v = torch.tensor([[1, 2, 3], [0, 0, 0], [2, 1, 0]])
    x = (v.float() - 0.5) / scale + 1e-6                                   # pos = x*scale + 0.5 == vertex coordinate
    out = ho.encode(x, params, *cfg)
    idx = v[:, 0] + v[:, 1] * res + v[:, 2] * res * res
    assert torch.allclose(out, torch.stack([params[idx * 2], params[idx * 2 + 1]], -1), atol=1e-3)
    # midpoint of an x-edge: average of the two end vertices
    xm = torch.tensor([[(1.5 - 0.5) / scale, (2 - 0.5) / scale + 1e-6, (3 - 0.5) / scale + 1e-6]])
    om = ho.encode(xm, params, *cfg)
    i0, i1 = 1 + 2 * res + 3 * res * res, 2 + 2 * res + 3 * res * res
    assert torch.allclose(om[0, 0], (params[i0 * 2] + params[i1 * 2]) / 2, atol=1e-2)


def test_partition_of_unity_and_gradients():
    metas, total = ho.level_meta(*CFG)
    x = torch.rand(64, 3, generator=torch.Generator().manual_seed(0)).requires_grad_(True)
    out = ho.encode(x, torch.ones(total), *CFG)
    assert torch.allclose(out, torch.ones_like(out), atol=1e-5)            # weights of the 8 corners sum to 1
    p = (torch.rand(total, generator=torch.Generator().manual_seed(1)) - 0.5).requires_grad_(True)
    ho.encode(x, p, *CFG).square().sum().backward()
    assert torch.isfinite(x.grad).all() and x.grad.abs().sum() > 0 and p.grad.abs().sum() > 0
