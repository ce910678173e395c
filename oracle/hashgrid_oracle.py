"""CPU oracle of the multiresolution hash-grid encoding (TEST INFRASTRUCTURE -- checker only).

PARITY UNPINNED: the reference uses tiny-cuda-nn (third-party, unpinned git HEAD per reference README.md:39, not in
/root/reference; configured at render/mlptexture.py:57-73).  Restated from tiny-cuda-nn's published algorithm
(Mueller et al. 2022, "Instant NGP", and the library's documented grid.h behaviour): see gshell_amd/csrc/hashgrid.hip
header for the formulas.  fp32 throughout.  Known answers pin this restatement in tests/test_oracle_hashgrid.py."""
import math

import numpy as np
import torch


def level_meta(n_levels, F, log2_T, base_res, per_level_scale):
    metas, off = [], 0
    for l in range(n_levels):
        pls32 = float(np.float32(per_level_scale))      # the C ABI takes per_level_scale as float
        scale = np.float32(2.0 ** (l * math.log2(pls32)) * base_res - 1.0)
        res = int(math.ceil(float(scale))) + 1
        n = min((res ** 3 + 7) // 8 * 8, 1 << log2_T)
        metas.append((float(scale), res, off, n))
        off += n
    return metas, off * F


def encode(x, params, n_levels, F, log2_T, base_res, per_level_scale):
    """x [N,3] in [0,1] (torch, may require grad), params [n] -> [N, n_levels*F]."""
    metas, total = level_meta(n_levels, F, log2_T, base_res, per_level_scale)
    assert params.numel() == total
    outs = []
    for scale, res, off, size in metas:
        pos = x * scale + 0.5
        fl = torch.floor(pos)
        w = pos - fl
        g0 = fl.long()
        acc = 0
        for c in range(8):
            d = torch.tensor([c & 1, (c >> 1) & 1, (c >> 2) & 1])
            g = (g0 + d) & 0xFFFFFFFF
            if res ** 3 <= size:
                idx = (g[:, 0] + g[:, 1] * res + g[:, 2] * res * res) & 0xFFFFFFFF
            else:
                idx = (g[:, 0] ^ ((g[:, 1] * 2654435761) & 0xFFFFFFFF) ^ ((g[:, 2] * 805459861) & 0xFFFFFFFF)) & 0xFFFFFFFF
            idx = idx % size
            wgt = torch.where(d.bool(), w, 1 - w).prod(-1, keepdim=True)
            acc = acc + wgt * params[(off + idx)[:, None] * F + torch.arange(F)[None]]
        outs.append(acc)
    return torch.cat(outs, -1)
