"""Times the SDF network's gradient passes on their bench sizes (HIP events around each C-ABI call): the row-sparse backward
over ~1.1e5 grid rows and the eikonal term over 50 000 samples (2e5 virtual rows).  GPU box.
usage: [GSHELL_HIP_LIB=...] python tools/chain_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib
from gshell_amd.geometry.mlp import MLP, eikonal_sq_sum, row_sparse_backward

torch.manual_seed(0)
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
N = 2282489
x = (torch.rand(N, 3, device="cuda") - 0.5).contiguous()
gy = torch.zeros(N, 1, device="cuda")
idx = torch.randperm(N, device="cuda")[:110000]
gy[idx, 0] = torch.randn(110000, device="cuda") * 1e-5
pts = (torch.rand(50000, 3, device="cuda") - 0.5).contiguous()


def once():
    row_sparse_backward(net, x, gy, True)
    eikonal_sq_sum(net, pts).backward()


for _ in range(2):
    once()
_lib.enable_op_timing(True)
_lib._timing["pending"].clear()
for _ in range(6):
    once()
torch.cuda.synchronize()
agg = {}
for name, e0, e1 in _lib._timing["pending"]:
    agg.setdefault(name, []).append(e0.elapsed_time(e1))
tot = 0.0
for name, v in agg.items():
    per_iter = sum(v) / 6
    tot += per_iter
    print(f"{name:28s} {per_iter:7.3f} ms / iteration in {len(v) // 6} calls ({' '.join('%.3f' % t for t in v[-(len(v) // 6):])})")
print(f"total {tot:.3f} ms")
