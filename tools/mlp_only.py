"""Runs only the fused SDF-MLP forward on the bench grid (for rocprofv3 --pmc passes).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import grid
from gshell_amd.geometry.mlp import MLP, fused_forward

torch.manual_seed(0)
verts, _ = grid.grid_for_res(256, device="cuda")
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        y = fused_forward(net, verts)
torch.cuda.synchronize()
print(float(y.abs().mean()))
