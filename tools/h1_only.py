"""Runs the one-product SDF forward kernel (k_h1_fwd) a few times on the bench grid (for rocprofv3 --pmc).  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, grid
from gshell_amd.geometry import mlp
from gshell_amd.geometry.mlp import MLP

torch.manual_seed(0)
verts, _ = grid.grid_for_res(256, device="cuda")
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
L = _lib.lib()
with torch.no_grad():
    packed, n_hidden, skip = mlp.pack_weights_h2(net)
    y = torch.empty(verts.shape[0], device="cuda")
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(verts), _lib.c_int64(verts.shape[0]), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip),
                                       _lib.ptr(y), _lib.c_void_p(0), _lib.c_void_p(0), _lib.stream()))
    torch.cuda.synchronize()
print(float(y.abs().mean()))
