"""k_h1_fwd of the library in GSHELL_HIP_LIB against a saved output of another build (bit equality): python tools/h1_compare.py save|check <file>.  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, grid
from gshell_amd.geometry import mlp
from gshell_amd.geometry.mlp import MLP

torch.manual_seed(0)
verts, _ = grid.grid_for_res(128, device="cuda")
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
L = _lib.lib()
with torch.no_grad():
    packed, n_hidden, skip = mlp.pack_weights_h2(net)
    y = torch.empty(verts.shape[0], device="cuda")
    occ = torch.zeros((verts.shape[0] + 63) // 64, dtype=torch.int64, device="cuda")
    _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(verts), _lib.c_int64(verts.shape[0]), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip),
                                   _lib.ptr(y), _lib.ptr(occ), _lib.c_void_p(0), _lib.stream()))
    torch.cuda.synchronize()
if sys.argv[1] == "save":
    torch.save((y.cpu(), occ.cpu()), sys.argv[2])
    print("saved", float(y.abs().mean()))
else:
    y0, o0 = torch.load(sys.argv[2])
    print("values bit-equal:", bool(torch.equal(y.cpu(), y0)), " sign words equal:", bool(torch.equal(occ.cpu(), o0)), " max |diff|", float((y.cpu() - y0).abs().max()))
