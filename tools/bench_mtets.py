"""Micro-benchmark of the G-MarchingTets extraction alone (fwd + bwd) on one GPU."""
import argparse
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gshell_amd import grid
from gshell_amd.geometry.gshell_tets import GShell_Tets

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=104)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--tangents", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
verts, tets = grid.bcc_grid(a.cells, device=dev)
r = torch.sqrt(verts[:, 0] ** 2 + verts[:, 2] ** 2)
sdf = torch.minimum(0.26 - 0.18 * verts[:, 1] - r, 0.36 - verts[:, 1].abs()).requires_grad_(True)
msdf = (0.12 - verts[:, 1] + 0.05 * torch.sin(8.0 * verts[:, 0])).requires_grad_(True)
pos = verts.clone().requires_grad_(True)
ext = GShell_Tets(compute_tangents=bool(a.tangents))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
t0.record(); topo = ext.topology(tets, verts.shape[0]); t1.record(); torch.cuda.synchronize()
print("topology build ms", t0.elapsed_time(t1), "E", topo.E)
def step():
    v, f, _, _, tng, extra = ext(pos, sdf, msdf, tets)
    gv = torch.ones_like(v); gm = torch.ones_like(extra["msdf"])
    torch.autograd.backward([v, extra["msdf"]], [gv, gm])
    return v, f
for _ in range(3): v, f = step()
torch.cuda.synchronize()
fw = []; 
for _ in range(a.steps):
    t0.record(); v, f, _, _, _, ex = ext(pos, sdf, msdf, tets); t1.record(); torch.cuda.synchronize(); fw.append(t0.elapsed_time(t1))
tot = []
for _ in range(a.steps):
    t0.record(); step(); t1.record(); torch.cuda.synchronize(); tot.append(t0.elapsed_time(t1))
N, F = verts.shape[0], tets.shape[0]
alg = 16 * F + 20 * N + 20 * v.shape[0] + 12 * f.shape[0]
print(json.dumps(dict(N=N, F=F, V_aug=v.shape[0], T=f.shape[0], fwd_ms=min(fw), fwd_bwd_ms=min(tot),
                      fwd_alg_GBps=alg / (min(fw) * 1e-3) / 1e9)))
