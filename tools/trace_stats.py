"""Diagnostic: BVH traversal statistics of shadow rays on the benchmark mesh (nodes visited / triangles tested per ray,
rays/s of the stand-alone any-hit kernel)."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd.render import optixutils as ou

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--rays", type=int, default=4_000_000)
a = ap.parse_args()
tr = workload.build(res=a.res, n_samples=8, batch=1, train_res=(64, 64), fit_steps=400)
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
v, f = m.v_pos.detach(), m.t_pos_idx
print("mesh", v.shape[0], f.shape[0], tr.geometry.optix_ctx.info())
g = torch.Generator(device="cuda").manual_seed(0)
fid = torch.randint(0, f.shape[0], (a.rays,), device="cuda", generator=g)
w = torch.rand(a.rays, 3, device="cuda", generator=g) + 0.05
w = w / w.sum(-1, keepdim=True)
p = (v[f[fid]] * w[..., None]).sum(1)
n = torch.linalg.cross(v[f[fid, 1]] - v[f[fid, 0]], v[f[fid, 2]] - v[f[fid, 0]])
n = n / n.norm(dim=-1, keepdim=True).clamp_min(1e-20)
d = torch.randn(a.rays, 3, device="cuda", generator=g)
d = d / d.norm(dim=-1, keepdim=True)
flip = (d * n).sum(-1, keepdim=True) < 0
d = torch.where(flip, -d, d)                       # hemisphere around the normal, like the shader's samples
o = (p + n * 1e-3).contiguous()
d = d.contiguous()
L = _lib.lib()
hit = torch.empty(a.rays, dtype=torch.uint8, device="cuda")
stats = torch.empty(a.rays, 2, dtype=torch.int32, device="cuda")
_lib.check(L.gs_bvh_any_hit_stats(tr.geometry.optix_ctx.handle, _lib.ptr(o), _lib.ptr(d), _lib.c_int64(a.rays), _lib.ptr(hit), _lib.ptr(stats), _lib.stream()))
torch.cuda.synchronize()
s = stats.float()
print(f"occluded {hit.float().mean():.3f}  nodes/ray mean {s[:,0].mean():.1f} p50 {s[:,0].median():.0f} p99 {s[:,0].quantile(0.99):.0f} max {s[:,0].max():.0f}"
      f"  tris/ray mean {s[:,1].mean():.1f} p99 {s[:,1].quantile(0.99):.0f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    ou.any_hit(tr.geometry.optix_ctx, o, d)
e0.record()
for _ in range(5):
    ou.any_hit(tr.geometry.optix_ctx, o, d)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"any-hit: {ms:.2f} ms for {a.rays} rays = {a.rays / ms / 1e6:.2f} Grays/s")
