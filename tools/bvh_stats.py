"""Shadow-ray traversal statistics on the bench mesh: nodes / triangles per ray, hit fraction, wave divergence,
and stand-alone any-hit throughput.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int64, check, ptr, stream
from gshell_amd.geometry.gshell_tets_geometry import sample_points
from gshell_amd.render import optixutils as ou

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = workload.build(res=res, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
ctx = tr.geometry.optix_ctx
T = m.t_pos_idx.shape[0]
info = [c_int64() for _ in range(4)]
import ctypes
check(_lib.lib().gs_bvh_info(ctx.handle, *[ctypes.byref(x) for x in info]))
print(f"T={T} depth={info[1].value} leaf={info[2].value} bytes={info[3].value / 1e6:.1f} MB")
g = torch.Generator(device="cuda").manual_seed(0)
P, S = 150000, 128
pts, fidx = sample_points(m.v_pos.detach(), m.t_pos_idx, P)
pts = pts.reshape(-1, 3)
f = m.t_pos_idx[fidx.reshape(-1)]
v0, v1, v2 = m.v_pos[f[:, 0]], m.v_pos[f[:, 1]], m.v_pos[f[:, 2]]
n = torch.nn.functional.normalize(torch.linalg.cross(v1 - v0, v2 - v0), dim=1)
# cosine-weighted hemisphere around n (two-sided shading flips n towards a random viewer; keep +n)
u1, u2 = torch.rand(P, S, device="cuda", generator=g), torch.rand(P, S, device="cuda", generator=g)
r, phi = u1.sqrt(), 2 * torch.pi * u2
a = torch.where(n[:, 0:1].abs() > 0.9, torch.tensor([0.0, 1.0, 0.0], device="cuda"), torch.tensor([1.0, 0.0, 0.0], device="cuda"))
t = torch.nn.functional.normalize(torch.linalg.cross(n, a.expand_as(n)), dim=1)
b = torch.linalg.cross(n, t)
d = (r * phi.cos())[..., None] * t[:, None] + (r * phi.sin())[..., None] * b[:, None] + (1 - u1).clamp_min(0).sqrt()[..., None] * n[:, None]
o = (pts + 1e-3 * n)[:, None].expand(P, S, 3)
o, d = o.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()
N = o.shape[0]
hit = torch.empty(N, dtype=torch.uint8, device="cuda")
stats = torch.empty(N, 2, dtype=torch.int32, device="cuda")
check(_lib.lib().gs_bvh_any_hit_stats(ctx.handle, ptr(o), ptr(d), c_int64(N), ptr(hit), ptr(stats), stream()))
torch.cuda.synchronize()
nodes, tris = stats[:, 0].float(), stats[:, 1].float()
wn = nodes.reshape(-1, 64)
print(f"rays={N / 1e6:.1f} M  hit={hit.float().mean():.3f}  nodes/ray={nodes.mean():.1f}  tris/ray={tris.mean():.1f}")
print(f"  miss rays: nodes={nodes[hit == 0].mean():.1f} tris={tris[hit == 0].mean():.1f};  hit rays: nodes={nodes[hit == 1].mean():.1f} tris={tris[hit == 1].mean():.1f}")
print(f"  wave max nodes (mean over waves)={wn.max(dim=1).values.mean():.1f}  -> SIMD efficiency {nodes.mean() / wn.max(dim=1).values.mean():.2f}")
for _ in range(2):
    ou.any_hit(ctx, o, d)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ou.any_hit(ctx, o, d)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"any_hit: {ms:.2f} ms for {N / 1e6:.1f} M rays = {N / ms / 1e6:.2f} G rays/s")
