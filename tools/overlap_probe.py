"""Do the shadow-ray traversal (VALU-issue bound) and the SDF gradient chain (HBM-plane / MFMA bound) overlap when they are launched on
two streams?  Times A = stand-alone any-hit over the bench mesh, B = eikonal forward + backward + weight gradients over 50 000 samples,
A then B on one stream, and A || B on two streams.  GPU box.
usage: python tools/overlap_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import workload
from gshell_amd.geometry.gshell_tets_geometry import sample_points
from gshell_amd.geometry.mlp import eikonal_sq_sum
from gshell_amd.render import optixutils as ou

tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
ctx = tr.geometry.optix_ctx
net = tr.geometry.sdf_net
g = torch.Generator(device="cuda").manual_seed(0)
P, S = 150000, 128
pts, fidx = sample_points(m.v_pos.detach(), m.t_pos_idx, P)
pts = pts.reshape(-1, 3)
f = m.t_pos_idx[fidx.reshape(-1)]
v0, v1, v2 = m.v_pos[f[:, 0]], m.v_pos[f[:, 1]], m.v_pos[f[:, 2]]
n = torch.nn.functional.normalize(torch.linalg.cross(v1 - v0, v2 - v0), dim=1)
u1, u2 = torch.rand(P, S, device="cuda", generator=g), torch.rand(P, S, device="cuda", generator=g)
r, phi = u1.sqrt(), 2 * torch.pi * u2
a = torch.where(n[:, 0:1].abs() > 0.9, torch.tensor([0.0, 1.0, 0.0], device="cuda"), torch.tensor([1.0, 0.0, 0.0], device="cuda"))
t = torch.nn.functional.normalize(torch.linalg.cross(n, a.expand_as(n)), dim=1)
b = torch.linalg.cross(n, t)
d = (r * phi.cos())[..., None] * t[:, None] + (r * phi.sin())[..., None] * b[:, None] + (1 - u1).clamp_min(0).sqrt()[..., None] * n[:, None]
o = (pts + 1e-3 * n)[:, None].expand(P, S, 3)
o, d = o.reshape(-1, 3).contiguous().detach(), d.reshape(-1, 3).contiguous().detach()
epts = pts[:50000].detach().contiguous()
side = torch.cuda.Stream()


def rays():
    ou.any_hit(ctx, o, d)


def chain():
    eikonal_sq_sum(net, epts).backward()


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both_serial():
    rays()
    chain()


def both_parallel():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain()
    rays()
    main.wait_stream(side)


ta, tb = timed(rays), timed(chain)
print(f"rays {ta:.3f} ms   chain {tb:.3f} ms   sum {ta + tb:.3f}")
print(f"one stream  {timed(both_serial):.3f} ms")
print(f"two streams {timed(both_parallel):.3f} ms")
