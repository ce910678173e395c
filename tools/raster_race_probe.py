"""Does the rasteriser's result change when ANOTHER queue is busy?  (round 6: with the eikonal chain on a side stream the chain tests saw 16-pixel row
strips of the id image resolve to another triangle in ~1 of 4 runs.)  The rasteriser is deterministic (64-bit atomicMin of (depth, id) keys), so the
stand-alone result is the reference; the same launch is then repeated while a side stream runs (a) nothing, (b) the eikonal chain forward + backward,
(c) a 2 GB device copy, and every differing pixel is reported with its run structure.  GPU box.
usage: [GSHELL_HIP_LIB=gshell_amd/lib/variants/rastcN.so] python tools/raster_race_probe.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd.geometry.mlp import eikonal_sq_sum
from gshell_amd.render import rast as dr
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
print("library:", _lib.LIB_PATH, "build flags:", repr(_lib.lib().gs_build_flags().decode()))
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
tgt = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
net = tr.geometry.sdf_net
tri = m.faces_i32().contiguous()
v_pos = m.v_pos.detach().contiguous()
mvp = tgt['mvp']
epts = v_pos[torch.randperm(v_pos.shape[0], device="cuda")[:50000]].contiguous()
big_a = torch.empty(1 << 28, dtype=torch.int64, device="cuda")       # 2 GiB
big_b = torch.empty_like(big_a)
side = torch.cuda.Stream()
H = W = 512


def frame():
    with torch.no_grad():
        clip = ru.xfm_points(v_pos[None], mvp)
        r, db, vis = dr.rasterize(None, clip, tri, [H, W], return_visible=True)
    return r, vis


def chain():
    eikonal_sq_sum(net, epts).backward()


def copy():
    big_b.copy_(big_a)


ref_r, ref_vis = frame()
torch.cuda.synchronize()
r2, vis2 = frame()
torch.cuda.synchronize()
assert torch.equal(ref_r, r2) and torch.equal(ref_vis, vis2), "the stand-alone rasteriser is not deterministic"
ids_ref = ref_r[..., 3]
print(f"mesh: V={v_pos.shape[0]} T={tri.shape[0]}; covered pixels {int((ids_ref > 0).sum())}; visible triangles {int(ref_vis.sum())}")

for label, load in (("idle side stream", None), ("eikonal chain fwd + bwd on the side stream", chain), ("2 GiB device copy on the side stream", copy)):
    bad_frames, bad_px, lines = 0, 0, []
    main = torch.cuda.current_stream()
    for it in range(reps):
        if load is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                load()
                if load is chain and it % 2:
                    load()
        r, vis = frame()
        torch.cuda.synchronize()
        diff = (r[..., 3] != ids_ref)
        if bool(diff.any()) or not torch.equal(vis, ref_vis):
            bad_frames += 1
            idx = diff.nonzero()
            bad_px += idx.shape[0]
            if len(lines) < 6:
                b, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
                key = (b * H + y) * (W // 16) + x // 16
                strips, counts = torch.unique(key, return_counts=True)
                empties = int((r[..., 3][diff] == 0).sum())
                lines.append(f"    rep {it}: {idx.shape[0]} pixels differ in {strips.numel()} aligned 16-pixel strips (pixels per strip: {counts.tolist()[:12]}), "
                             f"{empties} of them now EMPTY; visible-flag differences {int((vis != ref_vis).sum())}; first: view {int(b[0])} y {int(y[0])} x {int(x[0])}; "
                             f"ids got {r[..., 3][diff][:6].int().tolist()} want {ids_ref[diff][:6].int().tolist()}; others bit-equal: {bool(torch.equal(r[~diff], ref_r[~diff]))}")
        for p in net.parameters():
            p.grad = None
    print(f"  {label}: {bad_frames} of {reps} frames differ from the stand-alone result ({bad_px} pixels)")
    for l in lines:
        print(l)
