"""Device time of the torch-issued (non gs_*) kernels of a training iteration, attributed to code regions:
forward ops by the enclosing wrapped function, backward ops through the autograd sequence number of the forward op
they differentiate.  Run on the GPU box:  python tools/torch_kernel_regions.py"""
import collections
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function

from gshell_amd import workload
from gshell_amd.geometry import gshell_tets_geometry as G
from gshell_amd.geometry import mlp as M
from gshell_amd.render import light, mlptexture, regularizer, render, renderutils


def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    lab = "R:" + (label or name)

    @functools.wraps(fn)
    def inner(*a, **k):
        with record_function(lab):
            return fn(*a, **k)
    setattr(mod, name, inner)


for name in ("shade", "render_layer", "_sample_texture", "interpolate"):
    wrap(render, name)
for name in ("shading_loss", "material_smoothness_grad", "chroma_loss"):
    wrap(regularizer, name)
wrap(G, "compute_sdf_reg_loss")
wrap(G, "sample_points")
wrap(G, "eikonal_sq_sum")
wrap(M, "row_sparse_backward")
wrap(G.GShellTetsGeometry, "getMesh")
wrap(G.GShellTetsGeometry, "render", "geometry.render")
wrap(G.GShellTetsGeometry, "tick")
wrap(M.MLP, "forward", "MLP.forward(torch)")
wrap(mlptexture.MLPTexture3D, "sample", "MLPTexture3D.sample")
wrap(light.EnvironmentLight, "update_pdf")
wrap(torch.optim.Adam, "step", "Adam.step")
wrap(renderutils, "image_loss")

ITERS = 3
GEOM = sys.argv[1] if len(sys.argv) > 1 else "tets"
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if GEOM == "flexicubes":
    from gshell_amd.geometry import gshell_flexicubes as FC
    from gshell_amd.geometry import gshell_flexicubes_geometry as FG
    wrap(FG.GShellFlexiCubesGeometry, "getMesh")
    wrap(FG.GShellFlexiCubesGeometry, "render", "geometry.render")
    wrap(FG.GShellFlexiCubesGeometry, "tick")
    wrap(FC.GShellFlexiCubes, "__call__", "GShellFlexiCubes.__call__")
tr = workload.build(res=RES, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200, geometry=GEOM)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
tr.it = 1000          # steady-state schedule (sigma 2, shadow_scale 1), as bench.py
for _ in range(3):
    tr.step(tg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(ITERS):
        tr.step(tg)
    torch.cuda.synchronize()

evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]


def label_of(e):
    labs = []
    p = e
    while p is not None:
        if p.name.startswith("R:"):
            labs.append(p.name[2:])
        p = p.cpu_parent
    return "/".join(reversed(labs)) if labs else None


seq_label = {}
for e in evs:
    if e.sequence_nr is not None and e.sequence_nr >= 0 and not e.name.startswith("autograd::engine"):
        lab = label_of(e)
        if lab and e.sequence_nr not in seq_label:
            seq_label[e.sequence_nr] = lab


def bwd_label(e):
    p = e
    while p is not None:
        if p.name.startswith("autograd::engine::evaluate_function"):
            return "bwd of " + seq_label.get(p.sequence_nr, "<unlabelled> " + p.name.split(": ")[-1])
        p = p.cpu_parent
    return None


agg = collections.defaultdict(lambda: [0.0, 0])
ops = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
kern = collections.defaultdict(lambda: collections.defaultdict(float))
tot = 0.0
for e in evs:
    dt = e.self_device_time_total or 0.0
    if dt <= 0:
        continue
    lab = label_of(e) or bwd_label(e) or "<top level> " + e.name
    agg[lab][0] += dt
    agg[lab][1] += 1
    tot += dt
    ops[lab][e.name][0] += dt
    ops[lab][e.name][1] += 1
    for k in getattr(e, "kernels", []):
        kern[lab][k.name[:70]] += k.duration
print(f"device time per iteration: {tot / ITERS / 1e3:.2f} ms")
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{t / ITERS / 1e3:8.3f} ms {n / ITERS:7.1f} launches  {k}")
    if t / ITERS / 1e3 > 1.0:
        for kn, kt in sorted(kern[k].items(), key=lambda kv: -kv[1])[:6]:
            print(f"            {kt / ITERS / 1e3:8.3f} ms  {kn}")

if os.environ.get("GS_REGIONS_DETAIL"):
    # which torch operators issue the launches of every region (for hunting the small-kernel glue)
    print("\noperators per region:")
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n / ITERS:7.1f} launches {t / ITERS / 1e3:8.3f} ms  {k}")
        for on, (ot, oc) in sorted(ops[k].items(), key=lambda kv: -kv[1][1]):
            print(f"        {oc / ITERS:6.1f} x {ot / ITERS / 1e3:7.3f} ms  {on}")

# torch / rocPRIM / runtime kernels (everything that is not a hand-written kernel of this library), by total time
allk = collections.defaultdict(lambda: [0.0, 0])
for e in evs:
    for k in getattr(e, "kernels", []):
        hand = "anonymous namespace)::k_" in k.name or k.name.startswith("(anonymous namespace)::k_")
        if not hand:
            allk[k.name[:110]][0] += k.duration
            allk[k.name[:110]][1] += 1
tot_other = sum(v[0] for v in allk.values())
print(f"\nnot hand-written (ATen / rocPRIM / memcpy) kernels: {tot_other / ITERS / 1e3:.2f} ms per iteration in {sum(v[1] for v in allk.values()) / ITERS:.0f} launches")
for k, (t, n) in sorted(allk.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{t / ITERS / 1e3:8.3f} ms {n / ITERS:6.1f} x  {k}")
