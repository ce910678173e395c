"""Third probe of the two-queue rasteriser differences: WHAT does a sample that writes a wrong depth see?  Runs the GS_RAST_DEBUG build of raster.hip
(tools/build_variant.sh rastdbg raster.hip -DGS_RAST_DEBUG=1): k_rast_small logs, for every sample it issues, the pixel, the triangle, the depth it
computed and the operands it computed it from; the host recomputes each record from the clip-space vertices and prints the records that disagree.  GPU box.
usage: GSHELL_HIP_LIB=gshell_amd/lib/variants/rastdbg.so python tools/raster_race_probe3.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int64, check, ptr, stream
from gshell_amd.geometry.mlp import eikonal_sq_sum
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
L = _lib.lib()
assert "GS_RAST_DEBUG" in L.gs_build_flags().decode(), "run with GSHELL_HIP_LIB=gshell_amd/lib/variants/rastdbg.so"
MODE2 = "GS_RAST_DEBUG=2" in L.gs_build_flags().decode()
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
net = tr.geometry.sdf_net
tri = m.faces_i32().contiguous()
v_pos = m.v_pos.detach().contiguous()
mvp, _ = workload.views([0, 1, 2, 3], v_pos.device)
epts = v_pos[torch.randperm(v_pos.shape[0], device="cuda")[:50000]].contiguous()
side = torch.cuda.Stream()
B, H, W = 4, 512, 512
T, V = tri.shape[0], v_pos.shape[0]
with torch.no_grad():
    clip = ru.xfm_points(v_pos[None], mvp).contiguous()
nbytes = int(L.gs_rasterize_scratch_bytes(c_int64(B), c_int64(T), c_int64(H), c_int64(W)))
nscratch = (nbytes + 7) // 8
CAP = 1 << 20
dbg_off = B * H * W + 2 + B * T          # int64 words


def frame():
    scratch = torch.empty(nscratch, dtype=torch.int64, device="cuda")
    rast = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
    db = torch.empty_like(rast)
    vis = torch.zeros(T, dtype=torch.uint8, device="cuda")
    check(L.gs_rasterize_fwd(ptr(clip), c_int64(B), c_int64(V), ptr(tri), c_int64(T), c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(db), ptr(vis), stream()), "gs_rasterize_fwd")
    return rast, scratch


def records(scratch):
    d = scratch[dbg_off:].view(torch.int32)
    n = min(int(d[0]), CAP)
    return d[2:2 + 16 * n].view(n, 16), int(d[0])


def expected(rec):
    """the depth k_rast_small should have computed for each record (bary_eval's operation order, fp32)"""
    px, py, view, t = rec[:, 0].long(), rec[:, 1].long(), rec[:, 2].long(), rec[:, 3].long()
    fx = (px.float() + 0.5) * (2.0 / W) - 1.0
    fy = (py.float() + 0.5) * (2.0 / H) - 1.0
    p = clip[view[:, None], tri[t].long()]            # [n, 3, 4]
    p0, p1, p2 = p[:, 0], p[:, 1], p[:, 2]
    x = [q[:, 0] - fx * q[:, 3] for q in (p0, p1, p2)]
    y = [q[:, 1] - fy * q[:, 3] for q in (p0, p1, p2)]
    a0 = x[1] * y[2] - y[1] * x[2]
    a1 = x[2] * y[0] - y[2] * x[0]
    a2 = x[0] * y[1] - y[0] * x[1]
    z = p0[:, 2] * a0 + p1[:, 2] * a1 + p2[:, 2] * a2
    w = p0[:, 3] * a0 + p1[:, 3] * a1 + p2[:, 3] * a2
    return z / w, fx, fy, p, (a0, a1, a2)


xg = tr.geometry.verts.detach().contiguous()
n_rows = 110000
rows = torch.sort(torch.randperm(xg.shape[0], device="cuda")[:n_rows]).values.int().contiguous()
from gshell_amd.geometry import mlp as M
from gshell_amd._lib import c_int
saved = M._SavedChain(net, 1, xg, rows, n_rows)
g_out = torch.zeros(saved.Rpad, device="cuda")
g_out[:n_rows] = 1e-5
g_x = torch.zeros_like(xg)
Dpl = torch.empty_like(saved.A)


def chain():
    # gs_sdf_mlp_h2_bwd (k_h2_bwd<1>): the side load with the highest rate in tools/raster_race_probe6.py
    check(L.gs_sdf_mlp_h2_bwd(c_int(1), ptr(g_out), ptr(rows), c_int64(n_rows), ptr(None), ptr(saved.packed), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB), ptr(Dpl),
                              ptr(g_x), stream()), "bwd")


def audit2(scratch, label, show=12):
    """GS_RAST_DEBUG == 2: the records are the samples whose own re-evaluation disagreed"""
    rec, n_all = records(scratch)
    print(f"  {label}: {n_all} samples failed their self-check")
    f = lambda r, k: float(r[k:k + 1].view(torch.float32))
    for r in rec[:show]:
        print(f"    px {int(r[0])} py {int(r[1])} view {int(r[2])} tri {int(r[3])}: zw {f(r, 4):.7f}, from fresh operand copies {f(r, 5):.7f}, from z / w again {f(r, 6):.7f}; z {f(r, 7):.6e} w {f(r, 8):.6e} "
              f"(z / w = {f(r, 7) / f(r, 8) if f(r, 8) != 0 else float('nan'):.7f}); a0 {f(r, 9):.6e} / {f(r, 10):.6e}, a1 {f(r, 11):.6e} / {f(r, 12):.6e}, a2 {f(r, 13):.6e} / {f(r, 14):.6e}; s {f(r, 15):.6e}")
    return n_all


def audit(scratch, label, show=10):
    if MODE2:
        return audit2(scratch, label)
    rec, n_all = records(scratch)
    zw = rec[:, 4].view(torch.float32)
    e_zw, fx, fy, p, a = expected(rec)
    bad = ((zw - e_zw).abs() > 1e-3) | (zw != zw)
    print(f"  {label}: {n_all} samples logged ({rec.shape[0]} kept); samples whose depth differs from the recomputation by > 1e-3: {int(bad.sum())}")
    for i in bad.nonzero().reshape(-1)[:show].tolist():
        r = rec[i]
        f = lambda k: float(r[k:k + 1].view(torch.float32))
        print(f"    px {int(r[0])} py {int(r[1])} view {int(r[2])} tri {int(r[3])}: zw {f(4):.6f} (expected {float(e_zw[i]):.6f}); fx {f(5):.6f} ({float(fx[i]):.6f}) fy {f(6):.6f} ({float(fy[i]):.6f}); "
              f"p.z {f(7):.6f} {f(8):.6f} {f(9):.6f} ({[round(float(v), 6) for v in p[i, :, 2]]}); a {f(10):.4e} {f(11):.4e} {f(12):.4e} ({[f'{float(v[i]):.4e}' for v in a]}); "
              f"W {int(r[13])} H {int(r[14])}; p0.w {f(15):.6f} ({float(p[i, 0, 3]):.6f})")
    return int(bad.sum())


ref_r, ref_s = frame()
torch.cuda.synchronize()
audit(ref_s, "stand-alone frame")
ref_z = ref_s[:B * H * W].clone()
main = torch.cuda.current_stream()
shown = 0
nbad = 0
for it in range(reps):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain()
    r, s = frame()
    torch.cuda.synchronize()
    dz = int((s[:B * H * W] != ref_z).sum())
    if dz:
        nbad += 1
        if shown < 6:
            shown += 1
            audit(s, f"rep {it}: {dz} z-buffer words differ")
    elif MODE2 and int(s[dbg_off:].view(torch.int32)[0]):
        audit(s, f"rep {it}: z-buffer equal, yet")
print(f"{nbad} of {reps} frames with a different z-buffer")
