"""Sixth probe of the two-queue rasteriser differences: WHICH chain kernel on the side stream does it?  The row-sparse backward's three C-ABI calls
(gs_sdf_mlp_h2_save_fwd = k_h2_fwd<1>, gs_sdf_mlp_h2_bwd = k_h2_bwd<1>, gs_sdf_mlp_h2_wgrad = k_h2_wgrad16) run one at a time as the side load.  GPU box.
usage: python tools/raster_race_probe6.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int, c_int64, check, ptr, stream
from gshell_amd.geometry import mlp as M
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
L = _lib.lib()
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
net = tr.geometry.sdf_net
tri = m.faces_i32().contiguous()
v_pos = m.v_pos.detach().contiguous()
mvp, _ = workload.views([0, 1, 2, 3], v_pos.device)
side = torch.cuda.Stream()
B, H, W = 4, 512, 512
T, V = tri.shape[0], v_pos.shape[0]
with torch.no_grad():
    clip = ru.xfm_points(v_pos[None], mvp).contiguous()
nscratch = (int(L.gs_rasterize_scratch_bytes(c_int64(B), c_int64(T), c_int64(H), c_int64(W))) + 7) // 8
xg = tr.geometry.verts.detach().contiguous()
n = 110000
rows = torch.sort(torch.randperm(xg.shape[0], device="cuda")[:n]).values.int().contiguous()
saved = M._SavedChain(net, 1, xg, rows, n)
g_out = torch.zeros(saved.Rpad, device="cuda")
g_out[:n] = 1e-5
g_x = torch.zeros_like(xg)
D = torch.empty_like(saved.A)
params = list(net.parameters())
flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device="cuda")
grads = dict(zip((id(p) for p in params), M.split_param_grads(net, flat)))
dW = [grads[id(l.weight)] for l in saved.lin]
db = [grads[id(l.bias)] for l in saved.lin]
torch.cuda.synchronize()


def k_fwd():
    check(L.gs_sdf_mlp_h2_save_fwd(c_int(1), ptr(xg), ptr(rows), c_int64(n), ptr(None), ptr(saved.packed), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB),
                                   ptr(None), stream()), "save_fwd")


def k_bwd():
    check(L.gs_sdf_mlp_h2_bwd(c_int(1), ptr(g_out), ptr(rows), c_int64(n), ptr(None), ptr(saved.packed), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB), ptr(D),
                              ptr(g_x), stream()), "bwd")


def k_wgrad():
    check(L.gs_sdf_mlp_h2_wgrad(c_int(1), ptr(g_out), c_int64(n), ptr(None), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB), ptr(D), M._ptr_array(dW),
                                M._ptr_array(db), c_int(1 if M.SDF_MLP_WGRAD_FP32 else 0), stream()), "wgrad")


k_bwd()          # D exists before the weight-gradient pass runs alone
torch.cuda.synchronize()


def frame():
    scratch = torch.empty(nscratch, dtype=torch.int64, device="cuda")
    rast = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
    db_ = torch.empty_like(rast)
    vis = torch.zeros(T, dtype=torch.uint8, device="cuda")
    check(L.gs_rasterize_fwd(ptr(clip), c_int64(B), c_int64(V), ptr(tri), c_int64(T), c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(db_), ptr(vis), stream()), "gs_rasterize_fwd")
    return scratch[:B * H * W]


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


ref_z = frame().clone()
torch.cuda.synchronize()
main = torch.cuda.current_stream()
for label, load in (("gs_sdf_mlp_h2_save_fwd (k_h2_fwd<1>)", k_fwd), ("gs_sdf_mlp_h2_bwd (k_h2_bwd<1>)", k_bwd), ("gs_sdf_mlp_h2_wgrad (k_h2_wgrad16)", k_wgrad)):
    ms = timed(load)
    bad = 0
    for it in range(reps):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            load()
        z = frame()
        torch.cuda.synchronize()
        bad += int(bool((z != ref_z).any()))
    print(f"  side load = {label}, {ms:.3f} ms stand-alone: {bad} of {reps} frames with a different z-buffer")
