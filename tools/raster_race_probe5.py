"""Fifth probe of the two-queue rasteriser differences.  Probes 1-4: single samples of k_rast_small write a wrong DEPTH (right pixel, right triangle) while
row_sparse_backward / the eikonal chain run on another stream -- the side loads that contain a kernel with register spills (k_h2_bwd<1>, k_h2_fwd<4>:
private_segment_fixed_size 68 .. 108) -- and never under k_h1_fwd, hipBLASLt GEMMs or a device copy (no scratch); poisoning the register files before the
frame changes nothing.  Here the side load is a synthetic kernel whose only special property is a dynamically indexed per-lane array (scratch), against the
same arithmetic without it; victims: the rasteriser, gs_xfm_points_fwd and plain torch arithmetic.  GPU box.   usage: python tools/raster_race_probe5.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int64, check, ptr, stream
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib", "variants", "scratch_load.so"))
S.load_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
L = _lib.lib()
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
tri = m.faces_i32().contiguous()
v_pos = m.v_pos.detach().contiguous()
mvp, _ = workload.views([0, 1, 2, 3], v_pos.device)
B, H, W = 4, 512, 512
T, V = tri.shape[0], v_pos.shape[0]
with torch.no_grad():
    clip = ru.xfm_points(v_pos[None], mvp).contiguous()
nscratch = (int(L.gs_rasterize_scratch_bytes(c_int64(B), c_int64(T), c_int64(H), c_int64(W))) + 7) // 8
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
ta = torch.rand(1 << 22, device="cuda") + 0.5
tb = torch.rand(1 << 22, device="cuda") + 0.5


def raster():
    scratch = torch.empty(nscratch, dtype=torch.int64, device="cuda")
    rast = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
    db = torch.empty_like(rast)
    vis = torch.zeros(T, dtype=torch.uint8, device="cuda")
    check(L.gs_rasterize_fwd(ptr(clip), c_int64(B), c_int64(V), ptr(tri), c_int64(T), c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(db), ptr(vis), stream()), "gs_rasterize_fwd")
    return scratch[:B * H * W]


def xfm():
    with torch.no_grad():
        return ru.xfm_points(v_pos[None], mvp)


def torch_math():
    return (ta / tb + ta * tb).sqrt() / (tb + 1.0)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for use_scratch, rounds in ((1, 2000), (0, 20000)):
    ms = timed(lambda: S.load_launch(use_scratch, 4096, rounds, sink.data_ptr(), stream()))
    print(f"side load: synthetic kernel, scratch = {use_scratch}, 4096 x 256 lanes, {rounds} rounds: {ms:.2f} ms stand-alone")
    for name, victim in (("rasteriser (z-buffer words)", raster), ("gs_xfm_points_fwd", xfm), ("torch arithmetic (div, mul, add, sqrt over 4 M floats)", torch_math)):
        ref = victim().clone()
        torch.cuda.synchronize()
        t_v = timed(victim)
        bad, worst = 0, 0
        for it in range(reps):
            side.wait_stream(main)
            rc = S.load_launch(use_scratch, 4096, rounds, sink.data_ptr(), side.cuda_stream)
            assert rc == 0
            out = victim()
            torch.cuda.synchronize()
            n = int((out != ref).sum())
            bad += int(n > 0)
            worst = max(worst, n)
        print(f"    victim {name} ({t_v:.3f} ms): {bad} of {reps} runs differ from the stand-alone result (most differing elements in a run: {worst})")
