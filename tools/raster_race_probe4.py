"""Fourth probe of the two-queue rasteriser differences: does a raster kernel read a register (or LDS word) it never wrote?  A poison kernel
(tools/micro/vgpr_poison.hip: every wave fills 255 + 256 VGPRs, ~90 SGPRs and 64 KB of LDS with a pattern) runs on the SAME stream right before each
frame; a frame that then differs from the stand-alone result points at an uninitialised read (what a co-resident kernel of another queue would also
change).  GPU box.   usage: python tools/raster_race_probe4.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int64, check, ptr, stream
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
P = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH) if "variants" in _lib.LIB_PATH else os.path.join(os.path.dirname(_lib.LIB_PATH), "variants"), "vgpr_poison.so"))
P.poison_launch.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
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


def frame():
    scratch = torch.empty(nscratch, dtype=torch.int64, device="cuda")
    rast = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
    db = torch.empty_like(rast)
    vis = torch.zeros(T, dtype=torch.uint8, device="cuda")
    check(L.gs_rasterize_fwd(ptr(clip), c_int64(B), c_int64(V), ptr(tri), c_int64(T), c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(db), ptr(vis), stream()), "gs_rasterize_fwd")
    return rast, scratch[:B * H * W]


ref_r, ref_z = frame()
torch.cuda.synchronize()
for pattern in (0x7fc00000, 0xffffffff, 0x3f000000, 0x00000000, 0x12345678):
    bad = 0
    for it in range(reps):
        rc = P.poison_launch(pattern, stream())
        assert rc == 0, rc
        r, z = frame()
        torch.cuda.synchronize()
        bad += int(bool((z != ref_z).any()) or not torch.equal(r, ref_r))
    print(f"  poison pattern {pattern:#010x} before every frame: {bad} of {reps} frames differ from the stand-alone result")
