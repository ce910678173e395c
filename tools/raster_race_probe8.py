"""Eighth probe: synthetic side-queue loads of ONE instruction class each (tools/micro/aggressors.hip) against the rasteriser built WITH packed-fp32
instructions (GS_NO_FILEFLAGS=1 tools/build_variant.sh rastpk raster.hip).  GPU box.
usage: GSHELL_HIP_LIB=gshell_amd/lib/variants/rastpk.so python tools/raster_race_probe8.py [reps]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int, c_int64, check, ptr, stream
from gshell_amd.geometry import mlp as M
from gshell_amd.render import renderutils as ru

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib", "variants", "aggressors.so"))
A.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
NAMES = ["packed fp32 (v_pk_mul / add / fma with op_sel, neg)", "SDWA converts (v_cvt_f32_f16_sdwa, v_add_f32_sdwa, v_cvt_f16_f32_sdwa)", "MFMA f32_32x32x16_f16", "MFMA + packed fp32",
         "transcendentals (v_exp / log / rcp / sqrt)", "LDS reads + writes", "converts (v_cvt_pk_f16_f32, v_cvt_pk_bf16_f32, v_fma_mixlo_f16)",
         "epilogue mix (exp + pk converts + SDWA + packed + 16-bit LDS writes + MFMA)", "MFMA + SDWA converts"]
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
L = _lib.lib()
print("library", _lib.LIB_PATH, "build flags", repr(L.gs_build_flags().decode()))
with torch.no_grad():
    m = tr.geometry.getMesh(tr.mat)['imesh']
net = tr.geometry.sdf_net
tri = m.faces_i32().contiguous()
v_pos = m.v_pos.detach().contiguous()
mvp, _ = workload.views([0, 1, 2, 3], v_pos.device)
B, H, W = 4, 512, 512
T, V = tri.shape[0], v_pos.shape[0]
with torch.no_grad():
    clip = ru.xfm_points(v_pos[None], mvp).contiguous()
nscratch = (int(L.gs_rasterize_scratch_bytes(c_int64(B), c_int64(T), c_int64(H), c_int64(W))) + 7) // 8
sink = torch.zeros(16, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
xg = tr.geometry.verts.detach().contiguous()
n_rows = 110000
rows = torch.sort(torch.randperm(xg.shape[0], device="cuda")[:n_rows]).values.int().contiguous()
saved = M._SavedChain(net, 1, xg, rows, n_rows)
g_out = torch.zeros(saved.Rpad, device="cuda")
g_out[:n_rows] = 1e-5
g_x = torch.zeros_like(xg)
Dpl = torch.empty_like(saved.A)


def k_bwd():
    check(L.gs_sdf_mlp_h2_bwd(c_int(1), ptr(g_out), ptr(rows), c_int64(n_rows), ptr(None), ptr(saved.packed), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB), ptr(Dpl),
                              ptr(g_x), stream()), "bwd")


def frame():
    scratch = torch.empty(nscratch, dtype=torch.int64, device="cuda")
    rast = torch.empty((B, H, W, 4), dtype=torch.float32, device="cuda")
    db = torch.empty_like(rast)
    vis = torch.zeros(T, dtype=torch.uint8, device="cuda")
    check(L.gs_rasterize_fwd(ptr(clip), c_int64(B), c_int64(V), ptr(tri), c_int64(T), c_int64(H), c_int64(W), ptr(scratch), ptr(rast), ptr(db), ptr(vis), stream()), "gs_rasterize_fwd")
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


def run(label, load):
    ms = timed(load)
    bad = 0
    for it in range(reps):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            load()
        z = frame()
        torch.cuda.synchronize()
        bad += int(bool((z != ref_z).any()))
    print(f"  side load = {label} ({ms:.3f} ms stand-alone): {bad} of {reps} frames with a different z-buffer")


A.mix_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
BITS = ["v_exp_f32", "v_cvt_pk_f16_f32", "v_cvt_f32_f16_sdwa", "packed fp32", "16-bit LDS write", "MFMA"]
if len(sys.argv) > 2 and sys.argv[2] == "mix":
    masks = [int(a) for a in sys.argv[3:]] or [63] + [63 ^ (1 << b) for b in range(6)] + [32 | (1 << b) for b in range(5)] + [4 | 16, 4 | 16 | 32, 2 | 4 | 32, 1 | 4 | 32, 4 | 8 | 32]
    for mask in masks:
        rounds = 2000
        A.mix_launch(mask, 1024, rounds, sink.data_ptr(), stream())
        ms = timed(lambda: A.mix_launch(mask, 1024, rounds, sink.data_ptr(), stream()))
        rounds = max(50, int(rounds * 0.6 / max(ms, 1e-3)))
        run(f"mix {mask:2d} = {' + '.join(n for b, n in enumerate(BITS) if mask >> b & 1)}", lambda: A.mix_launch(mask, 1024, rounds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
    sys.exit(0)
run("gs_sdf_mlp_h2_bwd (k_h2_bwd<1>), the control", k_bwd)
for mode, name in enumerate(NAMES):
    rounds = 2000
    A.aggr_launch(mode, 1024, rounds, sink.data_ptr(), stream())
    ms = timed(lambda: A.aggr_launch(mode, 1024, rounds, sink.data_ptr(), stream()))
    rounds = max(50, int(rounds * 0.6 / max(ms, 1e-3)))              # ~0.6 ms stand-alone
    run(f"synthetic: {name}", lambda: A.aggr_launch(mode, 1024, rounds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
