"""Error / sign-flip / time table of the SDF-network forward kernels on the bench grid (tet-res 256, 2 282 489 rows).  GPU box.

    python tools/mlp_precision.py [res] > gpurun_out/r02_mlp_precision.json

Compares, against a float64 evaluation of the same network (fitted to the bench's capped-cone field):
  torch fp32 (hipBLASLt)  |  exact-fp32 MFMA kernel (csrc/mlp.hip)  |  fp16-pair "h2" kernel (csrc/mlp_h2.hip)
and counts the grid vertices whose SIGN differs (the sign decides the extracted topology)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool measures oracle / alternate-design kernels: they live in lib/variants/oracles.so (csrc/common.hpp GS_ORACLE_KERNELS), not in the shipped library
os.environ.setdefault("GSHELL_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gshell_amd", "lib", "variants", "oracles.so"))
import torch

from gshell_amd import grid, workload
from gshell_amd.geometry.mlp import MLP, fused_forward

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
verts, _ = grid.grid_for_res(res, device="cuda")
verts = ((verts - verts.mean(dim=0)) * 1.4).contiguous()


class _G:      # what workload.fit_sdf_net needs
    pass


g = _G()
g.verts = verts
g.sdf_net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
workload.fit_sdf_net(g, steps=400)
net = g.sdf_net


def chunks(fn, x, n=1 << 18):
    return torch.cat([fn(x[i:i + n]) for i in range(0, x.shape[0], n)])


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return y, e0.elapsed_time(e1) / reps


with torch.no_grad():
    net64 = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda().double()
    net64.load_state_dict({k: v.double() for k, v in net.state_dict().items()})
    y64 = chunks(lambda x: net64(x.double()), verts)[:, 0]
    y_t, ms_t = timed(lambda: chunks(net, verts)[:, 0], reps=2)
    y_f, ms_f = timed(lambda: fused_forward(net, verts, "fp32")[:, 0])
    y_h, ms_h = timed(lambda: fused_forward(net, verts, "h2")[:, 0])

scale = float(y64.abs().max())


def row(y):
    d = (y.double() - y64).abs()
    return {"max_abs_err": float(d.max()), "max_err_rel_to_max": float(d.max()) / scale, "rms_err": float(d.pow(2).mean().sqrt()),
            "sign_flips_vs_f64": int(((y > 0) != (y64 > 0)).sum())}


out = {
    "grid": f"tet-res{res}", "rows": int(verts.shape[0]), "max_abs_sdf": scale,
    "rows_with_abs_sdf_below_1e-6": int((y64.abs() < 1e-6).sum()), "rows_with_abs_sdf_below_1e-5": int((y64.abs() < 1e-5).sum()),
    "torch_fp32": dict(row(y_t), ms=round(ms_t, 3)),
    "mfma_fp32_kernel": dict(row(y_f), ms=round(ms_f, 3)),
    "h2_kernel": dict(row(y_h), ms=round(ms_h, 3)),
    "h2_vs_mfma_fp32": {"max_abs_diff": float((y_h - y_f).abs().max()), "max_diff_rel_to_max": float((y_h - y_f).abs().max()) / scale,
                        "sign_flips": int(((y_h > 0) != (y_f > 0)).sum())},
    "torch_fp32_vs_mfma_fp32": {"max_abs_diff": float((y_t - y_f).abs().max()), "sign_flips": int(((y_t > 0) != (y_f > 0)).sum())},
    "tflops_algorithmic": {"mfma_fp32_kernel": round(826880.0 * verts.shape[0] / ms_f / 1e9, 1), "h2_kernel": round(826880.0 * verts.shape[0] / ms_h / 1e9, 1)},
}
print(json.dumps(out, indent=1))
