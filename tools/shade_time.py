"""Times gs_env_shade_fwd / bwd on the bench workload (HIP events around the C-ABI calls over 6 training iterations).  GPU box.
usage: [GSHELL_HIP_LIB=...] python tools/shade_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload

tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
tg = workload.make_targets(tr, [0, 1, 2, 3], (512, 512))
tr.it = 1000
for _ in range(3):
    tr.step(tg)
names = {"gs_env_shade_fwd", "gs_env_shade_bwd_saved", "gs_bvh_build", "gs_hashgrid_encode_bwd", "gs_hashgrid_encode_bwd_binned", "gs_sdf_mlp_fwd_h1", "gs_sdf_mlp_h2_wgrad", "gs_sdf_mlp_h2_bwd",
         "gs_sdf_mlp_h2_save_fwd", "gs_sdf_eikonal_rr_fwd", "gs_sdf_eikonal_rr_bwd"}
_lib.enable_op_timing(True, only=names)
_lib.reset_op_timing()
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    tr.step(tg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 8 * 1e3
for k, v in sorted(_lib.op_timing_summary().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"{k:28s} {v['ms']:7.3f} ms x {v['n'] / 8:.1f}")
print(f"iteration {dt:.2f} ms")
