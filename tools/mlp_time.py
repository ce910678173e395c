"""Times the fused SDF-MLP forward kernels on the bench grid (HIP events, 10 launches each).  GPU box.
usage: python tools/mlp_time.py [h2|fp32|both] [res]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool measures oracle / alternate-design kernels: they live in lib/variants/oracles.so (csrc/common.hpp GS_ORACLE_KERNELS), not in the shipped library
os.environ.setdefault("GSHELL_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gshell_amd", "lib", "variants", "oracles.so"))
import torch

from gshell_amd import grid
from gshell_amd.geometry.mlp import MLP, fused_forward

which = sys.argv[1] if len(sys.argv) > 1 else "h2"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.manual_seed(0)
verts, _ = grid.grid_for_res(res, device="cuda")
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
with torch.no_grad():
    for prec in (("h2", "fp32") if which == "both" else (which,)):
        for _ in range(2):
            y = fused_forward(net, verts, prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = fused_forward(net, verts, prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{prec}: {ms:.3f} ms  ({826880.0 * verts.shape[0] / ms / 1e9:.1f} TFLOP/s algorithmic)  mean|y| {float(y.abs().mean()):.6f}")
