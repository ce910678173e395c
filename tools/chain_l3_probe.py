"""Would the SDF gradient chain gain from chunks whose saved planes fit the 256 MiB Infinity Cache?  Times the eikonal term
(gs_sdf_eikonal_rr_fwd / _bwd, HIP events around the C-ABI calls) for n samples, with the backward right behind the forward (planes of a small n are
still on chip) and with 1.5 GB streamed in between (they are not), and the row-sparse backward for r rows.  Planes: 2 x 9 x 2 n KB (eikonal),
2 x 9 x r KB (rows).  GPU box.   usage: [GSHELL_HIP_LIB=...] python tools/chain_l3_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib
from gshell_amd.geometry.mlp import MLP, eikonal_sq_sum, row_sparse_backward

torch.manual_seed(0)
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
flush_buf = torch.empty(1536 << 18, dtype=torch.float32, device="cuda")       # 1.5 GiB


def timed(fn, reps=6):
    for _ in range(2):
        fn()
    _lib.enable_op_timing(True)
    _lib._timing["pending"].clear()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    agg = {}
    for name, e0, e1 in _lib._timing["pending"]:
        agg.setdefault(name, 0.0)
        agg[name] += e0.elapsed_time(e1) / reps
    _lib.enable_op_timing(False)
    return agg


for n in (4096, 8192, 16384, 50000):
    pts = (torch.rand(n, 3, device="cuda") - 0.5).contiguous()

    def warm():
        eikonal_sq_sum(net, pts).backward()

    def cold():
        loss = eikonal_sq_sum(net, pts)
        flush_buf.fill_(1.0)
        loss.backward()

    for tag, fn in (("on chip", warm), ("flushed", cold)):
        a = timed(fn)
        f, b = a.get("gs_sdf_eikonal_rr_fwd", 0.0), a.get("gs_sdf_eikonal_rr_bwd", 0.0)
        print(f"eikonal n={n:6d} planes {2 * 9 * 2 * n / 1024:7.0f} MiB  {tag}:  fwd {f:.3f} ms  bwd {b:.3f} ms   {1e6 * (f + b) / n:7.1f} ns / sample")

N = 2282489
x = (torch.rand(N, 3, device="cuda") - 0.5).contiguous()
for r in (8192, 16384, 32768, 110000):
    gy = torch.zeros(N, 1, device="cuda")
    idx = torch.randperm(N, device="cuda")[:r]
    gy[idx, 0] = torch.randn(r, device="cuda") * 1e-5
    a = timed(lambda: row_sparse_backward(net, x, gy, True))
    t = sum(v for k, v in a.items() if k.startswith("gs_sdf_mlp_h2_"))
    print(f"rows    r={r:6d} planes {2 * 9 * r / 1024:7.0f} MiB:  " + "  ".join(f"{k[14:]} {v:.3f}" for k, v in a.items() if k.startswith("gs_sdf_mlp_h2_")) + f"   {1e6 * t / r:7.1f} ns / row")
