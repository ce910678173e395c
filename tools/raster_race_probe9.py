"""Ninth probe: are the kernels that disturb OTHER kernels' packed arithmetic (k_h2_fwd / k_h2_bwd: matrix + packed-fp32 instructions in one kernel) themselves
bit-reproducible?  Their outputs without atomics -- the saved activation planes A, the embedding, the reverse chain's delta planes D, d loss / d x -- are
recomputed `reps` times stand-alone and compared bit for bit with the first run; likewise the full-grid forward (k_h1_fwd + refinement, k_h2_fwd<GRID>).  GPU box.
usage: python tools/raster_race_probe9.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gshell_amd import _lib, workload
from gshell_amd._lib import c_int, c_int64, check, ptr, stream
from gshell_amd.geometry import mlp as M

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tr = workload.build(res=256, n_samples=8, batch=4, train_res=(512, 512), fit_steps=200)
L = _lib.lib()
net = tr.geometry.sdf_net
xg = tr.geometry.verts.detach().contiguous()
n = 110000
rows = torch.sort(torch.randperm(xg.shape[0], device="cuda")[:n]).values.int().contiguous()
g_out = None


def chain():
    global g_out
    saved = M._SavedChain(net, 1, xg, rows, n)
    if g_out is None:
        g_out = torch.zeros(saved.Rpad, device="cuda")
        g_out[:n] = torch.randn(n, device="cuda") * 1e-5
    D = torch.empty_like(saved.A)
    # g_x is accumulated with one plain store per row here (rows are distinct): zero-filled, deterministic
    g_x = torch.zeros_like(xg)
    check(L.gs_sdf_mlp_h2_bwd(c_int(1), ptr(g_out), ptr(rows), c_int64(n), ptr(None), ptr(saved.packed), c_int(saved.nf), c_int(saved.n_hidden), c_int(saved.skip), ptr(saved.A), ptr(saved.EMB), ptr(D),
                              ptr(g_x), stream()), "bwd")
    return {"A (k_h2_fwd<1>)": saved.A[:, :n], "EMB": saved.EMB[:n], "D (k_h2_bwd<1>)": D[:, :n], "d loss / d x": g_x}


def grid_forward(precision):
    with torch.no_grad():
        return {f"sdf over the grid, precision {precision}": M.fused_forward(net, xg, precision=precision)}


for label, fn in (("row-sparse chain over 110 000 rows", chain), ("full-grid forward h2 (k_h2_fwd<GRID>, three products)", lambda: grid_forward("h2"))):
    ref = {k: v.clone() for k, v in fn().items()}
    torch.cuda.synchronize()
    bad = {k: 0 for k in ref}
    worst = {k: 0 for k in ref}
    for it in range(reps):
        out = fn()
        torch.cuda.synchronize()
        for k in ref:
            d = int((out[k] != ref[k]).sum())
            bad[k] += int(d > 0)
            worst[k] = max(worst[k], d)
    print(f"  {label}: " + "; ".join(f"{k}: {bad[k]} of {reps} runs differ (most elements: {worst[k]} of {ref[k].numel()})" for k in ref))
