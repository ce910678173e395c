"""The two kernels behind gs_sdf_mlp_fwd_h1 (0: activations in LDS, 1: register-resident, round 5) on the bench grid: error against the
three-product kernel (gs_sdf_mlp_fwd_h2), sign disagreements, HIP-event time.  python tools/h1r_check.py [res] [fit_steps].  GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# this tool measures oracle / alternate-design kernels: they live in lib/variants/oracles.so (csrc/common.hpp GS_ORACLE_KERNELS), not in the shipped library
os.environ.setdefault("GSHELL_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gshell_amd", "lib", "variants", "oracles.so"))
import torch

from gshell_amd import _lib, grid
from gshell_amd.geometry import mlp
from gshell_amd.geometry.mlp import MLP

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fit = int(sys.argv[2]) if len(sys.argv) > 2 else 150
torch.manual_seed(0)
verts, _ = grid.grid_for_res(res, device="cuda")
net = MLP(n_freq=6, d_hidden=256, n_hidden=6, skip_in=[3]).cuda()
if fit:      # a fitted field like the bench's (workload.skirt_sdf)
    from gshell_amd import workload
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(fit):
        idx = torch.randint(0, verts.shape[0], (65536,), device="cuda", generator=g)
        loss = (net(verts[idx])[:, 0] - workload.skirt_sdf(verts[idx])).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
L = _lib.lib()
N = verts.shape[0]
with torch.no_grad():
    L.gs_sdf_mlp_h1_impl(_lib.c_int(1))          # the packer writes kernel 1's fragment section only while it is selected
    packed, n_hidden, skip = mlp.pack_weights_h2(net)
    ref = torch.empty(N, device="cuda")
    _lib.check(L.gs_sdf_mlp_fwd_h2(_lib.ptr(verts), _lib.c_int64(N), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip), _lib.ptr(ref),
                                   _lib.c_void_p(0), _lib.c_void_p(0), _lib.stream()))
    out = {}
    for impl in (0, 1):
        L.gs_sdf_mlp_h1_impl(_lib.c_int(impl))
        y = torch.empty(N, device="cuda")
        occ = torch.zeros((N + 63) // 64, dtype=torch.int64, device="cuda")
        st = torch.zeros(4, dtype=torch.int32, device="cuda")

        def run():
            _lib.check(L.gs_sdf_mlp_fwd_h1(_lib.ptr(verts), _lib.c_int64(N), _lib.ptr(packed), _lib.c_int(6), _lib.c_int(n_hidden), _lib.c_int(skip), _lib.ptr(y),
                                           _lib.ptr(occ), _lib.ptr(st), _lib.stream()))
        for _ in range(3):
            run()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            run()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        err = (y - ref).abs()
        bits = ((occ[:, None] >> torch.arange(64, device="cuda")[None]) & 1).reshape(-1)[:N].bool()
        flips = int(((y > 0) != (ref > 0)).sum())
        flips_far = int((((y > 0) != (ref > 0)) & (ref.abs() > 5e-4)).sum())
        out[impl] = y
        print(f"impl {impl}: {ms:.3f} ms = {826880.0 * N / ms / 1e9:.0f} TFLOP/s algorithmic; max |err| {float(err.max()):.3e}, rms {float(err.square().mean().sqrt()):.3e}, "
              f"max |ref| {float(ref.abs().max()):.3f}; sign flips {flips} (with |ref| > 5e-4: {flips_far}); occupancy words consistent with values: "
              f"{bool(torch.equal(bits, y > 0))}; status {st.tolist()}")
    print(f"impl 1 vs impl 0: max |diff| {float((out[1] - out[0]).abs().max()):.3e}")
